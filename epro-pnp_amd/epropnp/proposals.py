"""The AMIS proposal family in PyTorch: fits and distribution objects behind the reference's extension hooks
`allocate_buffer / initial_fit / gen_new_distr / gen_old_distr / estimate_params`
(epropnp/epropnp.py:64-82 abstract, :209-260 4-DoF, :282-342 6-DoF).

`monte_carlo_forward` does not call these: the fused sampler (csrc/amis_forward_mfma.hip, fits in csrc/amis_common.h)
evaluates the same formulas on the device in one launch.  They exist for code that subclasses the layer or inspects
the proposals (tests compare them with the kernel's own proposal records), and run wherever their tensors live.
"""
import math

import torch
from torch.distributions import constraints
from torch.distributions.distribution import Distribution

from .distributions import AngularCentralGaussian, VonMisesUniformMix


class MultivariateStudentT(Distribution):
    """Multivariate Student-t with `df` degrees of freedom, location `loc` (*,n) and scale factor `scale_tril` (*,n,n)
    (the reference takes this class from pyro; the arithmetic is the textbook one, see oracle/pyro_shim.py)."""

    arg_constraints = {'df': constraints.positive, 'loc': constraints.real_vector, 'scale_tril': constraints.lower_cholesky}
    support = constraints.real_vector
    has_rsample = True

    def __init__(self, df, loc, scale_tril, validate_args=None):
        n = loc.size(-1)
        assert scale_tril.shape[-2:] == (n, n)
        batch = torch.broadcast_shapes(loc.shape[:-1], scale_tril.shape[:-2])
        self.df = torch.as_tensor(df, dtype=loc.dtype, device=loc.device).expand(batch)
        self.loc = loc.expand(batch + (n,))
        self.scale_tril = scale_tril.expand(batch + (n, n))
        super().__init__(batch, (n,), validate_args=validate_args)

    def rsample(self, sample_shape=torch.Size()):
        shape = self._extended_shape(sample_shape)
        z = torch.randn(shape, dtype=self.loc.dtype, device=self.loc.device)
        chi2 = torch.distributions.Chi2(self.df).rsample(sample_shape)
        y = z * torch.rsqrt(chi2 / self.df).unsqueeze(-1)
        return self.loc + torch.matmul(self.scale_tril, y.unsqueeze(-1)).squeeze(-1)

    def log_prob(self, value):
        n = self.loc.size(-1)
        shape = torch.broadcast_shapes(value.shape[:-1], self.batch_shape)
        diff = (value - self.loc).expand(shape + (n,)).unsqueeze(-1)
        white = torch.linalg.solve_triangular(self.scale_tril.expand(shape + (n, n)), diff, upper=False).squeeze(-1)
        maha = white.square().sum(-1)
        half = 0.5 * (self.df + n)
        log_norm = (self.scale_tril.diagonal(dim1=-2, dim2=-1).log().sum(-1) + 0.5 * n * (self.df.log() + math.log(math.pi))
                    + torch.lgamma(0.5 * self.df) - torch.lgamma(half))
        return -half * torch.log1p(maha / self.df) - log_norm


def cholesky_or_default(mat, default_diag=None):
    """Batched Cholesky; where it fails (not positive definite / not finite) the factor is diag(default_diag) or I
    (cholesky_wrapper, epropnp.py:16-33) -- detected with cholesky_ex, no host round-trip."""
    tril, info = torch.linalg.cholesky_ex(mat)
    n = mat.size(-1)
    fallback = torch.diag(mat.new_tensor(default_diag)) if default_diag is not None \
        else torch.eye(n, dtype=mat.dtype, device=mat.device)
    bad = (info != 0) | ~torch.isfinite(tril).flatten(-2).all(-1)
    return torch.where(bad[..., None, None], fallback, tril)


def translation_moments(pose_samples, weights):
    """Weighted mean (B,3) and covariance (B,3,3) of the sample translations; weights (M,B) sum to one over M."""
    t = pose_samples[..., :3]
    mean = torch.einsum('mb,mbi->bi', weights, t)
    dev = t - mean
    return mean, torch.einsum('mb,mbi,mbj->bij', weights, dev, dev)


def acg_shape_factor(shape_mat, dispersion):
    """ACG shape matrix -> Cholesky factor of (shape + det^(1/4) * dispersion * I); identity where that fails."""
    eye = torch.eye(4, dtype=shape_mat.dtype, device=shape_mat.device)
    return cholesky_or_default(shape_mat + torch.det(shape_mat)[:, None, None] ** 0.25 * (dispersion * eye))


def acg_shape_from_laplace(quat, rot_cov, tangent_map):
    """Proposal 0 of the 6-DoF sampler: the 3x3 tangent covariance at `quat` lifted to a trace-one 4x4 ACG shape matrix,
    (T C^-1 T^T + I)^-1 / trace, with T = tangent_map(quat) (camera.py:145-165)."""
    T = tangent_map(quat)
    eye = torch.eye(4, dtype=quat.dtype, device=quat.device)
    shape = torch.linalg.inv(T @ torch.linalg.inv(rot_cov) @ T.transpose(-1, -2) + eye)
    return shape / shape.diagonal(dim1=-2, dim2=-1).sum(-1)[:, None, None]


def acg_shape_mle(quats, weights, num_iter, eps):
    """Weighted fixed-point iteration for the ACG maximum-likelihood shape matrix (Tyler's estimator):
    Sigma <- sum_j v_j q_j q_j^T + eps I,  v_j ~ w_j / max(q_j^T Sigma^-1 q_j, eps), normalised over j; Sigma_0 = I."""
    B = quats.size(1)
    eye = torch.eye(4, dtype=quats.dtype, device=quats.device)
    outer = quats.unsqueeze(-1) * quats.unsqueeze(-2)                      # (M,B,4,4)
    shape = eye.expand(B, 4, 4)
    for _ in range(num_iter):
        maha = torch.einsum('mbi,bij,mbj->mb', quats, torch.linalg.inv(shape), quats)
        v = weights / maha.clamp(min=eps)
        v = v / v.sum(dim=0, keepdim=True)
        shape = torch.einsum('mb,mbij->bij', v, outer) + eps * eye
    return shape


def yaw_concentration(pose_samples, weights, eps):
    """Weighted circular mean and the reference's concentration heuristic of the yaw samples -> (B,1), (B,1)."""
    yaw = pose_samples[..., 3:]
    ms = (weights.unsqueeze(-1) * yaw.sin()).sum(dim=0)
    mc = (weights.unsqueeze(-1) * yaw.cos()).sum(dim=0)
    r2 = ms.square() + mc.square()
    kappa = 0.33 * r2.sqrt().clamp(min=eps) * (2 - r2) / (1 - r2).clamp(min=eps)
    return torch.atan2(ms, mc), kappa


def student_t(df, mode, tril):
    return MultivariateStudentT(df, mode, tril)


__all__ = ['MultivariateStudentT', 'AngularCentralGaussian', 'VonMisesUniformMix', 'cholesky_or_default',
           'translation_moments', 'acg_shape_factor', 'acg_shape_from_laplace', 'acg_shape_mle', 'yaw_concentration']
