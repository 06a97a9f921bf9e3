"""Proposal distributions of the AMIS sampler as torch.distributions objects.

API mirror of the reference's epropnp/distributions.py (AngularCentralGaussian :15-52, VonMisesUniformMix :55-79),
kept for callers that build these objects directly (visualisation / diagnostics).  The sampler itself evaluates the
same densities inside the HIP kernel (csrc/amis_kernels.hip); it does not construct these classes.
Unlike the reference there is no pyro dependency (its base class is torch's own Distribution) and the von Mises
draw stays on the tensor's device (torch's sampler) instead of round-tripping through numpy on the host.
"""
import math

import torch
from torch.distributions import VonMises, constraints
from torch.distributions.distribution import Distribution


class AngularCentralGaussian(Distribution):
    """ACG on the unit sphere S^{q-1}, parameterised by the Cholesky factor of its shape matrix."""

    arg_constraints = {'scale_tril': constraints.lower_cholesky}
    has_rsample = True

    def __init__(self, scale_tril, validate_args=None, eps=1e-6):
        q = scale_tril.size(-1)
        assert q > 1 and scale_tril.shape[-2:] == (q, q)
        self.scale_tril = scale_tril
        self.q = q
        self.area = 2 * math.pi ** (0.5 * q) / math.gamma(0.5 * q)
        self.eps = eps
        super().__init__(scale_tril.shape[:-2], (q,), validate_args=validate_args)

    def log_prob(self, value):
        shape = torch.broadcast_shapes(value.shape[:-1], self.scale_tril.shape[:-2])
        rhs = value.expand(shape + (self.q,)).unsqueeze(-1)
        white = torch.linalg.solve_triangular(self.scale_tril.expand(shape + (self.q, self.q)), rhs, upper=False)
        maha = white.squeeze(-1).square().sum(-1)
        half_log_det = self.scale_tril.diagonal(dim1=-2, dim2=-1).log().sum(-1)
        return -0.5 * self.q * maha.log() - half_log_det - math.log(self.area)

    def rsample(self, sample_shape=torch.Size()):
        shape = self._extended_shape(sample_shape)
        g = torch.randn(shape, dtype=self.scale_tril.dtype, device=self.scale_tril.device)
        v = torch.matmul(self.scale_tril, g.unsqueeze(-1)).squeeze(-1)
        nrm = v.norm(dim=-1, keepdim=True)
        pole = torch.zeros_like(v)
        pole[..., 0] = 1.0
        return torch.where(nrm < self.eps, pole, v / nrm.clamp(min=1e-30))


class VonMisesUniformMix(VonMises):
    """(1 - uniform_mix) von Mises + uniform_mix uniform on the circle; the first round(uniform_mix * n) of n
    requested samples are the uniform ones (deterministic split, as in the reference)."""

    def __init__(self, loc, concentration, uniform_mix=0.25, **kwargs):
        super().__init__(loc, concentration, **kwargs)
        self.uniform_mix = uniform_mix

    @torch.no_grad()
    def sample(self, sample_shape=torch.Size()):
        assert len(sample_shape) == 1
        n_uniform = round(sample_shape[0] * self.uniform_mix)
        shape_u = self._extended_shape((n_uniform,))
        uni = (torch.rand(shape_u, dtype=self.loc.dtype, device=self.loc.device) * 2 - 1) * math.pi
        vm = super().sample((sample_shape[0] - n_uniform,))
        return torch.cat((uni, vm), dim=0)

    def log_prob(self, value):
        vm = super().log_prob(value) + math.log(1 - self.uniform_mix)
        return torch.logaddexp(vm, torch.full_like(vm, math.log(self.uniform_mix / (2 * math.pi))))
