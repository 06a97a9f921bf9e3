"""
oracle/epropnp_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain PyTorch tensor algebra, any float dtype, runs on CPU) of the
EPro-PnP hot path of tjiiv-cprg/EPro-PnP: the batched LM / GN PnP solver and the AMIS
Monte-Carlo pose sampler.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; the product package (epro-pnp_amd/) never does.

Every function cites the reference file:line it follows (paths relative to the reference
checkout).  The restatement is *pinned* against the unmodified reference code by
oracle/make_golden.py (run in the build container, where /root/reference exists): with the
same injected random draws both produce the fixtures under tests/golden/ to <= 1e-5.

Differences from the reference that are deliberate (and why):
  * all randomness is passed in explicitly (`noise` dicts), never drawn from a global RNG,
    so that oracle, reference and HIP kernels can consume identical draws;
  * the 4-DoF von Mises draw uses a bounded Best-Fisher rejection loop driven by injected
    uniforms instead of numpy.random.vonmises (epropnp/distributions.py:64-72) -- same
    target distribution, reproducible on a GPU;
  * Cholesky failure handling (epropnp/epropnp.py:16-33) is per-matrix instead of
    try/except around the batch.

Third-party arithmetic restated here because it is not under /root/reference:
  pyro-ppl 1.6.0 `MultivariateStudentT.rsample/log_prob` (EPro-PnP-Det/requirements.txt:6),
  torch.distributions.VonMises.log_prob (torch/distributions/von_mises.py:24-89,144-152).
"""
import math

import torch

# ----------------------------------------------------------------------------------------
# SE(3) helpers
# ----------------------------------------------------------------------------------------


def quat_to_rotmat(q):
    """epropnp/common.py:21-42.  Two algebraically equal forms, selected as in the reference by `q.requires_grad`
    (:30): the `1 - 2 (j^2 + k^2)` form (:31-36) when the quaternion is differentiated -- its derivative w.r.t. q differs
    from the other form's even on the unit sphere -- and R = 2(w[v]x + v v^T) + (w^2 - v.v) I (:37-41) otherwise.
    The quaternion is NOT normalised."""
    w, x, y, z = q.unbind(-1)
    if q.requires_grad:
        r00 = 1 - 2 * (y * y + z * z)
        r11 = 1 - 2 * (x * x + z * z)
        r22 = 1 - 2 * (x * x + y * y)
    else:
        dd = w * w - (x * x + y * y + z * z)
        r00 = 2 * (x * x) + dd
        r11 = 2 * (y * y) + dd
        r22 = 2 * (z * z) + dd
    r01 = 2 * (x * y - w * z)
    r02 = 2 * (x * z + w * y)
    r10 = 2 * (x * y + w * z)
    r12 = 2 * (y * z - w * x)
    r20 = 2 * (x * z - w * y)
    r21 = 2 * (y * z + w * x)
    return torch.stack((r00, r01, r02, r10, r11, r12, r20, r21, r22), -1).reshape(q.shape[:-1] + (3, 3))


def yaw_to_rotmat(yaw):
    """epropnp/common.py:45-64: rotation about the Y axis."""
    c, s = torch.cos(yaw), torch.sin(yaw)
    o, i = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack((c, o, s, o, i, o, -s, o, c), -1).reshape(yaw.shape + (3, 3))


def pose_to_rotmat(pose):
    return yaw_to_rotmat(pose[..., 3]) if pose.shape[-1] == 4 else quat_to_rotmat(pose[..., 3:])


def quat_tangent_map(q):
    """epropnp/camera.py:145-165: 4x3 map from a tangent-space rotation step to a quaternion increment."""
    w, i, j, k = q.unbind(-1)
    rows = (i, j, k, -w, -k, j, k, -w, -i, -j, i, -w)
    return torch.stack(rows, -1).reshape(q.shape[:-1] + (4, 3))


def pose_add(pose, step):
    """epropnp/levenberg_marquardt.py:255-265."""
    if pose.shape[-1] == 4:
        return pose + step
    q = pose[..., 3:] + (quat_tangent_map(pose[..., 3:]) @ step[..., 3:, None]).squeeze(-1)
    q = q / q.norm(dim=-1, keepdim=True).clamp(min=1e-12)  # F.normalize default eps
    return torch.cat((pose[..., :3] + step[..., :3], q), -1)


def pnp_normalize(x3d, pose=None):
    """epropnp/common.py:103-127 (detach_transformation=True)."""
    offset = x3d.detach().mean(dim=-2)
    x3d_n = x3d - offset.unsqueeze(-2)
    pose_n = None
    if pose is not None:
        shift = (pose_to_rotmat(pose) @ offset.unsqueeze(-1)).squeeze(-1)
        pose_n = torch.cat((pose[..., :3] + shift, pose[..., 3:]), -1)
    return offset, x3d_n, pose_n


def pnp_denormalize(offset, pose_n):
    """epropnp/common.py:130-136."""
    shift = (pose_to_rotmat(pose_n) @ offset.unsqueeze(-1)).squeeze(-1)
    return torch.cat((pose_n[..., :3] - shift, pose_n[..., 3:]), -1)


# ----------------------------------------------------------------------------------------
# camera + robust cost   (one "evaluate" = project -> clamp -> Huber -> optional J, r)
# ----------------------------------------------------------------------------------------


class Cam:
    """Plain parameter bundle mirroring PerspectiveCamera's state (epropnp/camera.py:35-62)."""

    def __init__(self, cam_mats, z_min=0.1, lb=None, ub=None):
        self.cam_mats, self.z_min, self.lb, self.ub = cam_mats, z_min, lb, ub

    @staticmethod
    def from_img_shape(cam_mats, img_shape, z_min=0.1, allowed_border=200):
        # camera.py:57-59: lb scalar, ub = [w, h] - 0.5 + border
        lb = -0.5 - allowed_border
        ub = img_shape[..., [1, 0]] + (-0.5 + allowed_border)
        return Cam(cam_mats, z_min, lb, ub)


def _clamp_bounds(p, cam):
    """camera.py:81-93.  Returns clamped projection (out of place, autograd-friendly)."""
    lb, ub = cam.lb, cam.ub
    if lb is None or ub is None:
        return p, None, None
    lbt = lb.unsqueeze(-2) if isinstance(lb, torch.Tensor) else torch.as_tensor(lb, dtype=p.dtype)
    ubt = ub.unsqueeze(-2) if isinstance(ub, torch.Tensor) else torch.as_tensor(ub, dtype=p.dtype)
    p = torch.minimum(torch.maximum(lbt, p), ubt)
    return p, lbt, ubt


def evaluate(x3d, x2d, w2d, pose, cam, delta, want_cost=False, want_resjac=False,
             clip_jac=True, eps=1e-10):
    """epropnp/common.py:67-100 -> camera.py:64-143 -> cost_fun.py:33-89.

    Returns (residual (*,2N) | None, cost (*) | None, jacobian (*,2N,d) | None).
    Leading dims of pose may broadcast against the points (AMIS passes (s,B,p) vs (B,N,.)).
    """
    K = cam.cam_mats
    R = pose_to_rotmat(pose)
    t = pose[..., :3]
    if want_resjac:  # project_a, camera.py:10-18
        xr = x3d @ R.transpose(-1, -2)
        h = (xr + t.unsqueeze(-2)) @ K.transpose(-1, -2)
    else:            # project_b, camera.py:21-30
        h = x3d @ (K @ R).transpose(-1, -2) + (K @ t.unsqueeze(-1)).squeeze(-1).unsqueeze(-2)
    z = h[..., 2:3].clamp(min=cam.z_min)
    p = h[..., :2] / z
    p, lbt, ubt = _clamp_bounds(p, cam)

    if not isinstance(delta, torch.Tensor):
        delta = torch.as_tensor(delta, dtype=x2d.dtype)
    delta = delta[..., None]                      # cost_fun.py:50
    r = (p - x2d) * w2d                           # :52
    rho = r.norm(dim=-1)                          # :53
    cost = None
    if want_cost:                                 # :55-61, huber_kernel :8-12
        cost = torch.where(rho <= delta, 0.5 * rho * rho, delta * rho - 0.5 * delta * delta).sum(-1)
    if not want_resjac:
        return None, cost, None

    dof = 4 if pose.shape[-1] == 4 else 6
    # camera.py:111-143
    Kb = K.unsqueeze(-3)                          # (*,1,3,3)
    d_xy = Kb[..., :2, :2] / z.unsqueeze(-1)
    d_z = (Kb[..., :2, 2:3] - p.unsqueeze(-1)) / z.unsqueeze(-1)
    D = torch.cat((d_xy, d_z), -1)                # (*,N,2,3)
    if dof == 4:
        yaw_dir = torch.stack((xr[..., 2], -xr[..., 0]), -1).unsqueeze(-1)     # :113-114
        J = torch.cat((D, D[..., ::2] @ yaw_dir), -1)
    else:
        a = 2 * xr
        o = torch.zeros_like(a[..., 0])
        S = torch.stack((o, -a[..., 2], a[..., 1], a[..., 2], o, -a[..., 0], -a[..., 1], a[..., 0], o),
                        -1).reshape(a.shape[:-1] + (3, 3))                        # skew(2 x_rot), :116
        J = torch.cat((D, D @ S), -1)
    if clip_jac:                                  # :100-105 (mask is per image row)
        mask = (z == cam.z_min).expand_as(p)
        if lbt is not None:
            mask = mask | (p == lbt) | (p == ubt)
        J = J.masked_fill(mask.unsqueeze(-1), 0)
    # robust rescaling, cost_fun.py:63-84, huber_d_kernel :15-20 (no-grad form)
    gam = (delta / rho.clamp(min=eps)).clamp(max=1.0).sqrt()
    res = (r * gam.unsqueeze(-1)).flatten(-2)
    J = (J * (w2d * gam.unsqueeze(-1)).unsqueeze(-1)).flatten(-3, -2)
    return res, cost, J


def adaptive_huber_delta(x2d, w2d, relative_delta=0.5):
    """epropnp/cost_fun.py:123-126 (differentiable)."""
    x2d_std = torch.var(x2d, dim=-2).sum(dim=-1).sqrt()
    return w2d.mean(dim=(-2, -1)) * x2d_std * relative_delta


# ----------------------------------------------------------------------------------------
# LM / GN solver
# ----------------------------------------------------------------------------------------

LM_DEFAULTS = dict(num_iter=10, min_lm_diagonal=1e-6, max_lm_diagonal=1e32, min_relative_decrease=1e-3,
                   initial_trust_region_radius=30.0, max_trust_region_radius=1e16, eps=1e-5)


def _solve(A, b):
    return torch.linalg.solve(A, b)       # levenberg_marquardt.py:15-19 (LU with pivoting)


@torch.no_grad()
def lm_solve(x3d, x2d, w2d, cam, delta, pose_init, fast_mode=False, with_pose_cov=False,
             with_cost=False, **kw):
    """epropnp/levenberg_marquardt.py:80-241 with pose_init given (no init solver).
    Returns pose_opt, pose_cov|None, cost|None (and the accept history as a 4th value)."""
    p = dict(LM_DEFAULTS)
    p.update(kw)
    eps = p['eps']
    pose = pose_init.clone()
    dof = 4 if pose.shape[-1] == 4 else 6
    eye = torch.eye(dof, dtype=x2d.dtype)
    history = []
    if fast_mode:   # :136-152 (Gauss-Newton, clip_jac off)
        for _ in range(p['num_iter']):
            res, cost, jac = evaluate(x3d, x2d, w2d, pose, cam, delta, True, True, clip_jac=False)
            jt = jac.transpose(-1, -2)
            jtj = jt @ jac + eps * eye
            grad = jt @ res.unsqueeze(-1)
            step = -_solve(jtj, grad).squeeze(-1)
            pose = pose_add(pose, step)
    else:           # :154-181 + _lm_iter :192-241
        res, cost, jac = evaluate(x3d, x2d, w2d, pose, cam, delta, True, True)
        B = pose.shape[0]
        radius = x2d.new_full((B,), p['initial_trust_region_radius'])
        dec = x2d.new_full((B,), 2.0)
        for _ in range(p['num_iter']):
            jt = jac.transpose(-1, -2)
            jtj = jt @ jac
            diag = torch.diagonal(jtj, dim1=-2, dim2=-1)
            jtj_lm = jtj + torch.diag_embed(
                diag.clamp(min=p['min_lm_diagonal'], max=p['max_lm_diagonal']) / radius[:, None] + eps)
            grad = jt @ res.unsqueeze(-1)
            step_ = -_solve(jtj_lm, grad)
            pose_new = pose_add(pose, step_.squeeze(-1))
            res_n, cost_n, jac_n = evaluate(x3d, x2d, w2d, pose_new, cam, delta, True, True)
            model_change = -(step_.transpose(-1, -2) @ ((jtj @ step_) / 2 + grad)).flatten()
            rel = (cost - cost_n) / model_change
            ok = (rel >= p['min_relative_decrease']) & (model_change > 0.0)
            history.append(ok.clone())
            pose = torch.where(ok[:, None], pose_new, pose)
            shrink = (1.0 - (2.0 * rel - 1.0) ** 3).clamp(min=1.0 / 3.0)
            radius = torch.where(ok, radius / shrink, radius)
            radius = radius.clamp(max=p['max_trust_region_radius'], min=eps)
            radius = torch.where(ok, radius, radius / dec)        # reject path not re-clamped (:239)
            dec = torch.where(ok, torch.full_like(dec, 2.0), dec * 2.0)
            jac = torch.where(ok[:, None, None], jac_n, jac)
            res = torch.where(ok[:, None], res_n, res)
            cost = torch.where(ok, cost_n, cost)
        jtj = jac.transpose(-1, -2) @ jac + eps * eye
    pose_cov = torch.inverse(jtj) if with_pose_cov else None    # :178-179
    return pose, pose_cov, (cost if with_cost else None), history


def gn_step(x3d, x2d, w2d, pose, cam, delta, eps=1e-5):
    """epropnp/levenberg_marquardt.py:243-253 (differentiable; autograd does the backward)."""
    res, _, jac = evaluate(x3d, x2d, w2d, pose, cam, delta, False, True)
    jt = jac.transpose(-1, -2)
    dof = jac.shape[-1]
    jtj = jt @ jac + torch.eye(dof, dtype=jac.dtype) * eps
    return -_solve(jtj, jt @ res.unsqueeze(-1)).squeeze(-1)


# ----------------------------------------------------------------------------------------
# RSLM initialiser (randomness injected)
# ----------------------------------------------------------------------------------------


def center_based_init(x2d, x3d, K, dof, eps=1e-6):
    """epropnp/levenberg_marquardt.py:283-298."""
    x2dh = torch.cat((x2d, torch.ones_like(x2d[..., :1])), -1)
    x2dc = torch.linalg.solve(K, x2dh.transpose(-1, -2)).transpose(-1, -2)
    x2dc = x2dc[..., :2] / x2dc[..., 2:].clamp(min=eps)
    x2dc_std, x2dc_mean = torch.std_mean(x2dc, dim=-2)
    x3d_std = torch.std(x3d, dim=-2)
    one = torch.ones_like(x2dc_mean[..., :1])
    if dof == 4:
        scale = x3d_std[..., 1] / x2dc_std[..., 1].clamp(min=eps)
    else:
        scale = math.sqrt(2 / 3) * x3d_std.norm(dim=-1) / x2dc_std.norm(dim=-1).clamp(min=eps)
    return torch.cat((x2dc_mean, one), -1) * scale.unsqueeze(-1)


@torch.no_grad()
def rslm_solve(x3d, x2d, w2d, cam, delta, inds, rot_init, dof, fast_mode=False, **kw):
    """epropnp/levenberg_marquardt.py:300-353.
    inds: (P,B,n) int64 sub-sample indices (reference: torch.multinomial :306-308);
    rot_init: (P,B) yaw or (P,B,4) unit quaternion (reference: torch.rand/randn :318-326).
    Returns pose (B,p), min_cost (B,)."""
    P, B, n = inds.shape
    bidx = torch.arange(B)[None, :, None]
    x2d_s, x3d_s, w2d_s = x2d[bidx, inds], x3d[bidx, inds], w2d[bidx, inds]
    t0 = center_based_init(x2d, x3d, cam.cam_mats, dof)
    rot = rot_init.unsqueeze(-1) if dof == 4 else rot_init
    pose0 = torch.cat((t0.expand(P, B, 3), rot), -1)
    rep = lambda v: v.repeat((P,) + (1,) * (v.dim() - 1)) if isinstance(v, torch.Tensor) else v
    cam_r = Cam(rep(cam.cam_mats), cam.z_min, rep(cam.lb), rep(cam.ub))
    pose, _, _, _ = lm_solve(x3d_s.reshape(P * B, n, 3), x2d_s.reshape(P * B, n, 2), w2d_s.reshape(P * B, n, 2),
                             cam_r, rep(delta), pose0.reshape(P * B, -1), fast_mode=fast_mode, **kw)
    pose = pose.reshape(P, B, -1)
    cost = evaluate(x3d, x2d, w2d, pose, cam, delta, want_cost=True)[1]          # :344
    min_cost, idx = cost.min(dim=0)
    return pose[idx, torch.arange(B)], min_cost


@torch.no_grad()
def lm_solve_with_init(x3d, x2d, w2d, cam, delta, pose_init, cost_init, rslm_noise, rslm_kw, dof,
                       fast_mode=False, **kw):
    """Initialisation logic of LMSolver.solve, levenberg_marquardt.py:115-130."""
    pose_s, cost_s = rslm_solve(x3d, x2d, w2d, cam, delta, rslm_noise['inds'], rslm_noise['rot'], dof,
                                fast_mode=fast_mode, **rslm_kw)
    if pose_init is not None:
        if cost_init is None:
            cost_init = evaluate(x3d, x2d, w2d, pose_init, cam, delta, want_cost=True)[1]
        use_init = cost_init < cost_s
        pose_s = torch.where(use_init[:, None], pose_init, pose_s)
    return lm_solve(x3d, x2d, w2d, cam, delta, pose_s, fast_mode=fast_mode, **kw)


# ----------------------------------------------------------------------------------------
# proposal distributions
# ----------------------------------------------------------------------------------------


def chol_or_default(mat, default_diag=None):
    """epropnp/epropnp.py:16-33: Cholesky; matrices that fail -> diag(default_diag) or I."""
    L, info = torch.linalg.cholesky_ex(mat)
    n = mat.shape[-1]
    dflt = torch.diag(torch.as_tensor(default_diag, dtype=mat.dtype)) if default_diag is not None \
        else torch.eye(n, dtype=mat.dtype)
    bad = (info != 0) | ~torch.isfinite(L).all(-1).all(-1)
    return torch.where(bad[..., None, None], dflt, L)


def student_t_sample(loc, L, z, chi2, df=3.0):
    """pyro MultivariateStudentT.rsample: loc + L (z * rsqrt(chi2/df)); call sites epropnp.py:146,224,306."""
    y = z * torch.rsqrt(chi2 / df).unsqueeze(-1)
    return loc + (L @ y.unsqueeze(-1)).squeeze(-1)


def student_t_logprob(x, loc, L, df=3.0):
    """pyro MultivariateStudentT.log_prob (n = 3)."""
    n = L.shape[-1]
    d = torch.linalg.solve_triangular(L, (x - loc).unsqueeze(-1), upper=False).squeeze(-1)
    maha = (d * d).sum(-1)
    Z = (torch.diagonal(L, dim1=-2, dim2=-1).log().sum(-1) + 0.5 * n * math.log(df) + 0.5 * n * math.log(math.pi)
         + math.lgamma(0.5 * df) - math.lgamma(0.5 * (df + n)))
    return -0.5 * (df + n) * torch.log1p(maha / df) - Z


def acg_sample(L, g, eps=1e-6):
    """epropnp/distributions.py:42-52."""
    v = (L @ g.unsqueeze(-1)).squeeze(-1)
    nrm = v.norm(dim=-1)
    out = v / nrm.unsqueeze(-1)
    e0 = torch.zeros_like(out)
    e0[..., 0] = 1
    return torch.where((nrm < eps).unsqueeze(-1), e0, out)


def acg_logprob(x, L):
    """epropnp/distributions.py:32-40, q = 4: area = 2 pi^2."""
    q = L.shape[-1]
    d = torch.linalg.solve_triangular(L, x.unsqueeze(-1), upper=False).squeeze(-1)
    maha = (d * d).sum(-1)
    half_log_det = torch.diagonal(L, dim1=-2, dim2=-1).log().sum(-1)
    area = 2 * math.pi ** (0.5 * q) / math.gamma(0.5 * q)
    return maha.log() * (-q / 2) - half_log_det - math.log(area)


_I0_SMALL = (1.0, 3.5156229, 3.0899424, 1.2067492, 0.2659732, 0.360768e-1, 0.45813e-2)
_I0_LARGE = (0.39894228, 0.1328592e-1, 0.225319e-2, -0.157565e-2, 0.916281e-2, -0.2057706e-1,
             0.2635537e-1, -0.1647633e-1, 0.392377e-2)


def log_i0(x):
    """torch/distributions/von_mises.py:24-89: polynomial log I0 (Abramowitz-Stegun 9.8.1/9.8.2), split at 3.75."""
    def poly(y, c):
        r = torch.full_like(y, c[-1])
        for ck in reversed(c[:-1]):
            r = ck + y * r
        return r
    ys = (x / 3.75) ** 2
    small = poly(ys, _I0_SMALL).log()
    yl = 3.75 / x
    large = x - 0.5 * x.log() + poly(yl, _I0_LARGE).log()
    return torch.where(x < 3.75, small, large)


def vm_mix_logprob(x, loc, kappa, uniform_mix=0.25):
    """epropnp/distributions.py:74-79 on top of VonMises.log_prob (von_mises.py:144-152)."""
    vm = kappa * torch.cos(x - loc) - math.log(2 * math.pi) - log_i0(kappa) + math.log(1 - uniform_mix)
    return torch.logaddexp(vm, torch.full_like(vm, math.log(uniform_mix / (2 * math.pi))))


VM_MAX_TRIES = 16


def vm_sample_bounded(loc, kappa, u):
    """Best & Fisher (1979) von Mises sampler with a bounded number of attempts.
    loc, kappa: (...,); u: (..., VM_MAX_TRIES, 3) uniforms in [0,1).  First accepted attempt wins;
    if none is accepted (p < 1e-7) the last candidate is used.  Stands in for numpy.random.vonmises
    (epropnp/distributions.py:70-72)."""
    k = kappa.double().clamp(min=1e-12)
    tau = 1 + (1 + 4 * k * k).sqrt()
    rho = (tau - (2 * tau).sqrt()) / (2 * k)
    r = torch.where(k < 1e-5, 1 / k + k, (1 + rho * rho) / (2 * rho))
    u = u.double()
    x = torch.zeros_like(k)
    done = torch.zeros_like(k, dtype=torch.bool)
    for a in range(u.shape[-2]):
        u1, u2, u3 = u[..., a, 0], u[..., a, 1], u[..., a, 2]
        zc = torch.cos(math.pi * u1)
        f = (1 + r * zc) / (r + zc)
        c = k * (r - f)
        acc = ((c * (2 - c) - u2) > 0) | ((c / u2.clamp(min=1e-300)).log() + 1 - c >= 0)
        cand = torch.where(u3 - 0.5 >= 0, 1.0, -1.0) * torch.acos(f.clamp(-1, 1))
        take = (~done) & (acc | (a == u.shape[-2] - 1))
        x = torch.where(take, cand, x)
        done = done | acc
    out = torch.remainder(x + math.pi + loc.double(), 2 * math.pi) - math.pi
    return out.to(loc.dtype)


def vm_mix_sample(loc, kappa, u_uniform, u_vm, n_total, uniform_mix=0.25):
    """epropnp/distributions.py:61-72: first round(0.25 s) rows uniform on [-pi, pi), the rest von Mises.
    loc, kappa: (B,1); u_uniform: (n_u,B,1) in [0,1); u_vm: (n_v,B,1,T,3)."""
    n_u = round(n_total * uniform_mix)
    xs = (u_uniform[:n_u] * 2 - 1) * math.pi
    xv = vm_sample_bounded(loc.expand(u_vm.shape[:3]), kappa.expand(u_vm.shape[:3]), u_vm)
    return torch.cat((xs, xv), 0)


# ----------------------------------------------------------------------------------------
# AMIS
# ----------------------------------------------------------------------------------------


def _inv(m):
    return torch.inverse(m)


@torch.no_grad()
def initial_fit_6dof(pose_opt, pose_cov, acg_dispersion=0.001):
    """epropnp/epropnp.py:288-302."""
    L_t = chol_or_default(pose_cov[:, :3, :3])
    eye4 = torch.eye(4, dtype=pose_opt.dtype)
    T = quat_tangent_map(pose_opt[:, 3:])
    rot_cov = _inv(T @ _inv(pose_cov[:, 3:, 3:]) @ T.transpose(-1, -2) + eye4)
    rot_cov = rot_cov / torch.diagonal(rot_cov, dim1=-2, dim2=-1).sum(-1)[:, None, None]
    L_r = chol_or_default(rot_cov + torch.det(rot_cov)[:, None, None] ** 0.25 * (acg_dispersion * eye4))
    return pose_opt[:, :3].clone(), L_t, L_r


@torch.no_grad()
def estimate_6dof(samples, logw, eps=1e-5, acg_mle_iter=3, acg_dispersion=0.001):
    """epropnp/epropnp.py:317-342.  samples (M,B,7), logw (M,B)."""
    w = torch.softmax(logw, dim=0)
    mean = (w[..., None] * samples[..., :3]).sum(0)
    dev = samples[..., :3] - mean
    cov = (w[..., None, None] * dev.unsqueeze(-1) * dev.unsqueeze(-2)).sum(0)
    L_t = chol_or_default(cov)
    eye4 = torch.eye(4, dtype=samples.dtype)
    rot = samples[..., 3:]
    rrt = rot[..., :, None] * rot[..., None, :]
    rot_cov = eye4.expand(samples.shape[1], 4, 4).clone()
    for _ in range(acg_mle_iter):
        M = rot[:, :, None, :] @ _inv(rot_cov) @ rot[:, :, :, None]
        iw = w[..., None, None] / M.clamp(min=eps)
        iw = iw / iw.sum(0)
        rot_cov = (iw * rrt).sum(0) + eye4 * eps
    L_r = chol_or_default(rot_cov + torch.det(rot_cov)[:, None, None] ** 0.25 * (acg_dispersion * eye4))
    return mean, L_t, L_r


@torch.no_grad()
def initial_fit_4dof(pose_opt, pose_cov, eps=1e-5):
    """epropnp/epropnp.py:216-220."""
    L_t = chol_or_default(pose_cov[:, :3, :3], [1.0, 1.0, 4.0])
    kappa = 0.33 / pose_cov[:, 3, 3, None].clamp(min=eps)
    return pose_opt[:, :3].clone(), L_t, pose_opt[:, 3:].clone(), kappa


@torch.no_grad()
def estimate_4dof(samples, logw, eps=1e-5):
    """epropnp/epropnp.py:238-260."""
    w = torch.softmax(logw, dim=0)
    mean = (w[..., None] * samples[..., :3]).sum(0)
    dev = samples[..., :3] - mean
    cov = (w[..., None, None] * dev.unsqueeze(-1) * dev.unsqueeze(-2)).sum(0)
    L_t = chol_or_default(cov, [1.0, 1.0, 4.0])
    ms = (w[..., None] * samples[..., 3:].sin()).sum(0)
    mc = (w[..., None] * samples[..., 3:].cos()).sum(0)
    mode = torch.atan2(ms, mc)
    r_sq = ms * ms + mc * mc
    kappa = 0.33 * r_sq.sqrt().clamp(min=eps) * (2 - r_sq) / (1 - r_sq).clamp(min=eps)
    return mean, L_t, mode, kappa


def amis(x3d, x2d, w2d, cam, delta, pose_opt, pose_cov, noise, mc_samples=512, num_iter=4,
         eps=1e-5, acg_mle_iter=3, acg_dispersion=0.001, return_proposals=False):
    """The AMIS loop of epropnp/epropnp.py:132-182 for dof = 6 (pose length 7) or 4.

    noise (6-DoF): {'z': (K,s,B,3) N(0,1), 'chi2': (K,s,B) Chi2(3), 'g': (K,s,B,4) N(0,1)}
    noise (4-DoF): {'z','chi2', 'u': (K,n_u,B,1) U[0,1), 'vm': (K,n_v,B,1,T,3) U[0,1)}
    Returns pose_samples (S,B,p) [no grad], logweights (S,B) [grad flows through the cost only].
    """
    K_it = num_iter
    s = mc_samples // num_iter
    B = x3d.shape[0]
    six = pose_opt.shape[-1] == 7
    if six:
        mode0, Lt0, Lr0 = initial_fit_6dof(pose_opt, pose_cov, acg_dispersion)
        modes, Lts, Lrs = [mode0], [Lt0], [Lr0]
    else:
        mode0, Lt0, rm0, kap0 = initial_fit_4dof(pose_opt, pose_cov, eps)
        modes, Lts, rmodes, kappas = [mode0], [Lt0], [rm0], [kap0]

    def logq(j, smp):   # log density of proposal j at samples smp (...,B,p)
        lp = student_t_logprob(smp[..., :3], modes[j], Lts[j])
        if six:
            return lp + acg_logprob(smp[..., 3:], Lrs[j])
        return lp + vm_mix_logprob(smp[..., 3:], rmodes[j], kappas[j]).squeeze(-1)

    samples, costs = [], []
    logprobs = [[None] * K_it for _ in range(K_it)]      # [proposal j][sample block k]
    logw = None
    for i in range(K_it):
        with torch.no_grad():
            t = student_t_sample(modes[i], Lts[i], noise['z'][i], noise['chi2'][i])
            if six:
                r = acg_sample(Lrs[i], noise['g'][i])
            else:
                r = vm_mix_sample(rmodes[i], kappas[i], noise['u'][i], noise['vm'][i], s)
            smp = torch.cat((t, r), -1)                  # (s,B,p)
        samples.append(smp)
        costs.append(evaluate(x3d, x2d, w2d, smp, cam, delta, want_cost=True)[1])     # epropnp.py:151
        with torch.no_grad():
            for k in range(i + 1):                       # :156-157
                logprobs[i][k] = logq(i, samples[k])
            for j in range(i):                           # :158-163
                logprobs[j][i] = logq(j, smp)
            mix = torch.stack([torch.logsumexp(torch.stack([logprobs[j][k] for j in range(i + 1)], 0), 0)
                               for k in range(i + 1)], 0) - math.log(i + 1)      # :165
        logw = -torch.stack(costs, 0) - mix             # :169  (i+1,s,B)
        if i == K_it - 1:
            break
        with torch.no_grad():
            all_s = torch.cat(samples, 0)
            lw = logw.detach().reshape(-1, B)
            if six:
                m, Lt, Lr = estimate_6dof(all_s, lw, eps, acg_mle_iter, acg_dispersion)
                modes.append(m); Lts.append(Lt); Lrs.append(Lr)
            else:
                m, Lt, rm, kp = estimate_4dof(all_s, lw, eps)
                modes.append(m); Lts.append(Lt); rmodes.append(rm); kappas.append(kp)
    pose_samples = torch.cat(samples, 0)
    logweights = logw.reshape(mc_samples, B)
    if return_proposals:
        props = dict(mode=torch.stack(modes), L_t=torch.stack(Lts))
        if six:
            props['L_r'] = torch.stack(Lrs)
        else:
            props['rmode'] = torch.stack(rmodes); props['kappa'] = torch.stack(kappas)
        return pose_samples, logweights, props
    return pose_samples, logweights


def monte_carlo_forward(x3d, x2d, w2d, cam, delta, pose_init, noise, mc_samples=512, num_iter=4,
                        lm_kw=None, normalize=False, fast_mode=False, with_pose_opt_plus=False,
                        rslm_noise=None, rslm_kw=None, eps=1e-5):
    """EProPnPBase.monte_carlo_forward, epropnp/epropnp.py:87-196.
    rslm_noise None  <=> force_init_solve=False.  Returns the reference's 6-tuple."""
    lm_kw = dict(lm_kw or {})
    if normalize:
        offset, x3d, pose_init = pnp_normalize(x3d, pose_init)
    dof = 4 if pose_init.shape[-1] == 4 else 6
    cost_init = evaluate(x3d, x2d, w2d, pose_init, cam, delta, want_cost=True)[1]       # :121-124
    with torch.no_grad():
        xd, ud, wd = x3d.detach(), x2d.detach(), w2d.detach()
        dd = delta.detach() if isinstance(delta, torch.Tensor) else delta
        if rslm_noise is None:
            pose_opt, pose_cov, cost, _ = lm_solve(xd, ud, wd, cam, dd, pose_init, fast_mode=fast_mode,
                                                   with_pose_cov=True, with_cost=True, **lm_kw)
        else:
            pose_opt, pose_cov, cost, _ = lm_solve_with_init(
                xd, ud, wd, cam, dd, pose_init, cost_init.detach(), rslm_noise, dict(rslm_kw or {}), dof,
                fast_mode=fast_mode, with_pose_cov=True, with_cost=True, **lm_kw)
    pose_opt_plus = None
    if with_pose_opt_plus:
        pose_opt_plus = pose_add(pose_opt, gn_step(x3d, x2d, w2d, pose_opt, cam, delta, lm_kw.get('eps', 1e-5)))
    pose_samples, logw = amis(x3d, x2d, w2d, cam, delta, pose_opt, pose_cov, noise, mc_samples, num_iter, eps)
    if normalize:
        pose_opt = pnp_denormalize(offset, pose_opt)
        pose_samples = pnp_denormalize(offset, pose_samples)
        if pose_opt_plus is not None:
            pose_opt_plus = pnp_denormalize(offset, pose_opt_plus)
    return pose_opt, cost, pose_opt_plus, pose_samples, logw, cost_init


def mc_pose_loss(logweights, cost_target, norm_factor=1.0):
    """EPro-PnP-6DoF/lib/models/monte_carlo_pose_loss.py:28-33 without the EMA state."""
    loss = cost_target + torch.logsumexp(logweights, dim=0)
    loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)
    return loss.mean() / norm_factor


# ----------------------------------------------------------------------------------------
# synthetic workloads + noise
# ----------------------------------------------------------------------------------------


def make_problem(B, N, dof=6, seed=0, dtype=torch.float32, cam_kind='pinhole800', bounds=None,
                 relative_delta=0.5, noise_px=1.0):
    """Seeded synthetic correspondences: x3d ~ N(0,0.5^2); gt pose t~N(0,I), t_z += 5 (6-DoF) / 10 (4-DoF);
    x2d = project(gt) + N(0, noise_px); w2d = softmax_N(U(0,1)) * 2; pose_init = perturbed gt."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    x3d = rn(B, N, 3) * 0.5
    t = rn(B, 3)
    t[:, 2] += 5 if dof == 6 else 10
    if dof == 6:
        q = rn(B, 4)
        q = q / q.norm(dim=-1, keepdim=True)
        pose_gt = torch.cat((t, q), -1)
    else:
        pose_gt = torch.cat((t, rn(B, 1)), -1)
    if cam_kind == 'pinhole800':
        Km = torch.tensor([[800., 0, 320], [0, 800., 240], [0, 0, 1]], dtype=torch.float64)
        px = noise_px
    else:   # notebook: K = I
        Km = torch.eye(3, dtype=torch.float64)
        px = noise_px / 800.0
    K = Km.expand(B, 3, 3).contiguous()
    cam = Cam(K, 0.1)
    x2d = evaluate_project(x3d, pose_gt, cam) + rn(B, N, 2) * px
    w2d = torch.softmax(torch.rand(B, N, 2, generator=g, dtype=torch.float64), dim=1) * 2.0
    if dof == 6:
        qi = pose_gt[:, 3:] + 0.05 * rn(B, 4)
        pose_init = torch.cat((pose_gt[:, :3] + 0.1 * rn(B, 3), qi / qi.norm(dim=-1, keepdim=True)), -1)
    else:
        pose_init = torch.cat((pose_gt[:, :3] + 0.1 * rn(B, 3), pose_gt[:, 3:] + 0.05 * rn(B, 1)), -1)
    out = dict(x3d=x3d, x2d=x2d, w2d=w2d, cam_mats=K, pose_gt=pose_gt, pose_init=pose_init)
    out = {k: v.to(dtype) for k, v in out.items()}
    if bounds == 'tensor':
        out['lb'] = torch.tensor([-0.5 - 200, -0.5 - 200], dtype=dtype).expand(B, 2).contiguous()
        out['ub'] = torch.tensor([640 - 0.5 + 200, 480 - 0.5 + 200], dtype=dtype).expand(B, 2).contiguous()
    elif bounds == 'tight':   # makes some projections hit the bounds (exercises clip_jac)
        out['lb'] = torch.tensor([100.0, 80.0], dtype=dtype).expand(B, 2).contiguous()
        out['ub'] = torch.tensor([540.0, 400.0], dtype=dtype).expand(B, 2).contiguous()
    out['delta'] = adaptive_huber_delta(out['x2d'], out['w2d'], relative_delta)
    return out


def evaluate_project(x3d, pose, cam):
    """project_b only (camera.py:21-30), used to synthesise x2d."""
    K = cam.cam_mats
    R = pose_to_rotmat(pose)
    h = x3d @ (K @ R).transpose(-1, -2) + (K @ pose[..., :3, None]).squeeze(-1).unsqueeze(-2)
    return h[..., :2] / h[..., 2:3].clamp(min=cam.z_min)


def make_noise(B, mc_samples, num_iter, dof=6, seed=1, dtype=torch.float32):
    """Base random draws for one AMIS run (layout matches the reference's (s,B,.) sampling shape)."""
    g = torch.Generator().manual_seed(seed)
    s = mc_samples // num_iter
    z = torch.randn(num_iter, s, B, 3, generator=g, dtype=torch.float64)
    chi2 = (torch.randn(num_iter, s, B, 3, generator=g, dtype=torch.float64) ** 2).sum(-1)   # Chi2(3)
    out = dict(z=z, chi2=chi2)
    if dof == 6:
        out['g'] = torch.randn(num_iter, s, B, 4, generator=g, dtype=torch.float64)
    else:
        n_u = round(s * 0.25)
        out['u'] = torch.rand(num_iter, n_u, B, 1, generator=g, dtype=torch.float64)
        out['vm'] = torch.rand(num_iter, s - n_u, B, 1, VM_MAX_TRIES, 3, generator=g, dtype=torch.float64)
    return {k: v.to(dtype) for k, v in out.items()}


def run_mc(prob, noise, dof, mc_samples, num_iter, lm_iter, normalize=False, relative_delta=0.5,
           rslm_kw=None, rslm_noise=None, with_pose_opt_plus=False, fast_mode=False, dtype=None):
    """monte_carlo_forward + MC loss (mean over objects) + backward, on the restatement.
    Same contract as oracle/ref_runner.py:run_mc, so the two can be diffed key by key."""
    cvt = (lambda v: v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) if dtype is not None \
        else (lambda v: v)
    prob = {k: cvt(v) for k, v in prob.items()}
    noise = {k: cvt(v) for k, v in noise.items()}
    x3d, x2d, w2d = (prob[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    cam = Cam(prob['cam_mats'], float(prob.get('z_min', 0.1)), prob.get('lb'), prob.get('ub'))
    delta = adaptive_huber_delta(x2d.detach(), w2d, relative_delta)
    rn = None
    if rslm_noise is not None:
        rn = dict(inds=rslm_noise['inds'], rot=cvt(rslm_noise['rot']))
    pose_opt, cost, pose_opt_plus, pose_samples, logw, cost_init = monte_carlo_forward(
        x3d, x2d, w2d, cam, delta, prob['pose_init'], noise, mc_samples, num_iter,
        lm_kw=dict(num_iter=lm_iter), normalize=normalize, fast_mode=fast_mode,
        with_pose_opt_plus=with_pose_opt_plus, rslm_noise=rn, rslm_kw=rslm_kw)
    loss_obj = cost_init + torch.logsumexp(logw, dim=0)
    total = loss_obj.mean()
    if with_pose_opt_plus:
        total = total + 0.1 * (pose_opt_plus * torch.linspace(0.5, 1.5, pose_opt_plus.shape[-1],
                                                              dtype=pose_opt_plus.dtype)).sum(-1).mean()
    total.backward()
    res = dict(pose_opt=pose_opt, cost=cost, pose_samples=pose_samples, logweights=logw, cost_init=cost_init,
               loss_obj=loss_obj, delta=delta, gx3d=x3d.grad, gx2d=x2d.grad, gw2d=w2d.grad)
    if with_pose_opt_plus:
        res['pose_opt_plus'] = pose_opt_plus
    return {k: v.detach().clone() for k, v in res.items()}


def perturb_inputs(prob, seed, ulps=3):
    """The same problem with every input (x3d, x2d, w2d, pose_init) moved by a random whole number of ulps in
    [-ulps, ulps] (relative 2^-23 steps).  Rounding-level: what separates two correct fp32 implementations."""
    g = torch.Generator().manual_seed(seed)
    out = dict(prob)
    for k in ('x3d', 'x2d', 'w2d', 'pose_init'):
        v = prob[k]
        kk = torch.randint(-ulps, ulps + 1, v.shape, generator=g).to(v.dtype)
        out[k] = v * (1 + kk * 2.0 ** -23)
    return out


SPREAD_KEYS = {  # key -> (object dim, relative to the object's largest magnitude?)
    'pose_opt': (0, False), 'cost': (0, True), 'cost_init': (0, True), 'loss_obj': (0, False),
    'pose_cov': (0, True), 'gx3d': (0, True), 'gx2d': (0, True), 'gw2d': (0, True), 'pose_opt_plus': (0, False)}


def per_object_diff(a, b, key):
    """max |a - b| over everything but the object axis; relative to max |b| of the object for the keys marked so."""
    od, rel = SPREAD_KEYS[key]
    d = (a.double() - b.double()).abs().movedim(od, 0)
    d = d.reshape(d.shape[0], -1).amax(1)
    if rel:
        m = b.double().abs().movedim(od, 0)
        d = d / m.reshape(m.shape[0], -1).amax(1).clamp(min=1e-30)
    return d.float()


def rounding_spread(run, prob, base, trials=8, ulps=3, seed=1000, extra=()):
    """How far `run(problem)` (a fp32 implementation of the path) moves, per object, when its inputs move by a few
    ulps: max over `trials` perturbed runs (and over the already-computed results in `extra`, e.g. the fp64 run) of
    |output - base|.  Two correct fp32 implementations differ by rounding-level perturbations inside the algorithm,
    so this is the yardstick the parity tests add to the north-star bars: it is large exactly where the reference
    itself is ill-conditioned (trust-region accept/reject flips at convergence, flat LM valleys, the cond-1e5 4x4
    proposal fits) and ~0 elsewhere.  -> {key: (B,) tensor}"""
    keys = [k for k in SPREAD_KEYS if k in base and base[k] is not None]
    sp = {k: torch.zeros(base[k].shape[SPREAD_KEYS[k][0]]) for k in keys}
    outs = [run(perturb_inputs(prob, seed + t, ulps)) for t in range(trials)] + list(extra)
    for o in outs:
        for k in keys:
            if k in o and o[k] is not None:
                sp[k] = torch.maximum(sp[k], per_object_diff(o[k], base[k], k))
    return sp


def make_rslm_noise(prob, dof, num_points, num_proposals, seed=2):
    """Sub-sample indices (weighted, without replacement: levenberg_marquardt.py:305-308) and random initial
    rotations (:318-326) for the RSLM initialiser."""
    g = torch.Generator().manual_seed(seed)
    B, N, _ = prob['x2d'].shape
    mw = prob['w2d'].mean(-1).double().reshape(1, B, N).expand(num_proposals, -1, -1)
    inds = torch.multinomial(mw.reshape(-1, N), num_points, generator=g).reshape(num_proposals, B, num_points)
    if dof == 4:
        rot = torch.rand(num_proposals, B, generator=g, dtype=torch.float64) * (2 * math.pi)
    else:
        rot = torch.randn(num_proposals, B, 4, generator=g, dtype=torch.float64)
        rot = rot / rot.norm(dim=-1, keepdim=True)
    return dict(inds=inds, rot=rot.to(prob['x2d'].dtype))
