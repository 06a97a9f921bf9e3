"""Tensor-level wrappers over the C ABI (include/epropnp_hip.h) + the autograd node of the AMIS path.

Everything here runs on the current HIP stream of the input tensors' device; nothing synchronises.
"""
import ctypes as C
import os
import threading

import torch

from . import _hip

tune = _hip.tune        # EPROPNP_TUNE keys (tools / tests only)


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise TypeError(f'{name}: the HIP path computes in fp32, got {t.dtype}')
    _hip.check_device(t, name)
    if not t.requires_grad and t.is_contiguous():      # nothing to cut off, nothing to copy: the tensor itself (two dispatcher calls less)
        return t
    return t.detach().contiguous()


# ---- device-side status word (include/epropnp_hip.h: epropnp_problem.status) -----------------------------------------
ST_LM_NOT_SPD, ST_NONFINITE_POSE, ST_CHOL_FALLBACK, ST_NONFINITE_WEIGHT, ST_SPLIT_TIMEOUT = 1, 2, 4, 8, 16
_status = threading.local()


def status_buffer(device):
    """The int32[2] status word kernels report numerical events into on `device`, or None when checking is off."""
    bufs = getattr(_status, 'bufs', None)
    if bufs is None:
        return None
    key = str(device)
    if key not in bufs:
        bufs[key] = torch.tensor([0, 2 ** 31 - 1], dtype=torch.int32, device=device)
    return bufs[key]


def flush_status():
    """Synchronise the current device and raise a pending numerical event (singular normal equations, non-finite pose) now."""
    _hip.flush_status()


class numerics_check:
    """Context manager: kernels launched inside record numerical events in a device-side status word (no host sync on
    the way); leaving the block -- or calling .check() -- synchronises once and raises what the reference would have
    raised where the event happened:

        with hip.numerics_check():
            pose_opt, *_ = solver(x3d, x2d, w2d, camera, cost_fun, pose_init=p0)

    Outside such a block the same two RuntimeErrors are still raised, asynchronously: kernels report into the library's
    host-mapped status word and every entry into the package polls it (`_hip.poll_status`, `flush_status()`).

    * a damped normal-equation system without a Cholesky factor (singular or NaN input) or a non-finite pose
      -> RuntimeError, as `torch.linalg.solve` / `torch.inverse` raise in levenberg_marquardt.py:15-19,178-181;
    * `strict=True` also reports what the reference tolerates silently: a proposal covariance replaced by its default
      (cholesky_wrapper, epropnp.py:16-33) and non-finite AMIS log-weights."""

    def __init__(self, strict=False):
        self.strict = strict

    @staticmethod
    def active():
        """True inside a `with numerics_check():` block of this thread."""
        return getattr(_status, 'bufs', None) is not None

    def __enter__(self):
        self.prev = getattr(_status, 'bufs', None)
        _status.bufs = {}
        return self

    def check(self):
        events = []
        for key, buf in _status.bufs.items():
            flags, first = buf.tolist()            # the one synchronisation
            buf.copy_(torch.tensor([0, 2 ** 31 - 1], dtype=torch.int32))
            if flags & ST_SPLIT_TIMEOUT:           # a performance event (a split kernel recomputed a sibling's share), never an error
                _hip._warn_split_degraded(first, key)
            if flags & ST_LM_NOT_SPD:
                events.append(f'linalg.solve: the damped normal equations of object {first} on {key} are singular or not finite')
            elif flags & ST_NONFINITE_POSE:
                events.append(f'the solver produced a non-finite pose for object {first} on {key}')
            if self.strict and flags & ST_CHOL_FALLBACK:
                events.append(f'cholesky: a proposal covariance of object {first} on {key} was replaced by its default')
            if self.strict and flags & ST_NONFINITE_WEIGHT:
                events.append(f'non-finite AMIS log-weight for object {first} on {key}')
        if events:
            raise RuntimeError('; '.join(events))

    def __exit__(self, exc_type, exc, tb):
        try:
            if exc_type is None:
                self.check()
        finally:
            _status.bufs = self.prev
        return False


class PnPProblem:
    """Contiguous fp32 device views of one batch of correspondences + camera + Huber threshold.

    Built from the duck-typed `camera` / `cost_fun` objects of the reference API:
    camera.cam_mats (*,3,3), camera.z_min, camera.lb / camera.ub (None | float | (*,2) tensor),
    cost_fun.delta (float | (*,) tensor).
    """

    def __init__(self, x3d, x2d, w2d, camera, cost_fun, dof):
        assert x3d.dim() == x2d.dim() == w2d.dim() == 3, 'x3d/x2d/w2d must be (num_obj, num_pts, C)'
        B, N, _ = x2d.shape
        self.B, self.N, self.dof = B, N, dof
        self.pose_len = 7 if dof == 6 else 4
        self.x3d, self.x2d, self.w2d = _f32c(x3d, 'x3d'), _f32c(x2d, 'x2d'), _f32c(w2d, 'w2d')
        dev = self.x2d.device
        self.device = dev
        self.huber_eps = float(getattr(cost_fun, 'eps', 1e-10))
        self.z_min = float(camera.z_min)
        if not self.z_min >= 0.0:
            raise ValueError(f'camera.z_min must be >= 0 (a depth clamp), got {self.z_min}')
        # contiguous (B,3,3) intrinsics and (B,2) bounds: callers pass an expanded view of one matrix / plain floats, and
        # a step builds several PnPProblems from the same camera object -- materialise them once per camera state
        srcs = (camera.cam_mats, camera.lb, camera.ub)
        vers = tuple(v._version if isinstance(v, torch.Tensor) else None for v in srcs)
        cached = getattr(camera, '_hip_views', None)
        # hit = the very same source objects (held alive by the cache, so their storage cannot be recycled), unmodified
        hit = (cached is not None and cached[0] == (B, dev) and cached[2] == vers
               and all((a is b) or (not isinstance(a, torch.Tensor) and a == b) for a, b in zip(cached[1], srcs)))
        if not hit:
            cam_c = _f32c(camera.cam_mats.to(dev).expand(B, 3, 3), 'cam_mats')
            lb, ub = camera.lb, camera.ub
            if lb is not None and ub is not None:
                lb_c, ub_c = self._bound(lb, B, dev), self._bound(ub, B, dev)
            else:
                lb_c = ub_c = None
            cached = ((B, dev), srcs, vers, (cam_c, lb_c, ub_c))
            try:
                camera._hip_views = cached
            except AttributeError:      # objects without a __dict__: just do not cache
                pass
        self.cam, self.lb, self.ub = cached[3]
        delta = cost_fun.delta
        if not isinstance(delta, torch.Tensor):
            delta = torch.full((B,), float(delta), dtype=torch.float32, device=dev)
        self.delta = _f32c(delta.to(dev).expand(B) if delta.dim() <= 1 else delta.reshape(B), 'delta')
        self.status = status_buffer(dev)          # None unless numerics checking is on (epropnp.status)
        if self.status is None:
            _hip.poll_status()                    # default: the library's host-mapped word, polled on entry (no sync)
        self.c = _hip.Problem(_hip.ptr(self.x3d), _hip.ptr(self.x2d), _hip.ptr(self.w2d), _hip.ptr(self.cam),
                              _hip.ptr(self.lb), _hip.ptr(self.ub), _hip.ptr(self.delta), self.z_min, B, N, dof,
                              self.huber_eps, _hip.ptr(self.status))
        self.stream = _hip.stream_of(self.x2d)
        self.delta_fold = None

    def fold_delta(self, stats, relative_delta):
        """`delta` is AdaptiveHuberPnPCost's threshold of THIS w2d (stats (B,4), relative_delta of that set_param): the
        backward entry points add its gradient to grad_w2d themselves (include/epropnp_hip.h: epropnp_problem.delta_stats) and
        the autograd nodes built on this problem return no gradient for delta."""
        self.delta_fold = (_f32c(stats, 'delta_stats'), float(relative_delta))
        self.c.delta_stats, self.c.delta_relative = self.delta_fold[0].data_ptr(), self.delta_fold[1]
        return self

    @staticmethod
    def _bound(v, B, dev):
        if isinstance(v, torch.Tensor):
            return _f32c(v.to(device=dev, dtype=torch.float32).expand(B, 2), 'bound')
        return torch.full((B, 2), float(v), dtype=torch.float32, device=dev)

    def new(self, *shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def with_points(self, x3d):
        """The same problem over other (contiguous fp32) 3D points -- the centred ones of pnp_normalize."""
        import copy
        other = copy.copy(self)
        other.x3d = x3d
        other.c = _hip.Problem.from_buffer_copy(self.c)
        other.c.x3d = x3d.data_ptr()
        return other


def _guard_inputs(ctx, prob):
    """Remember the version counters of the tensors the recompute backward will read again (the nodes keep raw device
    pointers, not autograd-saved tensors, so autograd's own in-place check does not see them)."""
    ctx.guarded = [(t, t._version) for t in (prob.x3d, prob.x2d, prob.w2d, prob.delta, prob.cam)]


def _check_inputs(ctx):
    for t, v in ctx.guarded:
        if t._version != v:
            raise RuntimeError('one of the variables needed for gradient computation has been modified by an inplace '
                               'operation (EPro-PnP recomputes its backward from x3d / x2d / w2d / delta / cam_mats: they '
                               'must stay unchanged between forward and backward)')


_shared = threading.local()


class share_problem:
    """For the duration of one `monte_carlo_forward`: the solver calls underneath (`LMSolver.solve`, the RSLM initialiser,
    `pose_opt_plus`) receive the very same (x3d, x2d, w2d, camera, cost_fun) objects, so they reuse its PnPProblem
    instead of re-deriving the contiguous views and the C struct (3-4 builds per step on the launch-bound shapes)."""

    def __init__(self, prob, x3d, x2d, w2d, camera, cost_fun):
        self.entry = (x3d, x2d, w2d, camera, cost_fun, cost_fun.delta, prob)

    def __enter__(self):
        self.prev = getattr(_shared, 'entry', None)
        _shared.entry = self.entry if self.entry[6] is not None else None
        return self

    def __exit__(self, *exc):
        _shared.entry = self.prev
        return False


def problem(x3d, x2d, w2d, camera, cost_fun, dof):
    """PnPProblem for these objects: the one shared by the enclosing monte_carlo_forward if they are the same objects."""
    e = getattr(_shared, 'entry', None)
    if e is not None and e[0] is x3d and e[1] is x2d and e[2] is w2d and e[3] is camera and e[4] is cost_fun \
            and e[5] is cost_fun.delta and e[6].dof == dof:
        return e[6]
    return PnPProblem(x3d, x2d, w2d, camera, cost_fun, dof)


def evaluate_cost(prob, poses):
    """poses (P,B,pose_len) or (B,pose_len)  ->  Huber cost (P,B) / (B,)."""
    squeeze = poses.dim() == 2
    ps = _f32c(poses.unsqueeze(0) if squeeze else poses, 'poses')
    P = ps.shape[0]
    assert ps.shape[1:] == (prob.B, prob.pose_len), f'poses shape {tuple(poses.shape)}'
    cost = prob.new(P, prob.B)
    _hip.call('epropnp_evaluate_cost', C.byref(prob.c), _hip.ptr(ps), P, _hip.ptr(cost), prob.stream)
    return cost[0] if squeeze else cost


def pose_cam_grad(prob, poses, weights, m_pose=-1, want_cam=True):
    """Gradients of sum_j weights[j] * cost(poses[j]) w.r.t. pose number `m_pose` (or None for -1) and w.r.t. the camera
    intrinsics (or None): what autograd records in the reference for `cost_init = evaluate_pnp(pose=pose_init,
    out_cost=True)` and for the AMIS log-weights w.r.t. pose_init and camera.cam_mats (epropnp.py:121-124,139-169).

    poses (P,B,pose_len), weights (P,B) -> (d/d poses[m_pose] (B,pose_len) | None, d/d cam_mats (B,3,3) | None).
    One sweep kernel (`epropnp_cost_pose_cam_grad`) reduces d/dK and M = a_m sum_n (d cost / d h_n)(X_n, 1)^T per object;
    with h = K (R X + t): d/dt = K^T M[:,3], d/dR = K^T M[:,:3], and d/dq (d/dyaw) through the derivative of the rotation
    the reference builds when the pose requires grad (common.py:30-36: the `1 - 2 (j^2 + k^2)` form, quaternion not
    normalised; :44-61 for yaw)."""
    ps = _f32c(poses, 'poses')
    P = ps.shape[0]
    assert ps.shape[1:] == (prob.B, prob.pose_len)
    wt = None if weights is None else _f32c(weights, 'weights')
    M = prob.new(prob.B, 3, 4) if m_pose >= 0 else None
    gk = prob.new(prob.B, 3, 3) if want_cam else None
    _hip.call('epropnp_cost_pose_cam_grad', C.byref(prob.c), _hip.ptr(ps), _hip.ptr(wt), P, int(m_pose), _hip.ptr(M),
              _hip.ptr(gk), prob.stream)
    if M is None:
        return None, gk
    K, pm = prob.cam, ps[m_pose]
    g_t = torch.einsum('bij,bi->bj', K, M[:, :, 3])                   # K^T M[:,3]
    G_R = torch.einsum('bij,bik->bjk', K, M[:, :, :3]).reshape(-1, 9)     # K^T M[:,:3]
    if prob.dof == 6:
        w, i, j, k = pm[:, 3], pm[:, 4], pm[:, 5], pm[:, 6]
        o = torch.zeros_like(w)
        dR = torch.stack((                                        # d R / d (w, i, j, k), each row-major (B,9)
            torch.stack((o, -k, j, k, o, -i, -j, i, o), -1),
            torch.stack((o, j, k, j, -2 * i, -w, k, w, -2 * i), -1),
            torch.stack((-2 * j, i, w, i, o, k, -w, k, -2 * j), -1),
            torch.stack((-2 * k, -w, i, w, -2 * k, j, i, j, o), -1)), 1) * 2          # (B,4,9)
        g_rot = torch.einsum('bqn,bn->bq', dR, G_R)
    else:
        yaw = pm[:, 3]
        c, s_ = torch.cos(yaw), torch.sin(yaw)
        o = torch.zeros_like(yaw)
        dR = torch.stack((-s_, o, c, o, o, o, -c, o, -s_), -1)                        # (B,9)
        g_rot = (dR * G_R).sum(-1, keepdim=True)
    return torch.cat((g_t, g_rot), -1), gk


class _PoseCamGrad(torch.autograd.Function):
    """Gradient-only node: contributes 0 to its consumer in the forward and, in the backward, the gradient of
    sum_j g[j] * sign * cost(poses[j]) w.r.t. the pose `m_pose` and the camera intrinsics -- the two inputs of the
    reference's evaluate_pnp that the kernel nodes (which differentiate w.r.t. the correspondences) do not cover."""

    @staticmethod
    def forward(ctx, pose_m, cam_mats, poses, prob, sign, m_pose):
        ctx.prob, ctx.poses, ctx.sign, ctx.m_pose = prob, poses.detach(), float(sign), int(m_pose)
        ctx.cam_shape = None if cam_mats is None else cam_mats.shape
        _guard_inputs(ctx, prob)
        return prob.new(poses.shape[0], prob.B).fill_(0.0)      # (fill_: a kernel; zero_() is a memset node under capture)

    @staticmethod
    def backward(ctx, g):
        _check_inputs(ctx)
        want_pose = ctx.needs_input_grad[0] and ctx.m_pose >= 0
        want_cam = ctx.needs_input_grad[1]
        gp, gk = pose_cam_grad(ctx.prob, ctx.poses, g.to(torch.float32) * ctx.sign, ctx.m_pose if want_pose else -1, want_cam)
        if gk is not None and tuple(ctx.cam_shape) != tuple(gk.shape):     # cam_mats broadcast over objects: (3,3) / (1,3,3)
            gk = gk.sum_to_size(ctx.cam_shape)
        return gp, gk, None, None, None, None


def pose_cam_grad_term(prob, poses, pose_m, cam_mats, sign, m_pose):
    """(P,B) zeros whose backward feeds d/d pose_m and d/d cam_mats of sum_j g[j] * sign * cost(poses[j])."""
    pm = pose_m if (pose_m is not None and pose_m.requires_grad) else None
    cm = cam_mats if (isinstance(cam_mats, torch.Tensor) and cam_mats.requires_grad) else None
    return _PoseCamGrad.apply(pm, cm, poses, prob, sign, m_pose if pm is not None else -1)


def normal_equations(prob, pose, clip_jac=True):
    """pose (B,pose_len) -> JtJ (B,d,d), Jtr (B,d), cost (B,) of the Huber-rescaled residual/Jacobian."""
    ps = _f32c(pose, 'pose')
    d = prob.dof
    jtj, jtr, cost = prob.new(prob.B, d, d), prob.new(prob.B, d), prob.new(prob.B)
    _hip.call('epropnp_normal_equations', C.byref(prob.c), _hip.ptr(ps), int(bool(clip_jac)), _hip.ptr(jtj),
              _hip.ptr(jtr), _hip.ptr(cost), prob.stream)
    return jtj, jtr, cost


def lm_solve(prob, pose_init, num_iter, fast_mode=False, with_pose_cov=False, with_cost=False, with_accepts=False,
             min_lm_diagonal=1e-6, max_lm_diagonal=1e32, min_relative_decrease=1e-3,
             initial_trust_region_radius=30.0, max_trust_region_radius=1e16, eps=1e-5):
    ps = _f32c(pose_init, 'pose_init')
    assert ps.shape == (prob.B, prob.pose_len)
    d = prob.dof
    pose_opt = prob.new(prob.B, prob.pose_len)
    cov = prob.new(prob.B, d, d) if with_pose_cov else None
    cost = prob.new(prob.B) if with_cost else None
    acc = prob.new(prob.B, dtype=torch.int32) if with_accepts else None
    par = _hip.LmParams(int(num_iter), int(bool(fast_mode)), min_lm_diagonal, max_lm_diagonal, min_relative_decrease,
                        initial_trust_region_radius, max_trust_region_radius, eps)
    scratch = lm_split_scratch(prob, par)
    _hip.call('epropnp_lm_solve', C.byref(prob.c), C.byref(par), _hip.ptr(ps), _hip.ptr(pose_opt), _hip.ptr(cov),
              _hip.ptr(cost), _hip.ptr(acc), _hip.ptr(scratch), 0 if scratch is None else scratch.numel() * 4, prob.stream)
    if with_accepts:
        return pose_opt, cov, cost, acc
    return pose_opt, cov, cost


def lm_split_scratch(prob, lm_par):
    """Scratch with which lm_solve deals the points of an object to several workgroups (few objects x many points), or None
    when the library would not; from torch's caching allocator."""
    if prob.B > 128:
        return None
    nbytes = int(_hip.lib().epropnp_lm_solve_split_bytes(C.byref(prob.c), C.byref(lm_par)))
    return prob.new((nbytes + 3) // 4) if nbytes > 0 else None


def noise_stride(dof):
    return _hip.lib().epropnp_noise_stride(dof)


def split_scratch(prob, mc_samples, num_iter):
    """Scratch for the AMIS forward's split over workgroups (few objects: csrc/amis_forward_mfma.hip), or None when the
    library would not split this problem.  From torch's caching allocator: no HIP allocation on the step's path, and inside
    a hipGraph capture no alloc / free nodes (which cost more than the split saves)."""
    if prob.B > 64:
        return None
    nbytes = int(_hip.lib().epropnp_amis_forward_split_bytes(C.byref(prob.c), int(mc_samples), int(num_iter)))
    return prob.new((nbytes + 3) // 4) if nbytes > 0 else None


def split_scratch_words(prob, lm_par, mc_samples, num_iter):
    """-> (words of the split LM solve's exchange scratch, words of the split forward's), 0 where the library would not split"""
    lm_bytes = 0 if prob.B > 128 else int(_hip.lib().epropnp_lm_solve_split_bytes(C.byref(prob.c), C.byref(lm_par)))
    fw_bytes = 0 if prob.B > 64 else int(_hip.lib().epropnp_amis_forward_split_bytes(C.byref(prob.c), int(mc_samples), int(num_iter)))
    return (lm_bytes + 3) // 4, (fw_bytes + 3) // 4


_NO_SCRATCH = object()


def _amis_struct(prob, S, K, eps, acg_mle_iter, acg_dispersion, seed, offset, offset_dev, scratch=_NO_SCRATCH):
    """-> (epropnp_amis_params, the scratch tensor to keep alive until the launch is enqueued)"""
    if scratch is _NO_SCRATCH:
        scratch = split_scratch(prob, S, K)
    return _hip.AmisParams(int(S), int(K), eps, int(acg_mle_iter), acg_dispersion, int(seed), int(offset), _hip.ptr(offset_dev),
                           _hip.ptr(scratch), 0 if scratch is None else scratch.numel() * 4), scratch


def amis_forward(prob, pose_opt, pose_cov, mc_samples, num_iter, eps=1e-5, acg_mle_iter=3, acg_dispersion=0.001,
                 noise=None, seed=0, offset=0, with_proposals=False, offset_dev=None):
    """-> pose_samples (S,B,pose_len), logweights (S,B) [, proposals (B,K,40)]."""
    po, pc = _f32c(pose_opt, 'pose_opt'), _f32c(pose_cov, 'pose_cov')
    S, B = int(mc_samples), prob.B
    samples, logw = prob.new(S, B, prob.pose_len), prob.new(S, B)
    props = prob.new(B, num_iter, 40) if with_proposals else None
    nz = None
    if noise is not None:
        nz = _f32c(noise, 'noise')
        assert nz.shape == (B, num_iter, S // num_iter, noise_stride(prob.dof)), f'noise shape {tuple(nz.shape)}'
    par, scratch = _amis_struct(prob, S, num_iter, eps, acg_mle_iter, acg_dispersion, seed, offset, offset_dev)
    _hip.call('epropnp_amis_forward', C.byref(prob.c), C.byref(par), _hip.ptr(po), _hip.ptr(pc), _hip.ptr(nz),
              _hip.ptr(samples), _hip.ptr(logw), _hip.ptr(props), prob.stream)
    del scratch          # stream-ordered reuse by the caching allocator: the launch is enqueued
    return (samples, logw, props) if with_proposals else (samples, logw)


BWD_SPLIT_MAX_SAMPLES = 2600     # the split kernel needs the LDS-resident pose table (csrc/amis_backward_mfma.hip)


def backward_split(B, N, S):
    """Workgroups per object for the backward sweep.  One object's S x N point-poses occupy a single CU (~70 us at
    512 x 512); below ~256 objects the point chunks of an object are dealt to several workgroups so the whole chip works
    (B = 32, N = 512: 8 workgroups of 64 points).  1 = the ordinary kernel."""
    env = os.environ.get('EPROPNP_BWD_SPLIT')
    if env is not None:
        return max(1, int(env))
    if S > BWD_SPLIT_MAX_SAMPLES or B >= 256 or (_hip.tune('bwd_impl') or '')[:1] == 'v':
        return 1
    n = 1
    while n < 8 and 2 * n * B <= 512 and 2 * n * 64 <= N:
        n *= 2
    if n == 8 and N >= 2048 and 16 * B <= 512:      # dense crops (32 x 4096): two workgroups per CU, 108.6 -> 93.4 us
        n = 16
    return n


def amis_backward(prob, pose_samples, grad_logweights, pose_init=None, grad_cost_init=None, nsplit=None, cstruct=None):
    """-> grad_x3d (B,N,3), grad_x2d (B,N,2), grad_w2d (B,N,2), grad_delta (B,).
    `cstruct`: a C problem struct to use instead of prob.c (same shapes; the fused path's centred points)."""
    B, N = prob.B, prob.N
    cs = prob.c if cstruct is None else cstruct
    S = 0 if pose_samples is None else pose_samples.shape[0]
    smp = None if S == 0 else _f32c(pose_samples, 'pose_samples')
    glw = None if S == 0 else _f32c(grad_logweights, 'grad_logweights')
    pin = gin = None
    if pose_init is not None and grad_cost_init is not None:
        pin, gin = _f32c(pose_init, 'pose_init'), _f32c(grad_cost_init, 'grad_cost_init')
    gx3d, gx2d, gw2d = prob.new(B, N, 3), prob.new(B, N, 2), prob.new(B, N, 2)
    nsplit = backward_split(B, N, S) if nsplit is None else int(nsplit)
    if nsplit > 1:
        parts = prob.new(B, nsplit)
        _hip.call('epropnp_amis_backward_split', C.byref(cs), _hip.ptr(smp), _hip.ptr(glw), S, _hip.ptr(pin),
                  _hip.ptr(gin), nsplit, _hip.ptr(gx3d), _hip.ptr(gx2d), _hip.ptr(gw2d), _hip.ptr(parts), prob.stream)
        return gx3d, gx2d, gw2d, parts.sum(dim=1)
    gdel = prob.new(B)
    _hip.call('epropnp_amis_backward', C.byref(cs), _hip.ptr(smp), _hip.ptr(glw), S, _hip.ptr(pin), _hip.ptr(gin),
              _hip.ptr(gx3d), _hip.ptr(gx2d), _hip.ptr(gw2d), _hip.ptr(gdel), prob.stream)
    return gx3d, gx2d, gw2d, gdel


class _MonteCarloCost(torch.autograd.Function):
    """Differentiable part of monte_carlo_forward: (x3d, x2d, w2d, delta) -> (logweights, cost_init).

    Forward runs the AMIS kernel (samples are drawn, never differentiated); backward recomputes the cost sweep
    (nothing but the pose samples is kept alive, vs ~12 MB/object of autograd state in the reference).
    """

    @staticmethod
    def forward(ctx, x3d, x2d, w2d, delta, prob, pose_opt, pose_cov, pose_init, cost_init_value, cfg):
        samples, logw = amis_forward(prob, pose_opt, pose_cov, **cfg)
        ctx.set_materialize_grads(False)      # no (S,B,7) zero tensor for the non-differentiable samples output
        ctx.prob, ctx.pose_init = prob, pose_init
        _guard_inputs(ctx, prob)
        ctx.save_for_backward(samples)
        ctx.mark_non_differentiable(samples)
        ctx.delta_shape = delta.shape if isinstance(delta, torch.Tensor) else None
        if cost_init_value is None:
            return samples, logw
        return samples, logw, cost_init_value.clone()

    @staticmethod
    def backward(ctx, _g_samples, g_logw, g_cost_init=None):
        (samples,) = ctx.saved_tensors
        prob = ctx.prob
        if g_logw is None and g_cost_init is None:
            return (None,) * 10
        _check_inputs(ctx)
        if g_logw is None:
            g_logw = torch.full(samples.shape[:2], 0.0, dtype=torch.float32, device=samples.device)
        gx3d, gx2d, gw2d, gdel = amis_backward(prob, samples, g_logw, ctx.pose_init, g_cost_init)
        gdelta = None
        if ctx.delta_shape is not None and ctx.needs_input_grad[3]:
            gdelta = gdel.sum() if len(ctx.delta_shape) == 0 else gdel.reshape(ctx.delta_shape)
        return (gx3d if ctx.needs_input_grad[0] else None, gx2d if ctx.needs_input_grad[1] else None,
                gw2d if ctx.needs_input_grad[2] else None, gdelta, None, None, None, None, None, None)


def monte_carlo_cost(x3d, x2d, w2d, delta, prob, pose_opt, pose_cov, pose_init, cost_init_value, cfg):
    return _MonteCarloCost.apply(x3d, x2d, w2d, delta, prob, pose_opt, pose_cov, pose_init, cost_init_value, cfg)


def _lm_struct(solver, fast_mode):
    return _hip.LmParams(int(solver.num_iter), int(bool(fast_mode)), solver.min_lm_diagonal, solver.max_lm_diagonal,
                         solver.min_relative_decrease, solver.initial_trust_region_radius,
                         solver.max_trust_region_radius, solver.eps)


class _FusedMonteCarlo(torch.autograd.Function):
    """monte_carlo_forward as ONE autograd node over ONE host call (epropnp_monte_carlo_forward, csrc/mc_forward.hip):
    [normalize] -> cost_init -> [RSLM, cheaper-of-two start] -> LM -> AMIS -> [denormalize] are enqueued from C++;
    (x3d, x2d, w2d, delta) -> (pose_opt, cost, pose_samples, logweights, cost_init), the last two differentiable.
    Backward: the same recompute kernel as _MonteCarloCost, in the solver's (normalised) frame."""

    @staticmethod
    def forward(ctx, x3d, x2d, w2d, delta, prob, pose_init, par, noise, with_cost):
        B, N, PL, d = prob.B, prob.N, prob.pose_len, prob.dof
        S = par.amis.mc_samples
        new = prob.new
        pin = None if pose_init is None else _f32c(pose_init, 'pose_init')
        nz = None
        if noise is not None:
            nz = _f32c(noise, 'noise')
            assert nz.shape == (B, par.amis.num_iter, S // par.amis.num_iter, noise_stride(d)), f'noise shape {tuple(nz.shape)}'
        normalize = bool(par.normalize)
        x3d_c = new(B, N, 3) if normalize else None
        offset = new(B, 3) if normalize else None
        pin_n = new(B, PL) if (normalize and pin is not None) else None
        start_pose, start_cost = (new(B, PL), new(B)) if par.init_mode else (None, None)
        pose_opt_n, pose_cov, samples_n, logw = new(B, PL), new(B, d, d), new(S, B, PL), new(S, B)
        cost = new(B) if with_cost else None
        cost_init = new(B) if pin is not None else None
        pose_opt, samples = (new(B, PL), new(S, B, PL)) if normalize else (None, None)     # caller's frame
        p = _hip.ptr
        _hip.call('epropnp_monte_carlo_forward', C.byref(prob.c), C.byref(par), p(pin), p(nz), p(x3d_c), p(offset), p(pin_n),
                  p(start_pose), p(start_cost), p(pose_opt_n), p(pose_cov), p(cost), p(samples_n), p(logw), p(cost_init),
                  p(pose_opt), p(samples), prob.stream)
        if normalize:      # the backward differentiates the cost in the solver frame: same problem, centred points
            bprob = _hip.Problem.from_buffer_copy(prob.c)       # (prob.fold_delta travels with the copy)
            bprob.x3d = x3d_c.data_ptr()
        else:
            bprob = prob.c
        ctx.set_materialize_grads(False)
        ctx.prob, ctx.bprob, ctx.keep = prob, bprob, (x3d_c, pin_n if normalize else pin)
        _guard_inputs(ctx, prob)
        ctx.save_for_backward(samples_n)
        ctx.delta_shape = delta.shape if isinstance(delta, torch.Tensor) else None
        ctx.mark_non_differentiable(*[t for t in (pose_opt_n, samples_n, cost, pose_opt, samples, x3d_c, offset) if t is not None])
        return pose_opt_n, samples_n, logw, cost, cost_init, pose_opt, samples, x3d_c, offset

    @staticmethod
    def backward(ctx, _gpn, _gsn, g_logw, _gc, g_cost_init, _gp, _gs, _gx, _go):
        (samples_n,) = ctx.saved_tensors
        prob = ctx.prob
        if g_logw is None and g_cost_init is None:
            return (None,) * 9
        _check_inputs(ctx)
        if g_logw is None:
            g_logw = torch.full(samples_n.shape[:2], 0.0, dtype=torch.float32, device=samples_n.device)
        pin = ctx.keep[1]
        gx3d, gx2d, gw2d, gdel = amis_backward(prob, samples_n, g_logw, pin if g_cost_init is not None else None,
                                               g_cost_init, cstruct=ctx.bprob)
        gdelta = None
        if ctx.delta_shape is not None and ctx.needs_input_grad[3]:
            gdelta = gdel.sum() if len(ctx.delta_shape) == 0 else gdel.reshape(ctx.delta_shape)
        return (gx3d if ctx.needs_input_grad[0] else None, gx2d if ctx.needs_input_grad[1] else None,
                gw2d if ctx.needs_input_grad[2] else None, gdelta, None, None, None, None, None)


def fused_monte_carlo(x3d, x2d, w2d, delta, prob, pose_init, par, noise, with_cost):
    """-> pose_opt, pose_samples, logweights, cost | None, cost_init | None, and the solver frame: pose_opt_n,
    (x3d_centered, offset) | (None, None)"""
    ext = _hip.torch_ext()
    if ext is not None:      # C++ autograd node over the same entry point (csrc/torch_binding.cpp)
        if noise is not None:
            assert noise.shape == (prob.B, par.amis.num_iter, par.amis.mc_samples // par.amis.num_iter, noise_stride(prob.dof)), \
                f'noise shape {tuple(noise.shape)}'
            _f32c(noise, 'noise')
        if pose_init is not None:
            _f32c(pose_init, 'pose_init')
        pose_opt_n, samples_n, logw, cost, cost_init, pose_opt, samples, x3d_c, offset = ext.fused_monte_carlo(
            x3d, x2d, w2d, delta, prob.x3d, prob.x2d, prob.w2d, prob.cam, prob.lb, prob.ub, prob.delta, prob.status,
            prob.z_min, prob.huber_eps, prob.dof, pose_init, noise, bytes(par), bool(with_cost),
            backward_split(prob.B, prob.N, par.amis.mc_samples), int(prob.stream or 0),
            None if prob.delta_fold is None else prob.delta_fold[0], 0.0 if prob.delta_fold is None else prob.delta_fold[1])
    else:
        assert prob.delta_fold is None or delta is None, 'a folded delta takes no gradient from this node'
        pose_opt_n, samples_n, logw, cost, cost_init, pose_opt, samples, x3d_c, offset = _FusedMonteCarlo.apply(
            x3d, x2d, w2d, delta, prob, pose_init, par, noise, with_cost)
    if pose_opt is None:
        pose_opt, samples = pose_opt_n, samples_n
    return pose_opt, samples, logw, cost, cost_init, pose_opt_n, x3d_c, offset


class _AdaptiveDelta(torch.autograd.Function):
    """delta_b = mean(w2d_b) * sqrt(sum_xy var_N(x2d_b)) * relative_delta in one pass over (x2d, w2d)
    (reference: ~8 ATen launches, epropnp/cost_fun.py:123-126).  Backward: closed-form broadcasts."""

    @staticmethod
    def forward(ctx, x2d, w2d, relative_delta):
        x, w = _f32c(x2d, 'x2d'), _f32c(w2d, 'w2d')
        B, N, _ = x.shape
        delta = torch.empty(B, dtype=torch.float32, device=x.device)
        stats = torch.empty(B, 4, dtype=torch.float32, device=x.device)
        _hip.call('epropnp_adaptive_delta', _hip.ptr(x), _hip.ptr(w), B, N, float(relative_delta), _hip.ptr(delta),
                  _hip.ptr(stats), _hip.stream_of(x))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, stats)
        ctx.rel, ctx.N = float(relative_delta), N
        ctx.mark_non_differentiable(stats)
        return delta, stats

    @staticmethod
    def backward(ctx, g, _g_stats=None):
        if g is None:
            return None, None, None
        x, stats = ctx.saved_tensors
        N, rel = ctx.N, ctx.rel
        mw, sd = stats[:, 0], stats[:, 1]
        gx = gw = None
        if ctx.needs_input_grad[1]:      # d delta / d w = std * rel / (2N), the same for every element of the object
            gw = (g * sd * (rel / (2 * N)))[:, None, None].expand(-1, N, 2)
        if ctx.needs_input_grad[0]:      # d delta / d x = mean_w * rel * (x - mean) / ((N-1) std)
            coef = g * mw * rel / ((N - 1) * sd.clamp(min=1e-30))
            gx = coef[:, None, None] * (x - stats[:, None, 2:4])
        return gx, gw, None


def adaptive_delta(x2d, w2d, relative_delta):
    """-> delta (B,) [differentiable], stats (B,4) = [mean_w, x2d_std, mean_x, mean_y] of the same pass"""
    ext = _hip.torch_ext()
    if ext is not None:
        _f32c(x2d, 'x2d'), _f32c(w2d, 'w2d')
        delta, stats = ext.adaptive_delta(x2d, w2d, float(relative_delta), int(_hip.stream_of(x2d) or 0))
        return delta, stats
    return _AdaptiveDelta.apply(x2d, w2d, relative_delta)


class _McPoseLoss(torch.autograd.Function):
    """Per-object Monte-Carlo pose loss cost_target + logsumexp_S(logweights), NaN -> 0, as two kernels (fwd / bwd)."""

    @staticmethod
    def forward(ctx, logw, cost_target):
        lw = _f32c(logw, 'pose_sample_logweights')
        S, B = lw.shape
        ct = None if cost_target is None else _f32c(cost_target, 'cost_target')
        loss = torch.empty(B, dtype=torch.float32, device=lw.device)
        lse = torch.empty_like(loss)
        _hip.call('epropnp_mc_loss_forward', _hip.ptr(lw), _hip.ptr(ct), S, B, _hip.ptr(loss), _hip.ptr(lse),
                  _hip.stream_of(lw))
        ctx.save_for_backward(lw, lse, loss)
        return loss

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        lw, lse, loss = ctx.saved_tensors
        S, B = lw.shape
        g = g.contiguous()
        glw = torch.empty_like(lw)
        gct = torch.empty_like(g) if ctx.needs_input_grad[1] else None
        _hip.call('epropnp_mc_loss_backward', _hip.ptr(lw), _hip.ptr(lse), _hip.ptr(loss), _hip.ptr(g), S, B,
                  _hip.ptr(glw), _hip.ptr(gct), _hip.stream_of(lw))
        return glw, gct


def mc_pose_loss(logweights, cost_target):
    ext = _hip.torch_ext()
    if ext is not None:
        _f32c(logweights, 'pose_sample_logweights')
        if cost_target is not None:
            _f32c(cost_target, 'cost_target')
        return ext.mc_pose_loss(logweights, cost_target, int(_hip.stream_of(logweights) or 0))
    return _McPoseLoss.apply(logweights, cost_target)


class _McPoseLossReduced(torch.autograd.Function):
    """The reduced Monte-Carlo pose loss (mc_pose_loss_reduced) as three kernels: per-object forward, reduce, backward."""

    @staticmethod
    def forward(ctx, logw, cost_target, weight, scale, momentum, nf_in, norm_factor):
        lw = _f32c(logw, 'pose_sample_logweights')
        S, B = lw.shape
        ct = None if cost_target is None else _f32c(cost_target, 'cost_target')
        loss = torch.empty(B, dtype=torch.float32, device=lw.device)
        lse = torch.empty_like(loss)
        out = torch.empty(2, dtype=torch.float32, device=lw.device)
        st = _hip.stream_of(lw)
        _hip.call('epropnp_mc_loss_forward', _hip.ptr(lw), _hip.ptr(ct), S, B, _hip.ptr(loss), _hip.ptr(lse), st)
        nf_count, nf_stride = (nf_in.numel(), nf_in.stride(0)) if (nf_in is not None and nf_in.dim() == 1 and nf_in.numel() > 1) else (1, 1)
        _hip.call('epropnp_mc_loss_reduce', _hip.ptr(loss), _hip.ptr(weight), B, float(scale), float(momentum),
                  _hip.ptr(nf_in), int(nf_count), int(nf_stride), _hip.ptr(norm_factor), _hip.ptr(out), st)
        ctx.save_for_backward(lw, lse, out)
        ctx.weight = weight
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 7
        lw, lse, out = ctx.saved_tensors
        S, B = lw.shape
        g = g.to(torch.float32).reshape(1).contiguous()
        glw = torch.empty_like(lw)
        gct = torch.empty(B, dtype=torch.float32, device=lw.device) if ctx.needs_input_grad[1] else None
        _hip.call('epropnp_mc_loss_reduce_backward', _hip.ptr(lw), _hip.ptr(lse), _hip.ptr(ctx.weight), _hip.ptr(out[1:]),
                  _hip.ptr(g), S, B, _hip.ptr(glw), _hip.ptr(gct), _hip.stream_of(lw))
        return glw, gct, None, None, None, None, None


def mc_pose_loss_reduced(logweights, cost_target, weight=None, scale=1.0, momentum=0.0, norm_factor_in=None, norm_factor=None):
    """-> the 0-dim loss  (sum_b weight[b] (cost_target[b] + logsumexp_S logweights[:, b])) * scale / norm_factor  of both
    reference loss modules (losses.MonteCarloPoseLoss), NaN objects zeroed.  `norm_factor` (1,)/0-dim device buffer, updated IN
    PLACE to (1 - momentum) norm_factor + momentum * norm_factor_in first when `norm_factor_in` (device scalar) is given.
    weight (B,) without grad, or None."""
    _f32c(logweights, 'pose_sample_logweights')
    if cost_target is not None:
        _f32c(cost_target, 'cost_target')
    w = None if weight is None else _f32c(weight, 'weight')
    if norm_factor_in is not None and norm_factor_in.dim() == 1 and norm_factor_in.numel() > 1:
        # one value per rank, strided through the receive buffer of the step's all-gather (sharding.ObjectExchange.scalar_slots):
        # the reduce kernel averages them itself -- no mean launch, no copy
        _hip.check_device(norm_factor_in, 'norm_factor_in')
        if norm_factor_in.dtype != torch.float32:
            raise TypeError('norm_factor_in must be float32')
        nf_in = norm_factor_in
    else:
        nf_in = None if norm_factor_in is None else _f32c(norm_factor_in, 'norm_factor_in')
    if norm_factor is not None and (norm_factor.dtype != torch.float32 or not norm_factor.is_contiguous()
                                    or norm_factor.device != logweights.device):
        raise ValueError('norm_factor must be a contiguous float32 scalar on the device of the log-weights')
    ext = _hip.torch_ext()
    if ext is not None:
        return ext.mc_pose_loss_reduced(logweights, cost_target, w, float(scale), float(momentum), nf_in, norm_factor,
                                        int(_hip.stream_of(logweights) or 0))
    return _McPoseLossReduced.apply(logweights, cost_target, w, scale, momentum, nf_in, norm_factor)


def exchange_pack(send, rows, scalars=None, sum_of=None, sum_scale=1.0, sum_row_weight=None):
    """send[:n_scalars] = scalars, send[n_scalars : n_scalars + rows.numel()] = rows.flatten() in ONE launch
    (epropnp_exchange_pack).  With `sum_of` the first scalar is `sum_scale * sum_of.sum()`, computed inside the same launch (fixed
    order: bit-reproducible) -- `scalars` then only supplies the others, if any.  All tensors fp32 on one HIP device."""
    rows = _f32c(rows, 'rows')
    n_scal = 0 if scalars is None else int(scalars.numel())
    row_len = 1
    if sum_of is not None:
        sum_of = _f32c(sum_of, 'sum_of')
        n_scal = max(n_scal, 1)
        if sum_row_weight is not None:        # sum_of (rows, ...) with one weight per row
            sum_row_weight = _f32c(sum_row_weight, 'sum_row_weight')
            assert sum_row_weight.dim() == 1 and sum_of.shape[0] == sum_row_weight.shape[0]
            row_len = max(1, sum_of.numel() // max(1, sum_of.shape[0]))
    if scalars is not None:
        scalars = _f32c(scalars.reshape(-1), 'scalars')
    assert send.is_contiguous() and send.dtype == torch.float32 and send.numel() >= n_scal + rows.numel()
    _hip.call('epropnp_exchange_pack', _hip.ptr(rows), int(rows.numel()), _hip.ptr(scalars), n_scal, _hip.ptr(sum_of),
              0 if sum_of is None else int(sum_of.numel()), float(sum_scale),
              _hip.ptr(sum_row_weight) if sum_of is not None else None, int(row_len), _hip.ptr(send), _hip.stream_of(rows))
    return send


def rslm_draw(w2d, num_proposals, num_points, seed, offset):
    """(B,N,2) weights -> (P,B,n) int64 indices, weighted sampling without replacement per (proposal, object)."""
    w = _f32c(w2d, 'w2d')
    B, N, _ = w.shape
    inds = torch.empty((num_proposals, B, num_points), dtype=torch.int64, device=w.device)
    _hip.call('epropnp_rslm_draw', _hip.ptr(w), B, N, int(num_proposals), int(num_points), int(seed), int(offset),
              _hip.ptr(inds), _hip.stream_of(w))
    return inds


class _CenterPoints(torch.autograd.Function):
    """x3d -> (offset, x3d - offset) with the offset treated as a constant (detach_transformation=True)."""

    @staticmethod
    def forward(ctx, x3d):
        x = _f32c(x3d, 'x3d')
        B, N, _ = x.shape
        offset, out = torch.empty((B, 3), dtype=x.dtype, device=x.device), torch.empty_like(x)
        _hip.call('epropnp_center_points', _hip.ptr(x), B, N, _hip.ptr(offset), _hip.ptr(out), _hip.stream_of(x))
        ctx.mark_non_differentiable(offset)
        ctx.set_materialize_grads(False)
        return offset, out

    @staticmethod
    def backward(ctx, _g_offset, g_out):
        return g_out


def center_points(x3d):
    return _CenterPoints.apply(x3d)


class _ShiftPoses(torch.autograd.Function):
    """pose translation += sign * R(pose) offset, differentiable w.r.t. the pose (offset is a constant)."""

    @staticmethod
    def forward(ctx, pose, offset, sign):
        ps, off = _f32c(pose.detach(), 'pose'), _f32c(offset.detach(), 'offset')
        B, pl = ps.shape[-2], ps.shape[-1]
        out = torch.empty_like(ps)
        _hip.call('epropnp_shift_poses', _hip.ptr(ps), _hip.ptr(off), ps.numel() // (B * pl), B, 6 if pl == 7 else 4,
                  float(sign), _hip.ptr(out), _hip.stream_of(ps))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(ps, off)
        ctx.sign = float(sign)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        ps, off = ctx.saved_tensors
        B, pl = ps.shape[-2], ps.shape[-1]
        g = g.contiguous()
        gp = torch.empty_like(ps)
        _hip.call('epropnp_shift_poses_backward', _hip.ptr(ps), _hip.ptr(off), _hip.ptr(g), ps.numel() // (B * pl), B,
                  6 if pl == 7 else 4, ctx.sign, _hip.ptr(gp), _hip.stream_of(ps))
        return gp, None, None


def shift_poses(pose, offset, sign):
    """pose (...,B,pose_len) translation += sign * R(pose) offset (B,3); differentiable w.r.t. the pose."""
    if pose.requires_grad and torch.is_grad_enabled():
        ext = _hip.torch_ext()
        if ext is not None:
            _f32c(pose, 'pose'), _f32c(offset, 'offset')
            return ext.shift_poses(pose, offset, float(sign), int(_hip.stream_of(pose) or 0))
        return _ShiftPoses.apply(pose, offset, sign)
    ps, off = _f32c(pose.detach(), 'pose'), _f32c(offset.detach(), 'offset')
    B, pl = ps.shape[-2], ps.shape[-1]
    out = torch.empty_like(ps)
    _hip.call('epropnp_shift_poses', _hip.ptr(ps), _hip.ptr(off), ps.numel() // (B * pl), B, 6 if pl == 7 else 4,
              float(sign), _hip.ptr(out), _hip.stream_of(ps))
    return out


RSLM_MAX_POINTS = 512      # epropnp_rslm_solve keeps an object's correspondences + 16 key rows in LDS


def rslm_solve(prob, num_proposals, num_points, num_iter, seed=0, offset=0, inds=None, rot=None, fast_mode=False,
               offset_dev=None,
               min_lm_diagonal=1e-6, max_lm_diagonal=1e32, min_relative_decrease=1e-3,
               initial_trust_region_radius=30.0, max_trust_region_radius=1e16, eps=1e-5):
    """The whole random-sample LM initialiser in one launch -> pose (B,pose_len), cost (B,).
    `inds` (P,B,n) int64 / `rot` (P,B[,4]) inject the random draws; None draws them on the device."""
    P, n = int(num_proposals), int(num_points)
    if inds is not None:
        inds = inds.contiguous()
        assert inds.dtype == torch.int64 and inds.shape == (P, prob.B, n), f'inds shape {tuple(inds.shape)}'
    if rot is not None:
        rot = _f32c(rot, 'rot')
        assert rot.numel() == P * prob.B * (prob.pose_len - 3), f'rot shape {tuple(rot.shape)}'
    pose, cost = prob.new(prob.B, prob.pose_len), prob.new(prob.B)
    par = _hip.LmParams(int(num_iter), int(bool(fast_mode)), min_lm_diagonal, max_lm_diagonal, min_relative_decrease,
                        initial_trust_region_radius, max_trust_region_radius, eps)
    scratch = rslm_scratch(prob, P)
    _hip.call('epropnp_rslm_solve', C.byref(prob.c), C.byref(par), P, n, int(seed), int(offset), _hip.ptr(offset_dev),
              _hip.ptr(inds), _hip.ptr(rot), _hip.ptr(pose), _hip.ptr(cost), _hip.ptr(scratch),
              0 if scratch is None else scratch.numel() * 4, prob.stream)
    return pose, cost


def rslm_scratch(prob, num_proposals):
    """Scratch with which the RSLM kernel deals an object's proposals to several workgroups (finer load balance), or None
    when the library would not; from torch's caching allocator (no HIP allocation, no alloc / free graph nodes)."""
    nbytes = int(_hip.lib().epropnp_rslm_solve_scratch_bytes(C.byref(prob.c), int(num_proposals)))
    return prob.new((nbytes + 3) // 4) if nbytes > 0 else None


class _GnStep(torch.autograd.Function):
    """step = -(J^T J + eps I)^-1 J^T r at `pose`, differentiable w.r.t. x3d, x2d, w2d, delta (one sweep forward,
    two sweeps backward, nothing materialised; reference: levenberg_marquardt.py:243-253 + autograd)."""

    @staticmethod
    def forward(ctx, x3d, x2d, w2d, delta, prob, pose, eps):
        ps = _f32c(pose, 'pose')
        step = prob.new(prob.B, prob.dof)
        _hip.call('epropnp_gn_step_forward', C.byref(prob.c), float(eps), _hip.ptr(ps), _hip.ptr(step), prob.stream)
        ctx.set_materialize_grads(False)
        ctx.prob, ctx.eps = prob, float(eps)
        _guard_inputs(ctx, prob)
        ctx.save_for_backward(ps)
        ctx.delta_shape = delta.shape if isinstance(delta, torch.Tensor) else None
        return step

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 7
        _check_inputs(ctx)
        (ps,) = ctx.saved_tensors
        prob = ctx.prob
        B, N = prob.B, prob.N
        g = g.contiguous()
        gx3d, gx2d, gw2d, gdel = prob.new(B, N, 3), prob.new(B, N, 2), prob.new(B, N, 2), prob.new(B)
        _hip.call('epropnp_gn_step_backward', C.byref(prob.c), ctx.eps, _hip.ptr(ps), _hip.ptr(g), _hip.ptr(gx3d),
                  _hip.ptr(gx2d), _hip.ptr(gw2d), _hip.ptr(gdel), prob.stream)
        gdelta = None
        if ctx.delta_shape is not None and ctx.needs_input_grad[3]:
            gdelta = gdel.sum() if len(ctx.delta_shape) == 0 else gdel.reshape(ctx.delta_shape)
        return (gx3d if ctx.needs_input_grad[0] else None, gx2d if ctx.needs_input_grad[1] else None,
                gw2d if ctx.needs_input_grad[2] else None, gdelta, None, None, None)


def _ext_gn_step(ext, x3d, x2d, w2d, delta, prob, pose, eps, with_plus):
    _f32c(pose, 'pose')
    fold = prob.delta_fold
    return ext.gn_step(x3d, x2d, w2d, None if fold else delta, prob.x3d, prob.x2d, prob.w2d, prob.cam, prob.lb, prob.ub,
                       prob.delta, prob.status, prob.z_min, prob.huber_eps, prob.dof, pose, float(eps), with_plus,
                       int(prob.stream or 0), None if fold is None else fold[0], 0.0 if fold is None else fold[1])


def gn_step(x3d, x2d, w2d, delta, prob, pose, eps):
    ext = _hip.torch_ext()
    if ext is not None:
        return _ext_gn_step(ext, x3d, x2d, w2d, delta, prob, pose, eps, False)
    return _GnStep.apply(x3d, x2d, w2d, None if prob.delta_fold else delta, prob, pose, eps)


class _PoseOptPlus(torch.autograd.Function):
    """pose_opt_plus = pose (+) gn_step(pose): the Gauss-Newton step and LMSolver.pose_add in one kernel each way
    (reference: levenberg_marquardt.py:70-72 -> :243-265 + autograd)."""

    @staticmethod
    def forward(ctx, x3d, x2d, w2d, delta, prob, pose, eps):
        ps = _f32c(pose, 'pose')
        plus = prob.new(prob.B, prob.pose_len)
        _hip.call('epropnp_pose_opt_plus_forward', C.byref(prob.c), float(eps), _hip.ptr(ps), _hip.ptr(plus), prob.stream)
        ctx.set_materialize_grads(False)
        ctx.prob, ctx.eps = prob, float(eps)
        _guard_inputs(ctx, prob)
        ctx.save_for_backward(ps)
        ctx.delta_shape = delta.shape if isinstance(delta, torch.Tensor) else None
        return plus

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 7
        _check_inputs(ctx)
        (ps,) = ctx.saved_tensors
        prob = ctx.prob
        B, N = prob.B, prob.N
        g = g.contiguous()
        gx3d, gx2d, gw2d, gdel = prob.new(B, N, 3), prob.new(B, N, 2), prob.new(B, N, 2), prob.new(B)
        _hip.call('epropnp_pose_opt_plus_backward', C.byref(prob.c), ctx.eps, _hip.ptr(ps), _hip.ptr(g), _hip.ptr(gx3d),
                  _hip.ptr(gx2d), _hip.ptr(gw2d), _hip.ptr(gdel), prob.stream)
        gdelta = None
        if ctx.delta_shape is not None and ctx.needs_input_grad[3]:
            gdelta = gdel.sum() if len(ctx.delta_shape) == 0 else gdel.reshape(ctx.delta_shape)
        return (gx3d if ctx.needs_input_grad[0] else None, gx2d if ctx.needs_input_grad[1] else None,
                gw2d if ctx.needs_input_grad[2] else None, gdelta, None, None, None)


def pose_opt_plus(x3d, x2d, w2d, delta, prob, pose, eps):
    ext = _hip.torch_ext()
    if ext is not None:
        return _ext_gn_step(ext, x3d, x2d, w2d, delta, prob, pose, eps, True)
    return _PoseOptPlus.apply(x3d, x2d, w2d, None if prob.delta_fold else delta, prob, pose, eps)
