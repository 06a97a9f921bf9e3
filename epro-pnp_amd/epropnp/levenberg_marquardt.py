"""Levenberg-Marquardt / Gauss-Newton PnP solver and the random-sample initialiser.

API mirror of the reference's epropnp/levenberg_marquardt.py (LMSolver :22-265, RSLMSolver :268-353): same
constructor arguments, call signatures and return tuples.  The iteration itself is ONE HIP kernel
(csrc/lm_kernel.hip via epropnp_lm_solve); there is no CPU implementation.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as hip
from ._dtype import fp32_boundary
from .common import evaluate_pnp, pnp_denormalize, pnp_normalize


def solve_wrapper(b, A):
    """A x = b for batched small systems; empty batches pass through keeping the graph (DDP callers rely on it)."""
    if A.numel() == 0:
        return b + A.reshape_as(b)
    return torch.linalg.solve(A, b)


class LMSolver(nn.Module):
    """Fixed-iteration Levenberg-Marquardt solver.
    4-DoF pose = [x, y, z, yaw] (yaw about the Y axis); 6-DoF pose = [x, y, z, w, i, j, k] (unit quaternion).

    `check_numerics` (attribute, not a constructor argument: the reference's signature is kept): set
    `solver.check_numerics = True` to get the reference's error convention -- a singular / non-finite damped system raises
    RuntimeError from the very call that produced it, as torch.linalg.solve / torch.inverse do in
    levenberg_marquardt.py:15-19,178-181 -- at the price of ONE host synchronisation per call.  Default False: the event
    is reported asynchronously at the next entry into the package (RuntimeWarning, or RuntimeError with
    EPROPNP_ASYNC_STATUS=raise), and nothing on the path synchronises."""

    check_numerics = False

    def __init__(self, dof=4, num_iter=10, min_lm_diagonal=1e-6, max_lm_diagonal=1e32, min_relative_decrease=1e-3,
                 initial_trust_region_radius=30.0, max_trust_region_radius=1e16, eps=1e-5, normalize=False,
                 init_solver=None):
        super().__init__()
        self.dof = dof
        self.num_iter = num_iter
        self.min_lm_diagonal = min_lm_diagonal
        self.max_lm_diagonal = max_lm_diagonal
        self.min_relative_decrease = min_relative_decrease
        self.initial_trust_region_radius = initial_trust_region_radius
        self.max_trust_region_radius = max_trust_region_radius
        self.eps = eps
        self.normalize = normalize
        self.init_solver = init_solver

    # ------------------------------------------------------------------------------------------------
    @fp32_boundary
    def forward(self, x3d, x2d, w2d, camera, cost_fun, with_pose_opt_plus=False, pose_init=None,
                normalize_override=None, **kwargs):
        """-> pose_opt, pose_cov | None, cost | None, pose_opt_plus | None"""
        normalize = normalize_override if isinstance(normalize_override, bool) else self.normalize
        if normalize:
            transform, x3d, pose_init = pnp_normalize(x3d, pose_init, detach_transformation=True)
        pose_opt, pose_cov, cost = self.solve(x3d, x2d, w2d, camera, cost_fun, pose_init=pose_init, **kwargs)
        pose_opt_plus = None
        if with_pose_opt_plus:
            from . import _hip
            if x2d.dim() == 3 and x2d.size(0) > 0 and not pose_opt.requires_grad and \
                    _hip.on_hip_path(x3d, x2d, w2d, pose_opt):
                # Gauss-Newton step + pose_add in one kernel each way (csrc/gn_step_kernel.hip)
                prob = hip.problem(x3d, x2d, w2d, camera, cost_fun, self.dof)
                delta = cost_fun.delta if isinstance(cost_fun.delta, torch.Tensor) else None
                pose_opt_plus = hip.pose_opt_plus(x3d, x2d, w2d, delta, prob, pose_opt, self.eps)
            else:
                pose_opt_plus = self.pose_add(pose_opt, self.gn_step(x3d, x2d, w2d, pose_opt, camera, cost_fun), camera)
        if normalize:
            pose_opt = pnp_denormalize(transform, pose_opt)
            if pose_cov is not None:
                raise NotImplementedError('Normalized covariance unsupported')
            if pose_opt_plus is not None:
                pose_opt_plus = pnp_denormalize(transform, pose_opt_plus)
        return pose_opt, pose_cov, cost, pose_opt_plus

    def _lm_kwargs(self):
        return dict(min_lm_diagonal=self.min_lm_diagonal, max_lm_diagonal=self.max_lm_diagonal,
                    min_relative_decrease=self.min_relative_decrease,
                    initial_trust_region_radius=self.initial_trust_region_radius,
                    max_trust_region_radius=self.max_trust_region_radius, eps=self.eps)

    @fp32_boundary
    def solve(self, x3d, x2d, w2d, camera, cost_fun, pose_init=None, cost_init=None, with_pose_cov=False,
              with_cost=False, force_init_solve=False, fast_mode=False):
        """x3d (B,N,3), x2d/w2d (B,N,2) -> pose_opt (B,4|7), pose_cov (B,d,d) | None, cost (B,) | None.
        Runs entirely without autograd."""
        if self.check_numerics and not hip.numerics_check.active() and x2d.size(0) > 0:
            with hip.numerics_check():          # raises here, after one synchronisation, what the reference raises here
                return self.solve(x3d, x2d, w2d, camera, cost_fun, pose_init=pose_init, cost_init=cost_init,
                                  with_pose_cov=with_pose_cov, with_cost=with_cost, force_init_solve=force_init_solve,
                                  fast_mode=fast_mode)
        with torch.no_grad():
            num_obj = x2d.size(0)
            pose_len = 4 if self.dof == 4 else 7
            if num_obj == 0:
                kw = dict(dtype=x2d.dtype, device=x2d.device)
                return (torch.empty((0, pose_len), **kw),
                        torch.empty((0, self.dof, self.dof), **kw) if with_pose_cov else None,
                        torch.empty((0,), **kw) if with_cost else None)
            prob = hip.problem(x3d, x2d, w2d, camera, cost_fun, self.dof)
            if pose_init is None or force_init_solve:
                assert self.init_solver is not None
                if pose_init is None:
                    pose_start, _, _ = self.init_solver.solve(x3d, x2d, w2d, camera, cost_fun, fast_mode=fast_mode)
                else:   # keep, per object, whichever of {given pose, random-sample solution} costs less
                    if cost_init is None:
                        cost_init = hip.evaluate_cost(prob, pose_init)
                    pose_start, _, cost_start = self.init_solver.solve(
                        x3d, x2d, w2d, camera, cost_fun, with_cost=True, fast_mode=fast_mode)
                    pose_start = torch.where((cost_init < cost_start).unsqueeze(-1), pose_init, pose_start)
            else:
                pose_start = pose_init
            return hip.lm_solve(prob, pose_start, self.num_iter, fast_mode=fast_mode, with_pose_cov=with_pose_cov,
                                with_cost=with_cost, **self._lm_kwargs())

    # ------------------------------------------------------------------------------------------------
    def gn_step(self, x3d, x2d, w2d, pose, camera, cost_fun, composite=False):
        """One differentiable Gauss-Newton step at `pose` (used for the derivative-regularisation loss).
        On HIP tensors: fused forward / backward kernels (csrc/gn_step_kernel.hip), differentiable w.r.t. the correspondences and
        cost_fun.delta; otherwise -- a pose that requires grad, `composite=True`, or camera.cam_mats requiring grad -- the PyTorch
        composite of the reference (levenberg_marquardt.py:243-253), which autograd differentiates w.r.t. everything."""
        from . import _hip
        cam_grad = isinstance(camera.cam_mats, torch.Tensor) and camera.cam_mats.requires_grad and torch.is_grad_enabled()
        if x2d.dim() == 3 and x2d.size(0) > 0 and not pose.requires_grad and not composite and not cam_grad \
                and _hip.on_hip_path(x3d, x2d, w2d, pose):
            prob = hip.problem(x3d, x2d, w2d, camera, cost_fun, self.dof)
            delta = cost_fun.delta if isinstance(cost_fun.delta, torch.Tensor) else None
            return hip.gn_step(x3d, x2d, w2d, delta, prob, pose, self.eps)
        residual, _, jac = evaluate_pnp(x3d, x2d, w2d, pose, camera, cost_fun, out_jacobian=True, out_residual=True)
        jac_t = jac.transpose(-1, -2)
        jtj = jac_t @ jac + self.eps * torch.eye(self.dof, device=jac.device, dtype=jac.dtype)
        return -solve_wrapper(jac_t @ residual.unsqueeze(-1), jtj).squeeze(-1)

    def pose_add(self, pose_opt, step, camera):
        if self.dof == 4:
            return pose_opt + step
        quat = pose_opt[..., 3:]
        dq = (camera.get_quaternion_transfrom_mat(quat) @ step[..., 3:, None]).squeeze(-1)
        return torch.cat((pose_opt[..., :3] + step[..., :3], F.normalize(quat + dq, dim=-1)), dim=-1)


class RSLMSolver(LMSolver):
    """Random-sample LM (a RANSAC generalisation): solve `num_proposals` small sub-problems of `num_points`
    correspondences from random rotations, keep the proposal with the lowest full-set cost."""

    def __init__(self, num_points=16, num_proposals=64, num_iter=3, **kwargs):
        super().__init__(num_iter=num_iter, **kwargs)
        self.num_points = num_points
        self.num_proposals = num_proposals

    def center_based_init(self, x2d, x3d, camera, eps=1e-6):
        """Translation guess from the centroid / spread of the back-projected 2D points."""
        # back-project through K^-1 (explicit 3x3 inverse applied element-wise: no batched LU / GEMM launches)
        kinv = torch.linalg.inv(camera.cam_mats)
        xh = torch.cat((x2d, torch.ones_like(x2d[..., :1])), dim=-1)
        rays = (xh.unsqueeze(-2) * kinv.unsqueeze(-3)).sum(dim=-1)
        rays = rays[..., :2] / rays[..., 2:].clamp(min=eps)
        ray_std, ray_mean = torch.std_mean(rays, dim=-2)
        x3d_std = torch.std(x3d, dim=-2)
        if self.dof == 4:
            depth = x3d_std[..., 1] / ray_std[..., 1].clamp(min=eps)
        else:
            depth = math.sqrt(2 / 3) * x3d_std.norm(dim=-1) / ray_std.norm(dim=-1).clamp(min=eps)
        return torch.cat((ray_mean, torch.ones_like(ray_mean[..., :1])), dim=-1) * depth.unsqueeze(-1)

    def draw(self, w2d):
        """Random part of the initialiser: sub-sample indices (P,B,n) ~ mean weight, w/o replacement, and random
        initial rotations (P,B) yaw in [0, 2 pi) or (P,B,4) unit quaternions.  Overridable for reproducibility."""
        bs, pn, _ = w2d.shape
        P = self.num_proposals
        from . import _hip
        if _hip.on_hip_path(w2d) and pn * 4 <= 64 * 1024:
            # one kernel (exponential-race keys + wave argmin) instead of torch.multinomial's top-k on (P*B, N)
            if not hasattr(self, '_draw_seed'):
                self._draw_seed, self._draw_calls = int(torch.randint(0, 2 ** 62, (1,)).item()), 0
            inds = hip.rslm_draw(w2d, P, self.num_points, self._draw_seed, self._draw_calls)
            self._draw_calls += 1
        else:
            mean_weight = w2d.mean(dim=-1).reshape(1, bs, pn).expand(P, -1, -1).reshape(-1, pn)
            inds = torch.multinomial(mean_weight, self.num_points).reshape(P, bs, self.num_points)
        if self.dof == 4:
            rot = torch.rand((P, bs), dtype=w2d.dtype, device=w2d.device) * (2 * math.pi)
        else:
            rot = torch.randn((P, bs, 4), dtype=w2d.dtype, device=w2d.device)
            nrm = rot.norm(dim=-1, keepdim=True)
            ident = rot.new_tensor([1.0, 0.0, 0.0, 0.0])
            rot = torch.where(nrm < self.eps, ident, rot / nrm.clamp(min=1e-30))
        return inds, rot

    @fp32_boundary
    def solve(self, x3d, x2d, w2d, camera, cost_fun, **kwargs):
        """-> pose (B,4|7), None, min_cost (B,)"""
        with torch.no_grad():
            bs, pn, _ = x2d.size()
            pose_len = 4 if self.dof == 4 else 7
            if bs == 0:
                return x2d.new_empty((0, pose_len)), None, x2d.new_empty((0,))
            P, n = self.num_proposals, self.num_points
            from . import _hip
            if n <= 16 and 2 <= pn <= hip.RSLM_MAX_POINTS and _hip.on_hip_path(x3d, x2d, w2d) \
                    and hip.tune('rslm_composite') is None:
                # one kernel: init translation, sub-sampling, P x B solves, scoring, argmin (csrc/rslm_kernel.hip)
                prob = hip.problem(x3d, x2d, w2d, camera, cost_fun, self.dof)
                if getattr(self.draw, '__func__', None) is RSLMSolver.draw:   # default sampler: device Philox
                    inds, rot = None, None
                else:                                       # overridden (reproducibility hooks): inject its draws
                    inds, rot = self.draw(w2d)
                if not hasattr(self, '_draw_seed'):
                    self._draw_seed, self._draw_calls = int(torch.randint(0, 2 ** 62, (1,)).item()), 0
                counter = getattr(self, 'rng_counter', None)       # device-side call counter (hipGraph replay)
                if counter is None:
                    self._draw_calls += 1
                pose, min_cost = hip.rslm_solve(prob, P, n, self.num_iter, self._draw_seed,
                                                0 if counter is not None else self._draw_calls - 1, inds, rot,
                                                fast_mode=bool(kwargs.get('fast_mode', False)), offset_dev=counter,
                                                **self._lm_kwargs())
                if counter is not None and inds is None:
                    counter.add_(1)
                return pose, None, min_cost
            inds, rot = self.draw(w2d)
            obj = torch.arange(bs, device=inds.device)[None, :, None]
            x3d_s, x2d_s, w2d_s = x3d[obj, inds], x2d[obj, inds], w2d[obj, inds]      # (P,B,n,.)
            t0 = self.center_based_init(x2d, x3d, camera).expand(P, bs, 3)
            pose0 = torch.cat((t0, rot.unsqueeze(-1) if self.dof == 4 else rot), dim=-1)

            cam_rep = camera.shallow_copy().repeat_(P)
            cost_rep = cost_fun.shallow_copy().repeat_(P)
            solve_kw = {k: v for k, v in kwargs.items() if k in ('fast_mode',)}
            pose, _, _ = LMSolver.solve(self, x3d_s.reshape(P * bs, n, 3), x2d_s.reshape(P * bs, n, 2),
                                        w2d_s.reshape(P * bs, n, 2), cam_rep, cost_rep,
                                        pose_init=pose0.reshape(P * bs, pose_len), **solve_kw)
            pose = pose.reshape(P, bs, pose_len)
            # score every proposal on the full correspondence set (cost-only sweep kernel)
            prob = hip.problem(x3d, x2d, w2d, camera, cost_fun, self.dof)
            cost = hip.evaluate_cost(prob, pose)
            min_cost, best = cost.min(dim=0)
            return pose[best, torch.arange(bs, device=pose.device)], None, min_cost
