"""EPro-PnP layer: PnP mode (LM solve) + Monte-Carlo pose distribution (AMIS), MI355X-native.

API mirror of the reference's epropnp/epropnp.py (EProPnPBase :36-196, EProPnP4DoF :199-260, EProPnP6DoF :263-342):
same constructor arguments, `forward` / `monte_carlo_forward` signatures and 6-tuple return.  Where the reference
runs a Python loop of ~150 ATen launches per AMIS iteration with host round-trips for every Cholesky, this layer
makes three HIP launches in total: cost of pose_init, the fused LM solve, and the fused AMIS sampler; the backward
is one more launch that recomputes the cost sweep (csrc/amis_kernels.hip).

Extra (non-reference) knobs, all optional:
  * `seed`            : Philox key of the on-device sampler (default: drawn once from torch's global generator)
  * `noise=` kwarg of monte_carlo_forward : injected base draws, for bit-reproducible comparisons with the oracle.
"""
import os

import torch

from . import _hip
from . import functional as hip
from ._dtype import fp32_boundary
from . import proposals
from .levenberg_marquardt import LMSolver, RSLMSolver
from .common import pnp_denormalize, pnp_normalize


def _DetachedCamera(camera):
    """Shallow copy of a camera object whose intrinsics do not require grad (same tensors otherwise)."""
    import copy
    cam = camera.shallow_copy() if hasattr(camera, 'shallow_copy') else copy.copy(camera)
    if isinstance(cam.cam_mats, torch.Tensor):
        cam.cam_mats = cam.cam_mats.detach()
    return cam


def cholesky_wrapper(mat, default_diag=None, force_cpu=True):
    """Batched Cholesky; matrices that are not positive definite yield diag(default_diag) (or I).
    Kept for API compatibility (reference: epropnp/epropnp.py:16-33); the AMIS kernel has its own in-register
    version and does not call this.  `force_cpu` is accepted and ignored: failures are detected on the device with
    cholesky_ex, there is no host round-trip to force."""
    return proposals.cholesky_or_default(mat, default_diag)


class EProPnPBase(torch.nn.Module):
    """End-to-End Probabilistic Perspective-n-Points.

    Args:
        mc_samples (int): total number of Monte Carlo samples
        num_iter (int): number of AMIS iterations
        normalize (bool): centre x3d before solving
        eps (float)
        solver: PnP solver module (LMSolver)
    """

    dof = None
    # `layer.check_numerics = True`: monte_carlo_forward / forward raise the reference's RuntimeError (singular normal
    # equations, non-finite pose: levenberg_marquardt.py:15-19,178-181) from the call that produced it, after ONE host
    # synchronisation at the end of the call.  Default False: asynchronous report at the next entry into the package.
    check_numerics = False

    def __init__(self, mc_samples=512, num_iter=4, normalize=False, eps=1e-5, solver=None, seed=None):
        super().__init__()
        assert num_iter > 0
        assert mc_samples % num_iter == 0
        self.mc_samples = mc_samples
        self.num_iter = num_iter
        self.iter_samples = mc_samples // num_iter
        self.eps = eps
        self.normalize = normalize
        if isinstance(solver, dict):     # EPro-PnP-Det style config: solver=dict(type='LMSolver', ...)
            from .builder import build_pnp
            solver = build_pnp(dict(solver), dof=self.dof)
        self.solver = solver
        self.seed = seed
        self._calls = 0
        self.rng_counter = None      # device int64 counter (enable_graph_safe_rng): fresh draws on hipGraph replays

    def enable_graph_safe_rng(self, device):
        """Keep the Philox call counter in DEVICE memory (and advance it with an in-stream add) instead of in this
        Python object, so that a step captured into a hipGraph (torch.cuda.CUDAGraph) draws fresh samples on every
        replay.  Also covers the RSLM initialiser of the solver.  Seeds are fixed now (no host sync later)."""
        if self.seed is None:
            self.seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        # [the sampler's call counter, the RSLM initialiser's]: one tensor, so that a forward that advances both does it with
        # ONE in-stream add (each tiny launch is ~2.5 us of a replayed Det step)
        # (third word: the ticket with which the sampler's launch advances the counters itself -- epropnp_amis_params.advance)
        self._rng_pair = torch.full((3,), 0, dtype=torch.int64, device=device)
        self.rng_counter = self._rng_pair[:1]
        init = getattr(self.solver, 'init_solver', None)
        if init is not None and hasattr(init, 'num_proposals'):
            if not hasattr(init, '_draw_seed'):
                init._draw_seed, init._draw_calls = int(torch.randint(0, 2 ** 62, (1,)).item()), 0
            init.rng_counter = self._rng_pair[1:2]      # its own call counter
        return self

    # Extension hooks of the reference (epropnp.py:64-82).  monte_carlo_forward runs the fused sampler and does not call
    # them; the 4-/6-DoF subclasses provide them in PyTorch (epropnp/proposals.py) for code that builds on them.
    def allocate_buffer(self, *args, **kwargs):
        raise NotImplementedError

    def initial_fit(self, *args, **kwargs):
        raise NotImplementedError

    def gen_new_distr(self, *args, **kwargs):
        raise NotImplementedError

    def gen_old_distr(self, *args, **kwargs):
        raise NotImplementedError

    def estimate_params(self, *args, **kwargs):
        raise NotImplementedError

    def _new(self, num_obj, dtype, device, *tails):
        return tuple(torch.empty((self.num_iter, num_obj) + t, dtype=dtype, device=device) for t in tails)

    def forward(self, *args, **kwargs):
        if self.check_numerics and not hip.numerics_check.active():
            with hip.numerics_check():
                return self.solver(*args, **kwargs)
        return self.solver(*args, **kwargs)

    def _amis_config(self, noise):
        if self.seed is None:   # one key per layer instance, derived from torch's RNG so torch.manual_seed governs it
            self.seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        cfg = dict(mc_samples=self.mc_samples, num_iter=self.num_iter, eps=self.eps, noise=noise, seed=self.seed,
                   offset=self._calls, acg_mle_iter=getattr(self, 'acg_mle_iter', 3),
                   acg_dispersion=getattr(self, 'acg_dispersion', 0.001))
        if self.rng_counter is not None:
            cfg.update(offset=0, offset_dev=self.rng_counter)
        else:
            self._calls += 1
        return cfg

    @fp32_boundary
    def monte_carlo_forward(self, x3d, x2d, w2d, camera, cost_fun, pose_init=None, force_init_solve=True,
                            noise=None, **kwargs):
        """Weighted pose samples from the pose distribution defined by the correspondences.

        x3d (B,N,3), x2d (B,N,2), w2d (B,N,2); pose_init (B,4|7) optional (the target pose for the MC loss).
        Tensors on a HIP device only (no CPU path); computed in fp32 -- other floating dtypes are cast on the way in and the
        outputs / gradients back (_dtype.py); differentiable w.r.t. x3d, x2d, w2d, a tensor-valued
        cost_fun.delta, and -- as in the reference -- w.r.t. pose_init (through cost_init) and camera.cam_mats (through
        cost_init and the log-weights).
        Limits: the samples of ONE iteration (mc_samples / num_iter) must fit the LDS pose table (<~ 1500); the total
        mc_samples is unbounded (the sampler state moves from LDS to a global scratch buffer beyond ~2500).  The
        RSLM initialiser's one-launch kernel takes <= 16 points per proposal and <= 512 points per object (beyond that
        the composite of the same kernels runs); any num_pts otherwise (LM streams beyond 8192 points).
        Returns: pose_opt (B,4|7), cost (B,)|None, pose_opt_plus (B,4|7)|None, pose_samples (S,B,4|7),
                 pose_sample_logweights (S,B) [differentiable], cost_init (B,)|None [differentiable].
        """
        assert x3d.dim() == x2d.dim() == w2d.dim() == 3
        if self.check_numerics and not hip.numerics_check.active() and x3d.size(0) > 0:
            with hip.numerics_check():
                return self.monte_carlo_forward(x3d, x2d, w2d, camera, cost_fun, pose_init=pose_init,
                                                force_init_solve=force_init_solve, noise=noise, **kwargs)
        cam_grad = isinstance(camera.cam_mats, torch.Tensor) and camera.cam_mats.requires_grad
        if torch.is_grad_enabled() and x3d.size(0) > 0 and ((pose_init is not None and pose_init.requires_grad) or cam_grad):
            # The reference's cost_init = evaluate_pnp(pose=pose_init) and its log-weights are recorded by autograd w.r.t.
            # pose_init and camera.cam_mats as well (epropnp.py:121-124,139-169).  The kernel nodes differentiate w.r.t. the
            # correspondences; these two inputs get their gradients from gradient-only terms (value 0) added to the outputs.
            out = list(self.monte_carlo_forward(x3d, x2d, w2d, _DetachedCamera(camera), cost_fun,
                                                pose_init=None if pose_init is None else pose_init.detach(),
                                                force_init_solve=force_init_solve, noise=noise, **kwargs))
            prob = hip.PnPProblem(x3d.detach(), x2d.detach(), w2d.detach(), _DetachedCamera(camera), cost_fun, self.dof)
            if pose_init is not None:
                out[5] = out[5] + hip.pose_cam_grad_term(prob, pose_init.detach().unsqueeze(0), pose_init, camera.cam_mats,
                                                         1.0, 0)[0]
            if cam_grad:        # logweights = -cost(samples) - log mixture: d/d cam_mats through the cost of every sample
                out[4] = out[4] + hip.pose_cam_grad_term(prob, out[3].detach(), None, camera.cam_mats, -1.0, -1)
                if kwargs.get('with_pose_opt_plus') and out[2] is not None:
                    # The reference's pose_opt_plus (LMSolver.gn_step + pose_add under autograd, levenberg_marquardt.py:70-72,
                    # 243-265) is differentiable w.r.t. camera.cam_mats as well; the fused Gauss-Newton kernels differentiate
                    # w.r.t. the correspondences only.  In this rare case (no caller of the reference learns intrinsics) the step
                    # is recomputed at the detached pose_opt with the PyTorch composite, which autograd differentiates w.r.t.
                    # the correspondences, delta AND the intrinsics -- it replaces the kernels' pose_opt_plus, same value to rounding.
                    sv = self.solver
                    x3d_s, pose_s, transform = x3d, out[0].detach(), None
                    if self.normalize:
                        transform, x3d_s, pose_s = pnp_normalize(x3d, pose_s, detach_transformation=True)
                    step = sv.gn_step(x3d_s, x2d, w2d, pose_s, camera, cost_fun, composite=True)
                    plus = sv.pose_add(pose_s, step, camera)
                    out[2] = pnp_denormalize(transform, plus) if self.normalize else plus
            return tuple(out)
        if self._fusable(x3d, x2d, w2d, pose_init, force_init_solve, kwargs):
            return self._fused_forward(x3d, x2d, w2d, camera, cost_fun, pose_init, force_init_solve, noise, **kwargs)
        if self.normalize:
            transform, x3d, pose_init = pnp_normalize(x3d, pose_init, detach_transformation=True)
        num_obj = x3d.size(0)

        prob = hip.PnPProblem(x3d, x2d, w2d, camera, cost_fun, self.dof) if num_obj > 0 else None
        cost_init_value = None
        if pose_init is not None and num_obj > 0:
            cost_init_value = hip.evaluate_cost(prob, pose_init)

        with hip.share_problem(prob, x3d, x2d, w2d, camera, cost_fun):
            pose_opt, pose_cov, cost, pose_opt_plus = self.solver(
                x3d, x2d, w2d, camera, cost_fun, pose_init=pose_init, cost_init=cost_init_value, with_pose_cov=True,
                force_init_solve=force_init_solve, normalize_override=False, **kwargs)

        if num_obj > 0:
            delta = cost_fun.delta
            out = hip.monte_carlo_cost(x3d, x2d, w2d, delta if isinstance(delta, torch.Tensor) else None, prob,
                                       pose_opt, pose_cov, pose_init, cost_init_value, self._amis_config(noise))
            pose_samples, pose_sample_logweights = out[0], out[1]
            cost_init = out[2] if pose_init is not None else None
            if self.rng_counter is not None and noise is None:
                self.rng_counter.add_(1)          # in-stream: part of a captured graph
        else:   # keep autograd connectivity for empty batches (DDP callers rely on it)
            pose_samples = x2d.new_zeros((self.mc_samples,) + pose_opt.size())
            pose_sample_logweights = x3d.reshape(self.mc_samples, 0) + x2d.reshape(self.mc_samples, 0) \
                + w2d.reshape(self.mc_samples, 0)
            cost_init = (x3d.sum((-1, -2)) + x2d.sum((-1, -2)) + w2d.sum((-1, -2))) if pose_init is not None else None

        if self.normalize:
            pose_opt = pnp_denormalize(transform, pose_opt)
            pose_samples = pnp_denormalize(transform, pose_samples)
            if pose_opt_plus is not None:
                pose_opt_plus = pnp_denormalize(transform, pose_opt_plus)
        return pose_opt, cost, pose_opt_plus, pose_samples, pose_sample_logweights, cost_init


    # ---- one host call for the whole forward (csrc/mc_forward.hip) -------------------------------------------------
    def _fusable(self, x3d, x2d, w2d, pose_init, force_init_solve, kwargs):
        """The fused entry serves the reference's own solver stack: LMSolver with no or an RSLMSolver initialiser whose
        sub-problems fit the one-launch kernel.  Anything else (a custom solver object, > 16 points per proposal, > 512
        points with RSLM) takes the composite path below, made of the same kernels."""
        sv = self.solver
        if type(sv) is not LMSolver or sv.dof != self.dof or x3d.size(0) == 0 or hip.tune('no_fused_forward') is not None:
            return False
        if set(kwargs) - {'with_pose_opt_plus', 'with_cost', 'fast_mode'}:
            return False
        ts = [x3d, x2d, w2d] + ([pose_init] if pose_init is not None else [])
        if not _hip.on_hip_path(*ts) or (pose_init is not None and pose_init.requires_grad and torch.is_grad_enabled()):
            return False
        if pose_init is None or force_init_solve:
            init = sv.init_solver
            if type(init) is not RSLMSolver or init.dof != self.dof or init.num_points > 16 \
                    or not (2 <= x2d.size(1) <= hip.RSLM_MAX_POINTS) or hip.tune('rslm_composite') is not None:
                return False
        return True

    def _fused_forward(self, x3d, x2d, w2d, camera, cost_fun, pose_init, force_init_solve, noise, with_pose_opt_plus=False,
                       with_cost=False, fast_mode=False):
        sv = self.solver
        prob = hip.PnPProblem(x3d, x2d, w2d, camera, cost_fun, self.dof)
        cfg = self._amis_config(noise)
        par = _hip.McParams()
        par.lm = hip._lm_struct(sv, fast_mode)
        # exchange scratch of the split LM solve and of the split forward as ONE block (LM first): the library then fills both
        # with one launch (csrc/mc_forward.hip)
        lm_words, fw_words = hip.split_scratch_words(prob, par.lm, cfg['mc_samples'], cfg['num_iter'])
        block = prob.new(lm_words + fw_words) if lm_words + fw_words else None
        lm_scratch = block[:lm_words] if lm_words else None
        split_scratch = block[lm_words:] if fw_words else None
        par.lm_scratch, par.lm_scratch_bytes = _hip.ptr(lm_scratch), lm_words * 4
        par.amis, _ = hip._amis_struct(prob, cfg['mc_samples'], cfg['num_iter'], cfg['eps'], cfg['acg_mle_iter'],
                                       cfg['acg_dispersion'], cfg['seed'], cfg['offset'], cfg.get('offset_dev'), scratch=split_scratch)
        par.normalize = int(bool(self.normalize))
        par.init_mode = 1 if pose_init is None else (2 if force_init_solve else 0)
        keep = (split_scratch, lm_scratch)
        if par.init_mode:
            init = sv.init_solver
            par.rslm_lm = hip._lm_struct(init, fast_mode)
            par.rslm_points, par.rslm_proposals = int(init.num_points), int(init.num_proposals)
            if getattr(init.draw, '__func__', None) is RSLMSolver.draw:          # default sampler: device Philox
                inds = rot = None
            else:                                                               # overridden draw(): inject its output
                inds, rot = init.draw(w2d)
                inds, rot = inds.contiguous(), hip._f32c(rot, 'rot')
                assert inds.dtype == torch.int64 and inds.shape == (par.rslm_proposals, prob.B, par.rslm_points)
            if not hasattr(init, '_draw_seed'):
                init._draw_seed, init._draw_calls = int(torch.randint(0, 2 ** 62, (1,)).item()), 0
            counter = getattr(init, 'rng_counter', None)                        # device-side call counter (hipGraph replay)
            if counter is None:
                init._draw_calls += 1
            par.rslm_seed = init._draw_seed
            par.rslm_offset = 0 if counter is not None else init._draw_calls - 1
            par.rslm_offset_dev, par.rslm_inds, par.rslm_rot = _hip.ptr(counter), _hip.ptr(inds), _hip.ptr(rot)
            rs = hip.rslm_scratch(prob, par.rslm_proposals)
            par.rslm_scratch, par.rslm_scratch_bytes = _hip.ptr(rs), 0 if rs is None else rs.numel() * 4
            keep = (inds, rot, counter, split_scratch, rs, lm_scratch)
        delta = cost_fun.delta if isinstance(cost_fun.delta, torch.Tensor) else None
        # AdaptiveHuberPnPCost.set_param on THIS w2d (x2d detached, as the reference's callers do): delta's gradient reaches
        # w2d as one number per object, which the backward kernel adds in its epilogue -- the node then owes delta nothing,
        # and autograd is spared two elementwise launches and the (B,N,2) add of AccumulateGrad (EPROPNP_DELTA_FOLD=0: off)
        src, fold = getattr(cost_fun, '_delta_src', None), None
        # (delta.requires_grad: a threshold that set_param computed under no_grad is a constant, as in the reference -- nothing
        # to fold then)
        if src is not None and delta is not None and delta.requires_grad and torch.is_grad_enabled() and w2d.requires_grad \
                and os.environ.get('EPROPNP_DELTA_FOLD', '1') != '0' and src[0]() is delta and src[1]() is w2d and not src[4]:
            fold = (src[2], src[3])
            prob.fold_delta(*fold)          # (every backward built on `prob`, pose_opt_plus below included)
        # Device-side call counters (enable_graph_safe_rng) are advanced by the sampler's own launch when its last workgroup
        # retires (epropnp_amis_params.advance): no add kernel in the step.  Counters that do not sit in the layer's own
        # [self, initialiser, ticket] triple (an initialiser shared between layers, say) keep the in-stream add.
        bump_init = bool(par.init_mode and inds is None and counter is not None)
        bump_self = self.rng_counter is not None and noise is None
        pair = getattr(self, '_rng_pair', None)
        in_kernel = pair is not None and pair.numel() == 3 and (bump_self or bump_init) \
            and (not bump_init or counter.data_ptr() == pair[1:2].data_ptr())
        if in_kernel:
            first = pair[0:1] if bump_self else pair[1:2]
            par.amis.advance, par.amis.advance_ticket = _hip.ptr(first), _hip.ptr(pair[2:3])
            par.amis.advance_count = 2 if (bump_self and bump_init) else 1
        try:
            pose_opt, samples, logw, cost, cost_init, pose_opt_n, x3d_c, offset = hip.fused_monte_carlo(
                x3d, x2d, w2d, None if fold else delta, prob, pose_init, par, noise, bool(with_cost))
        except Exception:
            # A call that failed between its launches may have left the advance ticket part-way (some workgroups counted, the last
            # one never arrived): every later step would then miss or mis-time the counters' increment and REUSE its samples in
            # silence.  Put the ticket back to zero in stream order before the error travels on.
            if in_kernel:
                pair[2:3].zero_()
            raise
        del keep
        if not in_kernel:
            if bump_init:
                counter.add_(1)
            if bump_self:
                self.rng_counter.add_(1)
        pose_opt_plus = None
        if with_pose_opt_plus:      # differentiable Gauss-Newton step at pose_opt, in the solver frame, then denormalised
            if self.normalize:      # d/dx3d passes through the (detached) centring unchanged: differentiate w.r.t. x3d itself
                plus_n = hip.pose_opt_plus(x3d, x2d, w2d, delta, prob.with_points(x3d_c), pose_opt_n, sv.eps)
                pose_opt_plus = hip.shift_poses(plus_n, offset, -1.0)
            else:
                pose_opt_plus = hip.pose_opt_plus(x3d, x2d, w2d, delta, prob, pose_opt_n, sv.eps)
        return pose_opt, cost, pose_opt_plus, samples, logw, cost_init


class EProPnP4DoF(EProPnPBase):
    """Pose = [x, y, z, yaw] (yaw about the Y axis, radians).
    Proposals: position ~ multivariate Student-t (3 dof); yaw ~ 0.75 von Mises + 0.25 uniform."""

    dof = 4
    _t_default = [1.0, 1.0, 4.0]      # fallback factor of the translation proposal (epropnp.py:217,248)

    def allocate_buffer(self, num_obj, dtype=torch.float32, device=None):
        """-> trans_mode (K,B,3), trans_cov_tril (K,B,3,3), rot_mode (K,B,1), rot_kappa (K,B,1)"""
        return self._new(num_obj, dtype, device, (3,), (3, 3), (1,), (1,))

    def initial_fit(self, pose_opt, pose_cov, camera, trans_mode, trans_cov_tril, rot_mode, rot_kappa):
        trans_mode[0], rot_mode[0] = pose_opt[:, :3], pose_opt[:, 3:]
        trans_cov_tril[0] = proposals.cholesky_or_default(pose_cov[:, :3, :3], self._t_default)
        rot_kappa[0] = 0.33 / pose_cov[:, 3, 3, None].clamp(min=self.eps)

    @staticmethod
    def gen_new_distr(iter_id, trans_mode, trans_cov_tril, rot_mode, rot_kappa):
        return (proposals.MultivariateStudentT(3, trans_mode[iter_id], trans_cov_tril[iter_id]),
                proposals.VonMisesUniformMix(rot_mode[iter_id], rot_kappa[iter_id]))

    @staticmethod
    def gen_old_distr(iter_id, trans_mode, trans_cov_tril, rot_mode, rot_kappa):
        return (proposals.MultivariateStudentT(3, trans_mode[:iter_id, None], trans_cov_tril[:iter_id, None]),
                proposals.VonMisesUniformMix(rot_mode[:iter_id, None], rot_kappa[:iter_id, None]))

    def estimate_params(self, iter_id, pose_samples, pose_sample_logweights, trans_mode, trans_cov_tril, rot_mode, rot_kappa):
        w = torch.softmax(pose_sample_logweights, dim=0)
        trans_mode[iter_id + 1], cov = proposals.translation_moments(pose_samples, w)
        trans_cov_tril[iter_id + 1] = proposals.cholesky_or_default(cov, self._t_default)
        rot_mode[iter_id + 1], rot_kappa[iter_id + 1] = proposals.yaw_concentration(pose_samples, w, self.eps)


class EProPnP6DoF(EProPnPBase):
    """Pose = [x, y, z, w, i, j, k] with a unit quaternion.
    Proposals: position ~ multivariate Student-t (3 dof); orientation ~ angular central Gaussian on S^3."""

    dof = 6

    def __init__(self, *args, acg_mle_iter=3, acg_dispersion=0.001, **kwargs):
        super().__init__(*args, **kwargs)
        self.acg_mle_iter = acg_mle_iter
        self.acg_dispersion = acg_dispersion

    def allocate_buffer(self, num_obj, dtype=torch.float32, device=None):
        """-> trans_mode (K,B,3), trans_cov_tril (K,B,3,3), rot_cov_tril (K,B,4,4)"""
        return self._new(num_obj, dtype, device, (3,), (3, 3), (4, 4))

    def initial_fit(self, pose_opt, pose_cov, camera, trans_mode, trans_cov_tril, rot_cov_tril):
        trans_mode[0] = pose_opt[:, :3]
        trans_cov_tril[0] = proposals.cholesky_or_default(pose_cov[:, :3, :3])
        shape = proposals.acg_shape_from_laplace(pose_opt[:, 3:], pose_cov[:, 3:, 3:], camera.get_quaternion_transfrom_mat)
        rot_cov_tril[0] = proposals.acg_shape_factor(shape, self.acg_dispersion)

    @staticmethod
    def gen_new_distr(iter_id, trans_mode, trans_cov_tril, rot_cov_tril):
        return (proposals.MultivariateStudentT(3, trans_mode[iter_id], trans_cov_tril[iter_id]),
                proposals.AngularCentralGaussian(rot_cov_tril[iter_id]))

    @staticmethod
    def gen_old_distr(iter_id, trans_mode, trans_cov_tril, rot_cov_tril):
        return (proposals.MultivariateStudentT(3, trans_mode[:iter_id, None], trans_cov_tril[:iter_id, None]),
                proposals.AngularCentralGaussian(rot_cov_tril[:iter_id, None]))

    def estimate_params(self, iter_id, pose_samples, pose_sample_logweights, trans_mode, trans_cov_tril, rot_cov_tril):
        w = torch.softmax(pose_sample_logweights, dim=0)
        trans_mode[iter_id + 1], cov = proposals.translation_moments(pose_samples, w)
        trans_cov_tril[iter_id + 1] = proposals.cholesky_or_default(cov)
        shape = proposals.acg_shape_mle(pose_samples[..., 3:], w, self.acg_mle_iter, self.eps)
        rot_cov_tril[iter_id + 1] = proposals.acg_shape_factor(shape, self.acg_dispersion)
