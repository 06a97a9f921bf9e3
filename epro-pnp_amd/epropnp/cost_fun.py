"""Huber-robustified reprojection cost (fixed or adaptive threshold).

API mirror of the reference's epropnp/cost_fun.py.  `compute` is the framework-level PyTorch statement used by
evaluate_pnp and the differentiable Gauss-Newton step; the hot path evaluates the same cost inside the HIP kernels.
"""
import torch

from ._batched import BatchedParams


def huber_kernel(s_sqrt, delta):
    """rho(s)/2 for s = s_sqrt^2: quadratic inside `delta`, linear outside."""
    quad = 0.5 * s_sqrt * s_sqrt
    lin = delta * s_sqrt - 0.5 * delta * delta
    return torch.where(s_sqrt <= delta, quad, lin)


def huber_d_kernel(s_sqrt, delta, eps: float = 1e-10):
    """sqrt(rho'(s)) = sqrt(min(delta / s_sqrt, 1)): scale that turns the residual into its robustified form."""
    if s_sqrt.requires_grad or delta.requires_grad:
        return (delta.clamp(min=eps).sqrt() * s_sqrt.clamp(min=eps).rsqrt()).clamp(max=1.0)
    return (delta / s_sqrt.clamp(min=eps)).clamp(max=1.0).sqrt()


class HuberPnPCost(BatchedParams):
    _batched = {'delta': 0}
    _plain = ('eps',)

    def __init__(self, delta=1.0, eps=1e-10):
        self.eps = eps
        self.delta = delta

    def set_param(self, *args, **kwargs):
        pass

    def compute(self, x2d_proj, x2d, w2d, jac_cam=None, out_residual=False, out_cost=False, out_jacobian=False):
        """x2d_proj/x2d/w2d (*,n,2), jac_cam (*,n,2,d).  out_* : False | True | preallocated tensor.
        -> residual (*,2n) | None, cost (*) | None, jacobian (*,2n,d) | None"""
        lead, n = x2d_proj.shape[:-2], x2d_proj.size(-2)
        delta = self.delta if isinstance(self.delta, torch.Tensor) else x2d.new_tensor(self.delta)
        delta = delta[..., None]
        weighted = (x2d_proj - x2d) * w2d
        norm = weighted.norm(dim=-1)

        cost = None
        if out_cost is not False:
            cost = huber_kernel(norm, delta).sum(dim=-1)
            if isinstance(out_cost, torch.Tensor):
                out_cost.copy_(cost)
                cost = out_cost
        residual = jacobian = None
        if out_residual is not False or out_jacobian is not False:
            scale = huber_d_kernel(norm, delta, eps=self.eps).unsqueeze(-1)
            if out_residual is not False:
                residual = (weighted * scale).reshape(*lead, n * 2)
                if isinstance(out_residual, torch.Tensor):
                    out_residual.view(*lead, n * 2).copy_(residual)
                    residual = out_residual.view(*lead, n * 2)
            if out_jacobian is not False:
                assert jac_cam is not None
                d = jac_cam.size(-1)
                jacobian = (jac_cam * (w2d * scale).unsqueeze(-1)).reshape(*lead, n * 2, d)
                if isinstance(out_jacobian, torch.Tensor):
                    out_jacobian.view(*lead, n * 2, d).copy_(jacobian)
                    jacobian = out_jacobian.view(*lead, n * 2, d)
        return residual, cost, jacobian


class AdaptiveHuberPnPCost(HuberPnPCost):
    """delta_b = mean(w2d_b) * sqrt(sum_xy var_n(x2d_b)) * relative_delta, recomputed by set_param (differentiable)."""

    _plain = ('eps', 'relative_delta')

    def __init__(self, delta=None, relative_delta=0.5, eps=1e-10):
        self.delta = delta
        self.relative_delta = relative_delta
        self.eps = eps
        self._delta_src = None

    def set_param(self, x2d, w2d):
        from . import _hip
        if x2d.dim() == 3 and x2d.shape == w2d.shape and x2d.size(0) > 0 and x2d.size(1) > 1 \
                and _hip.on_hip_path(x2d, w2d):
            from .functional import adaptive_delta        # one fused pass (fwd) instead of ~8 ATen launches
            self.delta, stats = adaptive_delta(x2d, w2d, self.relative_delta)
            # what the layer needs to add this threshold's gradient to grad_w2d inside its own backward kernel when it is
            # handed the SAME w2d (epropnp.py:_fused_forward; include/epropnp_hip.h: epropnp_problem.delta_stats)
            import weakref      # (weak: neither the caller's w2d nor delta's autograd graph is kept alive from here)
            self._delta_src = (weakref.ref(self.delta), weakref.ref(w2d), stats, float(self.relative_delta),
                               bool(x2d.requires_grad))
            return
        self._delta_src = None
        spread = torch.var(x2d, dim=-2).sum(dim=-1).sqrt()
        self.delta = w2d.mean(dim=(-2, -1)) * spread * self.relative_delta
