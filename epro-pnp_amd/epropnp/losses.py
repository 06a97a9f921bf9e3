"""Monte-Carlo pose loss (the KL-divergence loss of EPro-PnP), kept in PyTorch as in the reference.

Covers both reference variants with one class:
  * EPro-PnP-6DoF/lib/models/monte_carlo_pose_loss.py:8-35   forward(logweights, cost_target, norm_factor) -> scalar
  * EPro-PnP-Det/epropnp_det/models/losses/monte_carlo_pose_loss.py:30-66   + weight / avg_factor / reduction /
    loss_weight, and the world-mean of `norm_factor` (mmdet `reduce_mean`) when torch.distributed is initialised.
The loss is what seeds the backward of the HIP AMIS kernel: d loss / d logweights = softmax over samples.
"""
import os

import torch
import torch.nn as nn

from . import _hip


def monte_carlo_pose_loss(pose_sample_logweights, cost_target):
    """(S,B), (B,) -> per-object loss (B,): cost_target + logsumexp_S(logweights); NaN -> 0."""
    from . import _hip
    if pose_sample_logweights.dim() == 2 and pose_sample_logweights.numel() > 0 \
            and _hip.on_hip_path(pose_sample_logweights, cost_target):
        from .functional import mc_pose_loss              # fused forward / backward kernels
        return mc_pose_loss(pose_sample_logweights, cost_target)
    loss = cost_target + torch.logsumexp(pose_sample_logweights, dim=0)
    return torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)


def _world_mean(t):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return t
    t = t.clone()
    dist.all_reduce(t.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return t


class MonteCarloPoseLoss(nn.Module):

    def __init__(self, loss_weight=1.0, init_norm_factor=1.0, momentum=0.01, reduction='mean'):
        super().__init__()
        self.reduction = reduction
        self.loss_weight = loss_weight
        self.register_buffer('norm_factor', torch.tensor(init_norm_factor, dtype=torch.float))
        self.momentum = momentum

    def forward(self, pose_sample_logweights, cost_target, norm_factor, weight=None, avg_factor=None,
                reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        nf = slots = None
        scale = self._fused_scale(pose_sample_logweights, cost_target, weight, avg_factor, reduction)
        if self.training:
            with torch.no_grad():
                if hasattr(norm_factor, 'world_mean'):      # sharding.ObjectExchange: the scalar rode in the step's ONE
                    if scale is not None and hasattr(norm_factor, 'scalar_slots'):      # collective (no all-reduce here), and the
                        slots = norm_factor.scalar_slots()                               # fused reduce averages the ranks itself
                        nf = slots if slots.numel() > 1 else slots.reshape(1)
                    else:
                        nf = norm_factor.world_mean().to(self.norm_factor.device)
                else:
                    nf = _world_mean(torch.as_tensor(norm_factor, dtype=torch.float, device=self.norm_factor.device))
        if scale is not None:
            # the reduced loss as THREE launches (per-object loss, reduce + running estimate + scaling, backward) instead of the
            # ~15 elementwise / reduce launches of the statement below -- a quarter of the Det step when it is replayed from a
            # hipGraph (profiles/r03_det_loss_fused.txt); same value to rounding
            from .functional import mc_pose_loss_reduced
            return mc_pose_loss_reduced(pose_sample_logweights, cost_target, weight, scale, self.momentum,
                                        None if nf is None else (nf if slots is not None else nf.reshape(1)), self.norm_factor)
        if nf is not None:
            with torch.no_grad():
                self.norm_factor.mul_(1 - self.momentum).add_(self.momentum * nf)
        loss = monte_carlo_pose_loss(pose_sample_logweights, cost_target)
        if weight is not None:
            loss = loss * weight
        if avg_factor is None:
            loss = loss.mean() if reduction == 'mean' else (loss.sum() if reduction == 'sum' else loss)
        else:
            assert reduction in ('mean', 'none'), 'avg_factor can not be used with reduction="sum"'
            if reduction == 'mean':
                loss = loss.sum() / avg_factor
        return loss * (self.loss_weight / self.norm_factor)

    def _fused_scale(self, logw, cost_target, weight, avg_factor, reduction):
        """loss_weight / (what the sum over objects is divided by) when the fused kernels can produce the reduced loss, else
        None: fp32 (S,B) log-weights on the HIP path, a scalar reduction, a plain (B,) weight that needs no gradient, a plain
        number as avg_factor."""
        if os.environ.get('EPROPNP_LOSS_FUSED', '1') == '0':        # measurements: the composite statement
            return None
        if reduction not in ('mean', 'sum') or logw.dim() != 2 or logw.numel() == 0:
            return None
        tensors = [logw, self.norm_factor] + [t for t in (cost_target, weight) if t is not None]
        if not all(torch.is_tensor(t) for t in tensors) or not _hip.on_hip_path(*tensors):
            return None
        B = logw.shape[1]
        if cost_target is not None and cost_target.shape != (B,):
            return None
        if weight is not None and (weight.shape != (B,) or weight.requires_grad):
            return None
        if avg_factor is None:
            return self.loss_weight / B if reduction == 'mean' else float(self.loss_weight)
        if reduction != 'mean' or torch.is_tensor(avg_factor):
            return None
        return self.loss_weight / float(avg_factor)
