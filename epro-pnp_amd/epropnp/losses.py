"""Monte-Carlo pose loss (the KL-divergence loss of EPro-PnP), kept in PyTorch as in the reference.

Covers both reference variants with one class:
  * EPro-PnP-6DoF/lib/models/monte_carlo_pose_loss.py:8-35   forward(logweights, cost_target, norm_factor) -> scalar
  * EPro-PnP-Det/epropnp_det/models/losses/monte_carlo_pose_loss.py:30-66   + weight / avg_factor / reduction /
    loss_weight, and the world-mean of `norm_factor` (mmdet `reduce_mean`) when torch.distributed is initialised.
The loss is what seeds the backward of the HIP AMIS kernel: d loss / d logweights = softmax over samples.
"""
import torch
import torch.nn as nn


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
        if self.training:
            with torch.no_grad():
                if hasattr(norm_factor, 'world_mean'):      # sharding.ObjectExchange: the scalar rode in the step's ONE
                    nf = norm_factor.world_mean().to(self.norm_factor.device)       # collective (no all-reduce here)
                else:
                    nf = _world_mean(torch.as_tensor(norm_factor, dtype=torch.float, device=self.norm_factor.device))
                self.norm_factor.mul_(1 - self.momentum).add_(self.momentum * nf)
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
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
