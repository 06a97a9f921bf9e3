"""Correspondence pre-processing of the reference's training loops, as one fused op (forward + backward kernels).

The reference does this in its callers, not in `epropnp/`:
  * EPro-PnP-6DoF/lib/train.py:141,163-166   x3d = noc * dim;  w2d = exp(w2d - mean_N(w2d) - log N) * scale
    ("mean-normalised exp", the legacy alternative to softmax) -- mode='mean_exp'
  * EPro-PnP-Det/.../deform_pnp_head.py:418-423,873-875   w2d = softmax_N(w2d) * scale;  x3d = noc * dim -- mode='softmax'
Provided here because it is the data format on the input side of the layer; plain PyTorch for
tensors that are not on the HIP path.
"""
import math

import torch

MODES = {'softmax': 0, 'mean_exp': 1}


def _reference(noc, dim, logits, scale, mode):
    x3d = None if noc is None else noc * dim.unsqueeze(-2)
    if mode == 'softmax':
        w = logits.softmax(dim=-2)
    else:
        w = (logits - logits.mean(dim=-2, keepdim=True) - math.log(logits.size(-2))).exp()
    return x3d, (w if scale is None else w * scale.unsqueeze(-2))


class _Prepare(torch.autograd.Function):

    @staticmethod
    def forward(ctx, noc, dim, logits, scale, mode):
        from . import _hip
        from .functional import _f32c
        lg = _f32c(logits, 'w2d logits')
        B, N, _ = lg.shape
        nc = None if noc is None else _f32c(noc, 'noc')
        dm = None if dim is None else _f32c(dim, 'dim')
        sc = None if scale is None else _f32c(scale, 'scale')
        x3d = None if nc is None else torch.empty_like(nc)
        w2d = torch.empty_like(lg)
        stats = torch.empty((B, 4), dtype=torch.float32, device=lg.device)
        _hip.call('epropnp_prepare_forward', _hip.ptr(nc), _hip.ptr(dm), _hip.ptr(lg), _hip.ptr(sc), B, N, int(mode),
                  _hip.ptr(x3d), _hip.ptr(w2d), _hip.ptr(stats), _hip.stream_of(lg))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(nc, dm, lg, sc, stats)
        ctx.mode = int(mode)
        if x3d is None:
            return w2d
        return x3d, w2d

    @staticmethod
    def backward(ctx, *grads):
        from . import _hip
        nc, dm, lg, sc, stats = ctx.saved_tensors
        gx3d, gw2d = (None, grads[0]) if nc is None else grads
        if gx3d is None and gw2d is None:
            return None, None, None, None, None
        B, N, _ = lg.shape
        gw2d = torch.zeros_like(lg) if gw2d is None else gw2d.contiguous()
        gx3d = None if gx3d is None else gx3d.contiguous()
        gl = torch.empty_like(lg)
        gnoc = None if gx3d is None else torch.empty_like(nc)
        gdim = None if gx3d is None else torch.empty_like(dm)
        gsc = None if sc is None else torch.empty_like(sc)
        _hip.call('epropnp_prepare_backward', _hip.ptr(nc), _hip.ptr(dm), _hip.ptr(lg), _hip.ptr(sc), _hip.ptr(stats),
                  _hip.ptr(gx3d), _hip.ptr(gw2d), B, N, ctx.mode, _hip.ptr(gnoc), _hip.ptr(gdim), _hip.ptr(gl),
                  _hip.ptr(gsc), _hip.stream_of(lg))
        return gnoc, gdim, gl, gsc, None


def prepare_correspondences(noc, dim, w2d_logits, scale=None, mode='softmax'):
    """noc (B,N,3) | None, dim (B,3) | None, w2d_logits (B,N,2), scale (B,2) | None -> (x3d (B,N,3) | None, w2d (B,N,2)).

    x3d = noc * dim;  w2d = softmax over the N points (mode='softmax') or the mean-normalised exponential
    (mode='mean_exp') of the logits, times the per-object scale.  Differentiable w.r.t. all four inputs."""
    assert mode in MODES, f'mode must be one of {tuple(MODES)}'
    assert (noc is None) == (dim is None)
    from . import _hip
    ts = [t for t in (noc, dim, w2d_logits, scale) if t is not None]
    if w2d_logits.dim() == 3 and w2d_logits.size(0) > 0 and _hip.on_hip_path(*ts):
        out = _Prepare.apply(noc, dim, w2d_logits, scale, MODES[mode])
        return (None, out) if noc is None else out
    return _reference(noc, dim, w2d_logits, scale, mode)


def derivative_regularization_6dof(pose_opt_plus, pose_gt, beta=0.05):
    """The two losses fed by `pose_opt_plus` in EPro-PnP-6DoF/lib/train.py:185-193: smooth-L1 (Huber, threshold beta) on the
    translation error norm and 2 (1 - <q, q_gt>^2) on the orientation -> (loss_t, loss_r), each the batch mean."""
    err = (pose_opt_plus[:, :3] - pose_gt[:, :3]).norm(dim=-1)
    loss_t = torch.where(err < beta, 0.5 * err.square() / beta, err - 0.5 * beta).mean()
    dot = (pose_opt_plus[:, 3:] * pose_gt[:, 3:]).sum(dim=-1)
    return loss_t, ((1 - dot.square()) * 2).mean()
