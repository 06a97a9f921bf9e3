"""Correspondence pre-processing of the reference's training loops, as one fused op (forward + backward kernels).

The reference does this in its callers, not in `epropnp/`:
  * EPro-PnP-6DoF/lib/train.py:141,163-166   x3d = noc * dim;  w2d = exp(w2d - mean_N(w2d) - log N) * scale
    ("mean-normalised exp", the legacy alternative to softmax) -- mode='mean_exp'
  * EPro-PnP-Det/.../deform_pnp_head.py:418-423,873-875   w2d = softmax_N(w2d) * scale;  x3d = noc * dim -- mode='softmax'
  * EPro-PnP-6DoF/lib/train.py:143-162   the dense variant: pixel-grid x2d of the cropped box and a random subset of
    the out_res x out_res pixels gathered from the network's (bs, C, h, w) maps -- `prepare_dense_correspondences`
Provided here because it is the data format on the input side of the layer.  HIP kernels only:
tensors that are not fp32 on a HIP device raise (the PyTorch statement of the same maths is test infrastructure,
oracle/preprocess_oracle.py, pinned to the reference's source lines).
"""
import torch

MODES = {'softmax': 0, 'mean_exp': 1}


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
        gw2d = torch.full_like(lg, 0.0) if gw2d is None else gw2d.contiguous()
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
    assert w2d_logits.dim() == 3, 'w2d_logits must be (num_obj, num_pts, 2)'
    if not _hip.on_hip_path(*ts):
        raise RuntimeError('prepare_correspondences: fp32 tensors on a HIP device required (no CPU fallback)')
    if w2d_logits.size(0) == 0:        # empty batch: nothing to launch, keep autograd connectivity
        return (None if noc is None else noc * dim.unsqueeze(-2)), (w2d_logits if scale is None else w2d_logits * scale.unsqueeze(-2))
    out = _Prepare.apply(noc, dim, w2d_logits, scale, MODES[mode])
    return (None, out) if noc is None else out


def box_grid_params(c_box, s_box, out_res):
    """[wh_begin_x, wh_begin_y, wh_unit] per object from the crop centre / size, as EPro-PnP-6DoF/lib/train.py:143-145:
    s = s_box.long(); wh_begin = c_box.long() - s / 2.; wh_unit = s.float() / out_res  ->  (B,3) float32."""
    s = s_box.to(torch.int64)
    wh_begin = c_box.to(torch.int64) - s[:, None] / 2.
    wh_unit = s.to(torch.float32) / out_res
    return torch.cat((wh_begin.to(torch.float32), wh_unit[:, None]), dim=1)


class _PrepareDense(torch.autograd.Function):

    @staticmethod
    def forward(ctx, noc_map, dim, logit_map, scale, box, inds, mode):
        from . import _hip
        from .functional import _f32c
        lg = _f32c(logit_map, 'w2d logit map')
        B, _, H, W = lg.shape
        N = inds.shape[1]
        nc = None if noc_map is None else _f32c(noc_map, 'noc map')
        dm = None if dim is None else _f32c(dim, 'dim')
        sc = None if scale is None else _f32c(scale, 'scale')
        bx = _f32c(box, 'box')
        ix = inds.to(torch.int64).contiguous()
        x3d = None if nc is None else torch.empty((B, N, 3), dtype=torch.float32, device=lg.device)
        x2d = torch.empty((B, N, 2), dtype=torch.float32, device=lg.device)
        w2d = torch.empty((B, N, 2), dtype=torch.float32, device=lg.device)
        stats = torch.empty((B, 4), dtype=torch.float32, device=lg.device)
        _hip.call('epropnp_prepare_dense_forward', _hip.ptr(nc), _hip.ptr(dm), _hip.ptr(lg), _hip.ptr(sc), _hip.ptr(bx),
                  _hip.ptr(ix), B, N, H, W, int(mode), _hip.ptr(x3d), _hip.ptr(x2d), _hip.ptr(w2d), _hip.ptr(stats),
                  _hip.stream_of(lg))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(nc, dm, lg, sc, ix, stats)
        ctx.mode = int(mode)
        ctx.mark_non_differentiable(x2d)
        if x3d is None:
            return x2d, w2d
        return x3d, x2d, w2d

    @staticmethod
    def backward(ctx, *grads):
        from . import _hip
        nc, dm, lg, sc, ix, stats = ctx.saved_tensors
        gx3d, gw2d = (None, grads[1]) if nc is None else (grads[0], grads[2])
        if gx3d is None and gw2d is None:
            return (None,) * 7
        B, _, H, W = lg.shape
        N = ix.shape[1]
        gw2d = torch.full((B, N, 2), 0.0, dtype=torch.float32, device=lg.device) if gw2d is None else gw2d.contiguous()
        gx3d = None if gx3d is None else gx3d.contiguous()
        gl = torch.empty_like(lg)
        gnoc = None if gx3d is None else torch.empty_like(nc)
        gdim = None if gx3d is None else torch.empty_like(dm)
        gsc = None if sc is None else torch.empty_like(sc)
        _hip.call('epropnp_prepare_dense_backward', _hip.ptr(nc), _hip.ptr(dm), _hip.ptr(lg), _hip.ptr(sc), _hip.ptr(ix),
                  _hip.ptr(stats), _hip.ptr(gx3d), _hip.ptr(gw2d), B, N, H, W, ctx.mode, _hip.ptr(gnoc), _hip.ptr(gdim),
                  _hip.ptr(gl), _hip.ptr(gsc), _hip.stream_of(lg))
        return gnoc, gdim, gl, gsc, None, None, None


def prepare_dense_correspondences(noc_map, dim, w2d_logit_map, scale, box, sample_inds, mode='mean_exp'):
    """The 6-DoF training loop's correspondence set from the network's dense maps (EPro-PnP-6DoF/lib/train.py:141-166).

    noc_map (B,3,H,W) | None, dim (B,3) | None, w2d_logit_map (B,2,H,W), scale (B,2) | None,
    box (B,3) = `box_grid_params(c_box, s_box, out_res)`, sample_inds (B,N) int64 pixel indices (row * W + col; the
    reference draws `np.random.choice(H*W, N, replace=False)` per object)
      -> x3d (B,N,3) | None, x2d (B,N,2), w2d (B,N,2);  differentiable w.r.t. noc_map, dim, w2d_logit_map, scale."""
    assert mode in MODES, f'mode must be one of {tuple(MODES)}'
    assert (noc_map is None) == (dim is None)
    from . import _hip
    ts = [t for t in (noc_map, dim, w2d_logit_map, scale, box) if t is not None]
    assert w2d_logit_map.dim() == 4, 'w2d_logit_map must be (num_obj, 2, H, W)'
    if not _hip.on_hip_path(*ts) or sample_inds.device != w2d_logit_map.device:
        raise RuntimeError('prepare_dense_correspondences: fp32 tensors on a HIP device required (no CPU fallback)')
    if w2d_logit_map.size(0) == 0 or sample_inds.size(1) == 0:
        # empty detection batch / empty pixel subset: nothing to launch; correctly shaped empty tensors that stay connected to
        # autograd, as prepare_correspondences does (DDP callers rely on it, deform_pnp_head.py:913-920)
        B, N = w2d_logit_map.size(0), sample_inds.size(1)
        ix = sample_inds.to(torch.int64)
        gather = lambda m: m.flatten(2).gather(2, ix[:, None, :].expand(-1, m.size(1), -1)).transpose(1, 2)
        w2d = gather(w2d_logit_map) if scale is None else gather(w2d_logit_map) * scale.unsqueeze(-2)
        x2d = w2d_logit_map.new_zeros((B, N, 2))
        x3d = None if noc_map is None else gather(noc_map) * dim.unsqueeze(-2)
        return x3d, x2d, w2d
    out = _PrepareDense.apply(noc_map, dim, w2d_logit_map, scale, box, sample_inds, MODES[mode])
    return (None,) + tuple(out) if noc_map is None else out


def derivative_regularization_6dof(pose_opt_plus, pose_gt, beta=0.05):
    """The two losses fed by `pose_opt_plus` in EPro-PnP-6DoF/lib/train.py:185-193: smooth-L1 (Huber, threshold beta) on the
    translation error norm and 2 (1 - <q, q_gt>^2) on the orientation -> (loss_t, loss_r), each the batch mean."""
    err = (pose_opt_plus[:, :3] - pose_gt[:, :3]).norm(dim=-1)
    loss_t = torch.where(err < beta, 0.5 * err.square() / beta, err - 0.5 * beta).mean()
    dot = (pose_opt_plus[:, 3:] * pose_gt[:, 3:]).sum(dim=-1)
    return loss_t, ((1 - dot.square()) * 2).mean()
