"""
oracle/preprocess_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the correspondence pre-processing the reference does in its callers:
  * EPro-PnP-6DoF/lib/train.py:141        x3d = noc * dim
  * EPro-PnP-6DoF/lib/train.py:143-162    pixel grid of the crop box, random pixel subset gathered from the dense maps
  * EPro-PnP-6DoF/lib/train.py:165        w2d = exp(w2d - mean_N(w2d) - log N) * scale      (mode 'mean_exp')
  * EPro-PnP-Det/.../deform_pnp_head.py:418-421,873-874   w2d = softmax_N(w2d) (* mask) * scale; x3d = noc * dim  ('softmax')
Pinned by oracle/make_golden.py:case_preprocess, which exec's those literal source lines of the reference on seeded
inputs and asserts equality; the inputs/outputs travel as tests/golden/prep_dense.npz / prep_det.npz.
"""
import math

import torch


def prepare_ref(noc, dim, logits, scale, mode):
    """noc (B,N,3)|None, dim (B,3)|None, logits (B,N,2), scale (B,2)|None -> x3d (B,N,3)|None, w2d (B,N,2)."""
    x3d = None if noc is None else noc * dim.unsqueeze(-2)
    if mode == 'softmax':
        w = logits.softmax(dim=-2)
    else:
        w = (logits - logits.mean(dim=-2, keepdim=True) - math.log(logits.size(-2))).exp()
    return x3d, (w if scale is None else w * scale.unsqueeze(-2))


def box_grid_ref(c_box, s_box, out_res):
    """lib/train.py:143-145 -> (B,3) [wh_begin_x, wh_begin_y, wh_unit]."""
    s = s_box.to(torch.int64)
    wh_begin = c_box.to(torch.int64) - s[:, None] / 2.
    wh_unit = s.to(torch.float32) / out_res
    return torch.cat((wh_begin.to(torch.float32), wh_unit[:, None]), dim=1)


def prepare_dense_ref(noc_map, dim, logit_map, scale, box, inds, mode):
    """lib/train.py:141-166 on dense maps: noc_map (B,3,H,W)|None, logit_map (B,2,H,W), box (B,3), inds (B,N) int64
    -> x3d (B,N,3)|None, x2d (B,N,2), w2d (B,N,2)."""
    B, _, H, W = logit_map.shape
    ar_w = torch.arange(W, device=logit_map.device, dtype=torch.float32)
    ar_h = torch.arange(H, device=logit_map.device, dtype=torch.float32)
    y, x = torch.meshgrid(ar_h, ar_w, indexing='ij')
    box = box.to(torch.float32)
    x2d = torch.stack((box[:, 0, None, None] + x * box[:, 2, None, None],
                       box[:, 1, None, None] + y * box[:, 2, None, None]), dim=1)             # (B,2,H,W)
    bi = torch.arange(B, device=logit_map.device)[:, None]
    pick = lambda m: m.flatten(2).transpose(-1, -2)[bi, inds]
    x3d = None if noc_map is None else pick(noc_map * dim[..., None, None])
    _, w2d = prepare_ref(None, None, pick(logit_map), scale, mode)
    return x3d, pick(x2d).to(logit_map.dtype), w2d
