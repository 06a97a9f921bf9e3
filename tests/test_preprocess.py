"""Fused correspondence pre-processing (csrc/eval_kernels.hip: prepare_{forward,backward}_kernel) vs its PyTorch
definition, which restates EPro-PnP-6DoF/lib/train.py:141,163-166 and EPro-PnP-Det deform_pnp_head.py:418-423,873-875."""
import math

import pytest
import torch


@pytest.mark.parametrize('mode', ['softmax', 'mean_exp'])
@pytest.mark.parametrize('with_x3d,with_scale,N', [(True, True, 200), (False, True, 64), (True, False, 33)])
def test_prepare_matches_torch(backend, mode, with_x3d, with_scale, N):
    from epropnp.preprocess import _reference, prepare_correspondences
    g = torch.Generator().manual_seed(N)
    B = 5
    noc = (torch.rand(B, N, 3, generator=g) - 0.5) if with_x3d else None
    dim = (torch.rand(B, 3, generator=g) + 0.5) if with_x3d else None
    logits = torch.randn(B, N, 2, generator=g) * 2
    scale = (torch.rand(B, 2, generator=g) * 3 + 0.1) if with_scale else None
    up_w, up_x = torch.randn(B, N, 2, generator=g), torch.randn(B, N, 3, generator=g)

    def run(fn, dev, dtype):
        ins = [None if t is None else t.to(dev, dtype).requires_grad_(True) for t in (noc, dim, logits, scale)]
        x3d, w2d = fn(*ins, mode)
        loss = (w2d * up_w.to(dev, dtype)).sum()
        if x3d is not None:
            loss = loss + (x3d * up_x.to(dev, dtype)).sum()
        loss.backward()
        return x3d, w2d, [None if t is None else t.grad for t in ins]

    x_ref, w_ref, g_ref = run(_reference, 'cpu', torch.float64)
    x, w, gr = run(prepare_correspondences, backend, torch.float32)
    torch.testing.assert_close(w.detach().cpu().double(), w_ref.detach(), rtol=2e-5, atol=1e-7)
    if with_x3d:
        torch.testing.assert_close(x.detach().cpu().double(), x_ref.detach(), rtol=1e-6, atol=1e-7)
    for a, b in zip(gr, g_ref):
        assert (a is None) == (b is None)
        if a is not None:
            torch.testing.assert_close(a.cpu().double(), b, rtol=2e-4, atol=2e-5 * float(b.abs().max()))


def test_weights_sum_and_regularization_losses(backend):
    from epropnp.preprocess import derivative_regularization_6dof, prepare_correspondences
    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(4, 128, 2, generator=g) * 5).to(backend)
    scale = torch.tensor([[2.0, 3.0]]).expand(4, 2).contiguous().to(backend)
    _, w = prepare_correspondences(None, None, logits, scale, 'softmax')
    torch.testing.assert_close(w.sum(dim=1), scale, rtol=1e-5, atol=1e-6)
    _, w = prepare_correspondences(None, None, logits, None, 'mean_exp')
    ref = (logits - logits.mean(1, keepdim=True) - math.log(128)).exp()
    torch.testing.assert_close(w, ref, rtol=2e-5, atol=1e-8)
    pose = torch.tensor([[0.1, 0.0, 5.0, 1.0, 0.0, 0.0, 0.0]])
    gt = torch.tensor([[0.1, 0.02, 5.0, 0.0, 1.0, 0.0, 0.0]])
    lt, lr = derivative_regularization_6dof(pose, gt)
    assert abs(lt.item() - 0.5 * 0.02 ** 2 / 0.05) < 1e-7 and abs(lr.item() - 2.0) < 1e-6
