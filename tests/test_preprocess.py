"""Fused correspondence pre-processing (csrc/eval_kernels.hip: prepare_{forward,backward}_kernel) vs the reference's own
outputs (fixtures prep_dense / prep_det: produced by exec'ing EPro-PnP-6DoF/lib/train.py:141-165 and EPro-PnP-Det
deform_pnp_head.py:418-421,873-874, oracle/make_golden.py:case_preprocess) and vs the restatement pinned by them
(oracle/preprocess_oracle.py) for gradients and other shapes."""
import math

import pytest
import torch


@pytest.mark.parametrize('mode', ['softmax', 'mean_exp'])
@pytest.mark.parametrize('with_x3d,with_scale,N', [(True, True, 200), (False, True, 64), (True, False, 33)])
def test_prepare_matches_torch(backend, mode, with_x3d, with_scale, N):
    from epropnp.preprocess import prepare_correspondences
    from preprocess_oracle import prepare_ref as _reference
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


@pytest.mark.parametrize('mode', ['mean_exp', 'softmax'])
@pytest.mark.parametrize('with_x3d,with_scale,H,W,N', [(True, True, 16, 16, 32), (False, True, 8, 12, 96), (True, False, 64, 64, 512)])
def test_prepare_dense_matches_reference_composite(backend, mode, with_x3d, with_scale, H, W, N):
    """Dense maps + sampled pixels (EPro-PnP-6DoF/lib/train.py:141-166): the fused gather against the PyTorch composite
    (meshgrid, flatten/transpose/index, mean-normalised exp) -- x2d bit-exact, the rest and all gradients to fp32 rounding."""
    import numpy as np
    from epropnp.preprocess import box_grid_params, prepare_dense_correspondences
    from preprocess_oracle import prepare_dense_ref as _reference_dense
    g = torch.Generator().manual_seed(H * W + N)
    B = 3
    noc = (torch.rand(B, 3, H, W, generator=g) - 0.5) if with_x3d else None
    dim = (torch.rand(B, 3, generator=g) + 0.5) if with_x3d else None
    logits = torch.randn(B, 2, H, W, generator=g) * 2
    scale = (torch.rand(B, 2, generator=g) * 3 + 0.1) if with_scale else None
    c_box = torch.tensor([[320.7, 240.2], [100.0, 400.9], [55.5, 60.5]])
    s_box = torch.tensor([128.9, 77.0, 301.2])
    box = box_grid_params(c_box, s_box, W)
    rs = np.random.RandomState(N)
    inds = torch.tensor(np.stack([rs.choice(H * W, size=N, replace=False) for _ in range(B)]), dtype=torch.int64)
    up_w, up_x = torch.randn(B, N, 2, generator=g), torch.randn(B, N, 3, generator=g)

    def run(fn, dev, dtype):
        ins = [None if t is None else t.clone().to(dev, dtype).requires_grad_(True) for t in (noc, dim, logits, scale)]
        x3d, x2d, w2d = fn(*ins, box.to(dev), inds.to(dev), mode)
        loss = (w2d * up_w.to(dev, dtype)).sum()
        if x3d is not None:
            loss = loss + (x3d * up_x.to(dev, dtype)).sum()
        loss.backward()
        return x3d, x2d, w2d, [None if t is None else t.grad for t in ins]

    x_ref, p_ref, w_ref, g_ref = run(_reference_dense, 'cpu', torch.float64)
    _, p_ref32, _, _ = run(_reference_dense, 'cpu', torch.float32)
    x, p, w, gr = run(prepare_dense_correspondences, backend, torch.float32)
    assert not p.requires_grad
    assert torch.equal(p.cpu(), p_ref32)                       # begin + index * unit, rounded as mul then add
    torch.testing.assert_close(w.detach().cpu().double(), w_ref.detach(), rtol=2e-5, atol=1e-7)
    if with_x3d:
        torch.testing.assert_close(x.detach().cpu().double(), x_ref.detach(), rtol=1e-6, atol=1e-7)
    for a, b in zip(gr, g_ref):
        assert (a is None) == (b is None)
        if a is not None:
            assert a.shape == b.shape
            torch.testing.assert_close(a.cpu().double(), b, rtol=2e-4, atol=2e-5 * float(b.abs().max()))


def test_prepare_dense_repeated_pixels_accumulate(backend):
    """`inds` may repeat a pixel (sampling with replacement): the map gradients accumulate like index_put_(accumulate=True)."""
    from epropnp.preprocess import prepare_dense_correspondences
    from preprocess_oracle import prepare_dense_ref as _reference_dense
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(2, 2, 4, 4, generator=g)
    box = torch.tensor([[0.0, 0.0, 1.0], [10.0, 20.0, 2.5]])
    inds = torch.tensor([[0, 5, 5, 15, 0, 0], [3, 3, 3, 3, 7, 8]])
    up = torch.randn(2, 6, 2, generator=g)
    outs = []
    for fn, dev in ((_reference_dense, 'cpu'), (prepare_dense_correspondences, backend)):
        lg = logits.detach().clone().to(dev).requires_grad_(True)
        _, x2d, w2d = fn(None, None, lg, None, box.to(dev), inds.to(dev), 'mean_exp')
        (w2d * up.to(dev)).sum().backward()
        outs.append((x2d.cpu(), w2d.detach().cpu(), lg.grad.cpu()))
    assert torch.equal(outs[0][0], outs[1][0])
    torch.testing.assert_close(outs[1][1], outs[0][1], rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(outs[1][2], outs[0][2], rtol=1e-4, atol=1e-6)


def test_prepare_matches_reference_fixtures(backend):
    """The literal reference lines, executed in the build container: 6-DoF dense gather + mean-normalised exp, and the
    detection head's softmax x scale."""
    from helpers import load_golden
    from epropnp.preprocess import box_grid_params, prepare_correspondences, prepare_dense_correspondences
    g = load_golden('prep_dense')
    box = box_grid_params(g['c_box'], g['s_box'], 64)
    assert torch.equal(box, g['box'])
    x3d, x2d, w2d = prepare_dense_correspondences(g['noc'].to(backend), g['dim'].to(backend), g['logit'].to(backend),
                                                  g['scale'].to(backend), box.to(backend), g['inds'].to(backend), 'mean_exp')
    assert torch.equal(x2d.cpu(), g['x2d'])                                   # mul then add, un-fused: bit-exact
    torch.testing.assert_close(x3d.cpu(), g['x3d'], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(w2d.cpu(), g['w2d'], rtol=2e-5, atol=1e-7)
    d = load_golden('prep_det')
    x3d, w2d = prepare_correspondences(d['noc'].to(backend), d['dim'].to(backend), d['logits'].to(backend),
                                       d['scale'].to(backend), 'softmax')
    torch.testing.assert_close(x3d.cpu(), d['x3d'], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(w2d.cpu(), d['w2d'], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize('B,N', [(0, 8), (3, 0)])
def test_prepare_dense_empty_batch_keeps_autograd(backend, B, N):
    """An empty detection batch / empty pixel subset returns correctly shaped empty tensors connected to autograd (the
    non-dense variant did already): DDP callers back-propagate through them (deform_pnp_head.py:913-920)."""
    from epropnp.preprocess import prepare_dense_correspondences
    H = W = 4
    noc = torch.randn(B, 3, H, W, device=backend, requires_grad=True)
    dim = torch.randn(B, 3, device=backend, requires_grad=True)
    lg = torch.randn(B, 2, H, W, device=backend, requires_grad=True)
    sc = torch.randn(B, 2, device=backend, requires_grad=True)
    box = torch.zeros(B, 3, device=backend)
    inds = torch.zeros(B, N, dtype=torch.int64, device=backend)
    x3d, x2d, w2d = prepare_dense_correspondences(noc, dim, lg, sc, box, inds)
    assert x3d.shape == (B, N, 3) and x2d.shape == (B, N, 2) and w2d.shape == (B, N, 2)
    (x3d.sum() + w2d.sum()).backward()
    assert all(t.grad is not None and t.grad.shape == t.shape for t in (noc, dim, lg, sc))


def test_prepare_refuses_cpu_tensors():
    import install as emu
    from epropnp.preprocess import prepare_correspondences
    emu.uninstall()
    with pytest.raises(RuntimeError, match='HIP device'):
        prepare_correspondences(None, None, torch.zeros(2, 8, 2), None, 'softmax')
