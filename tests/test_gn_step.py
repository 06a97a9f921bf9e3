"""Fused Gauss-Newton step (csrc/gn_step_kernel.hip) vs the oracle's gn_step + torch autograd in fp64
(reference: epropnp/levenberg_marquardt.py:243-253)."""
import pytest
import torch

import epropnp_oracle as orc
from helpers import make_layer_objects


def _oracle(p, pose, dof, gvec, dtype=torch.float64):
    q = {k: (v.to(dtype) if isinstance(v, torch.Tensor) else v) for k, v in p.items()}
    leaves = {k: q[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d', 'delta')}
    cam = orc.Cam(q['cam_mats'], 0.1, q.get('lb'), q.get('ub'))
    step = orc.gn_step(leaves['x3d'], leaves['x2d'], leaves['w2d'], pose.to(dtype), cam, leaves['delta'])
    (step * gvec.to(dtype)).sum().backward()
    return step.detach(), {k: v.grad for k, v in leaves.items()}


@pytest.mark.parametrize('dof,N,bounds', [(6, 64, None), (6, 200, 'tight'), (6, 512, 'tensor'), (4, 37, None),
                                          (4, 300, 'tight'), (6, 9, None)])
def test_gn_step_forward_backward(backend, dof, N, bounds):
    from epropnp import functional as F
    B = 5
    p = orc.make_problem(B, N, dof=dof, seed=11 + N, bounds=bounds)
    pose = p['pose_init']
    g = torch.Generator().manual_seed(3)
    gvec = torch.randn(B, dof, generator=g)
    step64, grad64 = _oracle(p, pose, dof, gvec)

    d, cam, cf = make_layer_objects(p, backend)
    leaves = {k: d[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d', 'delta')}
    cf.delta = leaves['delta']
    prob = F.PnPProblem(leaves['x3d'], leaves['x2d'], leaves['w2d'], cam, cf, dof)
    step = F.gn_step(leaves['x3d'], leaves['x2d'], leaves['w2d'], leaves['delta'], prob, pose.to(backend), 1e-5)
    (step * gvec.to(backend)).sum().backward()

    # fp32 accumulation of 2N-term sums, solve with cond(JtJ) ~ 1e3..1e5
    scale = step64.abs().amax(-1, keepdim=True)
    assert ((step.detach().cpu() - step64).abs() / scale).max() < 5e-4
    for k in ('x3d', 'x2d', 'w2d', 'delta'):
        ref = grad64[k]
        got = leaves[k].grad.detach().cpu().double()
        if k == 'delta':
            tol = 2e-3 * ref.abs().max().clamp(min=1e-12) + 1e-9
            assert ((got - ref).abs() <= tol).all(), (k, (got - ref).abs().max(), ref.abs().max())
            continue
        den = ref.reshape(B, -1).abs().amax(-1).clamp(min=1e-12).reshape(B, 1, 1)
        err = ((got - ref).abs() / den).max()
        assert err < 2e-3, (k, err)


def _graph_has(fn, name, seen=None):
    seen = set() if seen is None else seen
    if fn is None or fn in seen:
        return False
    seen.add(fn)
    return name in type(fn).__name__ or name in fn.name() or any(_graph_has(f, name, seen) for f, _ in fn.next_functions)


def test_lmsolver_gn_step_uses_fused_kernel(backend):
    """LMSolver.forward(with_pose_opt_plus=True) routes through the fused kernel and matches the composite."""
    from epropnp.levenberg_marquardt import LMSolver
    dof, B, N = 6, 4, 96
    p = orc.make_problem(B, N, dof=dof, seed=5)
    d, cam, cf = make_layer_objects(p, backend)
    solver = LMSolver(dof=dof, num_iter=5)
    x3d, x2d, w2d = (d[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    pose_opt, _, cost, plus = solver(x3d, x2d, w2d, cam, cf, with_pose_opt_plus=True, pose_init=d['pose_init'],
                                     with_cost=True)
    assert plus.requires_grad and (_graph_has(plus.grad_fn, 'PoseOptPlus') or _graph_has(plus.grad_fn, 'GnStep'))   # ctypes / C++ node
    plus.square().sum().backward()
    cam64 = orc.Cam(p['cam_mats'].double(), 0.1)
    leaves = [p[k].double().clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d')]
    po = pose_opt.detach().cpu().double()
    ref = orc.pose_add(po, orc.gn_step(*leaves, po, cam64, p["delta"].double()))
    ref.square().sum().backward()
    assert (plus.detach().cpu() - ref.detach()).abs().max() < 1e-4
    for got, want in zip((x3d, x2d, w2d), leaves):
        den = want.grad.abs().max()
        assert ((got.grad.cpu() - want.grad).abs() / den).max() < 5e-3


@pytest.mark.parametrize('dof,bounds', [(6, None), (6, 'tight'), (4, 'tensor')])
def test_pose_opt_plus_kernel_matches_step_then_pose_add(backend, dof, bounds):
    """The fused pose_opt_plus (step + pose_add, adjoint of pose_add inside the backward kernel) against the two-stage
    path: gn_step kernel followed by the PyTorch LMSolver.pose_add, values and input gradients."""
    from epropnp import functional as F
    from epropnp.levenberg_marquardt import LMSolver
    B, N = 4, 80
    p = orc.make_problem(B, N, dof=dof, seed=60 + dof, bounds=bounds)
    d, cam, cf = make_layer_objects(p, backend)
    g = torch.Generator().manual_seed(2)
    up = torch.randn(B, 7 if dof == 6 else 4, generator=g).to(backend)
    pose = d['pose_init']
    solver = LMSolver(dof=dof, num_iter=1)
    outs = []
    for fused in (True, False):
        leaves = {k: d[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d', 'delta')}
        cf.delta = leaves['delta']
        prob = F.PnPProblem(leaves['x3d'], leaves['x2d'], leaves['w2d'], cam, cf, dof)
        if fused:
            plus = F.pose_opt_plus(leaves['x3d'], leaves['x2d'], leaves['w2d'], leaves['delta'], prob, pose, 1e-5)
        else:
            step = F.gn_step(leaves['x3d'], leaves['x2d'], leaves['w2d'], leaves['delta'], prob, pose, 1e-5)
            plus = solver.pose_add(pose, step, cam)
        (plus * up).sum().backward()
        outs.append((plus.detach(), {k: v.grad for k, v in leaves.items()}))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6)
    for k in ('x3d', 'x2d', 'w2d', 'delta'):
        a, b = outs[0][1][k], outs[1][1][k]
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max()))
