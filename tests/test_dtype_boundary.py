"""Floating dtypes other than fp32 at the tops of the API (epropnp/_dtype.py).

The reference is plain PyTorch and runs in the dtype of its inputs (fp64 in a gradcheck-style caller); the kernels compute in fp32, so
`LMSolver.forward / solve`, `RSLMSolver.solve` and `monte_carlo_forward` cast on the way in and the outputs / gradients back.  What is
checked: an fp64 (or bf16) call returns tensors and gradients of the caller's dtype whose values are the fp32 call's on the same
numbers, and the caller's camera / cost-function objects are not modified."""
import pytest
import torch

import epropnp_oracle as orc
from helpers import make_layer_objects, pack_noise


def _layer(dof, S=64, K=4, rslm=False):
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver, RSLMSolver
    init = RSLMSolver(dof=dof, num_points=8, num_proposals=16, num_iter=3) if rslm else None
    cls = EProPnP6DoF if dof == 6 else EProPnP4DoF
    return cls(mc_samples=S, num_iter=K, solver=LMSolver(dof=dof, num_iter=5, init_solver=init), seed=3)


def _objects(prob, dev, dtype):
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    p, _, _ = make_layer_objects(prob, dev)
    p = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in p.items()}
    cam = PerspectiveCamera(cam_mats=p['cam_mats'], z_min=0.1, lb=p.get('lb'), ub=p.get('ub'))
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)
    return p, cam, cf


@pytest.mark.parametrize('dof,dtype', [(6, torch.float64), (4, torch.float64), (6, torch.bfloat16)])
def test_monte_carlo_forward_in_the_callers_dtype(backend, dof, dtype):
    B, N, S, K = 6, 64, 64, 4
    prob = orc.make_problem(B, N, dof, seed=17, relative_delta=0.5)
    noise = pack_noise(orc.make_noise(B, S, K, dof, seed=2), dof).to(backend)
    outs, grads = {}, {}
    p32, _, cf32 = _objects({k: (v.to(dtype).float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in prob.items()},
                            backend, torch.float32)                     # the SAME numbers in both runs: rounded to `dtype` first
    cf32.set_param(p32['x2d'], p32['w2d'])
    for dt in (torch.float32, dtype):
        p, cam, cf = _objects(p32, backend, dt)
        x3d, x2d, w2d = (p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
        cf.set_param(x2d.detach(), w2d.detach())           # in the caller's dtype: the PyTorch composite beside the fp32 kernel
        assert cf.delta.dtype == dt and (cf.delta.float() - cf32.delta).abs().max() <= (1e-5 if dt != torch.bfloat16 else 2e-2) * cf32.delta.max()
        cf.delta = cf32.delta.to(dt)
        delta_before = cf.delta
        out = _layer(dof, S, K).monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=p['pose_init'], force_init_solve=False,
                                                    with_pose_opt_plus=True, with_cost=True, noise=noise.to(dt))
        assert all(o.dtype == dt for o in out), [o.dtype for o in out]
        assert cf.delta is delta_before and cf.delta.dtype == dt and cam.cam_mats.dtype == dt      # caller's objects untouched
        pose_opt, cost, plus, samples, logw, cost_init = out
        (cost_init + torch.logsumexp(logw.float(), dim=0).to(dt)).mean().add(0.1 * plus.sum(-1).mean()).backward()
        assert all(t.grad is not None and t.grad.dtype == dt for t in (x3d, x2d, w2d))
        outs[dt] = [o.detach().double().cpu() for o in out]
        grads[dt] = [t.grad.double().cpu() for t in (x3d, x2d, w2d)]
    # the fp32 run and the cast run compute the same fp32 numbers; they differ by the rounding of the OUTPUTS to `dtype`
    # (none for fp64) and, below fp32, by cost_fun.set_param running in the caller's dtype
    tol = 0.0 if dtype == torch.float64 else 0.1
    for a, b in zip(outs[torch.float32] + grads[torch.float32], outs[dtype] + grads[dtype]):
        if tol == 0.0:
            assert torch.equal(a, b), float((a - b).abs().max())
        else:
            assert torch.isfinite(b).all() and (a - b).abs().max() <= tol * (1.0 + a.abs().max())


def test_solver_entry_points_in_fp64(backend):
    prob = orc.make_problem(5, 48, 6, seed=9, relative_delta=0.5)
    res = {}
    for dt in (torch.float32, torch.float64):
        p, cam, cf = _objects(prob, backend, dt)
        cf.set_param(p['x2d'].float(), p['w2d'].float())
        cf.delta = cf.delta.to(dt)                  # one threshold for both runs (set_param in fp64 is the composite: last-bit different)
        layer = _layer(6, rslm=True)
        g = torch.Generator().manual_seed(5)
        inds = torch.stack([torch.stack([torch.randperm(48, generator=g)[:8] for _ in range(5)]) for _ in range(16)])
        rot = torch.randn(16, 5, 4, generator=g)
        layer.solver.init_solver.draw = lambda w2d: (inds.to(w2d.device), rot.to(w2d.device))
        pose_opt, pose_cov, cost, plus = layer(p['x3d'], p['x2d'], p['w2d'], cam, cf, with_pose_cov=True, with_cost=True,
                                               with_pose_opt_plus=True)          # LMSolver.forward through the RSLM initialiser
        assert pose_opt.dtype == pose_cov.dtype == cost.dtype == plus.dtype == dt
        start, _, start_cost = layer.solver.init_solver.solve(p['x3d'], p['x2d'], p['w2d'], cam, cf, with_cost=True)
        assert start.dtype == start_cost.dtype == dt
        solo = layer.solver.solve(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'], with_cost=True)
        assert solo[0].dtype == solo[2].dtype == dt and solo[1] is None
        res[dt] = [t.double().cpu() for t in (pose_opt, pose_cov, cost, plus, start, start_cost, solo[0], solo[2])]
    for a, b in zip(res[torch.float32], res[torch.float64]):
        assert torch.equal(a, b)
    # empty batch keeps the dtype
    z = _layer(6).solver.solve(torch.zeros(0, 8, 3, dtype=torch.float64, device=backend), torch.zeros(0, 8, 2, dtype=torch.float64, device=backend),
                               torch.zeros(0, 8, 2, dtype=torch.float64, device=backend), cam, cf,
                               pose_init=torch.zeros(0, 7, dtype=torch.float64, device=backend), with_cost=True)
    assert z[0].dtype == torch.float64 and z[0].shape == (0, 7)
