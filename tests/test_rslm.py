"""Fused random-sample LM initialiser (csrc/rslm_kernel.hip, epropnp_rslm_solve) vs the composite path it replaces
(rslm_draw + gather + row-variant LM + evaluate_cost + argmin) on identical random draws.  The reference-generated
fixtures mc6_demo / mc4_rslm go through the same kernel in tests/test_api_dropin.py."""
import math

import pytest
import torch

import epropnp_oracle as orc
from helpers import make_layer_objects


def _solvers(dof, P, n, L, inds, rot):
    from epropnp.levenberg_marquardt import RSLMSolver
    s = RSLMSolver(dof=dof, num_points=n, num_proposals=P, num_iter=L)
    s.draw = lambda w2d: (inds.to(w2d.device), rot.to(w2d.device))
    return s


@pytest.mark.parametrize('dof,N,P,n,bounds,fast', [(4, 100, 20, 16, 'tensor', False), (6, 64, 16, 16, None, False),
                                                   (6, 130, 35, 8, 'tight', False), (4, 33, 4, 12, None, True)])
def test_fused_matches_composite_on_same_draws(backend, monkeypatch, dof, N, P, n, bounds, fast):
    from epropnp import functional as F
    B, L = 6, 3
    p = orc.make_problem(B, N, dof=dof, seed=40 + N, bounds=bounds)
    d, cam, cf = make_layer_objects(p, backend)
    g = torch.Generator().manual_seed(N)
    inds = F.rslm_draw(d['w2d'], P, n, seed=99, offset=3).cpu()
    if dof == 4:
        rot = torch.rand(P, B, generator=g) * (2 * math.pi)
    else:
        rot = torch.nn.functional.normalize(torch.randn(P, B, 4, generator=g), dim=-1)
    solver = _solvers(dof, P, n, L, inds, rot)
    monkeypatch.setenv('EPROPNP_RSLM_COMPOSITE', '1')
    pose_c, _, cost_c = solver.solve(d['x3d'], d['x2d'], d['w2d'], cam, cf, fast_mode=fast)
    monkeypatch.delenv('EPROPNP_RSLM_COMPOSITE')
    pose_f, _, cost_f = solver.solve(d['x3d'], d['x2d'], d['w2d'], cam, cf, fast_mode=fast)
    torch.testing.assert_close(cost_f.cpu(), cost_c.cpu(), rtol=2e-4, atol=1e-5)
    # same winning proposal unless two proposals tie to rounding; then the costs above already agree
    same = (pose_f - pose_c).abs().max(-1).values.cpu() < 1e-3
    assert int(same.sum()) >= B - 1, (pose_f, pose_c)
    # device-drawn indices are the rslm_draw stream: only the rotations are injected here
    prob = F.PnPProblem(d['x3d'], d['x2d'], d['w2d'], cam, cf, dof)
    pose_d, cost_d = F.rslm_solve(prob, P, n, L, seed=99, offset=3, inds=None, rot=rot.to(backend), fast_mode=fast)
    pose_i, cost_i = F.rslm_solve(prob, P, n, L, seed=99, offset=3, inds=inds.to(backend), rot=rot.to(backend), fast_mode=fast)
    torch.testing.assert_close(cost_d, cost_i, rtol=0, atol=0)
    torch.testing.assert_close(pose_d, pose_i, rtol=0, atol=0)


@pytest.mark.parametrize('dof', [4, 6])
def test_device_draws_find_the_pose(backend, dof):
    """Production mode (everything drawn on the device): the initialiser recovers poses the plain LM start cannot,
    different calls draw different proposals, and the result is at least as good as the composite path's."""
    from epropnp.levenberg_marquardt import RSLMSolver
    B, N = 12, 96
    p = orc.make_problem(B, N, dof=dof, seed=7)
    d, cam, cf = make_layer_objects(p, backend)
    solver = RSLMSolver(dof=dof, num_points=16, num_proposals=64, num_iter=4)
    pose1, _, cost1 = solver.solve(d['x3d'], d['x2d'], d['w2d'], cam, cf)
    pose2, _, cost2 = solver.solve(d['x3d'], d['x2d'], d['w2d'], cam, cf)
    assert (pose1 - pose2).abs().max() > 0          # fresh draws per call
    # cost at the ground truth as the yard-stick: the best of 64 proposals gets within a small factor of it
    from epropnp import functional as F
    prob = F.PnPProblem(d['x3d'], d['x2d'], d['w2d'], cam, cf, dof)
    cost_gt = F.evaluate_cost(prob, d['pose_gt'])
    good = (torch.minimum(cost1, cost2) < 20 * cost_gt + 1e-3)
    assert int(good.sum()) >= B - 2, (cost1, cost2, cost_gt)
    if dof == 6:
        assert (pose1[:, 3:].norm(dim=-1) - 1).abs().max() < 1e-4
    # returned cost is the full-set cost of the returned pose
    torch.testing.assert_close(F.evaluate_cost(prob, pose1), cost1, rtol=1e-4, atol=1e-5)


def test_shapes_outside_the_fused_kernel_fall_back(backend):
    """num_points > 16 or N > 512 run the composite path (same API, same result type); bad arguments to the C entry
    point itself are errors."""
    from epropnp import functional as F
    from epropnp.levenberg_marquardt import RSLMSolver
    p = orc.make_problem(2, 600, dof=6, seed=1)
    d, cam, cf = make_layer_objects(p, backend)
    pose, _, cost = RSLMSolver(dof=6, num_points=16, num_proposals=4, num_iter=2).solve(d['x3d'], d['x2d'], d['w2d'], cam, cf)
    assert pose.shape == (2, 7) and cost.shape == (2,) and bool(torch.isfinite(cost).all())
    prob = F.PnPProblem(d['x3d'], d['x2d'], d['w2d'], cam, cf, 6)
    with pytest.raises(RuntimeError):
        F.rslm_solve(prob, 4, 16, 2)            # N = 600 > 512
    with pytest.raises(RuntimeError):
        F.rslm_solve(F.PnPProblem(d['x3d'][:, :64], d['x2d'][:, :64], d['w2d'][:, :64], cam, cf, 6), 4, 17, 2)
