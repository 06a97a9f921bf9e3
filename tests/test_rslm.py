"""Fused random-sample LM initialiser (csrc/rslm_kernel.hip, epropnp_rslm_solve) vs the composite path it replaces
(rslm_draw + gather + row-variant LM + evaluate_cost + argmin) on identical random draws.  The reference-generated
fixtures mc6_demo / mc4_rslm go through the same kernel in tests/test_api_dropin.py."""
import math

import pytest
import torch

import epropnp_oracle as orc
from helpers import make_layer_objects, set_tune


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
    set_tune(monkeypatch, rslm_composite=True)
    pose_c, _, cost_c = solver.solve(d['x3d'], d['x2d'], d['w2d'], cam, cf, fast_mode=fast)
    set_tune(monkeypatch)
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


@pytest.mark.parametrize('dof,N,P,n,bounds', [(4, 128, 64, 16, 'tensor'), (6, 96, 16, 16, None), (6, 200, 24, 8, 'tight')])
def test_fused_rslm_matches_oracle_on_injected_draws(backend, dof, N, P, n, bounds):
    """The one-launch initialiser against the ORACLE's restatement of RSLMSolver.solve (levenberg_marquardt.py:283-353,
    pinned to the reference by the fixtures mc6_demo / mc4_rslm / mc4_det) on the same injected sub-sample indices and
    initial rotations: the minimum full-set cost agrees; the winning pose agrees wherever the oracle's best and
    second-best proposals are not tied to rounding."""
    from epropnp import functional as F
    B, L = 10, 3
    prob = orc.make_problem(B, N, dof, seed=70 + N, bounds=bounds)
    rn = orc.make_rslm_noise(prob, dof, n, P, seed=71)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    pose, cost = F.rslm_solve(hp, P, n, L, inds=rn['inds'].to(backend), rot=rn['rot'].float().to(backend))
    ocam = orc.Cam(prob['cam_mats'], 0.1, prob.get('lb'), prob.get('ub'))
    o_pose, o_cost = orc.rslm_solve(prob['x3d'], prob['x2d'], prob['w2d'], ocam, prob['delta'], rn['inds'], rn['rot'].float(), dof,
                                    num_iter=L)
    torch.testing.assert_close(cost.cpu(), o_cost, rtol=5e-4, atol=1e-5)
    same = (pose.cpu() - o_pose).abs().max(-1).values < 1e-3
    assert int(same.sum()) >= B - 1, (pose.cpu() - o_pose).abs().max(-1).values


def test_rslm_draw_inclusion_probabilities_match_multinomial(backend):
    """The whole draw, not just its first pick: the probability that point i is among the n indices of a row equals that of
    torch.multinomial(weights, n, replacement=False) -- the reference's sampler (levenberg_marquardt.py:305-308) -- for
    very unequal weights, where sequential sampling without replacement differs most from independent draws."""
    from epropnp import functional as F
    B, N, n = 2, 12, 5
    P = 40000 if backend.type == 'cuda' else 2500
    w = torch.tensor([[8., 4, 2, 1, 1, 1, .5, .5, .25, .25, .1, 0.], [1.] * 6 + [3.] * 6]).unsqueeze(-1).expand(B, N, 2).contiguous()
    inds = F.rslm_draw(w.to(backend), P, n, seed=17, offset=2).cpu()                     # (P,B,n)
    g = torch.Generator().manual_seed(0)
    M = 400000
    for b in range(B):
        ref = torch.multinomial(w[b, :, 0].expand(M, N), n, replacement=False, generator=g)
        p_ref = torch.bincount(ref.flatten(), minlength=N).double() / M
        p_got = torch.bincount(inds[:, b].flatten(), minlength=N).double() / P
        sigma = (p_ref * (1 - p_ref) / P).sqrt()
        assert ((p_got - p_ref).abs() <= 5 * sigma + 3.0 / P).all(), (p_got, p_ref)
        # and the order statistics of the sequential draw: P(first pick = i, second pick = j) for the two heaviest points
        i, j = (0, 1) if b == 0 else (6, 7)
        pij_ref = ((ref[:, 0] == i) & (ref[:, 1] == j)).double().mean()
        pij_got = ((inds[:, b, 0] == i) & (inds[:, b, 1] == j)).double().mean()
        assert abs(pij_got - pij_ref) <= 5 * (pij_ref * (1 - pij_ref) / P).sqrt() + 3.0 / P
    assert not (inds[:, 0] == 11).any()                                                  # zero weight: never drawn
