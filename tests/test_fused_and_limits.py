"""The one-call forward (csrc/mc_forward.hip), the device-side status word, N beyond the register-resident limit and a
non-default HuberPnPCost.eps."""
import os

import pytest
import torch

import epropnp_oracle as orc
from helpers import make_layer_objects, pack_noise, set_tune


def _layer(dof, S, K, L, normalize, rslm):
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver, RSLMSolver
    init = RSLMSolver(dof=dof, num_points=8, num_proposals=12, num_iter=3) if rslm else None
    return (EProPnP6DoF if dof == 6 else EProPnP4DoF)(mc_samples=S, num_iter=K, normalize=normalize,
                                                      solver=LMSolver(dof=dof, num_iter=L, init_solver=init))


@pytest.mark.parametrize('dof,normalize,rslm,plus,bounds', [(6, False, False, False, None), (4, True, True, True, 'tensor'),
                                                             (6, True, False, True, 'tight'), (4, False, True, False, None),
                                                             (6, False, True, True, None)])
def test_fused_forward_equals_composite(backend, monkeypatch, dof, normalize, rslm, plus, bounds):
    """epropnp_monte_carlo_forward enqueues the same kernels the separate calls do: outputs and input gradients are
    bit-identical to the composite path (EPROPNP_TUNE=no_fused_forward), with and without pnp_normalize, RSLM
    initialisation (injected draws), pose_opt_plus and projection bounds; force_init_solve picks per object."""
    _fused_equals_composite(backend, monkeypatch, dof, normalize, rslm, plus, bounds, 5, 70, 32, 2, 3)


@pytest.mark.parametrize('dof,normalize,force,bounds', [(4, True, True, 'tensor'), (6, False, True, None), (6, True, False, 'tight')])
def test_rslm_split_winner_is_picked_inside_the_lm_launch(backend, monkeypatch, dof, normalize, force, bounds):
    """With the initialiser's proposals dealt to several workgroups per object (EPROPNP_TUNE=rslm_parts=.., the default at <= 768 objects on
    the GPU) the one-call forward leaves the reduce launch out: the LM kernel picks the winner over the parts -- and, with
    force_init_solve on a given pose_init, the cheaper of that and pose_init -- itself (lm_core.h: StartSelect).  Same bits as one
    workgroup per object, whose start goes through the plain pose array."""
    B, N, S, K, L = 5, 90, 32, 2, 3
    prob = orc.make_problem(B, N, dof, seed=61, bounds=bounds)
    prob['pose_init'][0, :3] += 3.0                        # object 0: a bad pose_init, so the initialiser's start wins there
    noise = pack_noise(orc.make_noise(B, S, K, dof, seed=62), dof).to(backend)
    outs = []
    for parts in ('1', '2', '4'):
        set_tune(monkeypatch, rslm_parts=parts)
        p, cam, cf = make_layer_objects(prob, backend, relative_delta=0.5)
        x3d, x2d, w2d = (p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
        cf.set_param(x2d.detach(), w2d)
        from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
        from epropnp.levenberg_marquardt import LMSolver, RSLMSolver
        init = RSLMSolver(dof=dof, num_points=12, num_proposals=64, num_iter=2)
        layer = (EProPnP6DoF if dof == 6 else EProPnP4DoF)(mc_samples=S, num_iter=K, normalize=normalize, seed=9,
                                                          solver=LMSolver(dof=dof, num_iter=L, init_solver=init))
        torch.manual_seed(5)                               # the initialiser's Philox offset comes from torch's generator
        o = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=p['pose_init'] if force else None,
                                      force_init_solve=force, noise=noise)
        (o[4].logsumexp(0).sum() + (o[5].sum() if o[5] is not None else 0.0)).backward()
        outs.append([t.detach().clone() for t in (o[0], o[3], o[4], x3d.grad, w2d.grad)])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def _sweep_cases(n, seed):
    import random
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        dof = rng.choice((6, 4))
        c = (dof, rng.random() < 0.5, rng.random() < 0.5, rng.random() < 0.5, rng.choice((None, 'tensor', 'tight')),
             rng.choice((1, 2, 6, 9)), rng.choice((24, 64, 100, 130, 300)), rng.choice((8, 16, 24)) * rng.choice((1, 2, 4)),
             0, rng.choice((1, 3, 5)))
        K = rng.choice([k for k in (1, 2, 4) if c[7] % k == 0])
        out.append(c[:8] + (K, c[9]))
    return out


@pytest.mark.parametrize('dof,normalize,rslm,plus,bounds,B,N,S,K,L', _sweep_cases(8, 11))
def test_fused_forward_equals_composite_sweep(backend, monkeypatch, poisoned_empty, dof, normalize, rslm, plus, bounds, B, N, S, K, L):
    _fused_equals_composite(backend, monkeypatch, dof, normalize, rslm, plus, bounds, B, N, S, K, L)


@pytest.mark.gpu
@pytest.mark.parametrize('dof,normalize,rslm,plus,bounds,B,N,S,K,L',
                         _sweep_cases(int(os.environ.get('EPROPNP_FUZZ_CASES', '40')), int(os.environ.get('EPROPNP_FUZZ_SEED', '12'))))
def test_fused_forward_equals_composite_sweep_gpu(monkeypatch, poisoned_empty, dof, normalize, rslm, plus, bounds, B, N, S, K, L):
    import install as emu
    emu.uninstall()
    _fused_equals_composite(torch.device('cuda:0'), monkeypatch, dof, normalize, rslm, plus, bounds, B, N, S, K, L)


def _fused_equals_composite(backend, monkeypatch, dof, normalize, rslm, plus, bounds, B, N, S, K, L):
    prob = orc.make_problem(B, N, dof, seed=3, bounds=bounds)
    prob['pose_init'][0, :3] += 3.0                        # object 0: a bad pose_init, so RSLM's start wins there
    noise = pack_noise(orc.make_noise(B, S, K, dof, seed=4), dof).to(backend)
    rn = orc.make_rslm_noise(prob, dof, 8, 12, seed=5)
    outs = []
    for fused in (True, False):
        if fused:
            set_tune(monkeypatch)
        else:
            set_tune(monkeypatch, no_fused_forward=True)
        p, cam, cf = make_layer_objects(prob, backend, relative_delta=0.5)
        x3d, x2d, w2d = (p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
        cf.set_param(x2d.detach(), w2d)
        layer = _layer(dof, S, K, L, normalize, rslm)
        if rslm:
            layer.solver.init_solver.draw = lambda w: (rn['inds'].to(backend), rn['rot'].float().to(backend))
        assert layer._fusable(x3d, x2d, w2d, p['pose_init'], rslm, dict(with_pose_opt_plus=plus)) == fused
        out = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=p['pose_init'], force_init_solve=rslm,
                                        with_pose_opt_plus=plus, with_cost=True, noise=noise)
        pose_opt, cost, pplus, samples, logw, cost_init = out
        loss = (cost_init + torch.logsumexp(logw, 0)).mean()
        if plus:
            loss = loss + 0.1 * pplus.sum()
        loss.backward()
        outs.append([t.detach().cpu() for t in (pose_opt, cost, samples, logw, cost_init, x3d.grad, x2d.grad, w2d.grad)]
                    + ([pplus.detach().cpu()] if plus else []))
    for i, (a, b) in enumerate(zip(*outs)):
        if i == 7:      # w2d.grad: the fused path adds the Huber threshold's term in its backward kernel (epropnp_problem.
            # delta_stats), the composite path leaves it to autograd -- the same products, summed in another order
            assert ((a - b).abs() / b.abs().amax(dim=(1, 2), keepdim=True).clamp(min=1e-30)).max().item() < 1e-6
        else:
            assert torch.equal(a, b)


def test_fused_forward_without_pose_init_and_fallback_paths(backend):
    """pose_init=None (RSLM only: no cost_init) through the fused entry; a solver the fused entry does not cover (a
    subclass) takes the composite path and still works."""
    from epropnp.levenberg_marquardt import LMSolver
    B, N, S, K = 4, 64, 32, 2
    prob = orc.make_problem(B, N, 4, seed=6)
    p, cam, cf = make_layer_objects(prob, backend, relative_delta=0.5)
    cf.set_param(p['x2d'], p['w2d'])
    layer = _layer(4, S, K, 3, True, True)
    out = layer.monte_carlo_forward(p['x3d'], p['x2d'], p['w2d'], cam, cf)
    assert out[5] is None and out[1] is None and out[3].shape == (S, B, 4) and torch.isfinite(out[4]).all()
    assert (out[0][:, :3].cpu() - prob['pose_gt'][:, :3]).norm(dim=-1).median() < 1.0
    # ... and its backward: no pose_init means no cost_init, so the backward kernels get NULL for both optional inputs
    leaves = [p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d')]
    out = layer.monte_carlo_forward(*leaves, cam, cf)
    torch.logsumexp(out[4], 0).mean().backward()
    assert all(t.grad is not None and bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().sum()) > 0 for t in leaves)

    class MySolver(LMSolver):
        pass
    layer.solver = MySolver(dof=4, num_iter=3, init_solver=layer.solver.init_solver)
    assert not layer._fusable(p['x3d'], p['x2d'], p['w2d'], None, True, {})
    out2 = layer.monte_carlo_forward(p['x3d'], p['x2d'], p['w2d'], cam, cf)
    assert out2[3].shape == (S, B, 4)


def test_numerics_check_reports_what_the_reference_raises(backend):
    """A NaN 3D point makes the damped normal equations non-finite: the reference's torch.linalg.solve raises
    (levenberg_marquardt.py:15-19); here the kernels record it in the device status word and `numerics_check` raises
    after ONE synchronisation, naming the first offending object.  Healthy inputs pass; strict mode also reports the
    Cholesky fallback the reference takes silently."""
    from epropnp import functional as F
    from epropnp.levenberg_marquardt import LMSolver
    B, N = 6, 40
    prob = orc.make_problem(B, N, 6, seed=8)
    p, cam, cf = make_layer_objects(prob, backend)
    solver = LMSolver(dof=6, num_iter=3)
    with F.numerics_check():
        solver.solve(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'], with_pose_cov=True)
    bad = p['x3d'].clone()
    bad[4, 7, 0] = float('nan')
    with pytest.raises(RuntimeError, match='object 4'):
        with F.numerics_check():
            solver.solve(bad, p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'])
    # outside such a block the call itself returns (nothing synchronises): every trust-region step of that object is
    # rejected (NaN comparisons are false), so it comes back at its starting pose while the other objects are solved ...
    pose_opt = solver.solve(bad, p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'])[0]
    assert torch.equal(pose_opt[4], p['pose_init'][4]) and not torch.equal(pose_opt[3], p['pose_init'][3])
    # ... and the event arrives asynchronously: the kernels reported into the library's host-mapped status word, which the
    # next entry into the package (or flush_status) reads with a plain host load -- a RuntimeWarning by default (the
    # reference hands NaN poses on silently and its losses zero them), the RuntimeError on request
    from epropnp import _hip
    with pytest.warns(RuntimeWarning, match='object 4'):
        F.flush_status()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        F.flush_status()                              # reported once
    solver.solve(bad, p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'])
    if p['x3d'].is_cuda:
        torch.cuda.synchronize()
    _hip.STATUS_MODE = 'raise'
    try:
        with pytest.raises(RuntimeError, match='singular or not finite'):
            solver.solve(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'])  # on entry to the NEXT call
    finally:
        _hip.STATUS_MODE = 'warn'
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, 6)
    po, cov, _ = F.lm_solve(hp, p['pose_init'], 3, with_pose_cov=True)
    cov = cov.clone()
    cov[2, 0, 0] = -1.0
    with F.numerics_check():                                  # tolerated, as in the reference
        F.amis_forward(F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, 6), po, cov, 32, 2, seed=1)
    with pytest.raises(RuntimeError, match='cholesky.*object 2'):
        with F.numerics_check(strict=True):
            F.amis_forward(F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, 6), po, cov, 32, 2, seed=1)


def test_check_numerics_attribute_raises_at_the_call(backend):
    """`solver.check_numerics = True` / `layer.check_numerics = True`: the reference's error convention without an environment
    variable -- torch.linalg.solve / torch.inverse raise inside LMSolver.solve (levenberg_marquardt.py:15-19,178-181), so the
    RuntimeError comes out of the call that produced the singular system (one synchronisation), for the solver called on its
    own, through EProPnP6DoF.forward and through monte_carlo_forward; healthy inputs pass and nothing is left pending."""
    import warnings
    from epropnp import functional as F
    from epropnp.epropnp import EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    B, N = 5, 40
    prob = orc.make_problem(B, N, 6, seed=18)
    p, cam, cf = make_layer_objects(prob, backend)
    bad = p['x3d'].clone()
    bad[3, 5, 1] = float('nan')
    assert LMSolver.check_numerics is False and EProPnP6DoF.check_numerics is False          # opt-in
    solver = LMSolver(dof=6, num_iter=3)
    solver.check_numerics = True
    solver.solve(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'], with_pose_cov=True)
    with pytest.raises(RuntimeError, match='object 3'):
        solver.solve(bad, p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'])
    layer = EProPnP6DoF(mc_samples=32, num_iter=2, solver=LMSolver(dof=6, num_iter=3))
    layer.check_numerics = True
    layer(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'])
    with pytest.raises(RuntimeError, match='singular or not finite'):
        layer(bad, p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'])
    out = layer.monte_carlo_forward(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'], force_init_solve=False)
    assert bool(torch.isfinite(out[4]).all())
    with pytest.raises(RuntimeError, match='object 3'):
        layer.monte_carlo_forward(bad, p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'], force_init_solve=False)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        F.flush_status()                  # the checked calls reported into their own status word: nothing pending


def test_lm_and_normal_equations_beyond_the_resident_limit(backend):
    """N = 8300 > 8192 points per object: the LM solve and the normal-equation sweep stream the points instead of
    refusing (the reference accepts any N)."""
    from epropnp import functional as F
    B, N = 2, 8300
    prob = orc.make_problem(B, N, 6, seed=9)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, 6)
    pose, cov, cost = F.lm_solve(hp, p['pose_init'], 3, with_pose_cov=True, with_cost=True)
    o = orc.lm_solve(prob['x3d'], prob['x2d'], prob['w2d'], orc.Cam(prob['cam_mats'], 0.1), prob['delta'], prob['pose_init'],
                     with_pose_cov=True, with_cost=True, num_iter=3)
    assert (pose.cpu() - o[0]).abs().max() <= 1e-4
    torch.testing.assert_close(cost.cpu(), o[2], rtol=1e-4, atol=1e-5)
    jtj, jtr, c = F.normal_equations(hp, p['pose_init'])
    res, cc, jac = orc.evaluate(*(prob[k].double() for k in ('x3d', 'x2d', 'w2d', 'pose_init')),
                                orc.Cam(prob['cam_mats'].double(), 0.1), prob['delta'].double(), True, True)
    ref = jac.transpose(-1, -2) @ jac
    assert ((jtj.cpu().double() - ref).abs() / ref.abs().amax(dim=(-1, -2), keepdim=True)).max() < 5e-5
    torch.testing.assert_close(c.cpu().double(), cc, rtol=1e-5, atol=1e-6)


def test_non_default_huber_eps(backend):
    """HuberPnPCost(eps=...) (cost_fun.py:18-22,29): the floor of |r| in the robust rescaling.  It only binds when the
    threshold delta is itself below eps -- then sqrt(min(delta / max(rho, eps), 1)) < 1 for every point."""
    from epropnp import functional as F
    from epropnp.cost_fun import HuberPnPCost
    B, N, eps = 3, 50, 5.0
    prob = orc.make_problem(B, N, 6, seed=10)
    prob['delta'] = torch.full((B,), 0.5)
    p, cam, _ = make_layer_objects(prob, backend)
    cf = HuberPnPCost(delta=p['delta'], eps=eps)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, 6)
    jtj, jtr, _ = F.normal_equations(hp, p['pose_init'])
    for e, should_match in ((eps, True), (1e-10, False)):
        res, _, jac = orc.evaluate(*(prob[k].double() for k in ('x3d', 'x2d', 'w2d', 'pose_init')),
                                   orc.Cam(prob['cam_mats'].double(), 0.1), prob['delta'].double(), True, True, eps=e)
        ref = jac.transpose(-1, -2) @ jac
        err = ((jtj.cpu().double() - ref).abs() / ref.abs().amax(dim=(-1, -2), keepdim=True)).max().item()
        assert (err < 5e-5) == should_match, (e, err)
    # gn_step with the same eps: value and gradient w.r.t. w2d against autograd of the oracle
    w2d = p['w2d'].clone().requires_grad_(True)
    hq = F.PnPProblem(p['x3d'], p['x2d'], w2d, cam, cf, 6)
    step = F.gn_step(p['x3d'], p['x2d'], w2d, None, hq, p['pose_init'], 1e-5)
    up = torch.linspace(0.5, 1.5, 6)
    (step * up.to(backend)).sum().backward()
    wr = prob['w2d'].double().clone().requires_grad_(True)
    res, _, jac = orc.evaluate(prob['x3d'].double(), prob['x2d'].double(), wr, prob['pose_init'].double(),
                               orc.Cam(prob['cam_mats'].double(), 0.1), prob['delta'].double(), True, True, eps=eps)
    jt = jac.transpose(-1, -2)
    ref_step = -torch.linalg.solve(jt @ jac + 1e-5 * torch.eye(6, dtype=torch.float64), jt @ res.unsqueeze(-1)).squeeze(-1)
    (ref_step * up.double()).sum().backward()
    assert (step.detach().cpu().double() - ref_step.detach()).abs().max() < 2e-4 * ref_step.abs().max()
    assert ((w2d.grad.cpu().double() - wr.grad).abs().max() / wr.grad.abs().max()) < 2e-3


def test_inplace_edit_between_forward_and_backward_is_detected(backend):
    """The backward recomputes from x3d / x2d / w2d / delta: editing one of them in place after the forward must raise
    (as autograd does for saved tensors in the reference), not yield silently wrong gradients."""
    from epropnp.epropnp import EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    prob = orc.make_problem(3, 40, 6, seed=12)
    p, cam, cf = make_layer_objects(prob, backend, relative_delta=0.5)
    x3d = p['x3d'].clone().requires_grad_(True)
    x2d = p['x2d'].clone()
    cf.set_param(x2d, p['w2d'])
    layer = EProPnP6DoF(mc_samples=32, num_iter=2, solver=LMSolver(dof=6, num_iter=3))
    out = layer.monte_carlo_forward(x3d, x2d, p['w2d'], cam, cf, pose_init=p['pose_init'], force_init_solve=False)
    x2d.add_(1.0)
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        (out[5] + torch.logsumexp(out[4], 0)).sum().backward()
    # a pose_init that requires grad receives its gradient (tests/test_pose_cam_grad.py), no warning any more
    import warnings
    pi = p['pose_init'].clone().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        out = layer.monte_carlo_forward(x3d, p['x2d'], p['w2d'], cam, cf, pose_init=pi, force_init_solve=False)
    out[5].sum().backward()
    assert pi.grad is not None and bool(torch.isfinite(pi.grad).all()) and pi.grad.abs().max() > 0


def _delta_fold_case(dev, monkeypatch, dof, B, N, S, K, bounds, normalize, nsplit_env=None, impl_env=None):
    """w2d / x3d / x2d gradients of  sum(loss)  through set_param -> monte_carlo_forward -> MC loss, with the Huber threshold's
    way into grad_w2d taken (a) by the backward kernel itself (epropnp_problem.delta_stats; default) and (b) by autograd through
    the AdaptiveDelta node (EPROPNP_DELTA_FOLD=0), on injected noise."""
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    from epropnp.losses import monte_carlo_pose_loss
    from helpers import pack_noise
    prob = orc.make_problem(B, N, dof, seed=5 + N, bounds=bounds)
    noise = pack_noise(orc.make_noise(B, S, K, dof, seed=6), dof).to(dev)
    p, cam, _ = make_layer_objects(prob, dev, relative_delta=0.5)
    layer = (EProPnP6DoF if dof == 6 else EProPnP4DoF)(mc_samples=S, num_iter=K, normalize=normalize,
                                                     solver=LMSolver(dof=dof, num_iter=3))
    if nsplit_env is not None:
        monkeypatch.setenv('EPROPNP_BWD_SPLIT', nsplit_env)
    if impl_env is not None:
        set_tune(monkeypatch, bwd_impl=impl_env)
    from epropnp import functional as F
    seen, real = [], F.PnPProblem.fold_delta
    monkeypatch.setattr(F.PnPProblem, 'fold_delta', lambda self, *a: (seen.append(True), real(self, *a))[1])
    res = []
    for fold in ('1', '0'):
        monkeypatch.setenv('EPROPNP_DELTA_FOLD', fold)
        cf = AdaptiveHuberPnPCost(relative_delta=0.5)
        leaves = [p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d')]
        cf.set_param(leaves[1].detach(), leaves[2])
        o = layer.monte_carlo_forward(*leaves, cam, cf, pose_init=p['pose_init'], force_init_solve=False, noise=noise,
                                      with_pose_opt_plus=True)
        # (pose_opt_plus: the derivative regularisation every training caller of the reference adds -- a second gradient
        #  w.r.t. delta, through epropnp_pose_opt_plus_backward)
        (monte_carlo_pose_loss(o[4], o[5]).sum() + 0.3 * (o[2] * o[2]).sum()).backward()
        res.append([t.grad.detach().cpu() for t in leaves] + [o[4].detach().cpu()])
    (gx3_a, gx2_a, gw_a, lw_a), (gx3_b, gx2_b, gw_b, lw_b) = res
    assert torch.equal(lw_a, lw_b) and torch.equal(gx3_a, gx3_b) and torch.equal(gx2_a, gx2_b)
    assert float(gw_b.abs().max()) > 0
    scale = gw_b.abs().amax(dim=(1, 2), keepdim=True).clamp(min=1e-20)
    assert seen == [True]                                             # (only the first run folded)
    assert ((gw_a - gw_b).abs() / scale).max().item() < 2e-6          # the same products, summed by another party
    # ... and the delta path matters: without it the gradient is visibly another one
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)
    leaves = [p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d')]
    cf.set_param(leaves[1].detach(), leaves[2].detach())
    o = layer.monte_carlo_forward(*leaves, cam, cf, pose_init=p['pose_init'], force_init_solve=False, noise=noise)
    monte_carlo_pose_loss(o[4], o[5]).sum().backward()
    assert ((leaves[2].grad.cpu() - gw_b).abs() / scale).max().item() > 1e-4


def test_delta_computed_without_grad_is_a_constant(backend, monkeypatch):
    """AdaptiveHuberPnPCost.set_param(x2d, w2d) under torch.no_grad(): delta is a constant, the reference sends no gradient
    through it.  A later monte_carlo_forward with grad enabled on the SAME w2d must not fold d delta / d w2d into grad_w2d:
    same gradients as with an explicitly detached w2d in set_param."""
    from epropnp import functional as F
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    from epropnp.epropnp import EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    from epropnp.losses import monte_carlo_pose_loss
    B, N, S, K = 4, 80, 32, 2
    prob = orc.make_problem(B, N, 6, seed=15)
    noise = pack_noise(orc.make_noise(B, S, K, 6, seed=16), 6).to(backend)
    p, cam, _ = make_layer_objects(prob, backend, relative_delta=0.5)
    layer = EProPnP6DoF(mc_samples=S, num_iter=K, solver=LMSolver(dof=6, num_iter=3))
    seen, real = [], F.PnPProblem.fold_delta
    monkeypatch.setattr(F.PnPProblem, 'fold_delta', lambda self, *a: (seen.append(True), real(self, *a))[1])
    grads = []
    for mode in ('no_grad', 'detached'):
        cf = AdaptiveHuberPnPCost(relative_delta=0.5)
        leaves = [p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d')]
        if mode == 'no_grad':
            with torch.no_grad():
                cf.set_param(leaves[1], leaves[2])
        else:
            cf.set_param(leaves[1].detach(), leaves[2].detach())
        assert not cf.delta.requires_grad
        o = layer.monte_carlo_forward(*leaves, cam, cf, pose_init=p['pose_init'], force_init_solve=False, noise=noise)
        monte_carlo_pose_loss(o[4], o[5]).sum().backward()
        grads.append([t.grad.detach().cpu() for t in leaves])
    assert seen == []
    for a, b in zip(*grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize('dof,B,N,S,K,bounds,normalize', [(6, 5, 96, 32, 2, None, False), (4, 4, 70, 48, 3, 'tensor', True)])
def test_delta_gradient_folded_into_the_backward_kernel(backend, monkeypatch, dof, B, N, S, K, bounds, normalize):
    _delta_fold_case(backend, monkeypatch, dof, B, N, S, K, bounds, normalize)


@pytest.mark.gpu
@pytest.mark.parametrize('dof,B,N,S,K,bounds,normalize,nsplit,impl', [
    (6, 600, 512, 64, 2, None, False, None, None),          # one workgroup per object: the kernel's epilogue
    (6, 32, 512, 128, 4, 'tensor', False, None, None),      # few objects: split over workgroups -> follow-up launch
    (4, 40, 128, 64, 2, 'tensor', True, '2', None),
    (6, 9, 300, 32, 2, None, False, None, 'valu')])         # the all-VALU backward
def test_delta_gradient_folded_into_the_backward_kernel_gpu(monkeypatch, dof, B, N, S, K, bounds, normalize, nsplit, impl):
    import install as emu
    emu.uninstall()
    _delta_fold_case(torch.device('cuda:0'), monkeypatch, dof, B, N, S, K, bounds, normalize, nsplit, impl)


def test_center_and_cost_of_pose_init_are_one_launch(backend, monkeypatch):
    """normalize=True with a pose_init: pnp_normalize of the points and of pose_init and the cost of pose_init in the centred frame
    leave the one-call forward as ONE launch (center_cost_kernel, csrc/eval_kernels.hip), and pnp_denormalize of pose_opt and of the
    samples rides in the AMIS launch (AmisParams.dn_*) -- counted through the library's own stage records;
    test_fused_forward_equals_composite holds the results against the separate launches bit for bit, and so does
    EPROPNP_TUNE=no_denorm_fold here."""
    from epropnp import _hip
    prob = orc.make_problem(7, 100, 4, seed=8, bounds='tensor')
    p, cam, cf = make_layer_objects(prob, backend, relative_delta=0.5)
    cf.set_param(p['x2d'], p['w2d'])
    noise = pack_noise(orc.make_noise(7, 32, 2, 4, seed=4), 4).to(backend)
    counts, outs = {}, {}
    for normalize, fold in ((True, True), (True, False), (False, True)):
        set_tune(monkeypatch, **({} if fold else {'no_denorm_fold': True}))
        layer = _layer(4, 32, 2, 3, normalize, False)
        _hip.profile(enable=True, reset=True)
        try:
            out = layer.monte_carlo_forward(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'], force_init_solve=False,
                                            noise=noise)
            counts[normalize, fold] = {s: _hip.profile_read(s)[1] for s in ('center_points', 'evaluate_cost', 'lm_solve', 'amis_forward',
                                                                            'shift_poses')}
        finally:
            _hip.profile(enable=False, reset=True)
        outs[normalize, fold] = [out[0].clone(), out[3].clone(), out[4].clone()]
    assert counts[True, True] == dict(center_points=1, evaluate_cost=0, lm_solve=1, amis_forward=1, shift_poses=0), counts
    assert counts[True, False] == dict(center_points=1, evaluate_cost=0, lm_solve=1, amis_forward=1, shift_poses=1), counts
    assert counts[False, True] == dict(center_points=0, evaluate_cost=1, lm_solve=1, amis_forward=1, shift_poses=0), counts
    for a, b in zip(outs[True, True], outs[True, False]):
        assert torch.equal(a, b)
    assert (outs[True, True][0] - outs[False, True][0]).abs().max() < 1e-2      # pose_opt in the caller's frame either way
