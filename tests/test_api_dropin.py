"""The Python boundary: same names, signatures, return arity and error behaviour as the reference's `epropnp` package
(SURVEY.md section 8b), exercised through the same fixtures the reference produced."""
import inspect

import pytest
import torch

import epropnp_oracle as orc
from helpers import assert_within_spread, load_golden, make_layer_objects, pack_noise, rel_per_object


def _resolve(qual):
    import importlib
    modname, *path = qual.split('.')
    modname = {'loss6dof': 'losses', 'lossdet': 'losses'}.get(modname, modname)
    obj = importlib.import_module('epropnp.' + modname)
    for part in path:
        obj = getattr(obj, part)
    return obj


def test_public_names_and_signatures():
    """tests/golden/signatures.json is read off the imported reference with `inspect` (oracle/make_golden.py:
    case_signatures): every public callable of the reference's `epropnp` package (and the two loss modules of its
    callers) must exist here under the same name, with the reference's parameters as a prefix -- same names, same order,
    same kinds, same defaults.  Anything this package adds must be optional (a default or *args / **kwargs)."""
    import json
    import os
    from helpers import GOLDEN
    table = json.load(open(os.path.join(GOLDEN, 'signatures.json')))
    assert len(table) >= 80
    problems = []
    for qual, ref_params in sorted(table.items()):
        try:
            fn = _resolve(qual)
        except AttributeError as e:
            problems.append(f'{qual}: missing ({e})')
            continue
        mine = [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
                for p in inspect.signature(fn).parameters.values()]
        if qual.startswith('lossdet') or qual.startswith('loss6dof'):
            # one class serves both callers: the 6-DoF signature is a prefix of the detection one
            ref_params = [p for p in ref_params]
        var_kinds = ('VAR_POSITIONAL', 'VAR_KEYWORD')
        ref_fixed = [p for p in ref_params if p[1] not in var_kinds]
        mine_fixed = [p for p in mine if p[1] not in var_kinds]
        if qual.startswith('loss6dof') and qual.endswith('__init__'):
            # 6-DoF loss: (init_norm_factor, momentum) by keyword in its only caller (lib/models/...:  built with defaults)
            if not {p[0] for p in ref_fixed} <= {p[0] for p in mine_fixed}:
                problems.append(f'{qual}: parameters {ref_fixed} not all accepted by {mine_fixed}')
            continue
        if mine_fixed[:len(ref_fixed)] != ref_fixed:
            problems.append(f'{qual}: reference {ref_fixed} is not a prefix of {mine_fixed}')
            continue
        for extra in mine_fixed[len(ref_fixed):]:
            if extra[2] is None:
                problems.append(f'{qual}: extra parameter {extra[0]} has no default')
        for vk in var_kinds:
            if any(p[1] == vk for p in ref_params) and not any(p[1] == vk for p in mine):
                problems.append(f'{qual}: reference accepts {vk}, this package does not')
    assert not problems, '\n'.join(problems)
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    layer = EProPnP6DoF(mc_samples=512, num_iter=4, solver=LMSolver(dof=6, num_iter=10))
    assert len(list(layer.parameters())) == 0 and len(list(layer.buffers())) == 0
    layer.solver.num_iter = 5          # Det overrides this at test time via rsetattr
    with pytest.raises(AssertionError):
        EProPnP4DoF(mc_samples=10, num_iter=4)


@pytest.mark.parametrize('name', ['eval6', 'eval6_clip', 'eval4_clip'])
def test_evaluate_pnp_composite_matches_reference(name):
    """The framework-level (PyTorch) evaluate_pnp / camera / cost objects reproduce the reference's outputs."""
    from epropnp.common import evaluate_pnp
    g = load_golden(name)
    p, cam, cf = make_layer_objects(g['prob'], 'cpu')
    res, cost, jac = evaluate_pnp(p['x3d'], p['x2d'], p['w2d'], g['pose'], cam, cf, out_jacobian=True, out_residual=True,
                                  out_cost=True)
    torch.testing.assert_close(res, g['res'], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(cost, g['cost'], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(jac, g['jac'], rtol=1e-5, atol=1e-5)
    buf_j, buf_r, buf_c = torch.empty_like(g['jac']), torch.empty_like(g['res']), torch.empty_like(g['cost'])
    with torch.no_grad():
        evaluate_pnp(p['x3d'], p['x2d'], p['w2d'], g['pose'], cam, cf, out_jacobian=buf_j, out_residual=buf_r, out_cost=buf_c)
    torch.testing.assert_close(buf_j, g['jac'], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(buf_r, g['res'], rtol=1e-6, atol=1e-7)
    costs = evaluate_pnp(p['x3d'], p['x2d'], p['w2d'], g['poses'], cam, cf, out_cost=True)[1]
    torch.testing.assert_close(costs, g['costs'], rtol=1e-5, atol=1e-6)


def _demo_layer(g, dof, draws, normalize=False):
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver, RSLMSolver
    n_pts, n_prop, n_it = g['rslm_cfg'].tolist()
    init = RSLMSolver(dof=dof, num_points=n_pts, num_proposals=n_prop, num_iter=n_it)
    init.draw = lambda w2d: (draws['inds'].to(w2d.device), draws['rot'].to(w2d.device))     # injected randomness
    cls = EProPnP6DoF if dof == 6 else EProPnP4DoF
    return cls(mc_samples=int(g['S']), num_iter=int(g['K']), normalize=normalize,
               solver=LMSolver(dof=dof, num_iter=int(g['lm_iter']), init_solver=init))


@pytest.mark.parametrize('name', ['mc6_demo', 'mc4_rslm', 'mc4_det'])
def test_demo_config_with_rslm_and_pose_opt_plus(backend, name):
    """BASELINE config[0] (demo/fit_identity.ipynb shape): RSLM initialisation + LM + AMIS + derivative-regularisation
    output, force_init_solve=True, against the reference's outputs and input gradients."""
    g = load_golden(name)
    dof = int(g['dof'])
    p, cam, cf = make_layer_objects(g['prob'], backend, relative_delta=0.5)
    x3d, x2d, w2d = (p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    cf.set_param(x2d.detach(), w2d)
    layer = _demo_layer(g, dof, g['rslm'], normalize=bool(g['normalize']))
    out = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=p['pose_init'], force_init_solve=True,
                                    with_pose_opt_plus=True, with_cost=True,
                                    noise=pack_noise(g['noise'], dof).to(backend))
    pose_opt, cost, pose_opt_plus, samples, logw, cost_init = out
    assert pose_opt_plus.requires_grad and logw.requires_grad and cost_init.requires_grad and not samples.requires_grad
    loss_obj = cost_init + torch.logsumexp(logw, dim=0)
    total = loss_obj.mean() + 0.1 * (pose_opt_plus * torch.linspace(0.5, 1.5, pose_opt_plus.shape[-1],
                                                                    device=backend)).sum(-1).mean()
    total.backward()
    r, sp = g['ref'], g['spread']      # bars widened only by the reference's own rounding spread (see tests/test_amis.py)
    assert_within_spread((pose_opt.detach().cpu() - r['pose_opt']).abs().max(-1).values, sp['pose_opt'], 1e-4, what='pose_opt')
    assert_within_spread((pose_opt_plus.detach().cpu() - r['pose_opt_plus']).abs().max(-1).values, sp['pose_opt_plus'], 1e-4,
                         what='pose_opt_plus')
    assert_within_spread((loss_obj.detach().cpu() - r['loss_obj']).abs(), sp['loss_obj'], 1e-3, what='loss_obj')
    assert abs(loss_obj.mean().item() - r['loss_obj'].mean().item()) <= 1e-3
    for k, t in (('gx3d', x3d), ('gx2d', x2d), ('gw2d', w2d)):
        assert_within_spread(rel_per_object(t.grad, r[k]), sp[k], 5e-4, what=k)       # incl. the gn_step backward


def test_inference_forward_and_empty_batch(backend):
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    from epropnp.epropnp import EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    prob = orc.make_problem(5, 80, 6, seed=3)
    p, cam, _ = make_layer_objects(prob, backend)
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)
    cf.set_param(p['x2d'], p['w2d'])
    layer = EProPnP6DoF(mc_samples=64, num_iter=4, solver=LMSolver(dof=6, num_iter=4))
    # non-contiguous inputs + an expanded single camera matrix, as the reference's callers pass them
    x3d_t = p['x3d'].transpose(0, 1).contiguous().transpose(0, 1)
    cam1 = PerspectiveCamera(cam_mats=p['cam_mats'][:1].expand(5, 3, 3))
    pose_opt, pose_cov, cost, plus = layer(x3d_t, p['x2d'], p['w2d'], cam1, cf, pose_init=p['pose_init'], fast_mode=True)
    assert pose_opt.shape == (5, 7) and pose_cov is None and cost is None and plus is None
    assert (pose_opt.cpu()[:, :3] - prob['pose_gt'][:, :3]).norm(dim=-1).max() < 0.2
    with pytest.raises(NotImplementedError):
        layer.solver(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'], with_pose_cov=True,
                     normalize_override=True)
    with pytest.raises(AssertionError):   # no init_solver but pose_init missing
        layer.solver.solve(p['x3d'], p['x2d'], p['w2d'], cam, cf)
    # B = 0 keeps the graph (DDP callers rely on it)
    e3, e2 = (torch.zeros(0, 16, c, device=backend, requires_grad=True) for c in (3, 2))
    ew = torch.zeros(0, 16, 2, device=backend, requires_grad=True)
    cam0 = PerspectiveCamera(cam_mats=torch.zeros(0, 3, 3, device=backend))
    out = layer.monte_carlo_forward(e3, e2, ew, cam0, cf, pose_init=torch.zeros(0, 7, device=backend),
                                    force_init_solve=False)
    assert out[3].shape == (64, 0, 7) and out[4].shape == (64, 0) and out[4].requires_grad
    (out[4].sum() + out[5].sum()).backward()
    assert e3.grad is not None and e3.grad.shape == e3.shape


def test_img_shape_bounds_and_scalar_lb(backend):
    """camera.set_param(img_shape) gives a python-float lb and a tensor ub (camera.py:57-59)."""
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import HuberPnPCost
    prob = orc.make_problem(4, 64, 4, seed=12)
    p = {k: v.to(backend) for k, v in prob.items()}
    img = torch.tensor([[480., 640.]], device=backend).expand(4, 2)
    cam = PerspectiveCamera(z_min=0.1, allowed_border=20)
    cam.set_param(p['cam_mats'], img_shape=img)
    assert isinstance(cam.lb, float) and isinstance(cam.ub, torch.Tensor)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, HuberPnPCost(delta=p['delta']), 4)
    cost = F.evaluate_cost(hp, p['pose_init'])
    ocam = orc.Cam.from_img_shape(prob['cam_mats'], img.cpu(), 0.1, 20)
    ref = orc.evaluate(prob['x3d'], prob['x2d'], prob['w2d'], prob['pose_init'], ocam, prob['delta'], want_cost=True)[1]
    torch.testing.assert_close(cost.cpu(), ref, rtol=2e-5, atol=1e-6)


def test_distributions_and_loss_modules():
    from epropnp.distributions import AngularCentralGaussian, VonMisesUniformMix
    from epropnp.epropnp import cholesky_wrapper
    from epropnp.losses import MonteCarloPoseLoss
    g = torch.Generator().manual_seed(0)
    A = torch.randn(5, 4, 4, generator=g)
    L = torch.linalg.cholesky(A @ A.transpose(-1, -2) + 0.1 * torch.eye(4))
    acg = AngularCentralGaussian(L)
    x = acg.rsample((7,))
    assert x.shape == (7, 5, 4) and (x.norm(dim=-1) - 1).abs().max() < 1e-5
    torch.testing.assert_close(acg.log_prob(x), orc.acg_logprob(x, L), rtol=1e-5, atol=1e-5)
    vm = VonMisesUniformMix(torch.tensor([[0.3], [-2.0]]), torch.tensor([[2.0], [9.0]]))
    xs = vm.sample((8,))
    assert xs.shape == (8, 2, 1)
    torch.testing.assert_close(vm.log_prob(xs), orc.vm_mix_logprob(xs, vm.loc, vm.concentration), rtol=1e-5, atol=1e-5)
    bad = torch.eye(3).repeat(2, 1, 1)
    bad[1, 0, 0] = -1.0
    tril = cholesky_wrapper(bad, [1.0, 1.0, 4.0])
    torch.testing.assert_close(tril[1], torch.diag(torch.tensor([1.0, 1.0, 4.0])))
    loss = MonteCarloPoseLoss(init_norm_factor=2.0, momentum=0.1)
    logw, ct = torch.randn(16, 3, generator=g), torch.rand(3, generator=g)
    v = loss(logw, ct, torch.tensor(4.0))
    torch.testing.assert_close(loss.norm_factor, torch.tensor(2.2))
    torch.testing.assert_close(v, orc.mc_pose_loss(logw, ct, 2.2), rtol=1e-6, atol=1e-6)
    v2 = loss(logw, ct, 4.0, weight=torch.tensor([1.0, 0.0, 2.0]), avg_factor=3.0)
    assert v2.dim() == 0


def test_loss_modules_match_reference_fixture():
    """Fixture `losses`: the UNMODIFIED loss modules of the reference's two callers (6-DoF lib/models/
    monte_carlo_pose_loss.py:9-35; detection models/losses/monte_carlo_pose_loss.py:31-66 behind oracle/mmdet_shim.py)
    -- values for every weight / avg_factor / reduction combination, loss_weight, the norm_factor EMA over two training
    calls, a NaN object."""
    from epropnp.losses import MonteCarloPoseLoss
    g = load_golden('losses')
    logw, ct, weight, out = g['logw'], g['cost_target'], g['weight'], g['out']
    close = lambda a, b: torch.testing.assert_close(torch.as_tensor(a), torch.as_tensor(b, dtype=torch.float32), rtol=1e-6, atol=1e-6)
    m6 = MonteCarloPoseLoss(init_norm_factor=2.0, momentum=0.1)            # the 6-DoF caller's constructor arguments
    close(m6(logw.clone(), ct, g['nf0']), out['six.call0'])
    close(m6(logw.clone(), ct, g['nf1']), out['six.call1'])
    close(m6.norm_factor, out['six.norm_factor'])
    m6.eval()
    close(m6(logw.clone(), ct, g['nf0']), out['six.eval'])
    md = MonteCarloPoseLoss(loss_weight=0.5, init_norm_factor=2.0, momentum=0.1)
    close(md(logw.clone(), ct, g['nf0']), out['det.call0'])
    close(md(logw.clone(), ct, g['nf1'], weight=weight, avg_factor=3.5), out['det.call1'])
    close(md.norm_factor, out['det.norm_factor'])
    md.eval()
    n = 0
    for red in ('mean', 'sum', 'none'):
        for wname, w in (('w0', None), ('w1', weight)):
            for aname, af in (('a0', None), ('a1', 3.5)):
                if af is not None and red == 'sum':
                    with pytest.raises((ValueError, AssertionError)):
                        md(logw.clone(), ct, g['nf0'], weight=w, avg_factor=af, reduction_override=red)
                    continue
                close(md(logw.clone(), ct, g['nf0'], weight=w, avg_factor=af, reduction_override=red),
                      out[f'det.{red}.{wname}.{aname}'])
                n += 1
    assert n == 10
    with pytest.raises(AssertionError):
        md(logw, ct, g['nf0'], reduction_override='max')


def test_loss_modules_fused_path_matches_reference_fixture(backend, monkeypatch):
    """The same fixture of the UNMODIFIED reference loss modules through the fused kernels (epropnp_mc_loss_forward +
    epropnp_mc_loss_reduce: per-object loss, then weight / reduction / avg_factor / loss_weight / norm_factor and the running
    estimate in one launch): every scalar reduction, the EMA over two training calls, the NaN object; 'none' stays on the
    composite statement.  Gradients w.r.t. the log-weights and cost_target against the composite statement's autograd."""
    from epropnp import functional as F
    from epropnp.losses import MonteCarloPoseLoss
    g = load_golden('losses')
    dev = backend
    logw, ct, weight, out = g['logw'].to(dev), g['cost_target'].to(dev), g['weight'].to(dev), g['out']
    calls = []
    real = F.mc_pose_loss_reduced
    monkeypatch.setattr(F, 'mc_pose_loss_reduced', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    close = lambda a, b: torch.testing.assert_close(torch.as_tensor(a).cpu(), torch.as_tensor(b, dtype=torch.float32), rtol=2e-6, atol=1e-6)
    m6 = MonteCarloPoseLoss(init_norm_factor=2.0, momentum=0.1).to(dev)
    close(m6(logw.clone(), ct, g['nf0']), out['six.call0'])
    close(m6(logw.clone(), ct, g['nf1']), out['six.call1'])
    close(m6.norm_factor, out['six.norm_factor'])
    m6.eval()
    close(m6(logw.clone(), ct, g['nf0']), out['six.eval'])
    close(m6.norm_factor, out['six.norm_factor'])                      # eval: the running estimate stays
    md = MonteCarloPoseLoss(loss_weight=0.5, init_norm_factor=2.0, momentum=0.1).to(dev)
    close(md(logw.clone(), ct, g['nf0']), out['det.call0'])
    close(md(logw.clone(), ct, g['nf1'], weight=weight, avg_factor=3.5), out['det.call1'])
    close(md.norm_factor, out['det.norm_factor'])
    assert len(calls) == 5
    md.eval()
    for red in ('mean', 'sum'):
        for wname, w in (('w0', None), ('w1', weight)):
            for aname, af in (('a0', None), ('a1', 3.5)):
                if af is not None and red == 'sum':
                    continue
                v = md(logw.clone(), ct, g['nf0'], weight=w, avg_factor=af, reduction_override=red)
                assert v.dim() == 0
                close(v, out[f'det.{red}.{wname}.{aname}'])
    assert len(calls) == 5 + 6
    close(md(logw.clone(), ct, g['nf0'], weight=weight, reduction_override='none'), out['det.none.w1.a0'])
    assert len(calls) == 11                                             # per-object losses: the composite statement
    # gradients: fused node vs autograd of the composite statement (a weight that requires grad keeps the fused path out)
    for w, af in ((None, None), (weight, 3.5), (weight, None)):
        grads = []
        for fused in (True, False):
            a, b = logw.clone().requires_grad_(True), ct.clone().requires_grad_(True)
            ww = w if (fused or w is None) else w.clone().requires_grad_(True)
            n0 = len(calls)
            v = md(a, b, g['nf0'], weight=ww, avg_factor=af)
            assert (len(calls) > n0) == (fused or w is None)
            (v * 1.7).backward()
            grads.append((v.detach().cpu(), a.grad.cpu(), b.grad.cpu()))
        if w is None:
            continue
        for x, y in zip(*grads):
            torch.testing.assert_close(x, y, rtol=2e-6, atol=1e-7)
    a, b = logw.clone().requires_grad_(True), ct.clone().requires_grad_(True)
    v = md(a, b, g['nf0'])
    v.backward()
    ref_a, ref_b = logw.clone().requires_grad_(True), ct.clone().requires_grad_(True)
    l = ref_b + torch.logsumexp(ref_a, 0)
    (torch.where(torch.isnan(l), torch.zeros_like(l), l).mean() * (0.5 / md.norm_factor)).backward()
    torch.testing.assert_close(a.grad.cpu(), torch.nan_to_num(ref_a.grad.cpu()), rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(b.grad.cpu(), torch.nan_to_num(ref_b.grad.cpu()), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('dof', [6, 4])
def test_amis_extension_hooks_reproduce_the_sampler(backend, dof):
    """allocate_buffer / initial_fit / gen_new_distr / gen_old_distr / estimate_params (epropnp.py:199-342): the AMIS
    loop of epropnp.py:132-182 written with the hooks (PyTorch), fed the kernel's own samples, arrives at the proposals
    the fused kernel fitted and at its log-weights."""
    import math
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    B, N, S, K = 4, 64, 64, 4
    s = S // K
    prob = orc.make_problem(B, N, dof, seed=17)
    noise = orc.make_noise(B, S, K, dof, seed=18)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    pose_opt, pose_cov, _ = F.lm_solve(hp, p['pose_init'], 4, with_pose_cov=True)
    samples, logw, props = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=pack_noise(noise, dof).to(backend),
                                          with_proposals=True)
    samples, logw, props = samples.cpu().double(), logw.cpu().double(), props.cpu()
    pose_opt, pose_cov = pose_opt.cpu().double(), pose_cov.cpu().double()
    layer = (EProPnP6DoF if dof == 6 else EProPnP4DoF)(mc_samples=S, num_iter=K)
    bufs = layer.allocate_buffer(B, dtype=torch.float64)
    assert [tuple(b.shape) for b in bufs][:2] == [(K, B, 3), (K, B, 3, 3)]
    layer.initial_fit(pose_opt, pose_cov, PerspectiveCamera(cam_mats=prob['cam_mats'].double()), *bufs)
    cost = orc.evaluate(prob['x3d'].double(), prob['x2d'].double(), prob['w2d'].double(), samples,
                        orc.Cam(prob['cam_mats'].double(), 0.1), prob['delta'].double(), want_cost=True)[1]
    logprobs = torch.zeros(K, K, s, B, dtype=torch.float64)
    blocks = samples.reshape(K, s, B, -1)
    for i in range(K):
        new_t, new_r = layer.gen_new_distr(i, *bufs)
        assert new_t.rsample((3,)).shape == (3, B, 3)
        seen = blocks[:i + 1]                                             # (i+1,s,B,p)
        logprobs[i, :i + 1] = new_t.log_prob(seen[..., :3]) + new_r.log_prob(seen[..., 3:]).reshape(i + 1, s, B)
        if i > 0:
            old_t, old_r = layer.gen_old_distr(i, *bufs)
            logprobs[:i, i] = old_t.log_prob(blocks[i][..., :3]) + old_r.log_prob(blocks[i][..., 3:]).reshape(i, s, B)
        mix = torch.logsumexp(logprobs[:i + 1, :i + 1], dim=0) - math.log(i + 1)
        lw = -cost.reshape(K, s, B)[:i + 1] - mix
        if i < K - 1:
            layer.estimate_params(i, samples[:(i + 1) * s], lw.reshape(-1, B), *bufs)
    assert (lw.reshape(S, B) - logw).abs().max().item() <= 5e-4 * max(1.0, logw.abs().max().item())

    def tril(v, n):
        L = torch.zeros(v.shape[:-1] + (n, n), dtype=torch.float64)
        idx = torch.tril_indices(n, n)
        L[..., idx[0], idx[1]] = v.double()
        return L
    rec = props.permute(1, 0, 2)                                          # (K,B,40)
    torch.testing.assert_close(rec[..., 0:3].double(), bufs[0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(tril(rec[..., 3:9], 3), bufs[1], rtol=2e-3, atol=1e-5)
    if dof == 6:
        torch.testing.assert_close(tril(rec[..., 16:26], 4), bufs[2], rtol=5e-3, atol=2e-5)
    else:
        d = (rec[..., 16:17].double() - bufs[2]).abs()
        assert torch.minimum(d, 2 * math.pi - d).max() < 1e-4
        torch.testing.assert_close(rec[..., 17:18].double(), bufs[3], rtol=2e-3, atol=1e-6)


def test_fused_delta_and_loss_match_torch(backend):
    """The fused set_param / Monte-Carlo-loss kernels against their PyTorch definitions, values and gradients."""
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    from epropnp.losses import monte_carlo_pose_loss
    g = torch.Generator().manual_seed(5)
    B, N, S = 7, 50, 24
    x = (torch.randn(B, N, 2, generator=g) * 80 + 300).to(backend).requires_grad_(True)
    w = (torch.rand(B, N, 2, generator=g) * 0.01).to(backend).requires_grad_(True)
    cf = AdaptiveHuberPnPCost(relative_delta=0.3)
    cf.set_param(x, w)
    up = torch.randn(B, generator=g).to(backend)
    (cf.delta * up).sum().backward()
    xr, wr = x.detach().cpu().double().requires_grad_(True), w.detach().cpu().double().requires_grad_(True)
    dref = orc.adaptive_huber_delta(xr, wr, 0.3)
    (dref * up.cpu().double()).sum().backward()
    torch.testing.assert_close(cf.delta.detach().cpu().double(), dref.detach(), rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(x.grad.cpu().double(), xr.grad, rtol=1e-4, atol=1e-9)
    torch.testing.assert_close(w.grad.cpu().double(), wr.grad, rtol=1e-4, atol=1e-9)
    logw = torch.randn(S, B, generator=g).mul(3).to(backend).requires_grad_(True)
    ct = torch.rand(B, generator=g).to(backend).requires_grad_(True)
    with torch.no_grad():
        logw[3, 2] = float('nan')        # NaN loss for object 2 -> 0, no gradient
    loss = monte_carlo_pose_loss(logw, ct)
    (loss * up).sum().backward()
    lr, cr = logw.detach().cpu().clone().requires_grad_(True), ct.detach().cpu().clone().requires_grad_(True)
    ref = cr + torch.logsumexp(lr, 0)
    ref = torch.where(torch.isnan(ref), torch.zeros_like(ref), ref)
    keep = torch.ones(B, dtype=torch.bool)
    keep[2] = False
    (ref * up.cpu())[keep].sum().backward()
    torch.testing.assert_close(loss.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    assert loss[2].item() == 0.0
    torch.testing.assert_close(logw.grad.cpu()[:, keep], lr.grad[:, keep], rtol=1e-4, atol=1e-6)
    assert (logw.grad[:, 2] == 0).all() and ct.grad[2].item() == 0.0
    torch.testing.assert_close(ct.grad.cpu()[keep], cr.grad[keep], rtol=1e-6, atol=1e-7)


def test_rslm_draw_is_weighted_sampling_without_replacement(backend):
    from epropnp import functional as F
    g = torch.Generator().manual_seed(9)
    B, N, P, n = 4, 40, 3000 if backend.type == 'cuda' else 300, 6
    w = torch.rand(B, N, 2, generator=g)
    w[:, :5] *= 8.0                       # five heavy points
    w[0, 7] = 0.0                         # never selected
    inds = F.rslm_draw(w.to(backend), P, n, seed=5, offset=0).cpu()
    assert inds.shape == (P, B, n) and inds.min() >= 0 and inds.max() < N
    srt = inds.sort(-1).values
    assert (srt[..., 1:] != srt[..., :-1]).all()                      # distinct within a row
    assert not (inds[:, 0] == 7).any()
    first = inds[:, :, 0]                                             # first pick ~ categorical(mean weight)
    pw = w.mean(-1)
    pw = pw / pw.sum(-1, keepdim=True)
    for b in range(B):
        freq = torch.bincount(first[:, b], minlength=N).float() / P
        assert (freq - pw[b]).abs().max() < 6 * (pw[b].max() / P) ** 0.5 + 2.0 / P
    assert not torch.equal(inds, F.rslm_draw(w.to(backend), P, n, seed=5, offset=1).cpu())


def test_mc_loss_kernel_edge_values(backend):
    """Several row batches per thread (S > 256), -inf weights (ignored), a +inf weight (loss +inf) and an all -inf
    column (loss -inf), against torch.logsumexp."""
    from epropnp.losses import monte_carlo_pose_loss
    g = torch.Generator().manual_seed(11)
    S, B = 700, 37
    logw = torch.randn(S, B, generator=g).mul(4)
    logw[5:40, 3] = float('-inf')
    logw[:, 8] = float('-inf')
    logw[123, 9] = float('inf')
    ct = torch.rand(B, generator=g)
    loss = monte_carlo_pose_loss(logw.to(backend), ct.to(backend)).cpu()
    ref = ct + torch.logsumexp(logw, 0)
    fin = torch.isfinite(ref)
    torch.testing.assert_close(loss[fin], ref[fin], rtol=1e-5, atol=1e-5)
    assert loss[8].item() == float('-inf') and loss[9].item() == float('inf')


def test_evaluate_pnp_cost_only_uses_the_sweep_kernel(backend):
    """evaluate_pnp(out_cost=True) on plain tensors (one pose per object, or a (P,B,p) grid of poses as in the Det head's
    orientation debug, deform_pnp_head.py:540-551) equals the PyTorch composite."""
    from epropnp.common import evaluate_pnp
    g = load_golden('eval4_clip')
    p, cam, cf = make_layer_objects(g['prob'], backend)
    poses = g['poses'].to(backend)                                   # (P,B,4)
    with torch.no_grad():
        fast = evaluate_pnp(p['x3d'], p['x2d'], p['w2d'], poses, cam, cf, out_cost=True)
        one = evaluate_pnp(p['x3d'], p['x2d'], p['w2d'], g['pose'].to(backend), cam, cf, out_cost=True)
    assert fast[0] is None and fast[2] is None
    torch.testing.assert_close(fast[1].cpu(), g['costs'], rtol=2e-5, atol=1e-5)
    torch.testing.assert_close(one[1].cpu(), g['cost'], rtol=2e-5, atol=1e-6)
    # with autograd in play the composite path runs (and is differentiable)
    x3d = p['x3d'].clone().requires_grad_(True)
    if backend.type == 'cpu':
        return          # the composite itself needs real tensors; on the emulation backend only the fast path exists
    c = evaluate_pnp(x3d, p['x2d'], p['w2d'], g['pose'].to(backend), cam, cf, out_cost=True)[1]
    c.sum().backward()
    assert torch.isfinite(x3d.grad).all()


def test_camera_view_cache_tracks_the_camera_state(backend):
    """PnPProblem materialises contiguous intrinsics / bounds once per camera state: replacing or modifying the
    camera's tensors must be seen by the next call."""
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    p = orc.make_problem(3, 40, 6, seed=1)
    d, cam, cf = make_layer_objects(p, backend)
    pose = d['pose_init']

    def cost():
        return F.evaluate_cost(F.PnPProblem(d['x3d'], d['x2d'], d['w2d'], cam, cf, 6), pose).clone()
    c0 = cost()
    torch.testing.assert_close(cost(), c0, rtol=0, atol=0)                     # cached views, same result
    k2 = d['cam_mats'].clone()
    k2[:, 0, 0] *= 1.1
    cam.set_param(k2)                                                          # new tensor object
    c1 = cost()
    assert (c1 - c0).abs().max() > 1e-3
    with torch.no_grad():
        cam.cam_mats[:, 1, 1] *= 0.9                                           # in-place edit of the same object
    assert (cost() - c1).abs().max() > 1e-3
    cam.set_param(d['cam_mats'][:1].expand(3, 3, 3))                           # expanded view of one matrix
    torch.testing.assert_close(cost(), c0, rtol=1e-6, atol=1e-6)
    cam2 = PerspectiveCamera(cam_mats=d['cam_mats'], lb=-50.0, ub=700.0)       # float bounds
    c_b = F.evaluate_cost(F.PnPProblem(d['x3d'], d['x2d'], d['w2d'], cam2, cf, 6), pose)
    cam2.ub = 300.0
    c_b2 = F.evaluate_cost(F.PnPProblem(d['x3d'], d['x2d'], d['w2d'], cam2, cf, 6), pose)
    assert torch.isfinite(c_b).all() and (c_b2 - c_b).abs().max() > 0


@pytest.mark.parametrize('dof,rslm,plus', [(6, False, True), (4, True, False), (4, True, True)])
def test_one_problem_per_monte_carlo_forward(backend, monkeypatch, dof, rslm, plus):
    """The solver calls under monte_carlo_forward (LM, RSLM initialiser, pose_opt_plus) reuse its PnPProblem; outside of
    it (or with different objects) they build their own."""
    from epropnp import functional as F
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver, RSLMSolver
    p = orc.make_problem(5, 40, dof, seed=21)
    d, cam, cf = make_layer_objects(p, backend, relative_delta=0.5)
    cf.set_param(d['x2d'], d['w2d'])
    init = RSLMSolver(dof=dof, num_points=8, num_proposals=4, num_iter=2) if rslm else None
    layer = (EProPnP6DoF if dof == 6 else EProPnP4DoF)(mc_samples=32, num_iter=4, normalize=(dof == 4),
                                                       solver=LMSolver(dof=dof, num_iter=3, init_solver=init))
    built = []
    real_init = F.PnPProblem.__init__

    def counting_init(self, *a, **k):
        built.append(1)
        real_init(self, *a, **k)
    monkeypatch.setattr(F.PnPProblem, '__init__', counting_init)
    x3d = d['x3d'].clone().requires_grad_(True)
    out = layer.monte_carlo_forward(x3d, d['x2d'], d['w2d'], cam, cf, pose_init=d['pose_init'], force_init_solve=rslm,
                                    with_pose_opt_plus=plus)
    assert len(built) == 1
    assert getattr(F._shared, 'entry', None) is None               # nothing outlives the call
    if plus:
        out[2].sum().backward()
        assert bool(torch.isfinite(x3d.grad).all())
    built.clear()
    layer.solver.solve(d['x3d'], d['x2d'], d['w2d'], cam, cf, pose_init=d['pose_init'])       # stand-alone: its own
    assert len(built) == 1


def test_the_device_query_is_asked_once(backend, monkeypatch):
    """`torch.cuda.is_available()` is a driver query (46 us per call on an MI355X box, profiles/r06_eager_host_time.txt): the status
    poll that every entry into the package performs asks it once per process, not once per call."""
    from epropnp import _hip
    from epropnp import functional as F
    asked = []
    real = torch.cuda.is_available
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: (asked.append(1), real())[1])
    monkeypatch.setattr(_hip, '_has_gpu', None)
    p = orc.make_problem(3, 24, 6, seed=5)
    d, cam, cf = make_layer_objects(p, backend)
    for _ in range(4):
        F.PnPProblem(d['x3d'], d['x2d'], d['w2d'], cam, cf, 6)
        _hip.poll_status()
    assert len(asked) == 1
