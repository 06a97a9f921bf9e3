"""
oracle/make_golden.py -- TEST INFRASTRUCTURE (build container only; needs /root/reference).

1. runs the UNMODIFIED reference (via oracle/ref_runner.py) on small seeded problems with injected noise,
2. runs the restatement (oracle/epropnp_oracle.py) on the same inputs,
3. asserts that they agree (this is what pins the oracle), and
4. writes inputs + reference outputs as small fixtures to tests/golden/*.npz.

Usage:  python oracle/make_golden.py            (regenerates every fixture, prints the max deviations)
The fixtures are what travels to the GPU box; /root/reference does not.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import epropnp_oracle as orc  # noqa: E402
import ref_runner as ref  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
torch.set_num_threads(4)
REPORT = []


def _maxdiff(a, b):
    return float((a.double() - b.double()).abs().max()) if a.numel() else 0.0


def check(name, key, a, b, tol):
    d = _maxdiff(a, b)
    REPORT.append((name, key, d, tol))
    assert d <= tol, f'{name}.{key}: reference vs restatement differ by {d:.3e} > {tol:.1e}'


def save(name, **arrays):
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f'{k}.{kk}'] = vv.numpy() if isinstance(vv, torch.Tensor) else np.asarray(vv)
        else:
            flat[k] = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **flat)


def cam_of(prob):
    return orc.Cam(prob['cam_mats'], 0.1, prob.get('lb'), prob.get('ub'))


def case_evaluate(name, dof, B, N, bounds, seed):
    prob = orc.make_problem(B, N, dof, seed=seed, bounds=bounds)
    cam = cam_of(prob)
    pose = prob['pose_init']
    r_res, r_cost, r_jac = ref.run_evaluate(prob, pose, jac=True, clip_jac=True)
    o_res, o_cost, o_jac = orc.evaluate(prob['x3d'], prob['x2d'], prob['w2d'], pose, cam, prob['delta'], True, True)
    check(name, 'res', r_res, o_res, 1e-6)
    check(name, 'cost', r_cost, o_cost, 1e-6)
    check(name, 'jac', r_jac, o_jac, 2e-5 * float(r_jac.abs().max()))
    # multi-pose cost-only sweep (the AMIS integrand), poses (P,B,p)
    g = torch.Generator().manual_seed(seed + 100)
    P = 5
    poses = pose.unsqueeze(0).repeat(P, 1, 1)
    poses[..., :3] += 0.3 * torch.randn(P, B, 3, generator=g)
    if dof == 6:
        q = poses[..., 3:] + 0.2 * torch.randn(P, B, 4, generator=g)
        poses[..., 3:] = q / q.norm(dim=-1, keepdim=True)
    else:
        poses[..., 3] += 0.5 * torch.randn(P, B, generator=g)
    poses[0, 0, 2] = -3.0    # one pose behind the camera: exercises the z_min clamp
    r_costs = ref.run_evaluate(prob, poses)[1]
    o_costs = orc.evaluate(prob['x3d'], prob['x2d'], prob['w2d'], poses, cam, prob['delta'], want_cost=True)[1]
    check(name, 'costs', r_costs, o_costs, 2e-5 * float(r_costs.abs().max()))
    jtj = r_jac.transpose(-1, -2).double() @ r_jac.double()
    jtr = (r_jac.transpose(-1, -2).double() @ r_res.double().unsqueeze(-1)).squeeze(-1)
    save(name, prob=prob, pose=pose, poses=poses, res=r_res, cost=r_cost, jac=r_jac, costs=r_costs,
         jtj=jtj.float(), jtr=jtr.float(), dof=dof)


def case_lm(name, dof, B, N, lm_iter, fast_mode, bounds, seed):
    prob = orc.make_problem(B, N, dof, seed=seed, bounds=bounds)
    r_pose, r_cov, r_cost = ref.run_lm(prob, dof, lm_iter, fast_mode)
    o_pose, o_cov, o_cost, hist = orc.lm_solve(prob['x3d'], prob['x2d'], prob['w2d'], cam_of(prob), prob['delta'],
                                               prob['pose_init'], fast_mode=fast_mode, with_pose_cov=True,
                                               with_cost=True, num_iter=lm_iter)
    check(name, 'pose_opt', r_pose, o_pose, 2e-5)
    check(name, 'cost', r_cost, o_cost, 1e-5 * max(1.0, float(r_cost.abs().max())))
    check(name, 'pose_cov', r_cov, o_cov, 2e-3 * float(r_cov.abs().max()))
    # fp64 reference of the same solve (for conditioning-aware tolerances in the tests)
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in prob.items()}
    d_pose, d_cov, d_cost, _ = orc.lm_solve(p64['x3d'], p64['x2d'], p64['w2d'], cam_of(p64), p64['delta'],
                                            p64['pose_init'], fast_mode=fast_mode, with_pose_cov=True,
                                            with_cost=True, num_iter=lm_iter)
    # rounding-level spread of the REFERENCE itself (inputs moved by <= 3 ulp, + its fp64 run): the tests' yardstick
    base = dict(pose_opt=r_pose, pose_cov=r_cov, cost=r_cost)
    spread = orc.rounding_spread(lambda q: dict(zip(('pose_opt', 'pose_cov', 'cost'), ref.run_lm(q, dof, lm_iter, fast_mode))),
                                 prob, base, extra=[dict(pose_opt=d_pose, pose_cov=d_cov, cost=d_cost)])
    save(name, prob=prob, pose_opt=r_pose, pose_cov=r_cov, cost=r_cost, pose_opt64=d_pose, pose_cov64=d_cov,
         cost64=d_cost, dof=dof, lm_iter=lm_iter, fast_mode=int(fast_mode), spread=spread,
         accepts=torch.stack(hist).to(torch.int32) if hist else torch.zeros(0, B, dtype=torch.int32))


def case_mc(name, dof, B, N, S, K, lm_iter, seed, normalize=False, rslm=None, with_pose_opt_plus=False,
            bounds=None, cam_kind='pinhole800', tol_logw=2e-3):
    prob = orc.make_problem(B, N, dof, seed=seed, bounds=bounds, cam_kind=cam_kind)
    noise = orc.make_noise(B, S, K, dof, seed=seed + 1)
    rn = orc.make_rslm_noise(prob, dof, rslm['num_points'], rslm['num_proposals'], seed + 2) if rslm else None
    r = ref.run_mc(prob, noise, dof, S, K, lm_iter, normalize=normalize, rslm=rslm, rslm_noise=rn,
                   with_pose_opt_plus=with_pose_opt_plus)
    rslm_kw = dict(num_iter=rslm['num_iter']) if rslm else None
    o = orc.run_mc(prob, noise, dof, S, K, lm_iter, normalize=normalize, rslm_kw=rslm_kw, rslm_noise=rn,
                   with_pose_opt_plus=with_pose_opt_plus)
    o64 = orc.run_mc(prob, noise, dof, S, K, lm_iter, normalize=normalize, rslm_kw=rslm_kw, rslm_noise=rn,
                     with_pose_opt_plus=with_pose_opt_plus, dtype=torch.float64)
    check(name, 'pose_opt', r['pose_opt'], o['pose_opt'], 2e-5)
    check(name, 'cost_init', r['cost_init'], o['cost_init'], 1e-5 * max(1.0, float(r['cost_init'].abs().max())))
    check(name, 'pose_samples', r['pose_samples'], o['pose_samples'], 5e-3)
    check(name, 'logweights', r['logweights'], o['logweights'], tol_logw * max(1.0, float(r['logweights'].abs().max())))
    check(name, 'loss_obj', r['loss_obj'], o['loss_obj'], 1e-3)
    for k in ('gx3d', 'gx2d', 'gw2d'):
        check(name, k, r[k], o[k], 2e-3 * float(r[k].abs().max()))
    if with_pose_opt_plus:
        check(name, 'pose_opt_plus', r['pose_opt_plus'], o['pose_opt_plus'], 5e-5)
    REPORT.append((name, 'loss_obj fp32-vs-fp64 oracle', _maxdiff(o['loss_obj'], o64['loss_obj']), float('nan')))
    REPORT.append((name, 'pose_opt fp32-vs-fp64 oracle', _maxdiff(o['pose_opt'], o64['pose_opt']), float('nan')))
    extra = {}
    if rn is not None:
        extra['rslm'] = rn
    # rounding-level spread of the REFERENCE itself (inputs moved by <= 3 ulp, + the fp64 run): the tests' yardstick
    extra['spread'] = orc.rounding_spread(
        lambda q: ref.run_mc(q, noise, dof, S, K, lm_iter, normalize=normalize, rslm=rslm, rslm_noise=rn,
                             with_pose_opt_plus=with_pose_opt_plus), prob, r, extra=[o64])
    save(name, prob=prob, noise=noise, ref=r, o64={k: v.float() for k, v in o64.items()}, dof=dof, S=S, K=K,
         lm_iter=lm_iter, normalize=int(normalize), with_pose_opt_plus=int(with_pose_opt_plus),
         rslm_cfg=np.array([rslm['num_points'], rslm['num_proposals'], rslm['num_iter']] if rslm else [0, 0, 0]),
         **extra)


def case_vm_numpy(name, seed):
    """Pins the bounded Best-Fisher sampler (orc.vm_sample_bounded, what the HIP kernel implements) against the
    reference's own sampler -- numpy.random.vonmises / numpy.random.uniform inside the UNMODIFIED
    VonMisesUniformMix.sample (epropnp/distributions.py:61-72) -- on a SHARED uniform stream: the reference draws from
    numpy's seeded global generator; the same seed is then replayed through random_sample() and dealt into the
    (attempt, 3) layout the bounded sampler consumes: numpy's legacy von Mises takes (U, V) per attempt until one is
    accepted and then one more double for the sign; uniform(-pi, pi) takes one double per element."""
    import math
    m = ref.load_reference()
    B, s = 10, 96
    n_u = round(0.25 * s)
    n_v = s - n_u
    g = torch.Generator().manual_seed(seed)
    var = torch.tensor([1e-5, 1e-4, 1e-3, 0.01, 0.05, 0.33, 2.0, 33.0, 330.0, 3300.0])
    kappa = (0.33 / var.clamp(min=1e-5)).reshape(B, 1)               # as epropnp.py:218 computes it (fp32)
    loc = (torch.rand(B, 1, generator=g) * 2 - 1) * 3.0
    dist = m['distributions'].VonMisesUniformMix(loc, kappa)
    np.random.seed(seed)
    x = m['vm_sample_unpatched'](dist, torch.Size([s]))               # (s,B,1) float32, the reference's draw
    np.random.seed(seed)
    u_uni = torch.from_numpy(np.random.random_sample((n_u, B, 1)))
    T = orc.VM_MAX_TRIES
    u_vm = torch.full((n_v, B, 1, T, 3), 0.5, dtype=torch.float64)
    worst = 0
    for i in range(n_v):
        for b in range(B):
            k = float(kappa[b, 0])
            if k < 1e-5:
                r = 1.0 / k + k
            else:
                tau = 1 + math.sqrt(1 + 4 * k * k)
                rho = (tau - math.sqrt(2 * tau)) / (2 * k)
                r = (1 + rho * rho) / (2 * rho)
            for a in range(T):
                U, V = np.random.random_sample(), np.random.random_sample()
                u_vm[i, b, 0, a, 0], u_vm[i, b, 0, a, 1] = U, V
                Z = math.cos(math.pi * U)
                W = (1 + r * Z) / (r + Z)
                Y = k * (r - W)
                if (Y * (2 - Y) - V >= 0) or (math.log(Y / V) + 1 - Y >= 0):
                    u_vm[i, b, 0, a, 2] = np.random.random_sample()
                    worst = max(worst, a + 1)
                    break
            else:
                raise AssertionError('numpy needed more than VM_MAX_TRIES attempts; pick another seed')
    got = orc.vm_mix_sample(loc.double(), kappa.double(), u_uni, u_vm, s)
    d = (got - x.double()).abs()
    d = torch.minimum(d, 2 * math.pi - d)
    REPORT.append((name, f'vm sample vs numpy stream (max tries {worst})', float(d.max()), 1e-6))
    assert float(d.max()) <= 1e-6, f'{name}: bounded sampler vs numpy.random.vonmises differ by {float(d.max()):.3e}'
    save(name, loc=loc, var=var, kappa=kappa, u_uniform=u_uni, u_vm=u_vm, x=x, seed=seed)


def _import_file(modname, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def case_losses(name, seed):
    """The two UNMODIFIED loss modules of the reference's callers on seeded inputs -> fixture `losses`:
    EPro-PnP-6DoF/lib/models/monte_carlo_pose_loss.py:9-35 and (behind oracle/mmdet_shim.py)
    EPro-PnP-Det/epropnp_det/models/losses/monte_carlo_pose_loss.py:31-66 with every weight / avg_factor / reduction
    combination mmdet accepts, loss_weight != 1, the EMA of norm_factor over two training calls, and a NaN object."""
    import mmdet_shim
    mmdet_shim.install()
    six = _import_file('ref_loss_6dof', os.path.join(ref.REF_ROOT, 'EPro-PnP-6DoF/lib/models/monte_carlo_pose_loss.py'))
    det = _import_file('ref_loss_det', os.path.join(ref.REF_ROOT, 'EPro-PnP-Det/epropnp_det/models/losses/monte_carlo_pose_loss.py'))
    g = torch.Generator().manual_seed(seed)
    S, B = 24, 6
    logw = torch.randn(S, B, generator=g) * 3
    logw[2, 4] = float('nan')                        # one object's loss is NaN -> 0
    ct = torch.rand(B, generator=g) * 5
    weight = torch.tensor([1.0, 0.0, 2.0, 0.5, 1.0, 3.0])
    nf_in = [torch.tensor(4.0), torch.tensor(2.5)]
    out = {}
    m6 = six.MonteCarloPoseLoss(init_norm_factor=2.0, momentum=0.1)
    out['six.call0'] = m6(logw.clone(), ct, nf_in[0]).detach()
    out['six.call1'] = m6(logw.clone(), ct, nf_in[1]).detach()
    out['six.norm_factor'] = m6.norm_factor.clone()
    m6.eval()
    out['six.eval'] = m6(logw.clone(), ct, nf_in[0]).detach()
    md = det.MonteCarloPoseLoss(loss_weight=0.5, init_norm_factor=2.0, momentum=0.1)
    out['det.call0'] = md(logw.clone(), ct, nf_in[0]).detach()
    out['det.call1'] = md(logw.clone(), ct, nf_in[1], weight=weight, avg_factor=3.5).detach()
    out['det.norm_factor'] = md.norm_factor.clone()
    md.eval()
    for red in ('mean', 'sum', 'none'):
        for wname, w in (('w0', None), ('w1', weight)):
            for aname, af in (('a0', None), ('a1', 3.5)):
                if af is not None and red == 'sum':
                    continue                          # mmdet raises ValueError; asserted separately in the test
                out[f'det.{red}.{wname}.{aname}'] = md(logw.clone(), ct, nf_in[0], weight=w, avg_factor=af,
                                                       reduction_override=red).detach()
    save(name, logw=logw, cost_target=ct, weight=weight, nf0=nf_in[0], nf1=nf_in[1], out=out)


def case_signatures(name):
    """Names, positional order and defaults of the reference's public API, read with `inspect` from the imported
    reference -> tests/golden/signatures.json (what tests/test_api_dropin.py compares the package against)."""
    import inspect
    import json
    m = ref.load_reference()
    table = {}

    def add(qual, fn):
        sig = inspect.signature(fn)
        table[qual] = [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
                       for p in sig.parameters.values()]
    E, LMm, C, CF, CO, D = (m[k] for k in ('epropnp', 'levenberg_marquardt', 'camera', 'cost_fun', 'common', 'distributions'))
    for cls in (E.EProPnPBase, E.EProPnP4DoF, E.EProPnP6DoF):
        for meth in ('__init__', 'forward', 'monte_carlo_forward', 'allocate_buffer', 'initial_fit', 'gen_new_distr',
                     'gen_old_distr', 'estimate_params'):
            add(f'epropnp.{cls.__name__}.{meth}', getattr(cls, meth))
    add('epropnp.cholesky_wrapper', E.cholesky_wrapper)
    for cls in (LMm.LMSolver, LMm.RSLMSolver):
        for meth in ('__init__', 'forward', 'solve', 'gn_step', 'pose_add'):
            add(f'levenberg_marquardt.{cls.__name__}.{meth}', getattr(cls, meth))
    add('levenberg_marquardt.RSLMSolver.center_based_init', LMm.RSLMSolver.center_based_init)
    add('levenberg_marquardt.solve_wrapper', LMm.solve_wrapper)
    for meth in ('__init__', 'set_param', 'project', 'project_jacobian', 'get_quaternion_transfrom_mat', 'reshape_',
                 'expand_', 'repeat_', 'shallow_copy'):
        add(f'camera.PerspectiveCamera.{meth}', getattr(C.PerspectiveCamera, meth))
    for fn in ('project_a', 'project_b'):
        add(f'camera.{fn}', getattr(C, fn))
    for cls in (CF.HuberPnPCost, CF.AdaptiveHuberPnPCost):
        for meth in ('__init__', 'set_param', 'compute', 'reshape_', 'expand_', 'repeat_', 'shallow_copy'):
            add(f'cost_fun.{cls.__name__}.{meth}', getattr(cls, meth))
    for fn in ('huber_kernel', 'huber_d_kernel'):
        add(f'cost_fun.{fn}', getattr(CF, fn))
    for fn in ('evaluate_pnp', 'pnp_normalize', 'pnp_denormalize', 'quaternion_to_rot_mat', 'yaw_to_rot_mat', 'skew'):
        add(f'common.{fn}', getattr(CO, fn))
    for cls in (D.AngularCentralGaussian, D.VonMisesUniformMix):
        for meth in ('__init__', 'log_prob') + (('rsample',) if cls is D.AngularCentralGaussian else ('sample',)):
            add(f'distributions.{cls.__name__}.{meth}', getattr(cls, meth))
    import mmdet_shim
    mmdet_shim.install()
    six = _import_file('ref_loss_6dof_sig', os.path.join(ref.REF_ROOT, 'EPro-PnP-6DoF/lib/models/monte_carlo_pose_loss.py'))
    det = _import_file('ref_loss_det_sig', os.path.join(ref.REF_ROOT, 'EPro-PnP-Det/epropnp_det/models/losses/monte_carlo_pose_loss.py'))
    for tag, mod in (('loss6dof', six), ('lossdet', det)):
        for meth in ('__init__', 'forward'):
            add(f'{tag}.MonteCarloPoseLoss.{meth}', getattr(mod.MonteCarloPoseLoss, meth))
    with open(os.path.join(OUT, name + '.json'), 'w') as f:
        json.dump(table, f, indent=1, sort_keys=True)


def _source_lines(path, first, last, must_contain):
    """The literal lines first..last (1-based, inclusive) of a reference source file, dedented; `must_contain` guards
    against the slice drifting if the reference checkout ever changes."""
    import textwrap
    with open(path) as f:
        lines = f.readlines()[first - 1:last]
    text = textwrap.dedent(''.join(lines))
    for needle in must_contain:
        assert needle in text, f'{path}:{first}-{last} no longer contains {needle!r}'
    return text


def case_preprocess():
    """Pins oracle/preprocess_oracle.py by EXECUTING the reference's own source lines (exec of the literal slice, inputs
    supplied through the namespace) -- fixtures prep_dense / prep_det."""
    import math
    import types
    import preprocess_oracle as pre
    # ---- 6-DoF training loop: lib/train.py:141-165 (x3d = noc * dim ... mean-normalised exp) ------------------------
    src = _source_lines(os.path.join(ref.REF_ROOT, 'EPro-PnP-6DoF/lib/train.py'), 141, 165,
                        ['x3d = noc * dim[..., None, None]', 'np.random.choice(64 * 64, size=64 * 64 // 8, replace=False)',
                         'w2d = (w2d - w2d.mean(dim=1, keepdim=True) - math.log(w2d.size(1))).exp() * scale[:, None, :]'])
    g = torch.Generator().manual_seed(50)
    bs, res = 2, 64                                   # the slice hard-codes a 64 x 64 map and 512 sampled pixels
    noc = torch.rand(bs, 3, res, res, generator=g) - 0.5
    dim = torch.rand(bs, 3, generator=g) + 0.5
    logit = torch.randn(bs, 2, res, res, generator=g) * 2
    scale = torch.rand(bs, 2, generator=g) * 3 + 0.1
    c_box = torch.tensor([[320.7, 240.2], [100.0, 400.9]])
    s_box = torch.tensor([128.9, 77.0])
    ns = dict(noc=noc, dim=dim, w2d=logit, scale=scale, s_box_var=s_box, c_box_var=c_box, bs=bs, torch=torch, np=np,
              math=math, cfg=types.SimpleNamespace(dataiter=types.SimpleNamespace(out_res=res)),
              pose_var=torch.zeros(bs, 3, 4), matrix_to_quaternion=lambda m: torch.zeros(m.shape[0], 4))
    np.random.seed(51)
    exec(compile(src, 'lib/train.py:141-165', 'exec'), ns)
    inds = ns['sample_inds']
    box = pre.box_grid_ref(c_box, s_box, res)
    o_x3d, o_x2d, o_w2d = pre.prepare_dense_ref(noc, dim, logit, scale, box, inds, 'mean_exp')
    check('prep_dense', 'x3d', ns['x3d'], o_x3d, 0.0)
    check('prep_dense', 'x2d', ns['x2d'], o_x2d, 0.0)
    check('prep_dense', 'w2d', ns['w2d'], o_w2d, 1e-7 * float(ns['w2d'].abs().max()))
    save('prep_dense', noc=noc, dim=dim, logit=logit, scale=scale, c_box=c_box, s_box=s_box, box=box, inds=inds,
         x3d=ns['x3d'], x2d=ns['x2d'], w2d=ns['w2d'])
    # ---- detection head: deform_pnp_head.py:418-421 (softmax over all heads' points, mask) and :873-874 ---------
    head = os.path.join(ref.REF_ROOT, 'EPro-PnP-Det/epropnp_det/models/dense_heads/deform_pnp_head.py')
    src1 = _source_lines(head, 418, 421, ['.softmax(dim=1)', 'w2d = w2d * mask_samples'])
    src2 = _source_lines(head, 873, 874, ['x3d = noc * dim_decoded[:, None]', 'w2d_scaled = w2d * scale[:, None, :]'])
    num_obj, heads, pts = 5, 4, 8
    raw = torch.randn(num_obj, heads, pts, 2, generator=g) * 2
    noc_d = torch.rand(num_obj, heads * pts, 3, generator=g) - 0.5
    dim_d = torch.rand(num_obj, 3, generator=g) + 0.5
    scale_d = torch.rand(num_obj, 2, generator=g) * 3 + 0.1
    ns1 = dict(w2d=raw.clone(), num_obj=num_obj, num_multihead_points=heads * pts, mask_samples=torch.ones(num_obj, heads, pts, 1),
               self=types.SimpleNamespace(num_heads=heads, num_points=pts))
    exec(compile(src1, 'deform_pnp_head.py:418-421', 'exec'), ns1)
    ns2 = dict(noc=noc_d, dim_decoded=dim_d, w2d=ns1['w2d'].reshape(num_obj, heads * pts, 2), scale=scale_d)
    exec(compile(src2, 'deform_pnp_head.py:873-874', 'exec'), ns2)
    o_x3d, o_w2d = pre.prepare_ref(noc_d, dim_d, raw.reshape(num_obj, heads * pts, 2), scale_d, 'softmax')
    check('prep_det', 'x3d', ns2['x3d'], o_x3d, 0.0)
    check('prep_det', 'w2d', ns2['w2d_scaled'], o_w2d, 1e-7 * float(ns2['w2d_scaled'].abs().max()))
    save('prep_det', logits=raw.reshape(num_obj, heads * pts, 2), noc=noc_d, dim=dim_d, scale=scale_d, x3d=ns2['x3d'],
         w2d=ns2['w2d_scaled'])


def case_callers():
    """tests/golden/callers_{notebook,linemod_train,det_head}.npz: the reference's OWN callers (literal source, read from the
    checkout at run time by oracle/run_callers.py) executed against the unmodified reference -- losses, pose_opt_plus, the
    gradients that reach the network outputs.  The GPU box runs the restated slices (oracle/callers_restated.py) on the package
    against these; here the restatement is pinned to the literal source on the reference itself: identical bits."""
    import subprocess
    import tempfile
    import run_callers as rcall
    with tempfile.TemporaryDirectory() as tmp:
        for scenario, fname in rcall.FIXTURES.items():
            lit = rcall.write_fixture(scenario, os.path.join(OUT, fname), tmp)
            run = rcall.FIXTURE_RUN[scenario]
            path = os.path.join(tmp, scenario + '_restated.npz')
            subprocess.run([sys.executable, os.path.join(HERE, 'run_callers.py'), '--side', 'reference', '--scenario', scenario, '--restated',
                            '--out', path, '--objects', str(run['objects']), '--steps', str(run['steps'])], check=True, capture_output=True,
                           env=dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES=''))
            res = dict(np.load(path))
            keys = [k for k in lit if not k.startswith(('spread.', 'meta.'))]
            assert set(keys) == set(res), (scenario, set(keys) ^ set(res))
            worst = max(float(np.abs(res[k].astype(lit[k].dtype) - lit[k]).max()) for k in keys)
            REPORT.append((fname[:-4], 'restated slice vs literal source (on the reference)', worst, 0.0))
            assert worst == 0.0, (scenario, worst)


def main():
    case_evaluate('eval6', 6, 6, 40, None, 10)
    case_evaluate('eval6_clip', 6, 6, 40, 'tight', 11)
    case_evaluate('eval4_clip', 4, 6, 40, 'tight', 12)
    case_lm('lm6_tr', 6, 8, 64, 5, False, None, 20)
    case_lm('lm6_gn', 6, 8, 64, 3, True, None, 21)
    case_lm('lm6_tr_clip', 6, 8, 48, 5, False, 'tight', 22)
    case_lm('lm4_tr', 4, 8, 64, 5, False, 'tensor', 23)
    case_lm('lm4_gn', 4, 8, 33, 5, True, None, 24)
    case_mc('mc6', 6, 4, 64, 64, 4, 3, 30)
    case_mc('mc6_n100', 6, 3, 100, 128, 4, 3, 31)
    case_mc('mc4', 4, 4, 64, 64, 4, 5, 32, bounds='tensor')
    case_mc('mc4_norm', 4, 3, 32, 32, 2, 5, 33, normalize=True, bounds='tensor')
    case_mc('mc6_demo', 6, 2, 64, 64, 4, 10, 34, rslm=dict(num_points=8, num_proposals=16, num_iter=5),
            with_pose_opt_plus=True, cam_kind='identity')
    case_mc('mc4_rslm', 4, 2, 48, 32, 4, 5, 35, rslm=dict(num_points=16, num_proposals=8, num_iter=3),
            with_pose_opt_plus=True, normalize=True, bounds='tensor')
    case_mc('mc6_tight', 6, 3, 48, 64, 4, 3, 36, bounds='tight')
    case_mc('mc6_k1', 6, 3, 40, 32, 1, 3, 38)
    case_mc('mc4_det', 4, 2, 64, 32, 4, 5, 37, rslm=dict(num_points=16, num_proposals=64, num_iter=3), normalize=True,
            bounds='tensor', with_pose_opt_plus=True)
    case_vm_numpy('vm_numpy', 7)
    case_losses('losses', 8)
    case_preprocess()
    case_signatures('signatures')
    case_callers()
    w = max(len(n) for n, *_ in REPORT)
    for n, k, d, tol in REPORT:
        print(f'{n:<{w}}  {k:<32} maxdiff {d:.3e}   tol {tol:.1e}')
    with open(os.path.join(OUT, 'PINNING_REPORT.txt'), 'w') as f:
        f.write('reference (/root/reference/epropnp, unmodified, injected noise) vs oracle/epropnp_oracle.py\n')
        for n, k, d, tol in REPORT:
            f.write(f'{n:<{w}}  {k:<32} maxdiff {d:.3e}   tol {tol:.1e}\n')


if __name__ == '__main__':
    main()
