"""AMIS sampler (forward + backward kernels) vs the reference fixtures and the restatement, with injected noise."""
import math

import pytest
import torch

import epropnp_oracle as orc
from helpers import assert_within_spread, load_golden, make_layer_objects, pack_noise, rel_per_object, set_tune

POSE_TOL = 1e-4     # north-star tolerance on the pose
KL_TOL = 1e-3       # north-star tolerance on the Monte-Carlo (KL) loss
GRAD_TOL = 2e-4     # backward kernel vs autograd of the oracle at fixed samples (test below) holds this bar outright


def _layer(dof, S, K, lm_iter, normalize=False, rslm=None):
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver, RSLMSolver
    init = RSLMSolver(dof=dof, **rslm) if rslm else None
    cls = EProPnP6DoF if dof == 6 else EProPnP4DoF
    return cls(mc_samples=S, num_iter=K, normalize=normalize, solver=LMSolver(dof=dof, num_iter=lm_iter, init_solver=init))


def run_layer(backend, prob, noise, dof, S, K, lm_iter, normalize=False, relative_delta=0.5):
    """monte_carlo_forward + MC loss + backward through the product API; same contract as orc.run_mc."""
    p, cam, cf = make_layer_objects(prob, backend, relative_delta=relative_delta)
    x3d, x2d, w2d = (p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    cf.set_param(x2d.detach(), w2d)
    layer = _layer(dof, S, K, lm_iter, normalize)
    out = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=p['pose_init'], force_init_solve=False,
                                    with_cost=True, noise=pack_noise(noise, dof).to(backend))
    pose_opt, cost, _, samples, logw, cost_init = out
    loss_obj = cost_init + torch.logsumexp(logw, dim=0)
    loss_obj.mean().backward()
    res = dict(pose_opt=pose_opt, cost=cost, pose_samples=samples, logweights=logw, cost_init=cost_init,
               loss_obj=loss_obj, gx3d=x3d.grad, gx2d=x2d.grad, gw2d=w2d.grad)
    return {k: v.detach().cpu() for k, v in res.items()}


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-12)).item()


@pytest.mark.parametrize('name', ['mc6', 'mc6_n100', 'mc4', 'mc4_norm', 'mc6_tight', 'mc6_k1'])
def test_monte_carlo_forward_backward_matches_reference(backend, name):
    g = load_golden(name)
    dof, S, K = int(g['dof']), int(g['S']), int(g['K'])
    r = run_layer(backend, g['prob'], g['noise'], dof, S, K, int(g['lm_iter']), normalize=bool(g['normalize']))
    ref, o64, sp = g['ref'], g['o64'], g['spread']
    # Yardstick for every bar below: `spread` = how far the REFERENCE's own fp32 result moves, per object, when its
    # inputs move by <= 3 ulp, plus its fp32-vs-fp64 drift (oracle/make_golden.py, orc.rounding_spread).  It is ~1e-6
    # for well-conditioned objects and large exactly where the reference is ill-conditioned: a trust-region
    # accept/reject decision on a knife edge (mc4_norm: 2.8e-4 on the pose), a flat LM valley under clip_jac whose
    # 3e-5 pose ambiguity moves proposal 0 and, through the cond-1e5 AMIS refits, the loss by 1.4e-3 (mc6_tight).
    # Errors and spreads are matched by rank over the objects (helpers.assert_within_spread).
    assert_within_spread((r['pose_opt'] - ref['pose_opt']).abs().max(-1).values, sp['pose_opt'], POSE_TOL, what='pose_opt')
    assert_within_spread((r['cost'] - ref['cost']).abs() / ref['cost'].abs().clamp(min=1e-30), sp['cost'], 1e-5, what='cost')
    torch.testing.assert_close(r['cost_init'], ref['cost_init'], rtol=2e-5, atol=1e-5)
    # KL / Monte-Carlo loss: per object, and the batch mean (what training sees) strictly within the north-star bar
    assert_within_spread((r['loss_obj'] - ref['loss_obj']).abs(), sp['loss_obj'], KL_TOL, what='loss_obj')
    assert abs(r['loss_obj'].mean().item() - ref['loss_obj'].mean().item()) <= KL_TOL
    assert abs(r['loss_obj'].mean().item() - o64['loss_obj'].mean().item()) <= KL_TOL
    # gradients of the loss, per object relative to the object's largest entry
    for k in ('gx3d', 'gx2d', 'gw2d'):
        assert_within_spread(rel_per_object(r[k], ref[k]), sp[k], GRAD_TOL, what=k)
    assert (r['pose_samples'] - ref['pose_samples']).abs().max().item() <= 2e-2


def _mixture_logq(samples, props, dof, K):
    """log of the equal-weight mixture of the K fitted proposals (kernel's own records) at `samples` (S,B,pose_len)."""
    def tril(v, n):
        L = torch.zeros(v.shape[:-1] + (n, n))
        idx = torch.tril_indices(n, n)
        L[..., idx[0], idx[1]] = v
        return L
    lq = []
    for j in range(K):
        rec = props[:, j]
        lp = orc.student_t_logprob(samples[..., :3], rec[:, 0:3], tril(rec[:, 3:9], 3))
        if dof == 6:
            lp = lp + orc.acg_logprob(samples[..., 3:], tril(rec[:, 16:26], 4))
        else:
            lp = lp + orc.vm_mix_logprob(samples[..., 3:], rec[:, 16:17], rec[:, 17:18]).squeeze(-1)
        lq.append(lp)
    return torch.logsumexp(torch.stack(lq, 0), 0) - torch.log(torch.tensor(float(K)))


@pytest.mark.parametrize('dof,S,K,N', [(6, 64, 4, 96), (4, 64, 4, 96), (6, 60, 3, 70), (6, 200, 2, 130), (4, 100, 1, 33),
                                       (6, 320, 2, 600), (6, 48, 2, 1300), (6, 32, 2, 2300), (6, 32, 2, 512), (4, 32, 2, 2040),
                                       (6, 32, 2, 800), (6, 40, 2, 2048),
                                       # sample counts after which the LDS layout is not a multiple of 4 floats (rred / cpart are read as float4)
                                       (4, 6, 2, 64), (4, 510, 2, 96), (6, 9, 3, 64), (6, 27, 3, 200)])
def test_logweights_consistent_with_own_samples(backend, dof, S, K, N):
    """Tight check that does not depend on the (ill-conditioned) proposal fit: recompute cost and proposal mixture
    density with the oracle AT THE KERNEL'S OWN samples and fitted proposals; log-weights must agree to 1e-4."""
    from epropnp import functional as F
    B = 3
    prob = orc.make_problem(B, N, dof, seed=7)
    noise = orc.make_noise(B, S, K, dof, seed=8)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    pose_opt, pose_cov, _ = F.lm_solve(hp, p['pose_init'], 3, with_pose_cov=True)
    samples, logw, props = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=pack_noise(noise, dof).to(backend),
                                          with_proposals=True)
    samples, logw, props = samples.cpu(), logw.cpu(), props.cpu()
    ocam = orc.Cam(prob['cam_mats'], 0.1)
    cost = orc.evaluate(prob['x3d'], prob['x2d'], prob['w2d'], samples, ocam, prob['delta'], want_cost=True)[1]
    mix = _mixture_logq(samples, props, dof, K)
    expect = -cost - mix
    assert (logw - expect).abs().max().item() <= 2e-4 * max(1.0, expect.abs().max().item())


@pytest.mark.parametrize('N,bounds', [(800, None), (1100, 'tight'), (2048, None)])
def test_forward_point_tiles_in_chunks_through_the_registers(backend, monkeypatch, N, bounds):
    """Beyond 48 point tiles per object the register-mode forward keeps 8 tiles per wave and takes the object's tiles through the
    registers in chunks of 32 per iteration (instead of 16 resident tiles per wave / 8-wave workgroups); EPROPNP_TUNE=fwd_no_chunks
    keeps the old shapes.  Same samples in the first iteration (they do not depend on the sweep), costs to summation order."""
    from epropnp import functional as F
    B, S, K, dof = 2, 64, 2, 6
    prob = orc.make_problem(B, N, dof, seed=51, bounds=bounds)
    noise = pack_noise(orc.make_noise(B, S, K, dof, seed=52), dof).to(backend)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    pose_opt, pose_cov, _ = F.lm_solve(hp, p['pose_init'], 3, with_pose_cov=True)
    set_tune(monkeypatch)
    s1, w1 = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise)
    again = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise)
    assert torch.equal(s1, again[0]) and torch.equal(w1, again[1])
    set_tune(monkeypatch, fwd_no_chunks=True)
    s2, w2 = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise)
    s = S // K
    assert torch.equal(s1[:s], s2[:s])
    assert (w1[:s] - w2[:s]).abs().max().item() <= 2e-5 * max(1.0, w2[:s].abs().max().item())
    assert (s1 - s2).abs().max().item() < 5e-4
    assert (torch.logsumexp(w1, 0) - torch.logsumexp(w2, 0)).abs().max().item() < 1e-3


@pytest.mark.parametrize('impl,N', [('valu', 150), ('mfma', 150), ('mfma', 300), ('mfma', 16), ('valu', 16), ('mfma', 17)])
def test_backward_matches_autograd_of_oracle_at_fixed_samples(backend, monkeypatch, impl, N):
    """The backward kernels alone (VALU sweep / MFMA projection; N=300 loops over two point chunks): arbitrary
    upstream gradients, samples fixed -> compare with autograd through the oracle's evaluate (which is what the
    reference's autograd replays)."""
    from epropnp import functional as F
    set_tune(monkeypatch, bwd_impl=impl)
    for dof, bounds in ((6, None), (4, 'tight'), (6, 'tight')):
        B, S = 3, 40
        if N == 17:                               # pose table larger than LDS: the MFMA launcher hands over to the VALU kernel
            B, S = 1, 2800
        prob = orc.make_problem(B, N, dof, seed=11, bounds=bounds)
        g = torch.Generator().manual_seed(5)
        poses = prob['pose_gt'].unsqueeze(0).repeat(S, 1, 1)
        poses[..., :3] += 0.2 * torch.randn(S, B, 3, generator=g)
        if dof == 6:
            q = poses[..., 3:] + 0.1 * torch.randn(S, B, 4, generator=g)
            poses[..., 3:] = q / q.norm(dim=-1, keepdim=True)
        else:
            poses[..., 3] += 0.3 * torch.randn(S, B, generator=g)
        poses[0, 0, 2] = -1.0                     # behind the camera: z clamp active
        g_logw = torch.randn(S, B, generator=g)
        g_logw[3] = 0.0                           # exact zeros are skipped by the kernel
        g_logw[5:9, B - 1] *= 1e-12               # below the 2^-30 relative drop threshold
        g_init = torch.randn(B, generator=g)
        x3d, x2d, w2d, delta = (prob[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d', 'delta'))
        ocam = orc.Cam(prob['cam_mats'], 0.1, prob.get('lb'), prob.get('ub'))
        c_s = orc.evaluate(x3d, x2d, w2d, poses, ocam, delta, want_cost=True)[1]
        c_i = orc.evaluate(x3d, x2d, w2d, prob['pose_init'], ocam, delta, want_cost=True)[1]
        ((-c_s) * g_logw).sum().add((c_i * g_init).sum()).backward()

        p, cam, cf = make_layer_objects(prob, backend)
        hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
        gx3d, gx2d, gw2d, gdel = F.amis_backward(hp, poses.to(backend), g_logw.to(backend), p['pose_init'],
                                                 g_init.to(backend))
        for mine, ref in ((gx3d, x3d.grad), (gx2d, x2d.grad), (gw2d, w2d.grad), (gdel, delta.grad)):
            assert _rel(mine.cpu(), ref) <= 2e-4, (dof, bounds, _rel(mine.cpu(), ref))


@pytest.mark.parametrize('delta,z_min', [(1e-4, 0.1), (3e3, 0.1), (0.7, 0.0), (0.7, 1e-6), (0.7, 4.5), (0.0, 0.1),
                                         (1e20, 0.1), (float('inf'), 0.1)])
def test_sweeps_over_the_range_of_delta_and_z_min(backend, delta, z_min):
    """Both AMIS sweeps carry the residuals in units of the Huber threshold (weights pre-divided by delta, min(rho, delta)
    through a clamp modifier) and take the depth clamp / its gradient mask from integer-max and clamped-fma tricks that
    assume z_min >= 0: cover a tiny and a huge threshold, z_min = 0 / tiny / beyond most of the points, and delta = 0
    (cost identically 0: finite outputs, no NaN from 1 / delta), and a threshold of 1e20 / inf (the Huber kernel switched off:
    1 / delta would be 0 and delta^2 inf; huber_scale caps the threshold at 1e12, where min(rho, delta) never binds)."""
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import HuberPnPCost
    B, N, S, K, dof = 3, 150, 48, 2, 6
    prob = orc.make_problem(B, N, dof, seed=23)
    prob['delta'] = torch.full((B,), delta)
    p, _, _ = make_layer_objects(prob, backend)
    cam = PerspectiveCamera(cam_mats=p['cam_mats'], z_min=z_min)
    cf = HuberPnPCost(delta=p['delta'])
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    # z_min = 0 divides by the raw depth: keep the proposal tight enough that no point crosses the image plane there
    # (the reference returns inf / nan for such samples as well)
    tz_var = 0.3 if z_min > 0 else 1e-3
    cov = (torch.eye(6) * torch.tensor([0.02, 0.02, tz_var, 1e-3, 1e-3, 1e-3])).expand(B, 6, 6).contiguous()
    noise = orc.make_noise(B, S, K, dof, seed=24)
    samples, logw, props = F.amis_forward(hp, p['pose_gt'], cov.to(backend), S, K, noise=pack_noise(noise, dof).to(backend),
                                          with_proposals=True)
    samples, logw, props = samples.cpu(), logw.cpu(), props.cpu()
    assert bool(torch.isfinite(logw).all())
    x3d, x2d, w2d, dl = (prob[k].double().clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d', 'delta'))
    if delta == float('inf'):     # autograd of where(rho <= delta, ., delta rho - delta^2 / 2) is inf * 0 = NaN at delta = inf (in
        dl = torch.full((B,), 1e20, dtype=torch.float64, requires_grad=True)    # the reference too): same quadratic at 1e20
    ocam = orc.Cam(prob['cam_mats'].double(), z_min)
    cost = orc.evaluate(x3d, x2d, w2d, samples.double(), ocam, dl, want_cost=True)[1]
    expect = -cost.detach().float() - _mixture_logq(samples, props, dof, K)
    assert (logw - expect).abs().max().item() <= 2e-4 * max(1.0, expect.abs().max().item())
    g = torch.Generator().manual_seed(3)
    g_logw = torch.randn(S, B, generator=g)
    ((-cost) * g_logw.double()).sum().backward()
    grads = F.amis_backward(hp, samples.to(backend), g_logw.to(backend), None, None)
    for i, (mine, ref) in enumerate(zip(grads, (x3d.grad, x2d.grad, w2d.grad, dl.grad))):
        assert bool(torch.isfinite(mine).all())
        if i == 3 and delta > 1e12:
            # every residual is an inlier: d cost / d delta is exactly 0 in the reference; the kernel leaves rounding residue
            # of rho in units of the capped threshold
            assert mine.abs().max().item() <= 1e-7 * 1e12 * N * S * 1e-6, (delta, mine)
        elif i == 3:
            # d cost / d delta = sum a max(rho - delta, 0): does not vanish at delta = 0, and is exactly 0 in the reference
            # when every residual is an inlier (delta = 3e3), where the kernel's fused rho - min(rho, delta) leaves the
            # rounding error of rho (<= ulp / 2, rho <= delta) per point-pose instead
            err = (mine.cpu().double() - ref).abs()
            bound = GRAD_TOL * ref.abs().max() + 1e-7 * max(delta, 1.0) * N * g_logw.abs().sum(0).double()
            assert bool((err <= bound).all()), (delta, z_min, err, bound)
        elif delta > 0:
            assert _rel(mine.cpu().double(), ref) <= GRAD_TOL, (delta, z_min, i, _rel(mine.cpu().double(), ref))
        else:                         # delta = 0 is carried as 1e-12 (huber_scale): gradients of that size instead of exact zeros
            assert mine.abs().max().item() <= 1e-5


def test_degenerate_objects_stay_finite_and_do_not_disturb_their_neighbours(backend):
    """One batch with a healthy object, an object whose weights are all zero (adaptive delta = 0), one with a single
    weighted point (pose undetermined), one with coordinates of 1e4 and one whose image points coincide (delta = 0 with
    non-zero residuals): nothing turns NaN / inf, and every object but the undetermined one matches the oracle."""
    B, N, S, K = 5, 100, 64, 2
    prob = orc.make_problem(B, N, 6, seed=5)
    prob['w2d'][1] = 0.0
    prob['w2d'][2] = 0.0
    prob['w2d'][2, 3] = 1.0
    prob['x3d'][3] *= 1e4
    prob['x2d'][4] = prob['x2d'][4, :1]
    noise = orc.make_noise(B, S, K, 6, seed=6)
    mine = run_layer(backend, prob, noise, 6, S, K, 3)
    ref = orc.run_mc(prob, noise, 6, S, K, 3)
    for k, v in mine.items():
        assert bool(torch.isfinite(v).all()), k
    ok = torch.tensor([0, 1, 3, 4])
    assert ((mine['loss_obj'][ok] - ref['loss_obj'][ok]).abs() <= 1e-4 * ref['loss_obj'][ok].abs() + 2e-3).all()
    assert (mine['pose_opt'][ok] - ref['pose_opt'][ok]).abs().max().item() <= 2e-2      # (object 3 sits 1e4 away)
    assert (mine['pose_opt'][0] - ref['pose_opt'][0]).abs().max().item() <= POSE_TOL
    for k in ('gx3d', 'gx2d', 'gw2d'):
        assert _rel(mine[k][0], ref[k][0]) <= 5e-3, k


@pytest.mark.parametrize('impl,nsplit', [('mfma', 1), ('mfma', 2), ('valu', 1)])
def test_backward_without_pose_init(backend, monkeypatch, impl, nsplit):
    """pose_init / grad_cost_init are optional (NULL in the C ABI): same gradients as a zero upstream gradient on the cost
    of pose_init.  (On the GPU the wave-uniform g_init[b] is a scalar load; it used to sit inside a per-lane conditional,
    where it is executed even when no lane takes the arm, and faulted on the NULL pointer.)"""
    from epropnp import functional as F
    set_tune(monkeypatch, bwd_impl=impl)
    B, N, S, dof = 3, 150, 48, 6
    prob = orc.make_problem(B, N, dof, seed=29)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    g = torch.Generator().manual_seed(2)
    poses = prob['pose_gt'].unsqueeze(0).repeat(S, 1, 1)
    poses[..., :3] += 0.1 * torch.randn(S, B, 3, generator=g)
    g_logw = torch.randn(S, B, generator=g).to(backend)
    without = F.amis_backward(hp, poses.to(backend), g_logw, None, None, nsplit=nsplit)
    zero = F.amis_backward(hp, poses.to(backend), g_logw, p['pose_init'], torch.zeros(B, device=backend), nsplit=nsplit)
    for a, b in zip(without, zero):
        assert bool(torch.isfinite(a).all())
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7 * float(b.abs().max()))


@pytest.mark.parametrize('dof,bounds,B,N,S,shape', [(6, None, 3, 300, 70, None), (4, 'tight', 5, 128, 33, None), (6, 'tight', 2, 512, 40, None),
                                                    (6, None, 4, 40, 200, '4,1'), (4, None, 3, 700, 64, '4,2')])
def test_backward_bf16_split_projection_against_the_fp32_matrix_path(backend, monkeypatch, dof, bounds, B, N, S, shape):
    """The MFMA backward projects on v_mfma_f32_16x16x32_bf16 with every fp32 operand split into three bf16 pieces (default for
    <= 4 resident point tiles); EPROPNP_BWD_PROJ=f32 keeps the fp32 MFMA.  Same gradients to fp32-level rounding (8 of the 9 cross
    products per real k are carried: ~1.4e-7 per projection), poses behind the camera and zero / negligible weights included,
    and run-to-run the same bits."""
    from epropnp import functional as F
    set_tune(monkeypatch, bwd_impl='mfma', bwd_mfma=shape or None)
    prob = orc.make_problem(B, N, dof, seed=41, bounds=bounds)
    g = torch.Generator().manual_seed(8)
    poses = prob['pose_gt'].unsqueeze(0).repeat(S, 1, 1)
    poses[..., :3] += 0.2 * torch.randn(S, B, 3, generator=g)
    if dof == 6:
        q = poses[..., 3:] + 0.1 * torch.randn(S, B, 4, generator=g)
        poses[..., 3:] = q / q.norm(dim=-1, keepdim=True)
    else:
        poses[..., 3] += 0.3 * torch.randn(S, B, generator=g)
    poses[0, 0, 2] = -1.0
    poses[1, :, :3] *= 300.0                       # far away: large (K t) entries against small rotation entries
    g_logw, g_init = torch.randn(S, B, generator=g), torch.randn(B, generator=g)
    g_logw[3] = 0.0
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    args = (hp, poses.to(backend), g_logw.to(backend), p['pose_init'], g_init.to(backend))
    monkeypatch.setenv('EPROPNP_BWD_PROJ', 'bf16')
    split = F.amis_backward(*args)
    again = F.amis_backward(*args)
    monkeypatch.setenv('EPROPNP_BWD_PROJ', 'f32')
    full = F.amis_backward(*args)
    for a, b, c in zip(split, again, full):
        assert torch.equal(a, b)
        assert _rel(a.cpu(), c.cpu()) <= 2e-5, _rel(a.cpu(), c.cpu())
    assert not all(torch.equal(a, c) for a, c in zip(split, full))       # the switch does select another kernel


@pytest.mark.parametrize('dof,bounds,N,S', [(6, None, 300, 70), (4, 'tight', 128, 33), (6, 'tight', 512, 40)])
def test_backward_split_over_workgroups(backend, dof, bounds, N, S):
    """Few objects: the point chunks of an object dealt to several workgroups (epropnp_amis_backward_split).  Per-point
    gradients are bit-identical to the one-workgroup kernel, grad_delta (added from per-workgroup partials) to rounding."""
    from epropnp import functional as F
    B = 3
    prob = orc.make_problem(B, N, dof, seed=13, bounds=bounds)
    g = torch.Generator().manual_seed(6)
    poses = prob['pose_gt'].unsqueeze(0).repeat(S, 1, 1)
    poses[..., :3] += 0.2 * torch.randn(S, B, 3, generator=g)
    if dof == 6:
        q = poses[..., 3:] + 0.1 * torch.randn(S, B, 4, generator=g)
        poses[..., 3:] = q / q.norm(dim=-1, keepdim=True)
    else:
        poses[..., 3] += 0.3 * torch.randn(S, B, generator=g)
    g_logw, g_init = torch.randn(S, B, generator=g), torch.randn(B, generator=g)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    args = (hp, poses.to(backend), g_logw.to(backend), p['pose_init'], g_init.to(backend))
    one = F.amis_backward(*args, nsplit=1)
    assert F.backward_split(32, 512, 512) == 8 and F.backward_split(600, 128, 128) == 1 and F.backward_split(4096, 512, 512) == 1
    assert F.backward_split(32, 512, 4096) == 1 and F.backward_split(2, 64, 64) == 1
    for nsplit in sorted(n for n in {2, 3, min(8, (N + 63) // 64)} if n <= (N + 63) // 64):
        many = F.amis_backward(*args, nsplit=nsplit)
        for a, b in zip(many[:3], one[:3]):
            assert torch.equal(a, b), nsplit
        assert _rel(many[3].cpu(), one[3].cpu()) <= 1e-5
    with pytest.raises(RuntimeError, match='nsplit'):
        F.amis_backward(*args, nsplit=(N + 63) // 64 + 1)


@pytest.mark.gpu
@pytest.mark.parametrize('N,nsplit', [(2048, 8), (2048, 16), (4096, 8), (4096, 16)])
def test_backward_split_dense_crops_matches_autograd_of_oracle(N, nsplit):
    """The backward's split over 8 / 16 workgroups at the dense LineMOD point counts (32 crops x 64 x 64 correspondences take
    nsplit = 16 in the layer, functional.backward_split): arbitrary upstream gradients at fixed samples against autograd
    through the oracle's evaluate (what the reference's autograd replays), with per-object tensor bounds and z_min 0.01 as
    lib/train.py:168-173 sets them, and bit-identical per-point gradients against the one-workgroup kernel."""
    import install as emu
    from epropnp import functional as F
    emu.uninstall()
    dev = torch.device('cuda:0')
    B, S, dof = 4, 96, 6
    prob = orc.make_problem(B, N, dof, seed=41, relative_delta=0.1)
    lo, hi = prob['x2d'].amin(1), prob['x2d'].amax(1)
    unit = (hi - lo).amax(-1, keepdim=True) / 64.0
    prob['lb'], prob['ub'], prob['z_min'] = (lo - 30 * unit).contiguous(), (hi + 30 * unit).contiguous(), 0.01
    g = torch.Generator().manual_seed(7)
    poses = prob['pose_gt'].unsqueeze(0).repeat(S, 1, 1)
    poses[..., :3] += 0.2 * torch.randn(S, B, 3, generator=g)
    q = poses[..., 3:] + 0.1 * torch.randn(S, B, 4, generator=g)
    poses[..., 3:] = q / q.norm(dim=-1, keepdim=True)
    poses[0, 0, 2] = -1.0                     # behind the camera: depth clamp active for every point of that sample
    g_logw, g_init = torch.randn(S, B, generator=g), torch.randn(B, generator=g)
    x3d, x2d, w2d, delta = (prob[k].double().clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d', 'delta'))
    ocam = orc.Cam(prob['cam_mats'].double(), 0.01, prob['lb'].double(), prob['ub'].double())
    c_s = orc.evaluate(x3d, x2d, w2d, poses.double(), ocam, delta, want_cost=True)[1]
    c_i = orc.evaluate(x3d, x2d, w2d, prob['pose_init'].double(), ocam, delta, want_cost=True)[1]
    ((-c_s) * g_logw.double()).sum().add((c_i * g_init.double()).sum()).backward()
    p, cam, cf = make_layer_objects(prob, dev)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    args = (hp, poses.to(dev), g_logw.to(dev), p['pose_init'], g_init.to(dev))
    many = F.amis_backward(*args, nsplit=nsplit)
    for mine, ref in zip(many, (x3d.grad, x2d.grad, w2d.grad, delta.grad)):
        assert _rel(mine.cpu().double(), ref) <= 2e-4, (N, nsplit, _rel(mine.cpu().double(), ref))
    one = F.amis_backward(*args, nsplit=1)
    for a, b in zip(many[:3], one[:3]):
        assert torch.equal(a, b)
    assert _rel(many[3].cpu(), one[3].cpu()) <= 1e-5


def test_philox_sampler_statistics(backend):
    """Production mode (on-device Philox): loss agrees with the injected-noise oracle within Monte-Carlo error,
    and two calls draw different samples."""
    B, N, S, K, dof = 6, 64, 256, 4, 6
    prob = orc.make_problem(B, N, dof, seed=21)
    p, cam, cf = make_layer_objects(prob, backend, relative_delta=0.5)
    cf.set_param(p['x2d'], p['w2d'])
    layer = _layer(dof, S, K, 3)
    layer.seed = 1234
    outs = [layer.monte_carlo_forward(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'],
                                      force_init_solve=False) for _ in range(2)]
    assert (outs[0][3] - outs[1][3]).abs().max() > 1e-3          # fresh draws per call
    q = outs[0][3][..., 3:]
    assert (q.norm(dim=-1) - 1).abs().max() < 1e-5
    lse = torch.stack([torch.logsumexp(o[4], 0) for o in outs]).cpu()
    noise = orc.make_noise(B, S, K, dof, seed=3)
    o = orc.run_mc(prob, noise, dof, S, K, 3)
    ref_lse = o['loss_obj'] - o['cost_init']
    # seed-to-seed std of the per-object estimate is ~0.05-0.1 at S=256
    assert (lse - ref_lse).abs().max().item() < 0.6
    assert (lse.mean(1) - ref_lse.mean()).abs().max().item() < 0.25


def test_forward_kernel_variants_agree(backend, monkeypatch):
    """EPROPNP_TUNE=fwd_impl=valu selects the register-resident VALU sweep kernel; same samples, same weights."""
    from epropnp import functional as F
    B, N, S, K, dof = 3, 200, 64, 4, 6
    prob = orc.make_problem(B, N, dof, seed=17)
    noise = pack_noise(orc.make_noise(B, S, K, dof, seed=18), dof).to(backend)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    pose_opt, pose_cov, _ = F.lm_solve(hp, p['pose_init'], 3, with_pose_cov=True)
    set_tune(monkeypatch)
    s1, w1 = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise)
    set_tune(monkeypatch, fwd_impl='valu')
    s2, w2 = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise)
    assert (s1 - s2).abs().max().item() < 5e-4
    assert (torch.logsumexp(w1, 0) - torch.logsumexp(w2, 0)).abs().max().item() < 1e-3
    # on-device Philox draws: the MFMA kernel pre-generates the base noise into LDS shared with its pose table, the VALU
    # kernel draws inline -- same counters, same samples
    set_tune(monkeypatch)
    p1, q1 = F.amis_forward(hp, pose_opt, pose_cov, S, K, seed=42, offset=7)
    set_tune(monkeypatch, fwd_impl='valu')
    p2, q2 = F.amis_forward(hp, pose_opt, pose_cov, S, K, seed=42, offset=7)
    assert (p1 - p2).abs().max().item() < 5e-4
    assert (torch.logsumexp(q1, 0) - torch.logsumexp(q2, 0)).abs().max().item() < 1e-3
    # MFMA kernel, LDS-chunk mode (used for N > 2048), forced on this small problem with 2 waves
    set_tune(monkeypatch)
    set_tune(monkeypatch, fwd_mfma='2,0')
    s3, w3 = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise)
    assert (s1 - s3).abs().max().item() < 5e-4
    assert (torch.logsumexp(w1, 0) - torch.logsumexp(w3, 0)).abs().max().item() < 1e-3
    # the projection on the fp32 MFMA instead of the bf16x3 split (register mode): the first iteration's samples do not depend on
    # the sweep at all and its log-weights only through the costs -- fp32-level agreement; later iterations through the refit
    set_tune(monkeypatch)
    monkeypatch.setenv('EPROPNP_FWD_PROJ', 'f32')
    s4, w4 = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise)
    monkeypatch.delenv('EPROPNP_FWD_PROJ')
    s = S // K
    assert torch.equal(s1[:s], s4[:s])
    assert not torch.equal(w1, w4)                                  # (the switch does select another kernel)
    assert (s1 - s4).abs().max().item() < 5e-4
    assert (torch.logsumexp(w1, 0) - torch.logsumexp(w4, 0)).abs().max().item() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('dof,bounds,B,N,S,K', [(6, None, 32, 512, 512, 4), (6, 'tight', 20, 300, 96, 3), (4, 'tensor', 64, 128, 128, 4),
                                               (6, None, 3, 1000, 64, 2), (4, 'tensor', 16, 512, 64, 2), (6, None, 8, 600, 1024, 2),
                                               (6, 'tensor', 5, 2048, 60, 1), (6, None, 32, 4096, 128, 4), (4, 'tensor', 9, 3000, 64, 2)])
def test_forward_split_over_workgroups(monkeypatch, dof, bounds, B, N, S, K):
    """Few objects: G workgroups share an object's point tiles and exchange partial costs through global memory behind a
    per-object arrival counter (csrc/amis_forward_mfma.hip).  Same samples as the one-workgroup kernel (the sampler runs
    redundantly in every part), log-weights equal up to the summation order of the partial costs, the oracle's log-weights
    at the kernel's own samples, bit-reproducible, and usable twice at once on two streams."""
    import install as emu
    emu.uninstall()
    from epropnp import functional as F
    dev = torch.device('cuda:0')
    prob = orc.make_problem(B, N, dof, seed=19, bounds=bounds)
    noise = pack_noise(orc.make_noise(B, S, K, dof, seed=20), dof).to(dev)
    p, cam, cf = make_layer_objects(prob, dev)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    pose_opt, pose_cov, _ = F.lm_solve(hp, p['pose_init'], 3, with_pose_cov=True)
    monkeypatch.setenv('EPROPNP_FWD_SPLIT', '1')
    s1, w1, pr1 = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise, with_proposals=True)
    monkeypatch.delenv('EPROPNP_FWD_SPLIT')
    outs = [F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise, with_proposals=True) for _ in range(2)]
    s2, w2, pr2 = outs[0]
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))                       # bit-reproducible
    # later iterations' samples come from proposals refitted to weights whose costs were summed in another order
    assert (s1 - s2).abs().max().item() <= 5e-4 * max(1.0, s1.abs().max().item())
    # (the costs behind the log-weights are sums over N points of magnitude |logw|: the two kernels add them in another order)
    assert (torch.logsumexp(w1, 0) - torch.logsumexp(w2, 0)).abs().max().item() < 1e-3 + 2e-6 * w1[torch.isfinite(w1)].abs().max().item()
    samples, logw, props = s2.cpu(), w2.cpu(), pr2.cpu()
    ocam = orc.Cam(prob['cam_mats'].double(), 0.1, None if bounds is None else prob['lb'].double(),
                   None if bounds is None else prob['ub'].double())
    cost = orc.evaluate(prob['x3d'].double(), prob['x2d'].double(), prob['w2d'].double(), samples.double(), ocam,
                        prob['delta'].double(), want_cost=True)[1]
    expect = -cost.float() - _mixture_logq(samples, props, dof, K)
    assert (logw - expect).abs().max().item() <= 2e-4 * max(1.0, expect.abs().max().item())
    # two split launches in flight at once (their parts wait for each other: both grids must fit the device together)
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    both = []
    for q in st:
        q.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(q):
            both.append(F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise))
    torch.cuda.synchronize()
    F.flush_status()
    assert all(torch.equal(o[1], w2) for o in both)


@pytest.mark.parametrize('dof,bounds,N,G,S,K', [(6, None, 300, 4, 48, 3), (4, 'tensor', 260, 4, 32, 2), (6, 'tight', 520, 8, 40, 2)])
def test_forward_split_recomputes_missing_parts(backend, monkeypatch, dof, bounds, N, G, S, K):
    """The split AMIS forward never depends on its sibling workgroups being resident: partial costs that are not there within
    EPROPNP_SPLIT_TIMEOUT_CYCLES are recomputed -- the missing part's point tiles loaded into this workgroup's registers and
    swept by the same lanes in the same order.  CPU emulation (workgroups run one after another): every part recomputes the
    later ones, the log-weights must still be the oracle's at the kernel's own samples.  GPU: a zero timeout forces the path
    wherever a sibling is late -- bit-identical to the patient run."""
    from epropnp import functional as F
    B = 2
    prob = orc.make_problem(B, N, dof, seed=29, bounds=bounds)
    noise = pack_noise(orc.make_noise(B, S, K, dof, seed=30), dof).to(backend)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    pose_opt, pose_cov, _ = F.lm_solve(hp, p['pose_init'], 3, with_pose_cov=True)
    monkeypatch.setenv('EPROPNP_FWD_SPLIT', '1')
    s1, w1 = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise)
    monkeypatch.setenv('EPROPNP_FWD_SPLIT', str(G))
    assert F.split_scratch(hp, S, K) is not None
    s2, w2, pr2 = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise, with_proposals=True)
    # the first iteration's samples are the same draws from the same initial proposal; later ones come from proposals refitted
    # to weights whose costs were summed in another order (with 16 samples per iteration a refit amplifies that: loose bar)
    s = S // K
    assert torch.equal(s1[:s], s2[:s])
    assert (s1 - s2).abs().max().item() <= 5e-3 * max(1.0, s1.abs().max().item())
    assert (torch.logsumexp(w1, 0) - torch.logsumexp(w2, 0)).abs().max().item() < 2e-2
    samples, logw, props = s2.cpu(), w2.cpu(), pr2.cpu()
    ocam = orc.Cam(prob['cam_mats'].double(), 0.1, None if bounds is None else prob['lb'].double(),
                   None if bounds is None else prob['ub'].double())
    cost = orc.evaluate(prob['x3d'].double(), prob['x2d'].double(), prob['w2d'].double(), samples.double(), ocam,
                        prob['delta'].double(), want_cost=True)[1]
    expect = -cost.float() - _mixture_logq(samples, props, dof, K)
    assert (logw - expect).abs().max().item() <= 2e-4 * max(1.0, expect.abs().max().item())
    if backend.type == 'cuda':
        monkeypatch.setenv('EPROPNP_SPLIT_TIMEOUT_CYCLES', '0')
        s3, w3 = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise)
        torch.cuda.synchronize()
        assert torch.equal(s3, s2) and torch.equal(w3, w2)


@pytest.mark.gpu
def test_split_kernels_beside_a_kernel_that_holds_the_cus():
    """LineMOD training runs the layer beside a backbone on other streams.  With large GEMMs in flight on a second stream the
    parts of a split launch are not all resident at once; they must neither hang nor return other numbers: the dense C3
    forward (split LM + 8-part split forward) and the C3-training forward give the quiet run's bits, every time."""
    import warnings
    import install as emu
    emu.uninstall()
    from epropnp import functional as F
    dev = torch.device('cuda:0')
    hog = torch.randn(8192, 8192, device=dev)
    side = torch.cuda.Stream()
    for B, N, S, K in ((32, 4096, 128, 4), (32, 512, 512, 4)):
        prob = orc.make_problem(B, N, 6, seed=31)
        noise = pack_noise(orc.make_noise(B, S, K, 6, seed=32), 6).to(dev)
        p, cam, cf = make_layer_objects(prob, dev)
        hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, 6)

        def body():
            pose_opt, pose_cov, cost = F.lm_solve(hp, p['pose_init'], 5, with_pose_cov=True, with_cost=True)
            return (pose_opt, cost) + tuple(F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise))
        quiet = [t.clone() for t in body()]
        torch.cuda.synchronize()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)          # "a sibling was recomputed" is the expected performance note
            for rnd in range(4):
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(6):
                        hog @ hog                                    # ~10 ms each: every CU busy
                for _ in range(3):
                    busy = body()
                    for a, b in zip(busy, quiet):
                        assert torch.equal(a, b), (N, rnd)
                torch.cuda.synchronize()
            F.flush_status()


@pytest.mark.gpu
def test_split_kernels_in_a_hipgraph_survive_eager_work_between_replays():
    """The workgroup-split forward and LM solve reset their exchange slots on the stream before every launch.  Captured into a
    hipGraph that reset must be a KERNEL node: on this ROCm stack a captured hipMemsetAsync node writes garbage once eager
    launches have run between two replays (tools/ubench/graph_memset_node.py reproduces it with PyTorch alone) -- stale
    slots would pass for arrived data.  Replays before and after eager work + host syncs equal the eager result bit for bit."""
    import install as emu
    emu.uninstall()
    from epropnp import functional as F
    dev = torch.device('cuda:0')
    for dof, B, N, S, K in ((6, 32, 512, 256, 4), (6, 32, 4096, 64, 2)):
        prob = orc.make_problem(B, N, dof, seed=23)
        noise = pack_noise(orc.make_noise(B, S, K, dof, seed=24), dof).to(dev)
        p, cam, cf = make_layer_objects(prob, dev)

        def body():
            hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)        # (binds the current stream)
            pose_opt, pose_cov, cost = F.lm_solve(hp, p['pose_init'], 4, with_pose_cov=True, with_cost=True)
            smp, logw = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=noise)
            return pose_opt, cost, smp, logw
        ref = [t.clone() for t in body()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = body()
        for rnd in range(3):
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize()
            F.flush_status()
            for a, b in zip(out, ref):
                assert torch.equal(a, b), (N, rnd)
            junk = torch.rand(1 << 20, device=dev)                # eager launches, a reduction with a memset, host syncs
            assert 0.0 < float(junk.sum()) < float(junk.numel())
            torch.cuda.synchronize()


def test_von_mises_draws_match_oracle_over_kappa_range(backend):
    """The device's bounded Best-Fisher sampler (fp32, cancellation-free form, fp64 only for borderline decisions) makes
    the same accept/reject decisions and returns the same angles as the oracle's fp64 procedure, from nearly uniform
    (kappa ~ 1e-5) to extremely peaked (kappa ~ 3e4) proposals.  One AMIS iteration: proposal 0 has
    kappa = 0.33 / cov[3,3]."""
    from epropnp import functional as F
    dof, S, K, N = 4, 512, 1, 32
    var = torch.tensor([1e-5, 3e-4, 0.01, 0.1, 1.0, 33.0, 3e3, 3e4])          # kappa 3.3e4 ... 1.1e-5
    B = var.numel()
    prob = orc.make_problem(B, N, dof, seed=31)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    cov = torch.diag_embed(torch.stack((torch.full((B,), 1e-2),) * 3 + (var,), -1))
    pose_opt = prob['pose_gt'].clone()
    noise = orc.make_noise(B, S, K, dof, seed=77)
    samples, _ = F.amis_forward(hp, pose_opt.to(backend), cov.to(backend), S, K, noise=pack_noise(noise, dof).to(backend))
    n_u = round(0.25 * S)
    yaw = samples[n_u:, :, 3].cpu()                                            # (n_v, B)
    kappa = (0.33 / var.clamp(min=1e-5)).reshape(1, B).expand(S - n_u, B)
    loc = pose_opt[:, 3].reshape(1, B).expand(S - n_u, B)
    ref = orc.vm_sample_bounded(loc, kappa, noise['vm'][0, :, :, 0])           # u: (n_v, B, T, 3)
    d = (yaw - ref).abs()
    d = torch.minimum(d, 2 * math.pi - d)                                      # wrap-around at +-pi
    assert d.max().item() < 2e-5, d.max(0).values
    uni = samples[:n_u, :, 3].cpu()
    torch.testing.assert_close(uni, ((noise['u'][0, :, :, 0] * 2 - 1) * math.pi).float(), rtol=0, atol=2e-6)


@pytest.mark.parametrize('dof', [6, 4])
def test_non_spd_pose_cov_falls_back_like_cholesky_wrapper(backend, dof):
    """cholesky_wrapper (epropnp/epropnp.py:16-33): a covariance block whose Cholesky fails (indefinite, NaN) is replaced
    by diag(default) -- I for 6-DoF, [1,1,4] for the 4-DoF translation proposal (:216-217), I for the ACG scatter matrix
    (:301-302).  Feed such pose_cov straight into the AMIS kernel: proposal 0 and the samples drawn from it must equal
    the oracle's (orc.chol_or_default inside initial_fit_*), object by object; healthy objects are untouched."""
    from epropnp import functional as F
    B, N, S, K = 5, 48, 32, 1
    prob = orc.make_problem(B, N, dof, seed=91)
    noise = orc.make_noise(B, S, K, dof, seed=92)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    pose_opt, cov, _ = F.lm_solve(hp, p['pose_init'], 3, with_pose_cov=True)
    pose_opt, cov = pose_opt.cpu(), cov.cpu().clone()
    cov[0, 0, 0] = -cov[0, 0, 0]                                  # indefinite translation block
    cov[1, :3, :3] = float('nan')                                 # NaN translation block
    if dof == 6:
        cov[2, 4, 4] = -1e-3                                      # indefinite rotation block -> T C^-1 T^T + I not SPD
        cov[3, 3:, 3:] = float('nan')                             # NaN rotation block
    else:
        cov[2] = float('nan')                                     # everything NaN (kappa = 0.33 / max(NaN, eps))
    samples, logw, props = F.amis_forward(hp, pose_opt.to(backend), cov.to(backend), S, K,
                                          noise=pack_noise(noise, dof).to(backend), with_proposals=True)
    samples, props = samples.cpu(), props.cpu()[:, 0]

    def tril(v, n):
        L = torch.zeros(v.shape[:-1] + (n, n))
        idx = torch.tril_indices(n, n)
        L[..., idx[0], idx[1]] = v
        return L
    if dof == 6:
        # fp64 oracle: the kernel fits in fp64, the reference's fp32 fit of a healthy object drifts by ~5e-4 (cond ~1e5)
        mode, Lt, Lr = (v.float() for v in orc.initial_fit_6dof(pose_opt.double(), cov.double()))
        torch.testing.assert_close(tril(props[:, 16:26], 4), Lr, rtol=2e-4, atol=1e-6)
        assert torch.equal(Lr[2], torch.eye(4)) and torch.equal(Lr[3], torch.eye(4))
        t = orc.student_t_sample(mode, Lt, noise['z'][0], noise['chi2'][0])
        want = torch.cat((t, orc.acg_sample(Lr, noise['g'][0])), -1)
        ok = [0, 1, 2, 3, 4]
    else:
        mode, Lt, rmode, kappa = orc.initial_fit_4dof(pose_opt, cov)
        assert torch.equal(Lt[0], torch.diag(torch.tensor([1.0, 1.0, 4.0])))      # the default IS the factor (scale_tril)
        t = orc.student_t_sample(mode, Lt, noise['z'][0], noise['chi2'][0])
        want = torch.cat((t, torch.zeros(S, B, 1)), -1)
        ok = [0, 1, 3, 4]                                         # object 2: kappa from a NaN variance (no reference value)
    torch.testing.assert_close(tril(props[:, 3:9], 3), Lt, rtol=2e-4, atol=1e-6)
    assert torch.equal(tril(props[:, 3:9], 3)[0], Lt[0]) and torch.equal(tril(props[:, 3:9], 3)[1], Lt[1])
    nt = 7 if dof == 6 else 3
    torch.testing.assert_close(samples[:, ok, :nt], want[:, ok, :nt], rtol=1e-4, atol=2e-5)
    assert torch.isfinite(logw.cpu()[:, ok]).all() and torch.isfinite(samples[:, ok]).all()


def test_von_mises_draws_match_numpy_stream(backend):
    """The device sampler against numpy.random.vonmises itself (fixture `vm_numpy`: the reference's unmodified
    VonMisesUniformMix.sample under a seeded numpy generator + the uniform stream it consumed): one AMIS iteration whose
    proposal 0 has loc = pose_opt yaw and kappa = 0.33 / cov[3,3] (epropnp.py:218), fed the same uniforms."""
    from epropnp import functional as F
    g = load_golden('vm_numpy')
    x, var = g['x'], g['var']
    s, B = x.shape[0], x.shape[1]
    n_u = g['u_uniform'].shape[0]
    prob = orc.make_problem(B, 32, 4, seed=31)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, 4)
    cov = torch.diag_embed(torch.stack((torch.full((B,), 1e-2),) * 3 + (var,), -1))
    pose_opt = prob['pose_gt'].clone()
    pose_opt[:, 3] = g['loc'][:, 0]
    noise = orc.make_noise(B, s, 1, 4, seed=5)
    noise['u'] = g['u_uniform'].float().unsqueeze(0)                     # (1,n_u,B,1)
    noise['vm'] = g['u_vm'].float().unsqueeze(0)                         # (1,n_v,B,1,T,3)
    samples, _ = F.amis_forward(hp, pose_opt.to(backend), cov.to(backend), s, 1, noise=pack_noise(noise, 4).to(backend))
    d = (samples[:, :, 3].cpu() - x[:, :, 0]).abs()
    d = torch.minimum(d, 2 * math.pi - d)
    # fp32 uniforms (the fixture's doubles rounded) move a draw by ~1e-7 * d(angle)/du; 2e-5 rad as in the oracle test above
    assert d[n_u:].max().item() < 2e-5, d[n_u:].max(0).values
    assert d[:n_u].max().item() < 2e-6


@pytest.mark.parametrize('impl', ['mfma', 'valu'])
def test_backward_drop_threshold_is_mass_bounded(backend, monkeypatch, impl):
    """The backward skips the low-weight tail whose TOTAL |weight| is below EPROPNP_BWD_DROP (default 2^-24) of the
    object's total (csrc/amis_common.h: mass_drop_threshold).  Heavy-tailed softmax weights (most samples negligible):
    the default differs from the exact sum (EPROPNP_BWD_DROP=0) by rounding only, a coarse 1e-3 budget by at most ~1e-3."""
    from epropnp import functional as F
    set_tune(monkeypatch, bwd_impl=impl)
    B, N, S, dof = 3, 100, 256, 6
    prob = orc.make_problem(B, N, dof, seed=23)
    g = torch.Generator().manual_seed(9)
    poses = prob['pose_gt'].unsqueeze(0).repeat(S, 1, 1)
    poses[..., :3] += 0.2 * torch.randn(S, B, 3, generator=g)
    q = poses[..., 3:] + 0.1 * torch.randn(S, B, 4, generator=g)
    poses[..., 3:] = q / q.norm(dim=-1, keepdim=True)
    w = torch.softmax(12.0 * torch.randn(S, B, generator=g), dim=0)          # log-weights spread over ~+-30 nats
    assert float((w < 1e-9 * w.max(0).values).float().mean()) > 0.3          # a third of the samples are negligible
    g_init = torch.randn(B, generator=g)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    outs = {}
    for eps in ('0', None, '1e-3'):
        if eps is None:
            monkeypatch.delenv('EPROPNP_BWD_DROP', raising=False)
        else:
            monkeypatch.setenv('EPROPNP_BWD_DROP', eps)
        outs[eps] = [t.cpu() for t in F.amis_backward(hp, poses.to(backend), (-w).to(backend), p['pose_init'], g_init.to(backend))]
    for a, e in zip(outs[None], outs['0']):
        assert _rel(a, e) <= 1e-6
    coarse = max(_rel(a, e) for a, e in zip(outs['1e-3'], outs['0']))
    assert 0 < coarse <= 3e-3, coarse
    # exact mode against autograd of the oracle
    x3d, x2d, w2d, delta = (prob[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d', 'delta'))
    ocam = orc.Cam(prob['cam_mats'], 0.1)
    c_s = orc.evaluate(x3d, x2d, w2d, poses, ocam, delta, want_cost=True)[1]
    c_i = orc.evaluate(x3d, x2d, w2d, prob['pose_init'], ocam, delta, want_cost=True)[1]
    ((c_s * w).sum() + (c_i * g_init).sum()).backward()
    for mine, ref in zip(outs['0'], (x3d.grad, x2d.grad, w2d.grad, delta.grad)):
        assert _rel(mine, ref) <= 2e-4
