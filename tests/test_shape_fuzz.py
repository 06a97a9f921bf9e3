"""Seeded sweep over problem shapes and optional arguments: every launch-shape heuristic, template instantiation and
NULL-pointer combination the hot path can take, checked with the comparisons that do not depend on the (chaotic) proposal
refit -- LM pose against the oracle's LM, log-weights against the oracle's cost / densities at the kernel's OWN samples and
fitted proposals, backward against autograd of the oracle at fixed samples.  (A NULL `pose_init` in the backward and a
hazard behind the bf16 MFMA were both shape / argument combinations no fixed-size test had reached.)"""
import os
import random

import pytest
import torch

import epropnp_oracle as orc
from helpers import make_layer_objects, pack_noise
from test_amis import GRAD_TOL, _mixture_logq, _rel


def _cases(n, seed, n_max, s_max):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        dof = rng.choice((6, 6, 4))
        K = rng.choice((1, 2, 3, 4))
        s = rng.choice((8, 16, 20, 32, 48, 64, 128)) if s_max >= 512 else rng.choice((8, 16, 20, 32))
        N = rng.choice((5, 16, 17, 33, 64, 100, 128, 129, 200, 256, 300, 511, 512, 513, 700, 1024, 1500, 2049, 2500))
        while N > n_max:
            N //= 2
        B = rng.choice((1, 2, 3, 5, 7))
        bounds = rng.choice((None, None, 'tensor', 'tight'))
        with_init = rng.random() < 0.6
        z_min = rng.choice((0.1, 0.1, 0.01, 2.0))
        out.append(pytest.param(dof, B, N, s * K, K, bounds, with_init, z_min, 1000 * seed + i,
                                id=f'{dof}dof-B{B}-N{N}-S{s * K}-K{K}-{bounds}-init{int(with_init)}-z{z_min}'))
    return out


def _check(backend, dof, B, N, S, K, bounds, with_init, z_min, seed):
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import HuberPnPCost
    prob = orc.make_problem(B, N, dof, seed=seed, bounds=bounds)
    p, _, _ = make_layer_objects(prob, backend)
    cam = PerspectiveCamera(cam_mats=p['cam_mats'], z_min=z_min, lb=p.get('lb'), ub=p.get('ub'))
    cf = HuberPnPCost(delta=p['delta'])
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    ocam = orc.Cam(prob['cam_mats'].double(), z_min, *(prob[k].double() if k in prob else None for k in ('lb', 'ub')))
    d64 = {k: prob[k].double() for k in ('x3d', 'x2d', 'w2d', 'delta', 'pose_init')}

    # ---- LM: pose and covariance against the oracle's solver (fp64) ----
    pose_opt, pose_cov, cost = F.lm_solve(hp, p['pose_init'], 10, with_pose_cov=True, with_cost=True)
    # The oracle in fp64 AND in fp32 (the reference's own precision): an accept / reject decision of the trust region that
    # sits on a knife-edge falls one way or the other with the rounding (the parity tests handle that with the
    # rounding-spread yardstick); per object the kernel has to agree with one of the two.  10 iterations, so that what is
    # compared is the minimum reached.  A clipped (tight bounds, binding depth clamp) or barely determined problem keeps
    # a flat direction along which solutions drift while reaching the same cost.
    well_posed = N >= 64 and bounds != 'tight' and z_min <= 0.1
    ocam32 = orc.Cam(prob['cam_mats'], z_min, prob.get('lb'), prob.get('ub'))
    refs = [orc.lm_solve(d64['x3d'], d64['x2d'], d64['w2d'], ocam, d64['delta'], d64['pose_init'], num_iter=10, with_cost=True),
            orc.lm_solve(prob['x3d'], prob['x2d'], prob['w2d'], ocam32, prob['delta'], prob['pose_init'], num_iter=10,
                         with_cost=True)]
    # pose error relative to the object's distance (4-DoF problems sit at t_z ~ 10, where the depth is the flat direction)
    perr = torch.stack([(pose_opt.cpu().double() - r[0].double()).abs().amax(-1) / (r[0].double().abs().amax(-1) / 4).clamp(min=1.0)
                        for r in refs]).amin(0)
    cerr = torch.stack([(cost.cpu().double() - r[2].double()).abs() / r[2].double().abs().clamp(min=1.0) for r in refs]).amin(0)
    # ... and an object where all three arithmetics part ways at a knife-edge still has to reach the same cost
    ok = (perr <= (2e-4 if well_posed else 2e-2)) | (cerr <= 1e-5)
    assert bool(ok.all()), (perr, cerr)
    assert cerr.max().item() <= 2e-3, cerr
    assert bool(torch.isfinite(pose_opt).all()) and bool(torch.isfinite(pose_cov).all())

    # ---- AMIS forward: log-weights against cost + mixture density at the kernel's own samples / proposals ----
    noise = orc.make_noise(B, S, K, dof, seed=seed + 1)
    samples, logw, props = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=pack_noise(noise, dof).to(backend),
                                          with_proposals=True)
    again = F.amis_forward(hp, pose_opt, pose_cov, S, K, noise=pack_noise(noise, dof).to(backend), with_proposals=True)
    for a, b in zip((samples, logw, props), again):      # fixed reduction order, no atomics: bit-reproducible (a race would not be)
        assert torch.equal(a, b)
    samples, logw, props = samples.cpu(), logw.cpu(), props.cpu()
    leaves = {k: d64[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d', 'delta')}
    c_s = orc.evaluate(leaves['x3d'], leaves['x2d'], leaves['w2d'], samples.double(), ocam, leaves['delta'], want_cost=True)[1]
    expect = -c_s.detach().float() - _mixture_logq(samples, props, dof, K)
    fin = torch.isfinite(expect)
    assert bool((torch.isfinite(logw) == fin).all())
    assert (logw[fin] - expect[fin]).abs().max().item() <= 2e-4 * max(1.0, expect[fin].abs().max().item())

    # ---- backward at those samples, softmax weights as the loss produces them, with / without the cost of pose_init ----
    g_logw = torch.softmax(logw, 0) / B
    g_init = torch.full((B,), 1.0 / B)
    obj = ((-c_s) * g_logw.double()).sum()
    if with_init:
        c_i = orc.evaluate(leaves['x3d'], leaves['x2d'], leaves['w2d'], d64['pose_init'], ocam, leaves['delta'], want_cost=True)[1]
        obj = obj + (c_i * g_init.double()).sum()
    obj.backward()
    grads = F.amis_backward(hp, samples.to(backend), g_logw.to(backend), p['pose_init'] if with_init else None,
                            g_init.to(backend) if with_init else None)
    again = F.amis_backward(hp, samples.to(backend), g_logw.to(backend), p['pose_init'] if with_init else None,
                            g_init.to(backend) if with_init else None)
    for name, mine, rerun in zip(('x3d', 'x2d', 'w2d', 'delta'), grads, again):
        assert torch.equal(mine, rerun), name
        want = leaves[name].grad
        assert bool(torch.isfinite(mine).all()), name
        if name == 'delta':      # sums of max(rho - delta, 0): a few point-poses near the threshold dominate; absolute floor
            err = (mine.cpu().double() - want).abs().max().item()
            assert err <= GRAD_TOL * want.abs().max().item() + 1e-6 * N, (name, err)
        else:
            # per point, relative to the tensor's largest entry.  With a projection clamp or a binding depth clamp a
            # point-pose whose projection sits within rounding of the bound passes its gradient in fp32 and not in fp64
            # (or the reverse): that moves ONE point's gradient by that pair's share -- allow two such points
            err = (mine.cpu().double() - want).abs().flatten(2).amax(-1) / want.abs().max().clamp(min=1e-12)
            flips = 2 if (bounds is not None or z_min > 0.1) else 0
            bad = err > GRAD_TOL
            assert int(bad.sum()) <= flips and err.max().item() <= (5e-2 if flips else GRAD_TOL), (name, err.max().item(), int(bad.sum()))


@pytest.mark.parametrize('dof,B,N,S,K,bounds,with_init,z_min,seed', _cases(10, 1, 320, 128))
def test_shape_sweep_small(backend, poisoned_empty, dof, B, N, S, K, bounds, with_init, z_min, seed):
    _check(backend, dof, B, N, S, K, bounds, with_init, z_min, seed)


@pytest.mark.gpu
@pytest.mark.parametrize('dof,B,N,S,K,bounds,with_init,z_min,seed', _cases(int(os.environ.get('EPROPNP_FUZZ_CASES', '80')), int(os.environ.get('EPROPNP_FUZZ_SEED', '2')), 4096, 512))
def test_shape_sweep_gpu(poisoned_empty, dof, B, N, S, K, bounds, with_init, z_min, seed):
    import install as emu
    emu.uninstall()
    _check(torch.device('cuda:0'), dof, B, N, S, K, bounds, with_init, z_min, seed)


def test_many_samples_small(backend, poisoned_empty):
    """More samples per iteration than the workgroup has lanes (several samples per lane in the draw, base noise no longer
    aliased onto the pose table)."""
    _check(backend, 6, 1, 40, 640, 1, None, True, 0.1, 4242)


def test_sampler_state_beyond_lds(backend, poisoned_empty):
    """mc_samples whose per-sample state no longer fits the 160 KiB of LDS: the forward keeps it in a global scratch buffer
    (the reference has no limit on mc_samples); the backward takes the all-VALU kernel beyond ~2700 poses."""
    _check(backend, 6, 1, 24, 4800, 8, None, True, 0.1, 777)


@pytest.mark.parametrize('dof,S,K', [(6, 8192, 4), (4, 3700, 1)])
def test_samples_per_iteration_beyond_lds(backend, poisoned_empty, dof, S, K):
    """More samples PER ITERATION than one LDS pose table holds (48 B per sample: ~3000 beside the point chunks; the reference
    has no limit, epropnp.py:55-59): an iteration's samples go through the table in tiles -- draw, sweep, costs to the global
    scratch, per tile.  EProPnP6DoF(mc_samples=8192, num_iter=4) used to be EINVAL."""
    _check(backend, dof, 1, 24, S, K, None, True, 0.1, 99 + S)
    # production mode (on-device Philox instead of injected draws: the base noise is then drawn inline, there is no room for
    # the ahead-of-time buffer): log-weights against the oracle's cost + mixture density at the kernel's own samples
    from epropnp import functional as F
    prob = orc.make_problem(1, 24, dof, seed=5)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    pose_opt, pose_cov, _ = F.lm_solve(hp, p['pose_init'], 5, with_pose_cov=True)
    samples, logw, props = (t.cpu() for t in F.amis_forward(hp, pose_opt, pose_cov, S, K, seed=3, with_proposals=True))
    assert bool(torch.isfinite(samples).all())
    cost = orc.evaluate(prob['x3d'].double(), prob['x2d'].double(), prob['w2d'].double(), samples.double(),
                        orc.Cam(prob['cam_mats'].double(), 0.1), prob['delta'].double(), want_cost=True)[1]
    expect = -cost.float() - _mixture_logq(samples, props, dof, K)
    assert (logw - expect).abs().max().item() <= 2e-4 * max(1.0, expect.abs().max().item())
    assert len(torch.unique(samples[:, 0, 0])) > S // 2                  # fresh draws for every sample of every tile


@pytest.mark.gpu
@pytest.mark.parametrize('dof,B,N,S,K', [(6, 2, 64, 2048, 4), (4, 2, 100, 2400, 3), (6, 1, 40, 1024, 1), (6, 2, 33, 2400, 4),
                                         (4, 3, 512, 1536, 2), (6, 3, 700, 2048, 4), (6, 2, 50, 3200, 16),
                                         (6, 2, 64, 4096, 4), (6, 3, 300, 6000, 4), (4, 2, 100, 4000, 4), (6, 1, 2100, 3000, 2),
                                         (6, 2, 1100, 16384, 4), (4, 2, 200, 12000, 2), (6, 1, 40, 9000, 1)])
def test_many_samples_gpu(poisoned_empty, dof, B, N, S, K):
    """... up to and beyond the LDS limit of the sampler state (6-DoF: 40 S + 96 S / K bytes + 2 KiB <= 160 KiB; past it
    the state moves to a global scratch buffer)."""
    import install as emu
    emu.uninstall()
    _check(torch.device('cuda:0'), dof, B, N, S, K, None, True, 0.1, 4242 + S)
