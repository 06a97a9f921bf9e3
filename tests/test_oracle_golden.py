"""The oracle (CPU restatement) against the fixtures produced by the UNMODIFIED reference (oracle/make_golden.py).
Runs without a GPU and without /root/reference: this is what keeps the checker itself pinned on the GPU box."""
import pytest
import torch

import epropnp_oracle as orc
from helpers import load_golden


def cam_of(p):
    return orc.Cam(p['cam_mats'], 0.1, p.get('lb'), p.get('ub'))


@pytest.mark.parametrize('name', ['eval6', 'eval6_clip', 'eval4_clip'])
def test_evaluate(name):
    g = load_golden(name)
    p = g['prob']
    res, cost, jac = orc.evaluate(p['x3d'], p['x2d'], p['w2d'], g['pose'], cam_of(p), p['delta'], True, True)
    torch.testing.assert_close(res, g['res'], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(cost, g['cost'], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(jac, g['jac'], rtol=1e-5, atol=1e-5)
    costs = orc.evaluate(p['x3d'], p['x2d'], p['w2d'], g['poses'], cam_of(p), p['delta'], want_cost=True)[1]
    torch.testing.assert_close(costs, g['costs'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name', ['lm6_tr', 'lm6_gn', 'lm6_tr_clip', 'lm4_tr', 'lm4_gn'])
def test_lm(name):
    g = load_golden(name)
    p = g['prob']
    pose, cov, cost, hist = orc.lm_solve(p['x3d'], p['x2d'], p['w2d'], cam_of(p), p['delta'], p['pose_init'],
                                         fast_mode=bool(g['fast_mode']), with_pose_cov=True, with_cost=True,
                                         num_iter=int(g['lm_iter']))
    torch.testing.assert_close(pose, g['pose_opt'], rtol=0, atol=2e-5)
    torch.testing.assert_close(cost, g['cost'], rtol=1e-5, atol=1e-6)
    if hist:
        assert torch.equal(torch.stack(hist).int(), g['accepts'].int())


@pytest.mark.parametrize('name', ['mc6', 'mc4', 'mc4_norm', 'mc6_demo', 'mc4_rslm', 'mc6_tight', 'mc6_k1', 'mc4_det'])
def test_monte_carlo(name):
    g = load_golden(name)
    dof, S, K = int(g['dof']), int(g['S']), int(g['K'])
    rs = g['rslm_cfg'].tolist()
    rn = g.get('rslm')
    o = orc.run_mc(g['prob'], g['noise'], dof, S, K, int(g['lm_iter']), normalize=bool(g['normalize']),
                   rslm_kw=dict(num_iter=rs[2]) if rn else None, rslm_noise=rn,
                   with_pose_opt_plus=bool(g['with_pose_opt_plus']))
    r = g['ref']
    torch.testing.assert_close(o['pose_opt'], r['pose_opt'], rtol=0, atol=2e-5)
    torch.testing.assert_close(o['cost_init'], r['cost_init'], rtol=1e-5, atol=1e-6)
    assert (o['loss_obj'] - r['loss_obj']).abs().max() < 1e-3
    if 'pose_opt_plus' in r:
        torch.testing.assert_close(o['pose_opt_plus'], r['pose_opt_plus'], rtol=0, atol=5e-5)


def test_von_mises_bounded_sampler_distribution():
    """The injected-uniform Best-Fisher sampler that stands in for numpy.random.vonmises (distributions.py:70-72)."""
    from scipy import stats
    g = torch.Generator().manual_seed(0)
    for kappa, loc in ((0.5, 0.3), (4.0, -1.0), (60.0, 2.5)):
        n = 20000
        u = torch.rand(n, orc.VM_MAX_TRIES, 3, generator=g, dtype=torch.float64)
        x = orc.vm_sample_bounded(torch.full((n,), loc), torch.full((n,), kappa), u).double().numpy()
        d = stats.kstest((x - loc + 3.141592653589793) % (2 * 3.141592653589793) - 3.141592653589793,
                         stats.vonmises(kappa).cdf)
        assert d.pvalue > 1e-3, (kappa, d)


def test_von_mises_sampler_matches_numpy_stream():
    """Fixture `vm_numpy`: the UNMODIFIED VonMisesUniformMix.sample (numpy.random.uniform / numpy.random.vonmises,
    epropnp/distributions.py:61-72) under a seeded global generator, together with the very doubles numpy consumed, dealt
    into the bounded sampler's (attempt, 3) layout (oracle/make_golden.py:case_vm_numpy).  The bounded Best-Fisher sampler
    fed that stream must return the reference's angles: kappa from 1e-4 to 3.3e4."""
    import math
    g = load_golden('vm_numpy')
    s = g['x'].shape[0]
    got = orc.vm_mix_sample(g['loc'].double(), g['kappa'].double(), g['u_uniform'], g['u_vm'], s)
    d = (got - g['x'].double()).abs()
    d = torch.minimum(d, 2 * math.pi - d)
    assert d.max().item() <= 1e-6, d.amax((0, 2))
