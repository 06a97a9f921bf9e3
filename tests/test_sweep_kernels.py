"""Sweep kernels vs the golden fixtures produced by the unmodified reference (oracle/make_golden.py)."""
import pytest
import torch

from helpers import load_golden, make_layer_objects


@pytest.mark.parametrize('name', ['eval6', 'eval6_clip', 'eval4_clip'])
def test_normal_equations_match_reference(backend, name):
    from epropnp import functional as F
    g = load_golden(name)
    p, cam, cf = make_layer_objects(g['prob'], backend)
    prob = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, int(g['dof']))
    jtj, jtr, cost = F.normal_equations(prob, g['pose'].to(backend), clip_jac=True)
    scale = g['jtj'].abs().amax(dim=(-1, -2), keepdim=True)
    assert ((jtj.cpu() - g['jtj']).abs() / scale).max() < 2e-5          # fp32 sums of 2N terms, different order
    assert ((jtr.cpu() - g['jtr']).abs() / g['jtr'].abs().amax(-1, keepdim=True).clamp(min=1e-3)).max() < 1e-4
    torch.testing.assert_close(cost.cpu(), g['cost'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name', ['eval6', 'eval6_clip', 'eval4_clip'])
def test_evaluate_cost_matches_reference(backend, name):
    from epropnp import functional as F
    g = load_golden(name)
    p, cam, cf = make_layer_objects(g['prob'], backend)
    prob = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, int(g['dof']))
    costs = F.evaluate_cost(prob, g['poses'].to(backend))
    torch.testing.assert_close(costs.cpu(), g['costs'], rtol=2e-5, atol=1e-5)
    one = F.evaluate_cost(prob, g['pose'].to(backend))
    torch.testing.assert_close(one.cpu(), g['cost'], rtol=2e-5, atol=1e-6)
