"""Device-side Philox call counter (EProPnPBase.enable_graph_safe_rng): same draws as the host-side counter, fresh
draws per call, and -- on the GPU -- a whole training step captured into a hipGraph and replayed."""
import pytest
import torch

import epropnp_oracle as orc
from helpers import make_layer_objects


def _layer(dof, rslm):
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver, RSLMSolver
    init = RSLMSolver(dof=dof, num_points=8, num_proposals=6, num_iter=2) if rslm else None
    cls = EProPnP6DoF if dof == 6 else EProPnP4DoF
    return cls(mc_samples=64, num_iter=4, solver=LMSolver(dof=dof, num_iter=3, init_solver=init), seed=1234)


@pytest.mark.parametrize('dof,rslm', [(6, False), (4, True)])
def test_device_counter_matches_host_counter(backend, dof, rslm):
    B, N = 4, 48
    p = orc.make_problem(B, N, dof, seed=2)
    d, cam, cf = make_layer_objects(p, backend, relative_delta=0.5)
    cf.set_param(d['x2d'], d['w2d'])
    host, devc = _layer(dof, rslm), _layer(dof, rslm)
    if rslm:
        host.solver.init_solver._draw_seed, host.solver.init_solver._draw_calls = 777, 0
        devc.solver.init_solver._draw_seed, devc.solver.init_solver._draw_calls = 777, 0
    devc.enable_graph_safe_rng(backend)
    kw = dict(pose_init=d['pose_init'], force_init_solve=rslm)
    outs = []
    for layer in (host, devc):
        outs.append([layer.monte_carlo_forward(d['x3d'], d['x2d'], d['w2d'], cam, cf, **kw) for _ in range(2)])
    for call in range(2):
        torch.testing.assert_close(outs[0][call][3], outs[1][call][3], rtol=0, atol=0)      # identical samples
        torch.testing.assert_close(outs[0][call][4], outs[1][call][4], rtol=0, atol=0)
    assert (outs[1][0][3] - outs[1][1][3]).abs().max() > 0                                   # fresh draws per call
    assert int(devc.rng_counter.item()) == 2
    if rslm:
        assert int(devc.solver.init_solver.rng_counter.item()) == 2


def _graph_replay_body():
    from epropnp.losses import monte_carlo_pose_loss
    dev = torch.device('cuda:0')
    dof, B, N = 4, 64, 96
    p = orc.make_problem(B, N, dof, seed=8)
    d, cam, cf = make_layer_objects(p, dev, relative_delta=0.5)
    layer = _layer(dof, True).enable_graph_safe_rng(dev)
    leaves = [d[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d')]
    out = {}

    def step():
        cf.set_param(leaves[1].detach(), leaves[2])
        o = layer.monte_carlo_forward(*leaves, cam, cf, pose_init=d['pose_init'], force_init_solve=True)
        monte_carlo_pose_loss(o[4], o[5]).mean().backward()
        out['samples'], out['logw'] = o[3], o[4].detach()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            for t in leaves:
                t.grad = None
            step()
    torch.cuda.current_stream().wait_stream(side)
    for t in leaves:
        t.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    seen = []
    for _ in range(3):
        before = int(layer.rng_counter.item())
        graph.replay()
        torch.cuda.synchronize()
        assert int(layer.rng_counter.item()) == before + 1
        seen.append((out['samples'].clone(), leaves[0].grad.clone()))
        assert bool(torch.isfinite(leaves[0].grad).all()) and bool(torch.isfinite(out['logw']).all())
    assert (seen[0][0] - seen[1][0]).abs().max() > 0 and (seen[1][0] - seen[2][0]).abs().max() > 0
    # a replay is the same computation as an eager call at the same counter value
    init = layer.solver.init_solver
    layer.rng_counter.fill_(100)
    init.rng_counter.fill_(100)
    graph.replay()
    torch.cuda.synchronize()
    s_graph, g_graph = out['samples'].clone(), leaves[0].grad.clone()
    layer.rng_counter.fill_(100)
    init.rng_counter.fill_(100)
    for t in leaves:
        t.grad = None
    step()
    torch.cuda.synchronize()
    torch.testing.assert_close(out['samples'], s_graph, rtol=0, atol=0)
    torch.testing.assert_close(leaves[0].grad, g_graph, rtol=0, atol=0)
    del graph                      # release the graph and its private pool before other tests allocate
    out.clear()
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_training_step_replays_from_a_hip_graph():
    """Runs in its own interpreter: graph capture switches the allocator into a private pool, and an isolated process
    keeps whatever torch does at graph / pool teardown away from the other GPU tests."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ('import sys; sys.path[:0] = [%r, %r, %r]; import test_graph_rng as t; t._graph_replay_body(); '
            'print("GRAPH-REPLAY-OK", flush=True)') % (here, os.path.join(os.path.dirname(here), 'oracle'),
                                                        os.path.join(os.path.dirname(here), 'epro-pnp_amd'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert 'GRAPH-REPLAY-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_failed_call_leaves_the_advance_ticket_clean(backend, monkeypatch):
    """The sampler's launch advances the device-side Philox counters itself, behind a ticket its workgroups count up (AmisParams.advance).
    A call that fails part-way must not leave that ticket half-counted: the two calls after an injected failure draw fresh, different
    samples and the counter has moved by exactly two."""
    from epropnp import functional as F
    from epropnp.epropnp import EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    p = orc.make_problem(5, 48, dof=6, seed=3)
    d, cam, cf = make_layer_objects(p, backend, relative_delta=0.5)
    cf.set_param(d['x2d'], d['w2d'])
    layer = EProPnP6DoF(mc_samples=32, num_iter=4, solver=LMSolver(dof=6, num_iter=3), seed=9).enable_graph_safe_rng(backend)
    call = lambda: layer.monte_carlo_forward(d['x3d'], d['x2d'], d['w2d'], cam, cf, pose_init=d['pose_init'], force_init_solve=False)
    first = call()[3].clone()
    assert int(layer._rng_pair[0]) == 1 and int(layer._rng_pair[2]) == 0
    real = F.fused_monte_carlo

    def failing(*a, **k):
        layer._rng_pair[2] = 3                      # as if three workgroups had counted and the launch then died
        raise RuntimeError('injected launch failure')
    monkeypatch.setattr(F, 'fused_monte_carlo', failing)
    with pytest.raises(RuntimeError, match='injected'):
        call()
    monkeypatch.setattr(F, 'fused_monte_carlo', real)
    assert int(layer._rng_pair[2]) == 0, 'the ticket was left part-way'
    second, third = call()[3].clone(), call()[3].clone()
    assert int(layer._rng_pair[0]) == 3 and int(layer._rng_pair[2]) == 0
    assert not torch.equal(first, second) and not torch.equal(second, third)
