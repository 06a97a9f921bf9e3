"""epropnp.graphed.GraphedLoss: a step segment (set_param -> monte_carlo_forward -> loss -> backward) replayed from a
hipGraph as a differentiable op == the same segment run eagerly at the same Philox counter values."""
import pytest
import torch

import epropnp_oracle as orc
from helpers import make_layer_objects


def _layer(dof=6):
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    cls = EProPnP6DoF if dof == 6 else EProPnP4DoF
    return cls(mc_samples=64, num_iter=4, solver=LMSolver(dof=dof, num_iter=3), seed=4321)


def _segment(layer, cam, cf):
    from epropnp.losses import monte_carlo_pose_loss

    def fn(x3d, x2d, w2d, pose_gt):
        cf.set_param(x2d.detach(), w2d)
        o = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=pose_gt, force_init_solve=False)
        return monte_carlo_pose_loss(o[4], o[5]).mean(), o[0], o[3]
    return fn


def test_runs_eagerly_off_device(backend):
    """No HIP tensors (here: the CPU emulation backend) -> the callable runs as is, autograd included."""
    from epropnp.graphed import GraphedLoss
    if backend.type != 'cpu':
        pytest.skip('covers the fallback')
    p = orc.make_problem(3, 32, 6, seed=1)
    d, cam, cf = make_layer_objects(p, backend, relative_delta=0.5)
    leaves = [d[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d')]
    g = GraphedLoss(_segment(_layer(), cam, cf), (*leaves, d['pose_init']), layers=[])
    assert not g.enabled
    loss, pose_opt, _ = g(*leaves, d['pose_init'])
    loss.backward()
    assert pose_opt.shape == (3, 7) and all(bool(torch.isfinite(t.grad).all()) for t in leaves)


def _graphed_body(B=16, N=64):
    from epropnp.graphed import GraphedLoss
    dev = torch.device('cuda:0')
    p = orc.make_problem(B, N, 6, seed=3)
    d, cam, cf = make_layer_objects(p, dev, relative_delta=0.5)
    _, cam_e, cf_e = make_layer_objects(p, dev, relative_delta=0.5)
    lg, le = _layer(), _layer()
    ins = [d[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d')]
    graphed = GraphedLoss(_segment(lg, cam, cf), (*ins, d['pose_init']), layers=[lg], warmup=3)
    assert graphed.enabled and int(lg.rng_counter.item()) == 3
    eager = _segment(le, cam_e, cf_e)
    for _ in range(3):                               # same number of draws as the warm-up runs
        eager(*[t.detach() for t in ins], d['pose_init'])
    scale = torch.tensor(0.37, device=dev)
    for call in range(3):
        # new data every call: the graph must read what is passed in, not what it was captured with
        new = [(t.detach() * (1 + 0.01 * call)).requires_grad_(True) for t in ins]
        ref = [t.detach().clone().requires_grad_(True) for t in new]
        loss, pose_opt, samples = graphed(*new, d['pose_init'])
        (loss * scale).backward()
        loss_e, pose_e, samples_e = eager(*ref, d['pose_init'])
        (loss_e * scale).backward()
        torch.cuda.synchronize()
        assert int(lg.rng_counter.item()) == 4 + call
        torch.testing.assert_close(samples, samples_e, rtol=0, atol=0)
        torch.testing.assert_close(pose_opt, pose_e, rtol=0, atol=0)
        torch.testing.assert_close(loss, loss_e.detach(), rtol=1e-6, atol=1e-7)
        for a, b in zip(new, ref):
            # the replay runs the captured backward at upstream 1 and scales the result; eager carries `scale` through
            # the kernels: same arithmetic, different rounding, so elements that are sums of cancelling per-sample terms
            # agree norm-wise (a few fp32 ulps of the largest entry), not element-relative
            torch.testing.assert_close(a.grad, b.grad, rtol=1e-5, atol=2e-6 * float(b.grad.abs().max()))
        assert not pose_opt.requires_grad and not samples.requires_grad
    stale, _, _ = graphed(*new, d['pose_init'])
    graphed(*new, d['pose_init'])
    with pytest.raises(RuntimeError, match='replayed again'):
        stale.backward()
    with pytest.raises(ValueError):
        graphed(new[0][:8], new[1][:8], new[2][:8], d['pose_init'][:8])
    del graphed
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_graphed_loss_matches_eager_segment():
    """Own interpreter, as tests/test_graph_rng.py: keeps graph / private-pool teardown away from the other GPU tests.
    16 x 64 and 32 x 512: the second shape runs the AMIS forward split over workgroups, whose exchange slots are reset on the
    stream before every launch -- by a kernel, because a captured memset node replays garbage after the eager work that sits
    between two replays here (tests/test_amis.py::test_split_kernels_in_a_hipgraph_survive_eager_work_between_replays)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ('import sys; sys.path[:0] = [%r, %r, %r]; import test_graphed as t; t._graphed_body(); t._graphed_body(32, 512); '
            'print("GRAPHED-OK", flush=True)') % (here, os.path.join(os.path.dirname(here), 'oracle'),
                                                   os.path.join(os.path.dirname(here), 'epro-pnp_amd'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert 'GRAPHED-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
