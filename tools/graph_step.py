#!/usr/bin/env python
"""A whole training step (set_param -> monte_carlo_forward -> MC loss -> backward) captured into a hipGraph
(torch.cuda.CUDAGraph) and replayed, against the same step launched eagerly.  The layer's Philox call counter lives in
device memory (EProPnPBase.enable_graph_safe_rng), so every replay draws fresh samples.

    python tools/graph_step.py            # C3 (32 x 512, 6-DoF), C4 (600 x 128, 4-DoF + RSLM + normalize), C2
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))
import bench  # noqa: E402
from epropnp.camera import PerspectiveCamera  # noqa: E402
from epropnp.cost_fun import AdaptiveHuberPnPCost  # noqa: E402
from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF  # noqa: E402
from epropnp.levenberg_marquardt import LMSolver, RSLMSolver  # noqa: E402
from epropnp.losses import monte_carlo_pose_loss  # noqa: E402


def build(name, dev):
    if name == 'C4':
        B, N, dof = 600, 128, 4
        p = bench.synth_problem(B, N, dev, seed=5, dof=4)
        cam = PerspectiveCamera(z_min=0.1, allowed_border=200)
        cam.set_param(p['cam_mats'], img_shape=torch.tensor([[480., 640.]], device=dev).expand(B, 2))
        layer = EProPnP4DoF(mc_samples=128, num_iter=4, normalize=True,
                            solver=LMSolver(dof=4, num_iter=5, init_solver=RSLMSolver(dof=4, num_points=16,
                                                                                       num_proposals=64, num_iter=3)))
        force = True
    else:
        B, N, dof = (32, 512, 6) if name == 'C3' else (4096, 512, 6)
        p = bench.synth_problem(B, N, dev, seed=4)
        cam = PerspectiveCamera(cam_mats=p['cam_mats'])
        layer = EProPnP6DoF(mc_samples=512, num_iter=4, solver=LMSolver(dof=6, num_iter=5 if name == 'C3' else 3))
        force = False
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)
    leaves = [p[k].requires_grad_(True) for k in ('x3d', 'x2d', 'w2d')]
    layer.enable_graph_safe_rng(dev)
    out = {}

    plus = os.environ.get('GRAPH_STEP_PLUS', '0') == '1'      # derivative regularisation through pose_opt_plus as well

    def step():
        cf.set_param(leaves[1].detach(), leaves[2])
        o = layer.monte_carlo_forward(*leaves, cam, cf, pose_init=p['pose_init'], force_init_solve=force,
                                      with_pose_opt_plus=plus)
        loss = monte_carlo_pose_loss(o[4], o[5]).mean()
        if plus:
            loss = loss + 0.1 * (o[2][:, :3] - p['pose_init'][:, :3]).norm(dim=-1).mean()
        loss.backward()
        out['loss'], out['samples'] = loss.detach(), o[3]
    return B, leaves, step, out, layer


def timed(fn, steps=50, repeats=5):
    ts = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device('cuda:0')
    for name in sys.argv[1:] or ['C3', 'C4', 'C2']:
        B, leaves, step, out, layer = build(name, dev)

        def eager():
            for t in leaves:
                t.grad = None
            step()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eager()
        torch.cuda.current_stream().wait_stream(side)
        t_eager = timed(eager)
        for t in leaves:
            t.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        graph.replay()
        torch.cuda.synchronize()
        s1, g1 = out['samples'].clone(), leaves[0].grad.clone()
        graph.replay()
        torch.cuda.synchronize()
        fresh = bool((out['samples'] - s1).abs().max() > 0)
        # gradients are accumulated into the static .grad buffers by the captured backward: zero them in-graph-free style
        t_graph = timed(graph.replay)
        print(json.dumps(dict(config=name, objects=B, eager_ms=round(t_eager * 1e3, 4), graph_ms=round(t_graph * 1e3, 4),
                              speedup=round(t_eager / t_graph, 2), fresh_samples_per_replay=fresh,
                              counter=int(layer.rng_counter.item()), grad_finite=bool(torch.isfinite(g1).all()))))


if __name__ == '__main__':
    main()
