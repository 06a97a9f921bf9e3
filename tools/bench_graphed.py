#!/usr/bin/env python
"""epropnp.graphed.GraphedLoss against the same segment run eagerly, at the launch-bound shapes (fwd + bwd to the inputs,
fresh input tensors every step as a backbone would produce them).   python tools/bench_graphed.py [C3 C4 C2]"""
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
from epropnp.graphed import GraphedLoss  # noqa: E402
from epropnp.levenberg_marquardt import LMSolver, RSLMSolver  # noqa: E402
from epropnp.losses import monte_carlo_pose_loss  # noqa: E402


def build(name, dev):
    if name == 'C4':
        B, N = 600, 128
        p = bench.synth_problem(B, N, dev, seed=5, dof=4)
        cam = PerspectiveCamera(z_min=0.1, allowed_border=200)
        cam.set_param(p['cam_mats'], img_shape=torch.tensor([[480., 640.]], device=dev).expand(B, 2))
        layer = EProPnP4DoF(mc_samples=128, num_iter=4, normalize=True,
                            solver=LMSolver(dof=4, num_iter=5, init_solver=RSLMSolver(dof=4, num_points=16, num_proposals=64, num_iter=3)))
        force = True
    else:
        B, N = (32, 512) if name == 'C3' else (4096, 512)
        p = bench.synth_problem(B, N, dev, seed=4)
        cam = PerspectiveCamera(cam_mats=p['cam_mats'])
        layer = EProPnP6DoF(mc_samples=512, num_iter=4, solver=LMSolver(dof=6, num_iter=5 if name == 'C3' else 3))
        force = False
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)

    def segment(x3d, x2d, w2d, pose_gt):
        cf.set_param(x2d.detach(), w2d)
        o = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=pose_gt, force_init_solve=force)
        return monte_carlo_pose_loss(o[4], o[5]).mean()
    return B, p, layer, segment


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
        B, p, layer, segment = build(name, dev)
        base = [p[k] for k in ('x3d', 'x2d', 'w2d')]

        def run(fn):
            ins = [t.detach().requires_grad_(True) for t in base]      # new leaves every step
            fn(*ins, p['pose_init']).backward()
            return ins
        for _ in range(5):
            run(segment)
        t_eager = timed(lambda: run(segment))
        graphed = GraphedLoss(segment, (*[t.detach().requires_grad_(True) for t in base], p['pose_init']), layers=[layer])
        ins = run(graphed)
        t_graph = timed(lambda: run(graphed))
        print(json.dumps(dict(config=name, objects=B, eager_ms=round(t_eager * 1e3, 4), graphed_ms=round(t_graph * 1e3, 4),
                              speedup=round(t_eager / t_graph, 2), inst_per_s=round(B / t_graph, 1),
                              grads_finite=all(bool(torch.isfinite(t.grad).all()) for t in ins))))
        del graphed
        torch.cuda.synchronize()


if __name__ == '__main__':
    main()
