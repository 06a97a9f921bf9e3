#!/usr/bin/env python
"""Host-side time of each phase of one eager step (no synchronisation inside the loop): where the Python / dispatcher /
autograd-engine time goes at the launch-bound shapes.   python tools/host_phases.py [c4|c3]"""
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


def main():
    dev = torch.device('cuda:0')
    which = sys.argv[1] if len(sys.argv) > 1 else 'c4'
    if which == 'c4':
        B, N = 600, 128
        p = bench.synth_problem(B, N, dev, seed=5, dof=4)
        cam = PerspectiveCamera(z_min=0.1, allowed_border=200)
        cam.set_param(p['cam_mats'], img_shape=torch.tensor([[480., 640.]], device=dev).expand(B, 2))
        layer = EProPnP4DoF(mc_samples=128, num_iter=4, normalize=True,
                            solver=LMSolver(dof=4, num_iter=5, init_solver=RSLMSolver(dof=4, num_points=16, num_proposals=64, num_iter=3)))
        kw = dict(force_init_solve=True)
    else:
        B, N = 32, 512
        p = bench.synth_problem(B, N, dev, seed=5, dof=6)
        cam = PerspectiveCamera(cam_mats=p['cam_mats'], z_min=0.01)
        layer = EProPnP6DoF(mc_samples=512, num_iter=4, solver=LMSolver(dof=6, num_iter=5))
        kw = dict(force_init_solve=False)
    x3d, x2d, w2d = (p[k].requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)
    acc = dict(zero=0.0, set_param=0.0, forward=0.0, loss=0.0, backward=0.0)
    n = 400
    for it in range(n + 50):
        if it == 50:
            torch.cuda.synchronize()
            for k in acc:
                acc[k] = 0.0
            t_all = time.perf_counter()
        t0 = time.perf_counter()
        for tt in (x3d, x2d, w2d):
            tt.grad = None
        t1 = time.perf_counter()
        cf.set_param(x2d.detach(), w2d)
        t2 = time.perf_counter()
        o = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=p['pose_init'], **kw)
        t3 = time.perf_counter()
        loss = monte_carlo_pose_loss(o[4], o[5]).mean()
        t4 = time.perf_counter()
        loss.backward()
        t5 = time.perf_counter()
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[k] += v
    host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    tot = time.perf_counter() - t_all
    print(f'{which}: host {host / n * 1e3:.3f} ms/step (with final sync {tot / n * 1e3:.3f}); phases [us]: '
          + ', '.join(f'{k} {v / n * 1e6:.0f}' for k, v in acc.items()))


if __name__ == '__main__':
    main()
