#!/usr/bin/env python
"""cProfile of the host side of one step at the launch-bound shapes (C4 nuScenes / C3 LineMOD training) on the GPU box:
where the Python + dispatcher time goes when the GPU is not the bottleneck.   python tools/host_profile.py [c4|c3]"""
import cProfile
import os
import pstats
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

    def step():
        for tt in (x3d, x2d, w2d):
            tt.grad = None
        cf.set_param(x2d.detach(), w2d)
        o = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=p['pose_init'], **kw)
        monte_carlo_pose_loss(o[4], o[5]).mean().backward()

    for _ in range(30):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(300):
        step()
    t_host = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(f'{which}: host-side {t_host / 300 * 1e3:.3f} ms/step, with the final sync {t_all / 300 * 1e3:.3f} ms/step')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(28)


if __name__ == '__main__':
    main()
