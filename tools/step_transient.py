#!/usr/bin/env python
"""Per-step GPU time of the first steps of the C2 workload in a fresh process (HIP events around each step): how long the
device takes to reach its steady state -- what a short warm-up leaves inside the timed region."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'epro-pnp_amd'), ROOT]
import torch
import bench
from epropnp.camera import PerspectiveCamera
from epropnp.cost_fun import AdaptiveHuberPnPCost
from epropnp.epropnp import EProPnP6DoF
from epropnp.levenberg_marquardt import LMSolver
from epropnp.losses import monte_carlo_pose_loss

dev = torch.device('cuda:0')
B, N, S, K, L = 4096, 512, 512, 4, 3
prob = bench.synth_problem(B, N, dev, seed=1000)
x3d, x2d, w2d = (prob[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
cam = PerspectiveCamera(cam_mats=prob['cam_mats'])
cf = AdaptiveHuberPnPCost(relative_delta=0.5)
layer = EProPnP6DoF(mc_samples=S, num_iter=K, solver=LMSolver(dof=6, num_iter=L), seed=1)
if len(sys.argv) > 1:          # optional generic spin-up, seconds of GEMM load
    import time
    t0, a = time.perf_counter(), torch.randn(2048, 2048, device=dev)
    while time.perf_counter() - t0 < float(sys.argv[1]):
        a = torch.mm(a, a).clamp_(-1, 1)
        torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(80)]
for e0, e1 in ev:
    e0.record()
    for t in (x3d, x2d, w2d):
        t.grad = None
    cf.set_param(x2d.detach(), w2d)
    out = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=prob['pose_init'], force_init_solve=False)
    monte_carlo_pose_loss(out[4], out[5]).mean().backward()
    e1.record()
torch.cuda.synchronize()
ms = [e0.elapsed_time(e1) for e0, e1 in ev]
print('steps 0-9  :', ' '.join(f'{m:.3f}' for m in ms[:10]))
print('steps 10-19:', ' '.join(f'{m:.3f}' for m in ms[10:20]))
print('steps 20-29:', ' '.join(f'{m:.3f}' for m in ms[20:30]))
print('steps 30-39:', ' '.join(f'{m:.3f}' for m in ms[30:40]))
print('steps 70-79:', ' '.join(f'{m:.3f}' for m in ms[70:80]))
