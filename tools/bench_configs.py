#!/usr/bin/env python
"""Timings of the other BASELINE.json configurations on one MI355X (C3 LineMOD shape, C4 nuScenes shape, one GPU's
shard of the C5 stress config).  One JSON line per configuration; bench.py remains the headline (C2) contract."""
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


def timed(fn, steps=20, warmup=3, repeats=5):
    """median over `repeats` of the mean step time of `steps` back-to-back steps (the small shapes are launch-bound and
    a single window varies by +-15 % with host jitter)"""
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device('cuda:0')
    out = []
    # ---- C3 inference: 32 objects x 4096 dense points, Gauss-Newton fast mode, 3 iterations, tensor bounds
    B, N = 32, 4096
    p = bench.synth_problem(B, N, dev, seed=3)
    lb = torch.tensor([-200.5, -200.5], device=dev).expand(B, 2).contiguous()
    ub = torch.tensor([839.5, 679.5], device=dev).expand(B, 2).contiguous()
    cam = PerspectiveCamera(cam_mats=p['cam_mats'], lb=lb, ub=ub)
    cf = AdaptiveHuberPnPCost(relative_delta=0.1)
    layer = EProPnP6DoF(mc_samples=512, num_iter=4, solver=LMSolver(dof=6, num_iter=3))

    def c3_infer():
        with torch.no_grad():
            cf.set_param(p['x2d'], p['w2d'])
            layer(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'], fast_mode=True)
    t = timed(c3_infer)
    out.append(dict(config='C3 LineMOD inference: 32 obj x 4096 pts, GN fast_mode 3 iters', ms=round(t * 1e3, 4),
                    objects_per_s=round(B / t, 1)))
    # ---- C3 training: 32 objects x 512 sub-sampled points, LM 5 + AMIS 512/4, fwd+bwd
    B, N = 32, 512
    p = bench.synth_problem(B, N, dev, seed=4)
    x3d, x2d, w2d = (p[k].requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    cam = PerspectiveCamera(cam_mats=p['cam_mats'], lb=lb, ub=ub)
    layer = EProPnP6DoF(mc_samples=512, num_iter=4, solver=LMSolver(dof=6, num_iter=5))

    def train_step(layer, x3d, x2d, w2d, cam, cf, pose_init, force):
        for tt in (x3d, x2d, w2d):
            tt.grad = None
        cf.set_param(x2d.detach(), w2d)
        o = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=pose_init, force_init_solve=force)
        monte_carlo_pose_loss(o[4], o[5]).mean().backward()
    t = timed(lambda: train_step(layer, x3d, x2d, w2d, cam, cf, p['pose_init'], False))
    out.append(dict(config='C3 LineMOD training: 32 obj x 512 pts, LM 5 + AMIS S=512 K=4, fwd+bwd', ms=round(t * 1e3, 4),
                    instances_per_s=round(B / t, 1)))
    # ---- C3 dense (BASELINE.json configs[2] as worded): 32 objects x ALL 4096 correspondences of the 64 x 64 crop through
    # 6-DoF LM + AMIS, fwd+bwd, tensor bounds (the reference trains on a 512-point subset and solves the dense set at test time)
    B, N = 32, 4096
    pd = bench.synth_problem(B, N, dev, seed=6)
    xd = [pd[k].requires_grad_(True) for k in ('x3d', 'x2d', 'w2d')]
    camd = PerspectiveCamera(cam_mats=pd['cam_mats'], lb=lb, ub=ub)
    t = timed(lambda: train_step(layer, *xd, camd, cf, pd['pose_init'], False))
    out.append(dict(config='C3 LineMOD dense: 32 obj x 4096 pts, LM 5 + AMIS S=512 K=4, fwd+bwd', ms=round(t * 1e3, 4),
                    instances_per_s=round(B / t, 1)))
    # ---- C4 nuScenes: 600 objects x 128 points, 4-DoF, S=128, K=4, normalize, RSLM(16,64,3) + LM 5, image bounds
    B, N = 600, 128
    p = bench.synth_problem(B, N, dev, seed=5, dof=4)
    x3d, x2d, w2d = (p[k].requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    cam = PerspectiveCamera(z_min=0.1, allowed_border=200)
    cam.set_param(p['cam_mats'], img_shape=torch.tensor([[480., 640.]], device=dev).expand(B, 2))
    cf4 = AdaptiveHuberPnPCost(relative_delta=0.5)
    layer4 = EProPnP4DoF(mc_samples=128, num_iter=4, normalize=True,
                         solver=LMSolver(dof=4, num_iter=5, init_solver=RSLMSolver(dof=4, num_points=16, num_proposals=64, num_iter=3)))
    t = timed(lambda: train_step(layer4, x3d, x2d, w2d, cam, cf4, p['pose_init'], True))
    out.append(dict(config='C4 nuScenes: 600 obj x 128 pts, 4-DoF, RSLM(16,64,3) + LM 5 + AMIS S=128 K=4, normalize, fwd+bwd',
                    ms=round(t * 1e3, 4), instances_per_s=round(B / t, 1)))

    def c4_infer():
        with torch.no_grad():
            cf4.set_param(p['x2d'], p['w2d'])
            layer4(p['x3d'].detach(), p['x2d'].detach(), p['w2d'].detach(), cam, cf4, fast_mode=True)
    layer4.solver.num_iter = 5
    t = timed(c4_infer)
    out.append(dict(config='C4 nuScenes inference: 600 obj x 128 pts, RSLM init + GN 5', ms=round(t * 1e3, 4),
                    objects_per_s=round(B / t, 1)))
    # ---- C5: one GPU's shard of the stress config
    B, N = 8192, 2048
    p = bench.synth_problem(B, N, dev, seed=6)
    x3d, x2d, w2d = (p[k].requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    cam = PerspectiveCamera(cam_mats=p['cam_mats'])
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)
    layer = EProPnP6DoF(mc_samples=1024, num_iter=4, solver=LMSolver(dof=6, num_iter=3))
    t = timed(lambda: train_step(layer, x3d, x2d, w2d, cam, cf, p["pose_init"], False), steps=5, warmup=2, repeats=3)
    out.append(dict(config='C5 stress shard: 8192 obj x 2048 pts, S=1024 K=4 L=3, fwd+bwd', ms=round(t * 1e3, 3),
                    instances_per_s=round(B / t, 1),
                    fp32_tflops=round((40 * 1024 + 80 * 1025) * 2048 * B / t / 1e12, 1)))
    for o in out:
        print(json.dumps(o), flush=True)


if __name__ == '__main__':
    main()
