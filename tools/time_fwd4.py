#!/usr/bin/env python
"""AMIS forward / backward / RSLM / LM kernel times at the Det shape (600 x 128 points, 4-DoF, S = 128, K = 4) under the current
env (EPROPNP_LIB, EPROPNP_TUNE=ablate=.. ...): median of 8 windows of 10 launches.   python tools/time_fwd4.py [B N S K]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))
sys.path.insert(0, ROOT)


def main():
    import bench
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    B, N, S, K = (int(v) for v in (sys.argv[1:5] + ['600', '128', '128', '4'][len(sys.argv) - 1:]))
    dev = torch.device('cuda:0')
    prob = bench.synth_problem(B, N, dev, seed=5, dof=4)
    cam = PerspectiveCamera(z_min=0.1, allowed_border=200)
    cam.set_param(prob['cam_mats'], img_shape=torch.tensor([[480., 640.]], device=dev).expand(B, 2))
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)
    cf.set_param(prob['x2d'], prob['w2d'])
    hp = F.PnPProblem(prob['x3d'], prob['x2d'], prob['w2d'], cam, cf, 4)

    def timeit(fn, inner=10, reps=8):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                out = fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / inner * 1e3)
        ts.sort()
        return round(ts[len(ts) // 2], 1), out
    t_rs, start = timeit(lambda: F.rslm_solve(hp, 64, 16, 3, seed=1, offset=7))
    t_lm, (pose_opt, cov, _) = timeit(lambda: F.lm_solve(hp, start[0], 5, with_pose_cov=True, with_cost=True))
    t_fw, (smp, logw) = timeit(lambda: F.amis_forward(hp, pose_opt, cov, S, K, seed=1))
    g = -torch.softmax(logw, 0) / B
    gi = torch.full((B,), 1.0 / B, device=dev)
    t_bw, grads = timeit(lambda: F.amis_backward(hp, smp, g, prob['pose_init'], gi))
    print(json.dumps(dict(B=B, N=N, S=S, K=K, rslm_us=t_rs, lm_us=t_lm, fwd_us=t_fw, bwd_us=t_bw,
                          lse=round(torch.logsumexp(logw, 0).mean().item(), 4), yaw_sum=round(float(smp[..., 3].double().abs().sum()), 4))))


if __name__ == '__main__':
    main()
