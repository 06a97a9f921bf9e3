#!/usr/bin/env python
"""Kernel times of the projection-clamp (BOUNDS) instantiations -- the ones the reference's real callers use (LineMOD crops
with tensor bounds, nuScenes with img_shape bounds): LM, AMIS forward, AMIS backward at 4096 x 512 x 512 (6-DoF) and
600 x 128 x 128 (4-DoF).  python tools/bounds_timing.py"""
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
    dev = torch.device('cuda:0')

    def timeit(fn):
        for _ in range(3):
            out = fn()
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                out = fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        return round(sorted(ts)[len(ts) // 2], 4), out
    for B, N, S, dof, L in ((4096, 512, 512, 6, 3), (600, 128, 128, 4, 5)):
        prob = bench.synth_problem(B, N, dev, seed=1000, dof=dof)
        for bounded in (False, True):
            kw = dict(lb=torch.tensor([-200.5, -200.5], device=dev).expand(B, 2).contiguous(),
                      ub=torch.tensor([839.5, 679.5], device=dev).expand(B, 2).contiguous()) if bounded else {}
            cam = PerspectiveCamera(cam_mats=prob['cam_mats'], **kw)
            cf = AdaptiveHuberPnPCost(relative_delta=0.5)
            cf.set_param(prob['x2d'], prob['w2d'])
            hp = F.PnPProblem(prob['x3d'], prob['x2d'], prob['w2d'], cam, cf, dof)
            t_lm, (po, cov, _) = timeit(lambda: F.lm_solve(hp, prob['pose_init'], L, with_pose_cov=True, with_cost=True))
            t_fw, (smp, logw) = timeit(lambda: F.amis_forward(hp, po, cov, S, 4, seed=1))
            g = -torch.softmax(logw, 0) / B
            gi = torch.full((B,), 1.0 / B, device=dev)
            t_bw, _ = timeit(lambda: F.amis_backward(hp, smp, g, prob['pose_init'], gi))
            print(json.dumps(dict(objects=B, points=N, samples=S, dof=dof, bounds=bounded, lm_ms=t_lm, fwd_ms=t_fw, bwd_ms=t_bw,
                                  lse=round(float(torch.logsumexp(logw, 0).mean()), 5))), flush=True)


if __name__ == '__main__':
    main()
