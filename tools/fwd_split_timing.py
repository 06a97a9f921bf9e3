#!/usr/bin/env python
"""amis_forward at few objects: one workgroup per object against G workgroups per object (EPROPNP_FWD_SPLIT), kernel time by
HIP events.  python tools/fwd_split_timing.py"""
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
    for B, N, S in ((32, 512, 512), (64, 512, 512), (128, 512, 512), (16, 512, 512), (32, 128, 128)):
        prob = bench.synth_problem(B, N, dev, seed=4)
        cam = PerspectiveCamera(cam_mats=prob['cam_mats'])
        cf = AdaptiveHuberPnPCost(relative_delta=0.5)
        cf.set_param(prob['x2d'], prob['w2d'])
        hp = F.PnPProblem(prob['x3d'], prob['x2d'], prob['w2d'], cam, cf, 6)
        pose_opt, cov, _ = F.lm_solve(hp, prob['pose_init'], 5, with_pose_cov=True)
        row = dict(objects=B, points=N, samples=S)
        for g in ('1', '2', '4', '8', 'auto'):
            if g == 'auto':
                os.environ.pop('EPROPNP_FWD_SPLIT', None)
            else:
                os.environ['EPROPNP_FWD_SPLIT'] = g
            for _ in range(5):
                F.amis_forward(hp, pose_opt, cov, S, 4, seed=1)
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    F.amis_forward(hp, pose_opt, cov, S, 4, seed=1)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            row[f'G={g}_us'] = round(sorted(ts)[len(ts) // 2], 1)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
