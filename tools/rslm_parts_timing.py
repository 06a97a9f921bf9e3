#!/usr/bin/env python
"""RSLM initialiser (600 x 128 points, 4-DoF, 16 points per proposal, 3 LM iterations): one workgroup per object against the
proposals of an object dealt to 2 / 4 workgroups (EPROPNP_TUNE=rslm_parts=..), kernel time by HIP events.  python tools/rslm_parts_timing.py"""
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
    for B, P in ((600, 64), (256, 64), (75, 64), (1024, 64), (2048, 64), (600, 128), (32, 4)):
        prob = bench.synth_problem(B, 128, dev, seed=5, dof=4)
        cam = PerspectiveCamera(z_min=0.1, allowed_border=200)
        cam.set_param(prob['cam_mats'], img_shape=torch.tensor([[480., 640.]], device=dev).expand(B, 2))
        cf = AdaptiveHuberPnPCost(relative_delta=0.5)
        cf.set_param(prob['x2d'], prob['w2d'])
        hp = F.PnPProblem(prob['x3d'], prob['x2d'], prob['w2d'], cam, cf, 4)
        row = dict(objects=B, proposals=P)
        ref = None
        for q in ('1', '2', '4', 'auto'):
            if q == 'auto':
                os.environ.pop('EPROPNP_TUNE', None)
            else:
                os.environ['EPROPNP_TUNE'] = 'rslm_parts=' + q
            for _ in range(3):
                out = F.rslm_solve(hp, P, 16, 3, seed=1, offset=7)
            ts = []
            for _ in range(8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    out = F.rslm_solve(hp, P, 16, 3, seed=1, offset=7)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            row[f'parts={q}_us'] = round(sorted(ts)[len(ts) // 2], 1)
            if ref is None:
                ref = out
            else:
                row[f'parts={q}_same_result'] = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
