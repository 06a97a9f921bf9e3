#!/usr/bin/env python
"""IC-cold timing of normal_equations_kernel (one Jacobian sweep) under workgroup-shape overrides (run on the GPU box).

    python tools/ne_shape_sweep.py [B N]          # driver: one subprocess per EPROPNP_TUNE=ne_shape=waves,ppl
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))
sys.path.insert(0, ROOT)


def worker(B, N):
    import torch
    import bench
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    dev = torch.device('cuda:0')
    prob = bench.synth_problem(B, N, dev, seed=1000)
    cam = PerspectiveCamera(cam_mats=prob['cam_mats'], z_min=0.1)
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)
    cf.set_param(prob['x2d'], prob['w2d'])
    mk = lambda: F.PnPProblem(prob['x3d'].clone(), prob['x2d'].clone(), prob['w2d'].clone(), cam, cf, 6)
    nbytes = B * (28.0 * N + 4.0 * (7 + 9 + 1 + 4) + 4.0 * (21 + 6 + 1))
    mean_ms, med_ms, sets = bench.single_sweep(F, mk, prob['pose_init'], nbytes, windows=12)
    hp = mk()
    out = F.normal_equations(hp, prob['pose_init'])
    chk = float(sum(t.double().abs().sum() for t in out))
    print(json.dumps(dict(mean_us=round(mean_ms * 1e3, 2), median_us=round(med_ms * 1e3, 2), sets=sets,
                          tbps=round(nbytes / (med_ms * 1e-3) / 1e12, 3), frac=round(nbytes / (med_ms * 1e-3) / 8e12, 4),
                          checksum=chk)))


def main():
    if '--worker' in sys.argv:
        worker(int(os.environ['NE_B']), int(os.environ['NE_N']))
        return
    args = [a for a in sys.argv[1:] if not a.startswith('-')]
    B, N = (int(args[0]), int(args[1])) if len(args) >= 2 else (4096, 512)
    shapes = os.environ.get('NE_SHAPES', 'default;1,8;2,4;4,2;8,1;2,8;4,4').split(';')
    for sh in shapes:
        e = dict(os.environ, NE_B=str(B), NE_N=str(N))
        if sh != 'default':
            e['EPROPNP_TUNE'] = 'ne_shape=' + sh
        r = subprocess.run([sys.executable, __file__, '--worker'], env=e, capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'FAILED ' + r.stderr.strip()[-300:]
        print(f'B={B} N={N} shape={sh:8s} {line}', flush=True)


if __name__ == '__main__':
    main()
