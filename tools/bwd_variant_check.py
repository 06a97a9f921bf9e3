#!/usr/bin/env python
"""Backward of the C2 workload under a build variant of the library (EPROPNP_LIB) against the default build: per-object
relative error of the three gradients and run-to-run reproducibility.   python tools/bwd_variant_check.py <variant.so> ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))
sys.path.insert(0, ROOT)


def worker(out):
    import torch
    import bench
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    dev = torch.device('cuda:0')
    B, N, S, K, L = 4096, 512, 512, 4, 3
    prob = bench.synth_problem(B, N, dev, seed=1000)
    cam = PerspectiveCamera(cam_mats=prob['cam_mats'])
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)
    cf.set_param(prob['x2d'], prob['w2d'])
    hp = F.PnPProblem(prob['x3d'], prob['x2d'], prob['w2d'], cam, cf, 6)
    pose_opt, cov, _ = F.lm_solve(hp, prob['pose_init'], L, with_pose_cov=True, with_cost=True)
    smp, logw = F.amis_forward(hp, pose_opt, cov, S, K, seed=1)
    g = -torch.softmax(logw, 0) / B
    gi = torch.full((B,), 1.0 / B, device=dev)
    runs = [F.amis_backward(hp, smp, g, prob['pose_init'], gi) for _ in range(3)]
    torch.cuda.synchronize()
    same = all(all(torch.equal(a, b) for a, b in zip(runs[0], r)) for r in runs[1:])
    torch.save({'grads': [t.cpu() for t in runs[0]], 'reproducible': same, 'smp_sum': float(smp.double().sum())}, out)


def main():
    if '--worker' in sys.argv:
        worker(sys.argv[-1])
        return
    import torch
    outs = {}
    for name, lib in [('default', None)] + [(os.path.basename(os.path.dirname(p)), p) for p in sys.argv[1:]]:
        e = dict(os.environ)
        if lib:
            e['EPROPNP_LIB'] = lib
        out = f'/tmp/bwd_check_{name}.pt'
        r = subprocess.run([sys.executable, __file__, '--worker', out], env=e, capture_output=True, text=True)
        if not os.path.exists(out):
            print(name, 'FAILED', r.stderr[-400:])
            continue
        outs[name] = torch.load(out)
    ref = outs['default']
    for name, o in outs.items():
        line = [f'{name:16s} reproducible={o["reproducible"]} same_forward={o["smp_sum"] == ref["smp_sum"]}']
        for label, a, b in zip(('x3d', 'x2d', 'w2d', 'delta'), o['grads'], ref['grads']):
            a, b = a.double(), b.double()
            if a.dim() > 1:
                scale = b.abs().amax(dim=tuple(range(1, b.dim())), keepdim=True).clamp(min=1e-30)
            else:
                scale = b.abs().clamp(min=1e-30)
            err = ((a - b).abs() / scale)
            line.append(f'{label}: max {err.max().item():.2e} objects>1e-4: {int((err.reshape(err.shape[0], -1).amax(1) > 1e-4).sum())}')
        print('  '.join(line), flush=True)


if __name__ == '__main__':
    main()
