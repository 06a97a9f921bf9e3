#!/usr/bin/env python
"""A/B of two forward-kernel variants selected by environment variables, on identical Philox counters (run on the GPU box).

    python tools/diag_fwd_variant.py EPROPNP_TUNE=fwd_mfma=8,4          # e.g. another launch shape; any KEY=VALUE pairs

The first AMIS iteration draws from the same proposal with the same counters in both variants, so its samples must be
bit-identical and its log-weights differ only by the sweep's arithmetic: the script prints where they differ (by sample
slot within a 16-pose tile, by pose tile, by object)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    env = dict(kv.split('=', 1) for kv in sys.argv[1:])
    dev = torch.device('cuda:0')
    torch.set_printoptions(precision=3, linewidth=200, sci_mode=True)
    for B, N, S, K in ((8, 64, 64, 4), (8, 128, 128, 4), (8, 512, 512, 4), (64, 512, 512, 4), (4096, 512, 512, 4)):
        prob = bench.synth_problem(B, N, dev, seed=1000)
        cam = PerspectiveCamera(cam_mats=prob['cam_mats'])
        cf = AdaptiveHuberPnPCost(relative_delta=0.5)
        cf.set_param(prob['x2d'], prob['w2d'])
        hp = F.PnPProblem(prob['x3d'], prob['x2d'], prob['w2d'], cam, cf, 6)
        pose_opt, cov, _ = F.lm_solve(hp, prob['pose_init'], 3, with_pose_cov=True, with_cost=True)
        out = []
        for e in ({}, env):
            os.environ.update(e)
            smp, logw = F.amis_forward(hp, pose_opt, cov, S, K, seed=1)
            torch.cuda.synchronize()
            out.append((smp.clone(), logw.clone()))
            for k in e:
                os.environ.pop(k)
        s = S // K
        (s0, l0), (s1, l1) = out
        print(f'--- B={B} N={N} S={S} s={s}: first-iteration samples identical: {bool((s0[:s] == s1[:s]).all())}, '
              f'lse {torch.logsumexp(l0, 0).mean().item():.4f} vs {torch.logsumexp(l1, 0).mean().item():.4f}')
        d = (l0[:s] - l1[:s]).abs()          # (s, B)
        print(f'    |dlogw| it0: max {d.max().item():.3e} mean {d.mean().item():.3e}  |logw| mean {l0[:s].abs().mean().item():.3e}')
        if s >= 16:
            print('    by slot in tile :', d.reshape(s // 16, 16, B).amax((0, 2)).cpu())
            print('    by pose tile    :', d.reshape(s // 16, 16, B).amax((1, 2)).cpu())
        print('    by object (<=16):', d.amax(0)[:16].cpu())
        print('    logw fp32 [0:6,0]:', l0[:6, 0].cpu(), ' variant:', l1[:6, 0].cpu())
        print('    dlogw signed [0:16,0]:', (l1[:16, 0] - l0[:16, 0]).cpu())


if __name__ == '__main__':
    main()
