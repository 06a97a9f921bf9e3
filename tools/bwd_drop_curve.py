#!/usr/bin/env python
"""Time and gradient error of the AMIS backward against its drop budget EPROPNP_BWD_DROP (the fraction of an object's total
|weight| the kernel may leave out, csrc/amis_common.h:mass_drop_threshold) at C2, with the weights of the real loss
(softmax of the log-weights): one line per budget.  python tools/bwd_drop_curve.py [B N S]"""
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
    B, N, S = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 512, 512)
    K, L = 4, 3
    dev = torch.device('cuda:0')
    prob = bench.synth_problem(B, N, dev, seed=1000)
    cam = PerspectiveCamera(cam_mats=prob['cam_mats'])
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)
    cf.set_param(prob['x2d'], prob['w2d'])
    hp = F.PnPProblem(prob['x3d'], prob['x2d'], prob['w2d'], cam, cf, 6)
    pose_opt, cov, _ = F.lm_solve(hp, prob['pose_init'], L, with_pose_cov=True, with_cost=True)
    smp, logw = F.amis_forward(hp, pose_opt, cov, S, K, seed=1)
    g = -torch.softmax(logw, 0) / B                    # d(mean KL loss) / d logweights
    gi = torch.full((B,), 1.0 / B, device=dev)
    a = g.abs()
    srt = torch.sort(a, dim=0).values
    cum = torch.cumsum(srt, 0) / srt.sum(0, keepdim=True)
    ess = (1.0 / (torch.softmax(logw, 0) ** 2).sum(0)).mean().item()

    def run(eps):
        os.environ['EPROPNP_BWD_DROP'] = repr(eps)
        for _ in range(3):
            out = F.amis_backward(hp, smp, g, prob['pose_init'], gi)
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                out = F.amis_backward(hp, smp, g, prob['pose_init'], gi)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        return sorted(ts)[len(ts) // 2], out
    t0, exact = run(0.0)
    print(json.dumps(dict(shape=[B, N, S], effective_sample_size_mean=round(ess, 1), exact_ms=round(t0, 4))))
    for eps in (2.0 ** -24, 1e-6, 1e-5, 1e-4, 1e-3):
        t, out = run(eps)
        errs = [float((o.double() - e.double()).abs().max() / e.double().abs().max()) for o, e in zip(out, exact)]
        per_obj = ((out[0].double() - exact[0].double()).abs().flatten(1).amax(1) / exact[0].double().abs().flatten(1).amax(1)).max().item()
        dropped = float((cum <= eps).float().mean())        # samples within the budget (the kernel's power-of-two bins drop fewer)
        print(json.dumps(dict(drop_eps=eps, bwd_ms=round(t, 4), vs_exact=round(t / t0, 4), samples_within_budget=round(dropped, 4),
                              rel_err_gx3d_gx2d_gw2d_gdelta=[float(f'{v:.2e}') for v in errs],
                              worst_object_rel_err_gx3d=float(f'{per_obj:.2e}'))))
    os.environ.pop('EPROPNP_BWD_DROP', None)


if __name__ == '__main__':
    main()
