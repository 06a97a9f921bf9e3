#!/usr/bin/env python
"""Per-phase shader-clock shares of the AMIS forward and the MFMA backward at a given shape, from a TUNING build of the library:

    python epro-pnp_amd/build.py --tag t -D PNP_TUNING            # lib/variants/t/
    EPROPNP_LIB=epro-pnp_amd/lib/variants/t/libepropnp_hip.so python tools/phase_shares.py [B N S K bounded]

forward phases  [initial fit + load | draw | sweep (+ exchange) | weights | refit | store]
backward phases [weights + max | drop threshold | compaction | pose rows | sweep + outputs | tail]
(cycles of thread 0 of every workgroup, summed: shares of a workgroup's lifetime, not of the launch).  Also prints the kernels'
launch times by HIP events under the same build (the counters cost a few percent)."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))
sys.path.insert(0, ROOT)


def main():
    import bench
    from epropnp import _hip
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    a = sys.argv[1:]
    B, N, S, K = (int(a[i]) if len(a) > i else d for i, d in enumerate((32, 4096, 512, 4)))
    bounded = (a[4] if len(a) > 4 else '1') == '1'
    dev = torch.device('cuda:0')
    prob = bench.synth_problem(B, N, dev, seed=1000)
    lb = ub = None
    if bounded:
        lo_, hi_ = prob['x2d'].amin(1), prob['x2d'].amax(1)
        unit = (hi_ - lo_).amax(-1, keepdim=True) / 64.0
        lb, ub = (lo_ - 30 * unit).contiguous(), (hi_ + 30 * unit).contiguous()
    cam = PerspectiveCamera(cam_mats=prob['cam_mats'], z_min=0.01, lb=lb, ub=ub)
    cf = AdaptiveHuberPnPCost(relative_delta=0.1)
    cf.set_param(prob['x2d'], prob['w2d'])
    hp = F.PnPProblem(prob['x3d'], prob['x2d'], prob['w2d'], cam, cf, 6)
    pose_opt, cov, _ = F.lm_solve(hp, prob['pose_init'], 5, with_pose_cov=True, with_cost=True)

    def timed(fn, reps=8, inner=10):
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
            ts.append(e0.elapsed_time(e1) / inner)
        return sorted(ts)[len(ts) // 2] * 1e3, out

    def shares(sym, fn):
        try:
            f = getattr(_hip.lib(), sym)
        except AttributeError:
            return None
        buf = (ctypes.c_ulonglong * 8)()
        if f(buf, 1) != 0:
            return None
        fn()
        torch.cuda.synchronize()
        f(buf, 0)
        tot = float(sum(buf[:6])) or 1.0
        return [round(v / tot, 3) for v in buf[:6]]
    t_fw, (smp, logw) = timed(lambda: F.amis_forward(hp, pose_opt, cov, S, K, seed=1))
    g = -torch.softmax(logw, 0) / B
    gi = torch.full((B,), 1.0 / B, device=dev)
    t_bw, _ = timed(lambda: F.amis_backward(hp, smp, g, prob['pose_init'], gi))
    t_lm, _ = timed(lambda: F.lm_solve(hp, prob['pose_init'], 5, with_pose_cov=True, with_cost=True))
    out = dict(B=B, N=N, S=S, K=K, bounded=bounded, lib=os.environ.get('EPROPNP_LIB', 'default'), lm_us=round(t_lm, 1), fwd_us=round(t_fw, 1),
               bwd_us=round(t_bw, 1), bwd_nsplit=F.backward_split(B, N, S),
               fwd_phases=shares('epropnp_tuning_phase_cycles', lambda: F.amis_forward(hp, pose_opt, cov, S, K, seed=1)),
               bwd_phases=shares('epropnp_tuning_bwd_cycles', lambda: F.amis_backward(hp, smp, g, prob['pose_init'], gi)))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
