#!/usr/bin/env python
"""Kernel-level timing of the C2 workload under shape / build variants (run on the GPU box).

    python tools/tune.py                 # driver: loops over VARIANTS in subprocesses, prints one line each
    python tools/tune.py --worker        # times lm / amis_fwd / amis_bwd once under the current env
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))
sys.path.insert(0, ROOT)


def worker(B, N, S, K, L, reps=8):
    import torch
    import bench
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    dev = torch.device('cuda:0')
    prob = bench.synth_problem(B, N, dev, seed=1000)
    cam = PerspectiveCamera(cam_mats=prob['cam_mats'])
    cf = AdaptiveHuberPnPCost(relative_delta=0.5)
    cf.set_param(prob['x2d'], prob['w2d'])
    hp = F.PnPProblem(prob['x3d'], prob['x2d'], prob['w2d'], cam, cf, 6)

    def timeit(fn, inner=10):
        """median over `reps` windows of `inner` back-to-back launches (a single launch from an idle queue reads ~5 us high)"""
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
        ts.sort()
        return ts[len(ts) // 2], out
    t_ne, _ = timeit(lambda: F.normal_equations(hp, prob['pose_init']))
    t_lm, (pose_opt, cov, _) = timeit(lambda: F.lm_solve(hp, prob['pose_init'], L, with_pose_cov=True, with_cost=True))
    t_fw, (smp, logw) = timeit(lambda: F.amis_forward(hp, pose_opt, cov, S, K, seed=1))
    phases = None
    try:      # tuning builds export per-phase cycle totals of the forward kernel
        import ctypes
        from epropnp import _hip
        fn = _hip.lib().epropnp_tuning_phase_cycles
        buf = (ctypes.c_ulonglong * 8)()
        fn(buf, 1)
        F.amis_forward(hp, pose_opt, cov, S, K, seed=1)
        torch.cuda.synchronize()
        fn(buf, 0)
        tot = float(sum(buf)) or 1.0
        tot = float(sum(buf[:6])) or 1.0
        phases = [round(v / tot, 3) for v in buf][:8]
    except AttributeError:
        pass
    g = -torch.softmax(logw, 0) / B
    gi = torch.full((B,), 1.0 / B, device=dev)
    t_bw, grads = timeit(lambda: F.amis_backward(hp, smp, g, prob['pose_init'], gi))
    gsum = sum(float(t.double().abs().sum()) for t in grads)
    lse = torch.logsumexp(logw, 0).mean().item()
    print(json.dumps(dict(ne_ms=round(t_ne, 4), lm_ms=round(t_lm, 4), fwd_ms=round(t_fw, 4), bwd_ms=round(t_bw, 4), lse=round(lse, 4), gsum=round(gsum, 6), fwd_phases=phases)))


VARIANTS = [
    ('default', {}),
]
# TUNE_VARIANTS="name:KEY=VAL+KEY=VAL|name2:KEY=VAL" adds variants without editing this file.  An UPPER-CASE key is an environment
# variable of its own (the user-facing knobs: EPROPNP_BWD_DROP, EPROPNP_FWD_PROJ ...), a lower-case key goes into the one tuning
# string the library reads, EPROPNP_TUNE="lm_shape=1,8;bwd_impl=valu;ablate=2" (csrc/pnp_host.h: tune_value):
#     TUNE_VARIANTS="norefit:ablate=2|valu:bwd_impl=valu+fwd_impl=valu|exact:EPROPNP_BWD_DROP=0" python tools/tune.py
for _spec in filter(None, os.environ.get('TUNE_VARIANTS', '').split('|')):
    _name, _, _kv = _spec.partition(':')
    _env, _tune = {}, []
    for kv in filter(None, _kv.split('+')):
        k, _, v = kv.partition('=')
        if k.isupper():
            _env[k] = v
        else:
            _tune.append(kv)
    if _tune:
        _env['EPROPNP_TUNE'] = ';'.join(_tune)
    VARIANTS.append((_name, _env))


def main():
    if '--worker' in sys.argv:
        B, N, S, K, L = (int(os.environ.get(k, d)) for k, d in (('TUNE_B', 4096), ('TUNE_N', 512), ('TUNE_S', 512),
                                                                 ('TUNE_K', 4), ('TUNE_L', 3)))
        worker(B, N, S, K, L)
        return
    libs = [('', None)]
    vdir = os.path.join(ROOT, 'epro-pnp_amd', 'lib', 'variants')
    if os.path.isdir(vdir):
        libs += [(t, os.path.join(vdir, t, 'libepropnp_hip.so')) for t in sorted(os.listdir(vdir))]
    for tag, lib in libs:
        for name, env in VARIANTS:
            e = dict(os.environ)
            e.update(env)
            if lib:
                e['EPROPNP_LIB'] = lib
            r = subprocess.run([sys.executable, __file__, '--worker'], env=e, capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'FAILED ' + r.stderr.strip()[-300:]
            print(f'{tag or "base":10s} {name:30s} {line}', flush=True)


if __name__ == '__main__':
    main()
