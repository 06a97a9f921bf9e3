#!/usr/bin/env python
"""TEST INFRASTRUCTURE, BUILD CONTAINER ONLY (needs /root/reference): the UNMODIFIED reference (oracle/ref_runner.py) and the oracle
restatement (oracle/epropnp_oracle.py) timed side by side on the sample bench.py's `cpu_baseline` leg uses -- 64 objects
of the C2 workload (N=512, S=512, K=4, L=3), monte_carlo_forward + MC loss + backward, same injected noise.
Shows that `cpu_baseline.kind: "port"` costs what the reference costs.  -> profiles/rNN_cpu_reference_vs_oracle.txt and, with
an output path, a small JSON bench.py quotes in its `cpu_baseline` object (`reference_vs_port`):
    python oracle/time_reference_vs_oracle.py [profiles/r05_cpu_reference_vs_oracle.json]"""
import json
import os
import sys
import time

import torch

sys.path[:0] = [os.path.dirname(os.path.abspath(__file__))]
import epropnp_oracle as orc  # noqa: E402
import ref_runner as ref  # noqa: E402


def best_of(fn, n=5):
    ts = []
    for _ in range(n + 1):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts[1:])


def main():
    B, N, S, K, L = 64, 512, 512, 4, 3
    prob = orc.make_problem(B, N, 6, seed=0)      # the SURVEY 8(d) generator bench.synth_problem also implements
    noise = orc.make_noise(B, S, K, 6, seed=1)
    ratios = {}
    lines = [f'host cores {os.cpu_count()}; {B} objects x N={N}, S={S}, K={K}, L={L}, fwd+bwd, best of 5 after 1 warm-up']
    for threads in (1, 4, 8):
        if threads > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(threads)
        t_ref = best_of(lambda: ref.run_mc(prob, noise, 6, S, K, L))
        t_orc = best_of(lambda: orc.run_mc(prob, noise, 6, S, K, L))
        ratios[str(threads)] = {'reference_inst_per_s': round(B / t_ref, 1), 'oracle_inst_per_s': round(B / t_orc, 1),
                                'oracle_over_reference_time': round(t_orc / t_ref, 3)}
        lines.append(f'threads {threads:2d}: reference {B / t_ref:8.1f} inst/s ({t_ref * 1e3:7.1f} ms)   '
                     f'oracle {B / t_orc:8.1f} inst/s ({t_orc * 1e3:7.1f} ms)   oracle/reference time {t_orc / t_ref:.3f}')
    r, o = ref.run_mc(prob, noise, 6, S, K, L), orc.run_mc(prob, noise, 6, S, K, L)
    lines.append(f'same outputs: |pose_opt| diff {float((r["pose_opt"] - o["pose_opt"]).abs().max()):.2e}, '
                 f'|loss_obj| diff {float((r["loss_obj"] - o["loss_obj"]).abs().max()):.2e}')
    print('\n'.join(lines))
    if len(sys.argv) > 1:
        json.dump({'where': 'build container (the reference checkout does not travel to the GPU box)', 'host_cores': os.cpu_count(),
                   'sample': f'{B} objects x N={N}, S={S}, K={K}, L={L}, fwd+bwd, best of 5 after 1 warm-up', 'threads': ratios,
                   'max_abs_diff': {'pose_opt': float((r['pose_opt'] - o['pose_opt']).abs().max()),
                                    'loss_obj': float((r['loss_obj'] - o['loss_obj']).abs().max())}},
                  open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
