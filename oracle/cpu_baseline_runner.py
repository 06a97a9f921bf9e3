#!/usr/bin/env python
"""TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.  One CPU-baseline measurement in a process of its own, for bench.py's
`cpu_baseline` leg: `objects` objects of the SURVEY 8(d) synthetic workload (monte_carlo_forward + MC loss + backward; reference
epropnp/epropnp.py:87-196 and the callers' loss) in chunks of `chunk`, one warm-up chunk, `passes` timed passes, on `threads` threads.

    cpu_baseline_runner.py reference|port N S K L objects chunk threads [passes]

  reference : the UNMODIFIED modules of a reference checkout ($EPROPNP_REFERENCE, default /root/reference) through oracle/ref_runner.py's
              loader (pyro absent -> oracle/pyro_shim.py; random draws injected so that both kinds do the same arithmetic).  The checkout
              does not travel to the GPU box: there this kind exits with status 3 and bench.py reports the port.
  port      : oracle/epropnp_oracle.py, the restatement.
A process of its own because the reference's package is called `epropnp`, like the product's.  Prints one JSON object."""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE]


def main(argv):
    kind, (N, S, K, L, objects, chunk, threads) = argv[1], (int(a) for a in argv[2:9])
    passes = int(argv[9]) if len(argv) > 9 else 3
    import epropnp_oracle as orc
    if kind == 'reference':
        root = os.environ.get('EPROPNP_REFERENCE', '/root/reference')
        if not os.path.isdir(os.path.join(root, 'epropnp')):
            print(json.dumps({'error': f'no reference checkout at {root}'}))
            return 3
        import ref_runner as impl
        impl.load_reference()
    else:
        impl = orc
    torch.set_num_threads(threads)
    prob = orc.make_problem(objects, N, 6, seed=0)
    noise = orc.make_noise(objects, S, K, 6, seed=1)

    def one_pass(count):
        t0 = time.perf_counter()
        for lo in range(0, count, chunk):
            hi = min(count, lo + chunk)
            impl.run_mc({k: v[lo:hi].contiguous() for k, v in prob.items()}, {k: v[:, :, lo:hi].contiguous() for k, v in noise.items()},
                        6, S, K, L)
        return time.perf_counter() - t0
    one_pass(min(chunk, objects))
    ts = [one_pass(objects) for _ in range(passes)]
    print(json.dumps({'kind': kind, 'objects': objects, 'chunk': chunk, 'threads': threads, 'pass_seconds': [round(t, 3) for t in ts],
                      'value': round(objects / sorted(ts)[len(ts) // 2], 2), 'unit': 'instances/s'}))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
