#!/bin/bash
# where the host time of an eagerly launched step goes: cProfile over bench.py's eager leg.  usage: tools/gpu_host_profile.sh <config> <tag>
export TMPDIR=/tmp
CFG=${1:-C4}; TAG=${2:-host}; O=/root/repo/gpurun_out; mkdir -p $O
STEPS=600
for i in 1 2 3; do python bench.py --config $CFG --launch eager --steps $STEPS --warmup 30 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('plain eager ms_per_step', d['ms_per_step'])"; done > $O/${TAG}_${CFG}.txt
python -m cProfile -o /tmp/prof.bin bench.py --config $CFG --launch eager --steps $STEPS --warmup 30 > /dev/null 2>&1
python - >> $O/${TAG}_${CFG}.txt <<PY
import pstats, io
p = pstats.Stats('/tmp/prof.bin')
rows = []
for (f, l, n), (cc, nc, tt, ct, callers) in p.stats.items():
    if nc >= $STEPS:
        rows.append((tt, ct, nc, f'{f.split("/")[-1]}:{l}({n})'))
rows.sort(reverse=True)
print('functions called at least once per step, by own time (us per step = total / steps incl. warmup+profile windows ~ %d calls of step)' % max(r[2] for r in rows if 'step' in r[3]))
for tt, ct, nc, name in rows[:70]:
    print(f'{nc:8d} own {tt*1e3:8.2f} ms  cum {ct*1e3:8.2f} ms   {name}')
import timeit, torch
print('torch.cuda.is_available() us', timeit.timeit(torch.cuda.is_available, number=2000) / 2000 * 1e6)
print('torch.cuda.current_device() us', timeit.timeit(torch.cuda.current_device, number=2000) / 2000 * 1e6)
x = torch.zeros(4, device='cuda')
print('torch.empty us', timeit.timeit(lambda: torch.empty(64, device='cuda'), number=2000) / 2000 * 1e6)
print('x.detach() us', timeit.timeit(x.detach, number=2000) / 2000 * 1e6)
print('torch.cuda.current_stream() us', timeit.timeit(torch.cuda.current_stream, number=2000) / 2000 * 1e6)
PY
cat $O/${TAG}_${CFG}.txt | cut -c1-200
