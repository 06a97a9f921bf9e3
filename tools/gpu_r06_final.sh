#!/bin/bash
# round 6, after the header clean-up: the whole GPU suite, the smoke, the bench line and the C2 kernel times on ONE box.  usage: tools/gpu_r06_final.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r06f}; O=/root/repo/gpurun_out; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/${TAG}_pytest_gpu.log; cat $O/${TAG}_pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $O/${TAG}_smoke.log; cat $O/${TAG}_smoke.log
(timeout 400 python bench.py 2>&1 | grep "^{" | tail -1) > $O/${TAG}_bench.json; python -c "
import json; d=json.load(open('$O/${TAG}_bench.json')); print(d['value'], d['ms_per_step'], d['kernel_ms']); print(d['roofline']['frac'], d['roofline']['clocks'])"
{ echo "# tools/tune.py at C2"; timeout 600 python tools/tune.py 2>&1 | cut -c1-150; } > $O/${TAG}_kernels.txt 2>&1; cat $O/${TAG}_kernels.txt
