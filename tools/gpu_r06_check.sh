#!/bin/bash
# round 6: kernel times at C2, RSLM initialiser, the other configurations, determinism + RSLM + parity tests, bench line.  usage: tools/gpu_r06_check.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r06x}; O=/root/repo/gpurun_out; mkdir -p $O
{
echo "# tools/tune.py at C2"; timeout 600 python tools/tune.py 2>&1 | cut -c1-150
echo "# tools/bounds_timing.py"; timeout 300 python tools/bounds_timing.py 2>&1 | tail -4
echo "# tools/rslm_parts_timing.py"; timeout 300 python tools/rslm_parts_timing.py 2>&1 | grep "^{" | cut -c1-260
echo "# tools/bench_configs.py"; timeout 600 python tools/bench_configs.py 2>&1 | grep "^{"
} > $O/${TAG}_kernels.txt 2>&1
cat $O/${TAG}_kernels.txt
(timeout 900 python -m pytest tests/test_rslm.py tests/test_api_dropin.py tests/test_determinism_gpu.py tests/test_baseline_shapes_gpu.py tests/test_amis.py tests/test_erratum_gpu.py tests/test_graph_rng.py -m gpu -q -x 2>&1 | tail -6) > $O/${TAG}_pytest_subset.log; cat $O/${TAG}_pytest_subset.log
(timeout 300 python bench.py 2>&1 | grep "^{" | tail -1) > $O/${TAG}_bench.json; python -c "
import json; d=json.load(open('$O/${TAG}_bench.json')); print(d['value'], d['ms_per_step'], d['kernel_ms']); print(d['roofline']['frac'], d['roofline']['clocks']); print(d['roofline']['large']['frac'], d['roofline']['large']['clocks'])"
