#!/bin/bash
# quick C4 loop: 4-DoF GPU tests, the Det step (replayed + eager), kernel stage times.   usage: tools/gpu_c4_check.sh [tag]
T=${1:-c4}
mkdir -p gpurun_out
O=gpurun_out
(timeout 900 python -m pytest tests/test_amis.py tests/test_graph_rng.py tests/test_api_dropin.py tests/test_rslm.py tests/test_fused_and_limits.py -m gpu -q -x 2>&1 | tail -3)
(RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 300 python bench.py --config C4 --steps 300 --warmup 10 --no-cpu-baseline 2>&1 | grep "^{" | tail -1) > $O/${T}_bench_C4.json
python - <<PY
import json
r = json.load(open('$O/${T}_bench_C4.json'))
print('C4 replayed', r['ms_per_step'], 'ms', r['value'], 'eager', r.get('eager', {}).get('ms_per_step'), 'kernel_ms', r['kernel_ms'])
PY
(timeout 300 python tools/graph_step.py C3 C4 2>&1 | grep "^{")
