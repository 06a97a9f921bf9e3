#!/bin/bash
# a long version of tests/test_erratum_gpu.py's whole-step soak (every output and gradient of every launch bit-identical to the first,
# alone and beside a bf16 GEMM stream), then the GPU suite and the bench line on the same box.  usage: tools/gpu_soak.sh <tag> [C2 launches] [C5 launches]
export TMPDIR=/tmp
TAG=${1:-soak}; N2=${2:-2000}; N5=${3:-300}; O=/root/repo/gpurun_out; mkdir -p $O
cd /root/repo
python - <<PY 2>&1 | grep -v Warning | tee $O/${TAG}_soak.txt
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'epro-pnp_amd'); sys.path.insert(0, '.')
import test_erratum_gpu as t
for name, B, N, S, n in (('C2', 4096, 512, 512, $N2), ('C5-shard', 8192, 2048, 1024, $N5)):
    t0 = time.time()
    t.test_whole_step_repeats_bit_for_bit_alone_and_beside_a_bf16_gemm(name, B, N, S, n)
    print(f'{name}: 2 x {n} whole steps (forward + loss + backward), alone and beside a bf16 GEMM stream: every output and gradient bit-identical to the first launch ({time.time() - t0:.0f} s)')
PY
(timeout 2000 python -m pytest tests -m gpu -q 2>&1 | tail -3) | tee $O/${TAG}_pytest.log
(timeout 400 python bench.py 2>&1 | grep "^{" | tail -1) > $O/${TAG}_bench.json; python -c "
import json; d=json.load(open('$O/${TAG}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['large']['frac'])" | tee -a $O/${TAG}_soak.txt
