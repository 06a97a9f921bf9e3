#!/bin/bash
# Round-4 GPU call B: A/B of the forward's refit (transposed reductions, L^-1 hand-over, paired fits) against the round-3
# library, the whole GPU suite without -x, the contract line.   usage: tools/gpu_r04_b.sh [parts: u t b]
PARTS=${1:-utb}
mkdir -p gpurun_out
export TMPDIR=/tmp
O=/root/repo/gpurun_out
T=r04b
if [[ $PARTS == *u* ]]; then
  (timeout 900 python tools/tune.py 2>&1) > $O/${T}_tune_refit.txt
  (TUNE_B=32 timeout 600 python tools/tune.py 2>&1) >> $O/${T}_tune_refit.txt
  (TUNE_B=600 TUNE_N=128 TUNE_S=128 timeout 600 python tools/tune.py 2>&1) >> $O/${T}_tune_refit.txt
  cat $O/${T}_tune_refit.txt
fi
if [[ $PARTS == *t* ]]; then
  rm -f $O/${T}_parity.jsonl
  (EPROPNP_PARITY_REPORT=$O/${T}_parity.jsonl timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60) > $O/${T}_pytest_gpu.log
  tail -12 $O/${T}_pytest_gpu.log
  (timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1) > $O/${T}_smoke.log
  cat $O/${T}_smoke.log
fi
if [[ $PARTS == *b* ]]; then
  (timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1) > $O/${T}_bench.json
  cut -c1-400 $O/${T}_bench.json; grep -o '"kernel_ms".*' $O/${T}_bench.json | cut -c1-400
fi
