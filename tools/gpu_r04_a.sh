#!/bin/bash
# Round-4 GPU call A: the whole GPU suite with the parity report (C3-dense records + fp64 tie-break), smoke, the contract line,
# the new bench configs, and the s_setprio tuning variants of the forward.   usage: tools/gpu_r04_a.sh [parts: t b c u]
PARTS=${1:-tbcu}
mkdir -p gpurun_out
export TMPDIR=/tmp
O=/root/repo/gpurun_out
T=r04a
if [[ $PARTS == *t* ]]; then
  rm -f $O/${T}_parity.jsonl
  (EPROPNP_PARITY_REPORT=$O/${T}_parity.jsonl timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -40) > $O/${T}_pytest_gpu.log
  tail -6 $O/${T}_pytest_gpu.log
  (timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1) > $O/${T}_smoke.log
  cat $O/${T}_smoke.log
fi
if [[ $PARTS == *b* ]]; then
  (timeout 900 python bench.py 2>&1 | tail -1) > $O/${T}_bench.json
  cut -c1-2500 $O/${T}_bench.json
fi
if [[ $PARTS == *c* ]]; then
  for c in C1 C3 C3-train; do
    (timeout 600 python bench.py --config $c 2>&1 | tail -1) > $O/${T}_bench_${c}.json
    cut -c1-900 $O/${T}_bench_${c}.json; grep -o '"eager".*' $O/${T}_bench_${c}.json | cut -c1-200
  done
fi
if [[ $PARTS == *u* ]]; then
  (timeout 900 python tools/tune.py 2>&1) > $O/${T}_tune_prio.txt
  (TUNE_B=32 timeout 600 python tools/tune.py 2>&1) >> $O/${T}_tune_prio.txt
  cat $O/${T}_tune_prio.txt
fi
