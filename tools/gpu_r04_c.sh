#!/bin/bash
# Round-4 GPU call C: forward phase shares (tuning build), backward Huber-weight variant A/B, full suite, contract line.
PARTS=${1:-utb}
mkdir -p gpurun_out
export TMPDIR=/tmp
O=/root/repo/gpurun_out
T=r04c
if [[ $PARTS == *u* ]]; then
  (TUNE_VARIANTS="norefit:EPROPNP_ABLATE=2;sweeponly:EPROPNP_ABLATE=38;nosweep:EPROPNP_ABLATE=1" timeout 900 python tools/tune.py 2>&1) > $O/${T}_tune.txt
  cat $O/${T}_tune.txt
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
