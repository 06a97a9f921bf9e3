#!/bin/bash
# rocprofv3 kernel sequences of ONE step of the launch-bound BASELINE configurations, as bench.py runs them:
#   <tag>_<cfg>_step_sequence.txt        one step replayed from the hipGraph (the timed region of `bench.py --config <cfg>`)
#   <tag>_<cfg>_step_sequence_eager.txt  the same step launched eagerly (bench.py's `eager` leg)
#   <tag>_<cfg>_kernel_stats.txt         per-kernel totals over the whole run
# usage: tools/gpu_step_sequences.sh [tag] [configs...]      (C4 runs under a one-rank nccl group, as the driver's N=1 line does not)
TAG=${1:-r05}; shift
CFGS=${@:-C1 C3 C3-train C4}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
W=10; STEPS=200
# dispatch order inside bench.py: 40 spin-up + W warm-up + 10 profiled eager steps + 3 side-stream steps, then 3 + STEPS replays,
# then (non-strong) W + STEPS eager steps  ->  step number 63 + 100 sits in the middle of the replays, -100 in the eager leg
REPLAY_IDX=$((40 + W + 10 + 3 + 3 + 100))
for c in $CFGS; do
  cd /tmp; rm -rf /tmp/prof_$c
  EAGER_IDX=-100
  if [ $c == C4 ]; then EAGER_IDX=$((-100 - W - STEPS)); fi      # C4 ends with W + STEPS replays of the step WITHOUT the exchange (A/B leg)
  if [ $c == C4 ]; then export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517; else unset RANK LOCAL_RANK WORLD_SIZE; fi
  (timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o s -- python $R/bench.py --config $c --steps $STEPS --warmup $W --no-cpu-baseline 2>&1 | grep '^{' | tail -1) > $O/${TAG}_bench_${c}_under_rocprof.json
  python $R/tools/rocprof_summary.py /tmp/prof_$c/s_results.db | cut -c1-200 > $O/${TAG}_${c}_kernel_stats.txt
  {
    echo "# $TAG: rocprofv3 kernel sequence of one REPLAYED step of \`bench.py --config $c\` (step $REPLAY_IDX of the run; tools/gpu_step_sequences.sh)."
    echo "# Durations under the profiler carry ~2.5 us of per-dispatch serialisation each; the un-profiled step time is in ${TAG}_bench_${c}.json."
    python $R/tools/rocprof_sequence.py /tmp/prof_$c/s_results.db adaptive_delta_kernel $REPLAY_IDX | cut -c1-170
  } > $O/${TAG}_${c}_step_sequence.txt
  {
    echo "# $TAG: the same step launched EAGERLY (bench.py's eager leg), same run."
    python $R/tools/rocprof_sequence.py /tmp/prof_$c/s_results.db adaptive_delta_kernel $EAGER_IDX | cut -c1-170
  } > $O/${TAG}_${c}_step_sequence_eager.txt
  cat $O/${TAG}_${c}_step_sequence.txt
  tail -1 $O/${TAG}_${c}_step_sequence_eager.txt
  cd $R
done
