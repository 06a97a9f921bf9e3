#!/bin/bash
# rocprofv3 kernel trace of the Det step as bench.py runs it (one-rank nccl group, whole step replayed from a hipGraph)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp; rm -rf /tmp/prof
(RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o c4 -- python $R/bench.py --config C4 --steps 200 --warmup 10 --no-cpu-baseline 2>&1 | tail -2)
python $R/tools/rocprof_summary.py /tmp/prof/c4_results.db | cut -c1-200 > $R/gpurun_out/c4_bench_kernel_stats.txt
python $R/tools/rocprof_sequence.py /tmp/prof/c4_results.db adaptive_delta_kernel -150 | cut -c1-150 > $R/gpurun_out/c4_bench_sequence.txt; cat $R/gpurun_out/c4_bench_sequence.txt
