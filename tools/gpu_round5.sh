#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > gpurun_out/smoke.log
cat gpurun_out/smoke.log
(timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -1) > gpurun_out/bench.log
cat gpurun_out/bench.log
cd /tmp
rm -rf /tmp/prof
(timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > /root/repo/gpurun_out/bench_under_rocprof.log
python /root/repo/tools/rocprof_summary.py /tmp/prof/bench_results.db | cut -c1-190 > /root/repo/gpurun_out/kernel_stats.txt
head -14 /root/repo/gpurun_out/kernel_stats.txt
