#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
(timeout 900 python tools/bench_configs.py 2>&1 | grep -v Warning) | tee gpurun_out/bench_configs.log
cd /tmp; rm -rf /tmp/prof
(timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o c4 -- python /root/repo/tools/profile_c4.py 2>&1 | tail -1)
python /root/repo/tools/rocprof_summary.py /tmp/prof/c4_results.db | cut -c1-180 > /root/repo/gpurun_out/c4_kernel_stats.txt
head -16 /root/repo/gpurun_out/c4_kernel_stats.txt
