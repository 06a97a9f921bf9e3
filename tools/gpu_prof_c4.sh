#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof
(timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o c4 -- python /root/repo/tools/profile_c4.py 2>&1 | tail -1)
python /root/repo/tools/rocprof_summary.py /tmp/prof/c4_results.db | cut -c1-180 > /root/repo/gpurun_out/c4_kernel_stats.txt
head -30 /root/repo/gpurun_out/c4_kernel_stats.txt
