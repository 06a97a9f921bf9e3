#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -m2 gfx
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > gpurun_out/smoke.log
cat gpurun_out/smoke.log
(timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -3) > gpurun_out/bench.log
cat gpurun_out/bench.log
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof -o bench -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -3)
cd /root/repo
ls -R gpurun_out/prof | head -20
