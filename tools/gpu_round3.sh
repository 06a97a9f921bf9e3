#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/smoke.log
cat gpurun_out/smoke.log
(timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -1) > gpurun_out/bench.log
cat gpurun_out/bench.log
(timeout 1200 python tools/tune.py 2>&1) | tee gpurun_out/tune.log
