#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
(timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -1) > gpurun_out/bench.log
cat gpurun_out/bench.log
(timeout 600 python tools/tune.py 2>&1) | tee gpurun_out/tune.log
