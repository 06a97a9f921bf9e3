#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
