#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python tools/bench_configs.py 2>&1 | grep -v Warning) | tee gpurun_out/bench_configs.log
