#!/bin/bash
mkdir -p gpurun_out
(timeout 1500 python tools/tune.py 2>&1) | tee gpurun_out/tune.log
