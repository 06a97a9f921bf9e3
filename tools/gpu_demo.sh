#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python demo/fit_identity.py --iters 400 2>&1 | grep -v Warning | tail -14) | tee gpurun_out/demo_fit_identity.log
