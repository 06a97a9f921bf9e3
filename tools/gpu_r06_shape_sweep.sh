#!/bin/bash
# round 6: launch shapes / projection flavours of the two AMIS kernels at C2 (occupancy against per-pair overhead), the RSLM initialiser
# with the new index draw, and the Det step.  -> gpurun_out/r06_shape_sweep.txt
export TMPDIR=/tmp
O=/root/repo/gpurun_out; mkdir -p $O
{
echo "# tools/tune.py at C2 (4096 x 512, S = 512): base = this tree; minw4 = backward compiled for four waves per SIMD (128 VGPRs)"
TUNE_VARIANTS="b_w4t2:bwd_mfma=4,2|b_w8t2:bwd_mfma=8,2|b_f32_w4t4:EPROPNP_BWD_PROJ=f32|b_f32_w4t2:EPROPNP_BWD_PROJ=f32+bwd_mfma=4,2|f_f32:EPROPNP_FWD_PROJ=f32|f_w8t4:fwd_mfma=8,4|f_f32_w8t4:EPROPNP_FWD_PROJ=f32+fwd_mfma=8,4" timeout 1200 python tools/tune.py 2>&1 | cut -c1-150
echo "# RSLM initialiser (tools/rslm_parts_timing.py)"
timeout 300 python tools/rslm_parts_timing.py 2>&1 | tail -8
echo "# Det step, LineMOD steps (tools/bench_configs.py)"
timeout 600 python tools/bench_configs.py 2>&1 | grep "^{"
} > $O/r06_shape_sweep.txt 2>&1
cat $O/r06_shape_sweep.txt
(timeout 600 python -m pytest tests/test_rslm.py tests/test_api_dropin.py tests/test_baseline_shapes_gpu.py -m gpu -q -x 2>&1 | tail -5) > $O/r06c_pytest_rslm.log; cat $O/r06c_pytest_rslm.log
