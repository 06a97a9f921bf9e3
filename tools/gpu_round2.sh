#!/bin/bash
# GPU visit 2: PMC counters for the three hot kernels + shape/occupancy tuning sweep.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > /root/repo/gpurun_out/counters_list.txt 2>&1
B="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /root/repo/gpurun_out/pmc1 -o p -- $B 2>&1 | tail -2)
(timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM -d /root/repo/gpurun_out/pmc2 -o p -- $B 2>&1 | tail -2)
(timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d /root/repo/gpurun_out/pmc3 -o p -- $B 2>&1 | tail -2)
cd /root/repo
(timeout 1200 python tools/tune.py 2>&1) | tee gpurun_out/tune.log
