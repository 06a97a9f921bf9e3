#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
B="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "WRITE_SIZE SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  (timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc$i -o p -- $B 2>&1 | grep -E "rror|ailed" | head -3)
  python /root/repo/tools/pmc_csv_summary.py /tmp/pmc$i > /root/repo/gpurun_out/pmc_pass$i.txt 2>&1
done
cd /root/repo
cat gpurun_out/pmc_pass*.txt
