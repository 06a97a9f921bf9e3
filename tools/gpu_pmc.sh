#!/bin/bash
# PMC counters for the hot kernels; only text summaries are kept (the rocpd databases are too big to ship back).
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>&1 | grep -E "^\s*(Name|gpu-agent)|SQ_|TCC_|GRBM|FETCH|WRITE" | head -400 > /root/repo/gpurun_out/counters_list.txt
B="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS" \
           "WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  (timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc$i -o p -- $B 2>&1 | grep -v simple_timer | tail -4)
  python /root/repo/tools/pmc_csv_summary.py /tmp/pmc$i > /root/repo/gpurun_out/pmc_pass$i.txt 2>&1; ls -R /tmp/pmc$i | head -8
done
cd /root/repo
