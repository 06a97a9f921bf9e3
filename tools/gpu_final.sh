#!/bin/bash
# Final evidence run: GPU tests, smoke, bench (HIP-event kernel times), rocprofv3 kernel stats of the same command,
# PMC passes (text summaries only), other configurations.
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1) > gpurun_out/smoke.log
cat gpurun_out/smoke.log
(timeout 600 python bench.py 2>&1 | tail -1) > gpurun_out/bench.log
cat gpurun_out/bench.log
cd /tmp; rm -rf /tmp/prof
(timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-hipgraph 2>&1 | tail -1) > /root/repo/gpurun_out/bench_under_rocprof.log
python /root/repo/tools/rocprof_summary.py /tmp/prof/bench_results.db | cut -c1-190 > /root/repo/gpurun_out/kernel_stats.txt
head -12 /root/repo/gpurun_out/kernel_stats.txt
B="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hipgraph"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "WRITE_SIZE SQ_BUSY_CU_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  (timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc$i -o p -- $B 2>&1 | grep -E "rror|ailed" | head -3)
  python /root/repo/tools/pmc_csv_summary.py /tmp/pmc$i > /root/repo/gpurun_out/pmc_pass$i.txt 2>&1
done
cd /root/repo
(timeout 900 python tools/bench_configs.py 2>&1 | grep -v Warning) > gpurun_out/bench_configs.log
cat gpurun_out/bench_configs.log
grep -A4 "lm_solve_kernel" gpurun_out/pmc_pass3.txt gpurun_out/pmc_pass4.txt | head -20
