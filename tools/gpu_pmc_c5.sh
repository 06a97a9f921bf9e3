#!/bin/bash
# rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) + kernel trace of the C5 stress shard
# (8192 objects x 2048 points x 1024 samples): HBM bytes per launch -> profiles/r02_pmc_traffic.json (key C5:...).
export TMPDIR=/tmp
O=/root/repo/gpurun_out
B="python /root/repo/bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-hipgraph"
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmc5_$i
  (timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc5_$i -o p -- $B 2>&1 | grep -E "rror|ailed" | head -3)
done
cp /root/repo/profiles/r02_pmc_traffic.json $O/r02_pmc_traffic_with_c5.json
python /root/repo/tools/pmc_traffic.py /tmp/pmc5_1 /tmp/pmc5_2 C5:B8192:N2048:S1024:K4:L3 $O/r02_pmc_traffic_with_c5.json
rm -rf /tmp/prof5
(timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof5 -o bench -- $B 2>&1 | tail -1) > $O/r02_bench_C5_under_rocprof.json
python /root/repo/tools/rocprof_summary.py /tmp/prof5/bench_results.db | cut -c1-190 > $O/r02_c5_kernel_stats.txt
head -8 $O/r02_c5_kernel_stats.txt
