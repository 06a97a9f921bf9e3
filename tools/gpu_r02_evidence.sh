#!/bin/bash
# Round-2 evidence run on the GPU box: GPU tests (+ parity report), smoke, bench (C2) with HIP-event kernel times,
# rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE passes -> traffic json, C4 / C5 harness lines.
# usage: tools/gpu_r02_evidence.sh [tag]      (outputs under gpurun_out/<tag>_*)
TAG=${1:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
O=/root/repo/gpurun_out
rm -f $O/${TAG}_parity.jsonl
(EPROPNP_PARITY_REPORT=$O/${TAG}_parity.jsonl timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > $O/${TAG}_pytest_gpu.log
tail -4 $O/${TAG}_pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1) > $O/${TAG}_smoke.log
cat $O/${TAG}_smoke.log
(timeout 900 python bench.py 2>&1 | tail -1) > $O/${TAG}_bench.json
cut -c1-1500 $O/${TAG}_bench.json
cd /tmp; rm -rf /tmp/prof
(timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-hipgraph 2>&1 | tail -1) > $O/${TAG}_bench_under_rocprof.json
python /root/repo/tools/rocprof_summary.py /tmp/prof/bench_results.db | cut -c1-190 > $O/${TAG}_kernel_stats.txt
head -14 $O/${TAG}_kernel_stats.txt
B="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hipgraph"
i=0
for set in "FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "WRITE_SIZE SQ_BUSY_CU_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  (timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc$i -o p -- $B 2>&1 | grep -E "rror|ailed" | head -3)
  python /root/repo/tools/pmc_csv_summary.py /tmp/pmc$i > $O/${TAG}_pmc_pass$i.txt 2>&1
done
python /root/repo/tools/pmc_traffic.py /tmp/pmc1 /tmp/pmc2 C2:B4096:N512:S512:K4:L3 $O/${TAG}_pmc_traffic.json
cd /root/repo
(timeout 900 python tools/bench_configs.py 2>&1 | grep "^{") > $O/${TAG}_other_configs.json
cat $O/${TAG}_other_configs.json
(python tools/host_phases.py c4 2>&1 | tail -1; python tools/host_phases.py c3 2>&1 | tail -1) > $O/${TAG}_host_phases.txt
cat $O/${TAG}_host_phases.txt
(EPROPNP_NO_FUSED_FORWARD=1 python tools/host_phases.py c4 2>&1 | tail -1 | sed 's/^/composite path (EPROPNP_NO_FUSED_FORWARD=1) /') >> $O/${TAG}_host_phases.txt
(TUNE_VARIANTS="bwd_exact:EPROPNP_BWD_DROP=0" python tools/tune.py 2>&1) > $O/${TAG}_tune.txt
cat $O/${TAG}_tune.txt
for c in C4 C5; do
  (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/${TAG}_bench_$c.json
  cut -c1-900 $O/${TAG}_bench_$c.json
done
