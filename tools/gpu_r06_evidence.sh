#!/bin/bash
# Round-6 evidence run on the GPU box.  usage: tools/gpu_r06_evidence.sh [tag] [parts]   parts: any of t(ests) b(ench) p(mc) o(ther) c(onfigs + C5)
TAG=${1:-r06}; PARTS=${2:-tbpoc}
mkdir -p gpurun_out
export TMPDIR=/tmp
O=/root/repo/gpurun_out
if [[ $PARTS == *t* ]]; then
  rm -f $O/${TAG}_parity.jsonl
  (EPROPNP_PARITY_REPORT=$O/${TAG}_parity.jsonl timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60) > $O/${TAG}_pytest_gpu.log
  tail -4 $O/${TAG}_pytest_gpu.log
  (timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1) > $O/${TAG}_smoke.log
  cat $O/${TAG}_smoke.log
fi
if [[ $PARTS == *b* ]]; then
  (timeout 900 python bench.py 2>&1 | grep "^{" | tail -1) > $O/${TAG}_bench.json
  cut -c1-1800 $O/${TAG}_bench.json
  cd /tmp; rm -rf /tmp/prof
  (timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-hipgraph 2>&1 | grep "^{" | tail -1) > $O/${TAG}_bench_under_rocprof.json
  python /root/repo/tools/rocprof_summary.py /tmp/prof/bench_results.db | cut -c1-190 > $O/${TAG}_kernel_stats.txt
  head -14 $O/${TAG}_kernel_stats.txt
  cd /root/repo
  for c in C4 C5; do
    (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --config $c --steps 200 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" | tail -1) > $O/${TAG}_bench_${c}_torchrun1.json
    cut -c1-400 $O/${TAG}_bench_${c}_torchrun1.json; grep -o '"collective.*' $O/${TAG}_bench_${c}_torchrun1.json | cut -c1-600
  done
  (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --config C4 --launch eager --steps 200 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" | tail -1) > $O/${TAG}_bench_C4_eager_torchrun1.json
  cut -c1-300 $O/${TAG}_bench_C4_eager_torchrun1.json; grep -o '"collective.*' $O/${TAG}_bench_C4_eager_torchrun1.json | cut -c1-600
  (BENCH_SELF_LAUNCH=1 timeout 600 python bench.py --gpus 1 --config C2 --steps 50 --warmup 5 --no-cpu-baseline --no-hipgraph 2>&1 | grep "^{" | tail -1) > $O/${TAG}_bench_C2_selflaunch.json
  cut -c1-700 $O/${TAG}_bench_C2_selflaunch.json
fi
if [[ $PARTS == *p* ]]; then
  cd /tmp
  B="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hipgraph --no-large-sweep"      # every normal_equations_kernel dispatch has the C2 launch size
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT" \
             "FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "WRITE_SIZE SQ_BUSY_CU_CYCLES"; do
    i=$((i+1)); rm -rf /tmp/pmc$i
    (timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc$i -o p -- $B 2>&1 | grep -E "rror|ailed" | head -3)
    python /root/repo/tools/pmc_csv_summary.py /tmp/pmc$i > $O/${TAG}_pmc_pass$i.txt 2>&1
  done
  python /root/repo/tools/pmc_traffic.py /tmp/pmc3 /tmp/pmc4 C2:B4096:N512:S512:K4:L3 $O/${TAG}_pmc_traffic.json
  cat $O/${TAG}_pmc_pass1.txt $O/${TAG}_pmc_pass2.txt | grep -A9 "amis_backward_mfma\|amis_forward_mfma" | head -50
  cd /root/repo
fi
if [[ $PARTS == *o* ]]; then
  (timeout 900 python tools/bench_configs.py 2>&1 | grep "^{") > $O/${TAG}_other_configs.json
  cat $O/${TAG}_other_configs.json
  (timeout 600 python tools/graph_step.py C3 C4 2>&1 | grep "^{") > $O/${TAG}_hipgraph_step.txt
  cat $O/${TAG}_hipgraph_step.txt
fi
if [[ $PARTS == *c* ]]; then
  # driver-reachable configs other than C2, and the C5 shard (8192 x 2048, IC-cold by construction): kernel trace + FETCH/WRITE
  for c in C1 C3 C3-train; do
    (timeout 600 python bench.py --config $c --no-cpu-baseline 2>&1 | grep "^{" | tail -1) > $O/${TAG}_bench_${c}.json
    cut -c1-500 $O/${TAG}_bench_${c}.json
  done
  bash tools/gpu_step_sequences.sh $TAG C1 C3 C3-train C4 > /dev/null 2>&1
  tail -1 $O/${TAG}_C4_step_sequence.txt
  cd /tmp
  B="python /root/repo/bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-hipgraph"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf /tmp/pmc5_$i
    (timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc5_$i -o p -- $B 2>&1 | grep -E "rror|ailed" | head -3)
  done
  # (without the `p` part in the same call the C2 entry is carried over from the committed profile of this round, not re-measured)
  [ -f $O/${TAG}_pmc_traffic.json ] || cp /root/repo/profiles/${TAG}_pmc_traffic.json $O/${TAG}_pmc_traffic.json 2>/dev/null || true
  python /root/repo/tools/pmc_traffic.py /tmp/pmc5_1 /tmp/pmc5_2 C5:B8192:N2048:S1024:K4:L3 $O/${TAG}_pmc_traffic.json
  rm -rf /tmp/prof5
  (timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof5 -o bench -- $B 2>&1 | tail -1) > $O/${TAG}_bench_C5_under_rocprof.json
  python /root/repo/tools/rocprof_summary.py /tmp/prof5/bench_results.db | cut -c1-190 > $O/${TAG}_c5_kernel_stats.txt
  head -8 $O/${TAG}_c5_kernel_stats.txt
  cd /root/repo
fi
