#!/bin/bash
# Round-3 first GPU pass: GPU tests, C2 bench line (IC-cold single sweep), Det step (C4) eager and graph-replayed under a
# one-rank nccl group in both launch forms.   usage: tools/gpu_r03_a.sh [tag]
TAG=${1:-r03a}
mkdir -p gpurun_out
export TMPDIR=/tmp
O=/root/repo/gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $O/${TAG}_pytest_gpu.log
tail -4 $O/${TAG}_pytest_gpu.log
(timeout 900 python bench.py --no-hipgraph 2>&1 | tail -1) > $O/${TAG}_bench.json
cut -c1-2500 $O/${TAG}_bench.json
for mode in eager graph; do
  (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --config C4 --steps 200 --warmup 20 --launch $mode 2>&1 | tail -3) > $O/${TAG}_bench_C4_$mode.json
  cut -c1-3000 $O/${TAG}_bench_C4_$mode.json
done
(BENCH_SELF_LAUNCH=1 timeout 600 python bench.py --gpus 1 --config C4 --steps 200 --warmup 20 2>&1 | tail -2) > $O/${TAG}_bench_C4_plain.json
cut -c1-600 $O/${TAG}_bench_C4_plain.json
