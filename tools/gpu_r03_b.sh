#!/bin/bash
O=/root/repo/gpurun_out; mkdir -p $O
T="python -X faulthandler -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29551"
for mode in eager graph; do
  (timeout 600 $T bench.py --gpus 1 --config C4 --steps 200 --warmup 20 --launch $mode > $O/r03c_C4_$mode.log 2>&1)
  grep -E "^\{" $O/r03c_C4_$mode.log | cut -c1-300; grep -E "^\{" $O/r03c_C4_$mode.log | grep -o '"kernel_ms.*' | cut -c1-1200
  grep -B2 -A25 "Fatal Python\|Traceback" $O/r03c_C4_$mode.log | head -40
done
(timeout 900 python tools/tune.py 2>&1) > $O/r03c_tune.txt
cat $O/r03c_tune.txt
