#!/bin/bash
# host time of an eagerly launched step by segment.  usage: tools/gpu_host_segments.sh <tag> [configs...]
export TMPDIR=/tmp
TAG=${1:-host}; shift; O=/root/repo/gpurun_out; mkdir -p $O
for CFG in ${@:-C4 C3-train}; do
  for i in 1 2; do
    BENCH_HOST_SEGMENTS=1 python bench.py --config $CFG --launch eager --steps 600 --warmup 30 2> /tmp/err.txt | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$CFG eager ms_per_step', d['ms_per_step'])"
    grep host_segments /tmp/err.txt
  done
done > $O/${TAG}_segments.txt 2>&1
cat $O/${TAG}_segments.txt
