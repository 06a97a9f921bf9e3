cd /root/repo
mkdir -p gpurun_out; O=/root/repo/gpurun_out; export TMPDIR=/tmp
bash tools/gpu_r05_evidence.sh r05b t
for c in C3 C3-train; do
  (timeout 600 python bench.py --config $c --no-cpu-baseline 2>&1 | grep "^{" | tail -1) > $O/r05b_bench_${c}.json
  python -c "import json,sys; d=json.load(open('$O/r05b_bench_${c}.json')); print('$c', d['ms_per_step'], d.get('kernel_ms'))"
done
for t in "" "no_loss_ticket" "no_denorm_fold"; do
  (EPROPNP_TUNE=$t timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --config C4 --steps 200 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" | tail -1) > $O/r05b_bench_C4_torchrun1_$t.json
  python -c "import json,sys; d=json.load(open('$O/r05b_bench_C4_torchrun1_$t.json')); print('C4 [$t]', d['ms_per_step'], d.get('kernel_ms'))"
done
(timeout 600 python tools/graph_step.py C3 C4 2>&1 | grep "^{") > $O/r05b_hipgraph_step.txt
cat $O/r05b_hipgraph_step.txt
bash tools/gpu_step_sequences.sh r05b C3 C3-train C4 > /dev/null 2>&1
for c in C3 C3-train C4; do head -24 $O/r05b_${c}_step_sequence.txt | cut -c1-110; tail -1 $O/r05b_${c}_step_sequence.txt; done
(timeout 900 python bench.py 2>&1 | grep "^{" | tail -1) > $O/r05b_bench.json
python -c "import json; d=json.load(open('$O/r05b_bench.json')); print('C2', d['ms_per_step'], d['value'], d['kernel_ms'])"
