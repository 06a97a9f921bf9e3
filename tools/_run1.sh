cd /root/repo
mkdir -p gpurun_out; O=/root/repo/gpurun_out; export TMPDIR=/tmp
bash tools/gpu_r05_evidence.sh r05c t
for c in C1 C3 C3-train; do
  (timeout 600 python bench.py --config $c --no-cpu-baseline 2>&1 | grep "^{" | tail -1) > $O/r05c_bench_${c}.json
  python -c "import json,sys; d=json.load(open('$O/r05c_bench_${c}.json')); print('$c', d['ms_per_step'], d.get('kernel_ms'))"
done
for t in "" "no_denorm_fold"; do
  (EPROPNP_TUNE=$t timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --config C4 --steps 200 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" | tail -1) > $O/r05c_bench_C4_torchrun1_$t.json
  python -c "import json,sys; d=json.load(open('$O/r05c_bench_C4_torchrun1_$t.json')); print('C4 [$t]', d['ms_per_step'], d.get('kernel_ms'))"
done
(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --config C4 --launch eager --steps 200 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" | tail -1) > $O/r05c_bench_C4_eager_torchrun1.json
python -c "import json,sys; d=json.load(open('$O/r05c_bench_C4_eager_torchrun1.json')); print('C4 eager', d['ms_per_step'])"
(timeout 600 python tools/graph_step.py C3 C4 2>&1 | grep "^{") > $O/r05c_hipgraph_step.txt
cat $O/r05c_hipgraph_step.txt
bash tools/gpu_step_sequences.sh r05c C1 C3 C3-train C4 > /dev/null 2>&1
for c in C3 C4; do head -20 $O/r05c_${c}_step_sequence.txt | cut -c1-110; done
tail -1 $O/r05c_C3-train_step_sequence.txt $O/r05c_C1_step_sequence.txt
