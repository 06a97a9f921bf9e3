# A/B of the Det step (bench.py --config C4, one-rank nccl group): fused reduced loss vs the composite PyTorch statement
for rep in 1 2; do for f in 1 0; do
  for launch in graph eager; do
  EPROPNP_LOSS_FUSED=$f RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --config C4 --steps 300 --warmup 20 --no-cpu-baseline --launch $launch 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('fused=$f', '$launch', d['ms_per_step'], d['value'], d.get('loss'))"
  done
done; done
