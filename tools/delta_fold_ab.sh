# A/B: the Huber threshold's gradient added to grad_w2d by the backward kernel (default) vs by autograd (EPROPNP_DELTA_FOLD=0)
for rep in 1 2; do for f in 1 0; do
  EPROPNP_DELTA_FOLD=$f python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hipgraph 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('C2 fold=$f', d['ms_per_step'], d['value'], d['kernel_ms']['amis_backward'], d['loss'])"
  EPROPNP_DELTA_FOLD=$f RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --config C4 --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('C4 fold=$f', d['ms_per_step'], d['value'], d['loss'])"
done; done
