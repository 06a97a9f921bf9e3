# A/B of two builds of the library on the roofline kernel: the bench's own IC-cold single-sweep measurement, same box
for rep in 1 2 3; do
for lib in "" epro-pnp_amd/lib/variants/head/libepropnp_hip.so; do
  EPROPNP_LIB=$lib python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-hipgraph 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('${lib:-new}'.split('/')[-2] if '/' in '${lib:-new}' else 'new', d['ms_per_step'], r['launch_ms'], r['launch_ms_median'], r['frac'])"
done; done
