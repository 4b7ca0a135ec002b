python -m pytest tests/test_gpu_pbr.py tests/test_gpu_headline_call.py tests/test_gpu_train.py -q -x 2>&1 | tail -3
for v in 0 1; do
  if [ $v = 1 ]; then export IA_ENV_GRAD_UNBINNED=1; fi
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-config2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unbinned $v', d['ms_per_step'], d['value'], d['kernel_breakdown_ms_per_step'].get('ia_pbr_shade_bwd'))"
done
