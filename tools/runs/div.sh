python -m pytest tests/test_gpu_parity.py tests/test_gpu_spec_search.py tests/test_gpu_train.py -q -x 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['kernel_breakdown_ms_per_step'].get('ia_fuse_broyden_spec_rows'), d['roofline']['frac'])"
