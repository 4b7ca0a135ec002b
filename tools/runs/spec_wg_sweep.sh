mkdir -p gpurun_out
for cfg in "256 192" "512 192" "640 192" "1024 192" "1024 96" "640 96" "512 96" "1024 320"; do
  set -- $cfg
  IA_BR_SPEC_WG=$1 IA_BR_SPEC_PTS=$2 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-config2 --no-breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wg $1 pts $2', d['ms_per_step'], d['value'])"
done
