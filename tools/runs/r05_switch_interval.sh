#!/bin/bash
# two host threads feed the two streams of the secondary march: does the interpreter's switch interval (GIL hand-over) matter?
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_switch_interval.jsonl
: > $OUT
for rep in 1 2; do
for si in 0.005 0.0005 0.00005; do
    IA_SWITCH_INTERVAL=$si timeout 300 python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config2 --no-breakdown --no-search-modes 2>/dev/null | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read())
print(json.dumps(dict(switch_interval=$si, ms_per_step=b['ms_per_step'], taken=b['config']['secondary_march_streams_taken'], host_cpus_busy=b['config']['host_cpus_busy'])))" >> $OUT
done
done
cat $OUT
