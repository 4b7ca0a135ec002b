#!/bin/bash
# frame pipelining (two half-frames in flight) at several secondary chunk sizes, against the default step, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_frame_pipeline_chunks.jsonl
: > $OUT
run() {
    timeout 400 python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config2 --no-breakdown --no-search-modes 2>/dev/null | tail -1 | python -c "
import sys, json, os
b = json.loads(sys.stdin.read())
print(json.dumps(dict(frame_pipeline=int(os.environ.get('IA_FRAME_PIPELINE', '0')), chunk=os.environ.get('IA_SECONDARY_CHUNK'), ms_per_step=b['ms_per_step'], live_GiB=b['config']['peak_device_memory_GiB'], reserved_GiB=b['config']['peak_reserved_memory_GiB'])))" >> $OUT
}
unset IA_SECONDARY_CHUNK; IA_FRAME_PIPELINE=0 run
for ch in 8388608 10485760 12582912; do IA_SECONDARY_CHUNK=$ch IA_FRAME_PIPELINE=2 run; done
unset IA_SECONDARY_CHUNK; IA_FRAME_PIPELINE=0 run
cat $OUT
