#!/bin/bash
# the secondary march on 1 / 2 host threads + HIP streams (IA_SECONDARY_STREAMS), per-stream chunk size, capacity of the fused traversal
run() { IA_SECONDARY_STREAMS=$1 IA_SECONDARY_CHUNK=$2 IA_TRAVERSE_CAP_PER_RAY=$3 timeout 900 python bench.py --steps 12 --warmup 6 --pose ${POSE:-male-3-casual:0} --no-search-modes --no-cpu-baseline --no-config2 --no-breakdown 2>gpurun_out/streams.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $1 chunk $2 cap $3', d['ms_per_step'], d['value'], 'alloc', d['config']['peak_device_memory_GiB'], 'reserved', d['config']['peak_reserved_memory_GiB'])" || tail -3 gpurun_out/streams.err; }
run 2 16777216 24
run 2 16777216 66
run 1 16777216 24
run 2 26843545 24
POSE=synthetic:0 run 2 16777216 24
POSE=synthetic:0 run 1 16777216 24
