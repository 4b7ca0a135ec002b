#!/bin/bash
# the secondary march on 1 / 2 host threads + HIP streams (IA_SECONDARY_STREAMS) and the per-stream chunk size: headline step, same box
run() { IA_SECONDARY_STREAMS=$1 IA_SECONDARY_CHUNK=$2 timeout 900 python bench.py --steps 12 --warmup 6 --no-search-modes --no-cpu-baseline --no-config2 --no-breakdown 2>gpurun_out/streams.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $1 chunk $2', d['ms_per_step'], d['value'], 'alloc', d['config']['peak_device_memory_GiB'], 'reserved', d['config']['peak_reserved_memory_GiB'])" || tail -3 gpurun_out/streams.err; }
run 1 16777216
run 2 16777216
run 2 12582912
run 2 10485760
run 1 16777216
run 2 14000000
