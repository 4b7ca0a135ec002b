#!/bin/bash
# memory / speed trade of the headline step: secondary-ray chunk size x march streams, small up-front arena so that
# `peak_reserved` is what the step needs, not what the bench reserved.  One line per setting -> gpurun_out/r06_memory_speed_sweep.jsonl
R=$GRAFT_REPO_ROOT
F="--steps 4 --warmup 2 --no-search-modes --no-cpu-baseline --no-config2 --no-config4 --no-breakdown"
: > $R/gpurun_out/r06_memory_speed_sweep.jsonl
for chunk in 16777216 8388608 4194304; do for streams in 2 1; do
  IA_BENCH_ARENA_GIB=8 IA_SECONDARY_CHUNK=$chunk IA_SECONDARY_STREAMS=$streams IA_MAX_SEARCH_POINTS=$((chunk * 14)) timeout 300 python $R/bench.py $F 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(dict(secondary_chunk=$chunk, streams=$streams, streams_taken=d['config']['secondary_march_streams_taken'], ms_per_step=d['ms_per_step'], rays_per_s=d['value'], peak_live_GiB=d['config']['peak_device_memory_GiB'], peak_reserved_GiB=d['config']['peak_reserved_memory_GiB'])))" | tee -a $R/gpurun_out/r06_memory_speed_sweep.jsonl
done; done
