#!/bin/bash
# two chunks of 20.8 M rays (one per stream) instead of four of 10.4 M: search batches of ~155 M points, split at the 2^31-item bound
R=$GRAFT_REPO_ROOT
F="--steps 4 --warmup 2 --no-search-modes --no-cpu-baseline --no-config2 --no-config4 --no-breakdown"
for cfg in "16777216 2 150000000" "33554432 2 165000000" "33554432 2 120000000" "16777216 2 150000000" "33554432 2 165000000"; do
  set -- $cfg
  IA_BENCH_ARENA_GIB=8 IA_SECONDARY_CHUNK=$1 IA_SECONDARY_STREAMS=$2 IA_MAX_SEARCH_POINTS=$3 timeout 400 python $R/bench.py $F 2>$R/gpurun_out/big_chunk.err | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(dict(secondary_chunk=$1, streams=$2, max_search_points=$3, streams_taken=d['config']['secondary_march_streams_taken'], ms_per_step=d['ms_per_step'], rays_per_s=d['value'], peak_live_GiB=d['config']['peak_device_memory_GiB'], peak_reserved_GiB=d['config']['peak_reserved_memory_GiB'])))" | tee -a $R/gpurun_out/r06_big_chunk_probe.jsonl
  grep -E "Error|error" $R/gpurun_out/big_chunk.err | tail -2 | cut -c1-300
done
