#!/bin/bash
# do secondary-ray chunks above the default 16 Mi still pay?  (4 -> 8 -> 16 Mi x 2 streams: 319 -> 311 -> 304 ms; 288 GB of HBM)
R=$GRAFT_REPO_ROOT
F="--steps 4 --warmup 2 --no-search-modes --no-cpu-baseline --no-config2 --no-config4 --no-breakdown"
: > $R/gpurun_out/r06_big_chunk_probe.jsonl
for cfg in "16777216 2 150000000" "25165824 2 240000000" "33554432 2 300000000" "33554432 1 300000000" "16777216 2 150000000"; do
  set -- $cfg
  IA_BENCH_ARENA_GIB=8 IA_SECONDARY_CHUNK=$1 IA_SECONDARY_STREAMS=$2 IA_MAX_SEARCH_POINTS=$3 timeout 400 python $R/bench.py $F 2>$R/gpurun_out/big_chunk.err | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(dict(secondary_chunk=$1, streams=$2, max_search_points=$3, streams_taken=d['config']['secondary_march_streams_taken'], ms_per_step=d['ms_per_step'], rays_per_s=d['value'], peak_live_GiB=d['config']['peak_device_memory_GiB'], peak_reserved_GiB=d['config']['peak_reserved_memory_GiB'])))" | tee -a $R/gpurun_out/r06_big_chunk_probe.jsonl
  tail -2 $R/gpurun_out/big_chunk.err | cut -c1-300
done
