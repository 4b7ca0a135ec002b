#!/bin/bash
# secondary-march chunk size x streams on the headline step: ms per step, peak live / reserved memory (same box, back to back)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_chunk_streams_sweep.jsonl
: > $OUT
for streams in 1 2; do
  for chunk in 4194304 8388608 16777216; do
    IA_SECONDARY_STREAMS=$streams IA_SECONDARY_CHUNK=$chunk timeout 300 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-config2 --no-breakdown --no-search-modes 2>/dev/null | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read())
print(json.dumps(dict(streams=$streams, chunk=$chunk, ms_per_step=b['ms_per_step'], taken=b['config']['secondary_march_streams_taken'], live_GiB=b['config']['peak_device_memory_GiB'], reserved_GiB=b['config']['peak_reserved_memory_GiB'])))" >> $OUT
  done
done
cat $OUT
