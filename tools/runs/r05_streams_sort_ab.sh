#!/bin/bash
# two streams vs one with three sort builds (same box, back to back): does the sort decide whether the second stream pays?
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_streams_sort_ab.jsonl
: > $OUT
for lib in tree t256 rocprim; do
  for streams in 2 1; do
    if [ $lib = tree ]; then unset IA_AMD_LIB; else export IA_AMD_LIB=$R/intrinsicavatar_amd/_ab/libia_amd_$lib.so; fi
    IA_SECONDARY_STREAMS=$streams timeout 300 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-config2 --no-breakdown --no-search-modes 2>/dev/null | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read())
print(json.dumps(dict(lib='$lib', streams=$streams, ms_per_step=b['ms_per_step'], taken=b['config']['secondary_march_streams_taken'], live_GiB=b['config']['peak_device_memory_GiB'], reserved_GiB=b['config']['peak_reserved_memory_GiB'])))" >> $OUT
  done
done
cat $OUT
