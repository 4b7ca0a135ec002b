#!/bin/bash
# two half-frames in flight (train_phys.forward_backward_phys_pipelined) against the default step, same box, interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_frame_pipeline_ab.jsonl
: > $OUT
for rep in 1 2; do
for fp in 0 2; do
    if [ $fp = 2 ]; then export IA_SECONDARY_CHUNK=6291456; else unset IA_SECONDARY_CHUNK; fi      # two half-frames in flight: half the chunk each
    IA_FRAME_PIPELINE=$fp timeout 400 python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config2 --no-breakdown --no-search-modes 2>$R/gpurun_out/fp_err_$fp.txt | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read())
print(json.dumps(dict(frame_pipeline=$fp, ms_per_step=b['ms_per_step'], rays_per_s=b['value'], live_GiB=b['config']['peak_device_memory_GiB'], reserved_GiB=b['config']['peak_reserved_memory_GiB'], samples=b['config']['samples'])))" >> $OUT
done
done
cat $OUT; tail -3 $R/gpurun_out/fp_err_2.txt
