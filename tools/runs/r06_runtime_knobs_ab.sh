# two more runtime knobs, alternating on one box: HSA_ENABLE_SDMA=0 (copies and read-backs through blit kernels instead of the SDMA engines),
# GPU_MAX_HW_QUEUES=8 (default 4; the step uses three)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -f $O/r06_runtime_knobs_ab.jsonl
F="--steps 5 --warmup 3 --no-cpu-baseline --no-config2 --no-config4 --no-search-modes --no-breakdown"
for round in 1 2; do
  for cfg in "default" "HSA_ENABLE_SDMA=0" "GPU_MAX_HW_QUEUES=8"; do
    if [ "$cfg" = "default" ]; then e=""; else e="$cfg"; fi
    c4=$(env $e timeout 200 python $R/tools/config4_bench.py --steps 30 --repeat 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    hl=$(env $e timeout 300 python $R/bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "{\"env\": \"$cfg\", \"round\": $round, \"config4_ms_per_step\": $c4, \"headline_ms_per_step\": $hl}" >> $O/r06_runtime_knobs_ab.jsonl
  done
done
cat $O/r06_runtime_knobs_ab.jsonl
