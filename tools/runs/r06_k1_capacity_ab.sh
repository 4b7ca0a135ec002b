# K1's total merged into the foreground-count read-back (capacity-sized K1 outputs) against its own read-back, alternating on one box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -f $O/r06_k1_capacity_ab.jsonl
F="--steps 5 --warmup 3 --no-cpu-baseline --no-config2 --no-config4 --no-search-modes --no-breakdown"
for round in 1 2; do
  for v in 0 1; do
    c4=$(IA_K1_CAPACITY=$v timeout 200 python $R/tools/config4_bench.py --steps 30 --repeat 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(dict(ms=d['ms_per_step'], runs=d['ms_per_step_runs'], readbacks=d['readbacks'], launches=d['launches'])))")
    hl=$(IA_K1_CAPACITY=$v timeout 300 python $R/bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(dict(ms=d['ms_per_step'], live=d['config']['peak_device_memory_GiB'], reserved=d['config']['peak_reserved_memory_GiB'])))")
    echo "{\"k1_capacity\": $v, \"round\": $round, \"config4\": $c4, \"headline\": $hl}" >> $O/r06_k1_capacity_ab.jsonl
  done
done
cat $O/r06_k1_capacity_ab.jsonl
