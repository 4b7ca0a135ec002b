# K2's edge count + the sample count of its result in ONE read-back (capacity-sized edge buffers) against two, alternating on one box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -f $O/r06_k2_readback_ab.jsonl
F="--steps 5 --warmup 3 --no-cpu-baseline --no-config2 --no-config4 --no-search-modes --no-breakdown"
for round in 1 2 3; do
  for v in 0 1; do
    c4=$(IA_K2_MERGED_READBACK=$v timeout 200 python $R/tools/config4_bench.py --steps 30 --repeat 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(dict(ms=d['ms_per_step'], runs=d['ms_per_step_runs'], readbacks=d['readbacks'], launches=d['launches'])))")
    echo "{\"merged\": $v, \"round\": $round, \"config4\": $c4}" >> $O/r06_k2_readback_ab.jsonl
  done
done
for v in 0 1; do
  hl=$(IA_K2_MERGED_READBACK=$v timeout 300 python $R/bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "{\"merged\": $v, \"headline_ms_per_step\": $hl}" >> $O/r06_k2_readback_ab.jsonl
done
cat $O/r06_k2_readback_ab.jsonl
