# workgroup size of the search (one point per lane, chunk = the workgroup's lanes) in the headline step: 256 (default) / 128 / 64, same box, alternating
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_search_wg_step.jsonl
: > $O
for rep in 1 2; do for wg in 256 64 128; do
  IA_BR_SPEC_WG=$wg timeout 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-config4 --no-breakdown --no-search-modes 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(dict(wg=$wg, ms_per_step=d['ms_per_step'])))" >> $O
done; done
cat $O
