# one search on the device at a time (IA_SEARCH_TOKEN) x search at four workgroups per CU (IA_BR_SPEC_PAD_LDS): does another stream's hash gather
# run in the slots the search leaves?  Same box, alternating; headline step.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_search_token_ab.jsonl
: > $O
run() {
  timeout 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-config4 --no-breakdown --no-search-modes 2>/dev/null | tail -1 | \
    python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print(json.dumps(dict(cfg=os.environ.get('CFG'), ms_per_step=d['ms_per_step'], streams=d.get('secondary_streams_taken'))))" >> $O
}
for rep in 1 2; do
  CFG=default run
  CFG=pad8k IA_BR_SPEC_PAD_LDS=8192 run
  CFG=token IA_SEARCH_TOKEN=1 run
  CFG=token+pad8k IA_SEARCH_TOKEN=1 IA_BR_SPEC_PAD_LDS=8192 run
  CFG=token+pad20k IA_SEARCH_TOKEN=1 IA_BR_SPEC_PAD_LDS=20480 run
done
cat $O
