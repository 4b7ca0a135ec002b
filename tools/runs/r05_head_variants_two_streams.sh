# does a smaller SDF-head workgroup (8 / 12 waves instead of 16: 50 / 70 KB of LDS instead of 90) overlap better with the OTHER stream's kernels
# in the two-stream secondary march?  (the 16-wave workgroup needs a CU almost to itself)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_head_variants_two_streams.jsonl
: > $O
run() {
  timeout 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-config4 --no-breakdown --no-search-modes 2>/dev/null | tail -1 | \
    python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print(json.dumps(dict(cfg=os.environ.get('CFG'), ms_per_step=d['ms_per_step'])))" >> $O
}
for rep in 1 2; do
  CFG=default run
  CFG=pipe2w12 IA_SDF_HEAD=pipe2w12 run
  CFG=pipe2w8 IA_SDF_HEAD=pipe2w8 run
done
cat $O
