# VERDICT r05 item 7, "a measured row each": the lower-traffic alternative of each of the three kernels, on one box
#   hash_fwd_kernel<true>  (flat Jacobian gather, 2.2 x)   vs IA_HASH_FWD=xcd     (XCD-partitioned gather + transpose: 8 x less fabric traffic)
#   hash-backward record fill (staged, 1.8 x)               vs IA_HASHBWD_FILL=direct (per-workgroup runs, scattered appends)
#   select_min_split_kernel (permuted 4-byte writes, 7 x)   vs in-order writes (tools/select_min_probe.py: the ceiling of any re-ordering)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
F="--steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-config4 --no-search-modes"
rm -f $O/r06_traffic_rows.jsonl
for round in 1 2; do
  for cfg in "default" "IA_HASH_FWD=xcd" "IA_HASHBWD_FILL=direct"; do
    if [ "$cfg" = "default" ]; then e=""; else e="$cfg"; fi
    env $e timeout 400 python $R/bench.py $F 2>/dev/null | tail -1 > /tmp/line.json
    python - "$cfg" $round >> $O/r06_traffic_rows.jsonl <<'P'
import json, sys
d = json.loads(open("/tmp/line.json").read())
kb = d["kernel_breakdown_ms_per_step"]
keep = {k: v["ms_per_step"] for k, v in kb.items() if "hashgrid" in k}
print(json.dumps(dict(env=sys.argv[1], round=int(sys.argv[2]), ms_per_step=d["ms_per_step"], hashgrid_entry_points_ms_per_step=keep)))
P
  done
done
cat $O/r06_traffic_rows.jsonl
timeout 600 python $R/tools/select_min_probe.py --repeat 3 > $O/r06_select_min_probe.json 2> $O/r06_select_min_probe.err
tail -40 $O/r06_select_min_probe.json; tail -5 $O/r06_select_min_probe.err
