# read-back latency knobs of the runtime against the config-4 step (15 size read-backs per step), alternating on one box
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 300 python $R/tools/launch_audit.py --out $O/r06_launch_audit_gaps.json > /dev/null 2>&1
rm -f $O/r06_host_wait_ab.jsonl
for round in 1 2; do
  for cfg in "default" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=1000 HSA_ENABLE_INTERRUPT=0" "HSA_ENABLE_MWAITX=1"; do
    if [ "$cfg" = "default" ]; then e=""; else e="$cfg"; fi
    out=$(env $e timeout 200 python $R/tools/config4_bench.py --steps 30 --repeat 2 2>/dev/null | tail -1)
    echo "{\"env\": \"$cfg\", \"round\": $round, \"result\": $out}" >> $O/r06_host_wait_ab.jsonl
  done
done
cat $O/r06_host_wait_ab.jsonl | cut -c1-400
