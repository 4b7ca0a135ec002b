#!/bin/bash
# PMC passes (separate from any tracing other than --kernel-trace, as the pool requires); summaries -> gpurun_out/pmc_*.csv
cd /tmp && export TMPDIR=/tmp
CMD="$1"; TAG="$2"
pass() { # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -- $CMD > /dev/null 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$name" <<'PY'
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r.get("Dispatch_Id"))
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:14]:
    print(name, "|", k, "| launches", cnt[k], "|", " ".join(f"{c}={v / cnt[k]:.4g}" for c, v in sorted(acc[k].items())))
PY
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass grbm GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
