#!/bin/bash
# usage: tools/pmc_probe.sh "<command>" <kernel-substring> <pass-name>:<CTR,CTR,...> [<pass-name>:<...> ...]
# One rocprofv3 --pmc pass per counter group (own run each, --kernel-trace only, as the pool requires); prints the
# per-launch average of every counter for the kernels whose name contains <kernel-substring>.
cd /tmp && export TMPDIR=/tmp
CMD="$1"; SUB="$2"; shift 2
for spec in "$@"; do
  name="${spec%%:*}"; ctrs="${spec#*:}"
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc ${ctrs//,/ } --kernel-trace --output-format csv -d /tmp/pmc_$name -- $CMD > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$name" "$SUB" <<'PY'
import csv, sys, collections
f, name, sub = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if sub not in k:
        continue
    k = k[:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    seen[k].add(r.get("Dispatch_Id"))
for k in acc:
    n = len(seen[k])
    print(name, "|", k, "| launches", n, "|", " ".join(f"{c}={v / n:.5g}" for c, v in sorted(acc[k].items())))
PY
done
