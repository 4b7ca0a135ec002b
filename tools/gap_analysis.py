#!/usr/bin/env python3
"""GPU idle time between kernels of the training step, from a rocprofv3 --kernel-trace CSV.
usage: python tools/gap_analysis.py <kernel_trace.csv> [n_tail_kernels]   (analyses the tail = steady-state steps)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
rows = rows[-tail:]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gaps = collections.defaultdict(lambda: [0, 0])
tot_gap = 0
hist = collections.Counter()
for a, b in zip(rows[:-1], rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g <= 0:
        continue
    tot_gap += g
    hist[min(int(g / 1000).bit_length(), 12)] += 1
    if g > 15000:
        k = a["Kernel_Name"].replace("(anonymous namespace)::", "")[:50] + "  ->  " + b["Kernel_Name"].replace("(anonymous namespace)::", "")[:50]
        gaps[k][0] += 1
        gaps[k][1] += g
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {tot_gap / 1e6:.2f} ms ({100.0 * tot_gap / span:.1f} %)")
print("gap histogram (us, power-of-two buckets):", {f"<{2 ** k}": v for k, v in sorted(hist.items())})
print("largest idle contributors (gaps > 15 us):")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / 1e6:7.3f} ms  x{c:4d}  avg {t / c / 1e3:7.1f} us   {k}")
