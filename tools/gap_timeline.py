#!/usr/bin/env python3
"""GPU idle gaps of one step from a rocprofv3 --kernel-trace CSV: which kernels the device waits in front of, and for how long.
    python tools/gap_timeline.py <kernel_trace.csv> [n_last_kernels]      (analyses the last n kernels: one step's worth)"""
import csv, sys, json, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy_end = t0
gaps = []
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        gaps.append((s - busy_end, r["Kernel_Name"][:70]))
        busy += e - s
    else:
        busy += max(0, e - busy_end)
    busy_end = max(busy_end, e)
tot_gap = sum(g for g, _ in gaps)
by = collections.Counter()
for g, k in gaps:
    by[k] += g
print(json.dumps(dict(kernels=len(rows), span_ms=round((t1 - t0) / 1e6, 3), busy_ms=round(busy / 1e6, 3), idle_ms=round(tot_gap / 1e6, 3),
                      gaps_over_20us=sum(1 for g, _ in gaps if g > 20000), gaps_over_100us=sum(1 for g, _ in gaps if g > 100000),
                      idle_ms_in_gaps_over_20us=round(sum(g for g, _ in gaps if g > 20000) / 1e6, 3),
                      top_gaps_us_before=[(round(g / 1e3, 1), k) for g, k in sorted(gaps, reverse=True)[:25]])))
