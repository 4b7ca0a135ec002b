#!/usr/bin/env python3
"""print the top kernels of a rocprofv3 kernel_stats.csv: python tools/kstats.py <dir-or-csv> [substring] [top]"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = glob.glob(p + "/**/*kernel_stats.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
k = 0
for r in csv.DictReader(open(p)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if sub and sub not in n:
        continue
    print(f"{n[:64]:64s} x{int(r['Calls']):4d}  avg {float(r['AverageNs']) / 1e6:8.3f} ms  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms")
    k += 1
    if k >= top:
        break
