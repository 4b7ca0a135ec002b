#!/usr/bin/env python3
"""memory-side traffic per kernel launch of the bench step (default: the headline workload) -> gpurun_out/<tag>_pmc_traffic.json
(copied to profiles/).

Three separate rocprofv3 --pmc passes (counters only next to --kernel-trace, as the pool requires) over
`bench.py --steps 1 --warmup 1`:  FETCH_SIZE;  WRITE_SIZE TCC_HIT_sum TCC_MISS_sum;  TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCC_READ_REQ_sum (vector-L1 accesses and the read requests it sends on to the L2, 128 B each: the L1 hit rate and the
L2 -> L1 line rate of the gather kernels).  Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE /
WRITE_SIZE are in KB and FETCH_SIZE reports half of the bytes on gfx950."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get("IA_PROFILE_TAG", "r05")
CMD = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-config2", "--no-config4",
       "--no-breakdown", "--no-search-modes"] + sys.argv[1:]


def run_pass(name, counters):
    d = f"/tmp/pmc_{name}"
    subprocess.run(["rm", "-rf", d])
    env = dict(os.environ, TMPDIR="/tmp", IA_SECONDARY_STREAMS="1")      # per-launch counters of kernels that have the device to themselves
    subprocess.run(["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + CMD,
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=900)
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen, cnt = set(), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
            try:                                            # duration of the dispatch under counter collection (for clocks = cycles / ns)
                acc[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
    return acc, cnt


def main():
    fa, fc = run_pass("fetch", ["FETCH_SIZE"])
    wa, wc = run_pass("write", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"])
    try:
        la, lc = run_pass("l1", ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "GRBM_GUI_ACTIVE"])
    except Exception:
        la, lc = {}, {}
    kernels = {}
    for k in fa:
        if k not in wa or fc[k] == 0:
            continue
        fetch = 2.0 * fa[k]["FETCH_SIZE"] * 1024.0 / fc[k]
        write = wa[k]["WRITE_SIZE"] * 1024.0 / wc[k]
        hit, miss = wa[k]["TCC_HIT_sum"], wa[k]["TCC_MISS_sum"]
        kernels[k] = dict(launches=fc[k], fetch_bytes=fetch, write_bytes=write, hbm_side_bytes_per_launch=fetch + write,
                          l2_hit_rate=round(hit / max(hit + miss, 1.0), 3))
        if k in la and lc[k]:
            acc_, req = la[k]["TCP_TOTAL_CACHE_ACCESSES_sum"] / lc[k], la[k]["TCP_TCC_READ_REQ_sum"] / lc[k]
            kernels[k].update(l1_accesses_per_launch=acc_, l1_read_requests_to_l2_per_launch=req,
                              l1_hit_rate=round(1.0 - req / max(acc_, 1.0), 3), l2_to_l1_bytes_per_launch=req * 128.0)
            if la[k].get("_ns", 0) > 0 and la[k].get("GRBM_GUI_ACTIVE", 0) > 0:
                # GRBM_GUI_ACTIVE counts shader-engine cycles while the dispatch runs: cycles / ns = the clock the kernel ran at
                g = la[k]["GRBM_GUI_ACTIVE"] / la[k]["_ns"]              # the counter is summed over the 8 XCDs when it reads ~8 x a plausible clock
                kernels[k]["clock_GHz"] = round(g / 8.0 if g > 6.0 else g, 3)
    top = dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_side_bytes_per_launch"] * kv[1]["launches"])[:20])
    out = dict(note="rocprofv3 --pmc passes on `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config2 --no-breakdown` " + " ".join(sys.argv[1:]) + " (tools/pmc_traffic.py): FETCH_SIZE / WRITE_SIZE are "
                    "KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of the bytes of a streaming read); values "
                    "are per launch, averaged over the launches of the run; Infinity-Cache hits are included (memory-side "
                    "requests of the L2)", kernels=top, all_kernels_by_traffic=sorted(kernels, key=lambda k: -kernels[k]["hbm_side_bytes_per_launch"] * kernels[k]["launches"])[:40])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{TAG}_pmc_traffic.json"), "w"), indent=1)
    for k, v in top.items():
        print(f"{k:62s} x{v['launches']:3d}  fetch {v['fetch_bytes'] / 1e9:7.2f} GB  write {v['write_bytes'] / 1e9:6.2f} GB  L2 hit {v['l2_hit_rate']}")


if __name__ == "__main__":
    main()
