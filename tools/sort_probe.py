"""ia_morton_order on the points the step sorts: march points of the headline frame's secondary rays (tools/spec_search_probe.march_points)
and uniform points, several sizes; time per call (HIP events, 5 repeats after 2 warm-ups) and GB/s of algorithmic traffic (68 B / point for the
three 10-bit passes).  IA_AMD_LIB selects the library (tools/ab_build.sh): the rocPRIM build of rounds 2-4 for the A/B.
  python tools/sort_probe.py [--check]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import _lib as L, synthetic as S      # noqa: E402

DEV = "cuda:0"


def order_of(pts, origin, inv_cell, drop=0):
    lib, st = L.lib(), L.stream()
    n = pts.shape[0]
    order = torch.empty(n, dtype=torch.int32, device=DEV)
    nb = int(lib.ia_morton_order_tmp_bytes(L.i64(n)))
    tmp = L.scratch("morton", nb, pts.device)
    L.check(lib.ia_morton_order(L.i64(n), L.ptr(pts), origin, L.f32(inv_cell), L.i32(drop), L.ptr(order), L.ptr(tmp), C.c_size_t(nb), st), "ia_morton_order")
    return order


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    from tools import spec_search_probe as SP
    rs, rays, _ = S.build_frame(DEV, 256, 256, pose_seed=0, beta=0.01)
    lo, inv_cell = rs._sort_grid_params()
    origin = (C.c_float * 3)(*lo)
    march = SP.march_points(rs, rays, 1 << 21)
    g = torch.Generator(device=DEV).manual_seed(0)
    out = dict(lib=os.environ.get("IA_AMD_LIB", "tree"), rows=[])
    # march_points() returns its points sorted: the step's input is in between the sorted and the shuffled set (ray-major, samples in t order)
    sets = [("march_sorted", march), ("march_shuffled", march[torch.randperm(march.shape[0], device=DEV)].contiguous())]
    for n in (4_000_000, 40_000_000, 140_000_000):
        if n > march.shape[0]:
            reps = -(-n // march.shape[0])
            big = (march.repeat_interleave(reps, 0)[:n] + (torch.rand((n, 3), device=DEV, generator=g) - 0.5) * 0.02).contiguous()
            sets.append((f"march_x{reps}_jitter", big))
    sets.append(("uniform", (torch.rand((40_000_000, 3), device=DEV, generator=g) * 2.4 - 1.2).contiguous()))
    for name, pts in sets:
        n = pts.shape[0]
        for _ in range(2):
            o = order_of(pts, origin, inv_cell)
        if args.check:
            keys = torch.empty(n, dtype=torch.int32, device=DEV)
            L.check(L.lib().ia_morton_keys(L.i64(n), L.ptr(pts), origin, L.f32(inv_cell), L.ptr(keys), L.stream()), "keys")
            want = torch.sort(keys, stable=True)[1]
            assert torch.equal(o.long(), want), name
            del keys, want
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            o = order_of(pts, origin, inv_cell)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out["rows"].append(dict(points=name, n=n, ms=round(ms, 3), ns_per_point=round(ms * 1e6 / n, 3), GBps_68B=round(68.0 * n / ms / 1e6, 1)))
        del pts, o
    print(json.dumps(out))


if __name__ == "__main__":
    main()
