#!/usr/bin/env python3
"""What does the permuted write of the min-select cost?  (`select_min_split_kernel` writes 7 x its algorithmic bytes.)

    python tools/select_min_probe.py [--repeat 3]

The SDF-only query of the headline frame's primary march edges (RenderStep._sdf_at: search -> candidate list -> hash gather -> head -> min per
point) run twice over the SAME points in the SAME (Morton) evaluation order:
  permuted   the product path: the search reads the caller's points through the permutation, the select writes sdf[order[p]]
  in_order   the points gathered into Morton order first (a [n,3] copy), order = None: the select writes sdf[p] -- coalesced
The difference of the select's time is the CEILING of what any re-ordering of its writes can return; the gathered copy the in_order
variant needs is timed next to it (round 2 removed exactly that copy: 4.7 x its algorithmic bytes)."""
import argparse
import json
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    import bench as B
    from intrinsicavatar_amd import build, nerfacc, _lib as L
    from intrinsicavatar_amd.render import ray_points
    build.build()
    dev = "cuda:0"
    rs, rays, export, mat, sg = B.build_headline(dev, 540, 1024, 0, "male-3-casual:0")
    dfm = rs.deformer
    with torch.no_grad():
        r = dfm.transform_rays_w2s(rays.float())
        n_rays = r.shape[0]
        ro, rd = r[:, 0:3].contiguous(), r[:, 3:6].contiguous()
        near = torch.zeros(n_rays, device=dev)
        far = torch.full((n_rays,), 1e10, device=dev)
        # a finer march than the frame's own (step / 8): ~10^8 points, the size of one secondary-march search batch
        intervals, samples, _ = nerfacc.traverse_grids(ro, rd, rs.binaries, rs.aabbs, near, far, rs.render_step_size / 8, 0.0,
                                                       grid_bits=rs.grid_bits, termination_planes=False)
        pts = ray_points(ro, rd, intervals.ray_indices, intervals.vals)
        n = pts.shape[0]
        if n > rs.MAX_SEARCH_POINTS:
            pts = pts[:rs.MAX_SEARCH_POINTS].contiguous()
            n = pts.shape[0]
        order = rs._spatial_order(pts)
        lib = L.lib()
        rows = []
        for rep in range(args.repeat):
            res = {}
            for name in ("permuted", "in_order"):
                torch.cuda.synchronize()
                lib.start()
                if name == "permuted":
                    sdf = dfm.deform_sdf(pts, rs.geometry, order=order)
                    t_copy = 0.0
                else:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ps = pts[order.long()].contiguous()
                    e1.record()
                    sdf2 = dfm.deform_sdf(ps, rs.geometry, order=None)
                det = lib.report(detail=True)
                if name == "in_order":
                    t_copy = e0.elapsed_time(e1)
                sel = sum(c[0] for k in det for c in det[k] if k.startswith("ia_deform_select_min"))
                tot = sum(c[0] for k in det for c in det[k])
                res[name] = dict(select_ms=round(sel, 3), all_entry_points_ms=round(tot, 3), gathered_copy_ms=round(t_copy, 3),
                                 search_ms=round(sum(c[0] for k in det for c in det[k] if "broyden" in k), 3))
            same = bool(torch.equal(sdf[order.long()], sdf2))
            rows.append(dict(points=n, repeat=rep, values_identical=same, **res))
    out = dict(rows=rows, bytes_algorithmic_select=int(n) * 4 + int(n) * 12,
               note="select_ms: the min-select alone (live HIP events around the entry point); in_order needs the gathered [n,3] copy of the points "
                    "(gathered_copy_ms, torch index) and hands the search a plain list")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
