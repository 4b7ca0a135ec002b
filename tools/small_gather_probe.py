import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, _lib as L
dev = "cuda:0"
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01)
geo = rs.geometry
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for n in (140_000, 206_000, 557_000, 1_000_000, 2_000_000, 4_000_000, 10_000_000):
    x = (geo.center + (torch.rand((n, 3), device=dev, generator=g) - 0.5) * geo.scale * 0.5).contiguous()
    x = x[rs._spatial_order(x).long()].contiguous() if n >= (1 << 20) else x
    for _ in range(3): y = geo.sdf_only(x)
    lib = L.lib(); lib.start()
    for _ in range(10): y = geo.sdf_only(x)
    per = lib.report()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(20): y = geo.sdf_only(x)
    torch.cuda.synchronize()
    out[n] = dict(gather_ms=round(per["ia_hashgrid_fwd_xcd"][1] / per["ia_hashgrid_fwd_xcd"][0], 4) if "ia_hashgrid_fwd_xcd" in per else None,
                  wall_ms=round((time.perf_counter() - t0) / 20 * 1e3, 4))
print(json.dumps(dict(plan=os.environ.get("IA_HASH_XCD_PLAN", "level"), rows=out)))
