import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import fields
dev = "cuda:0"
n = 4_400_000
g = torch.Generator().manual_seed(0)
base = torch.rand((n // 16, 1, 3), generator=g) * 0.5 + 0.25
dirs = torch.nn.functional.normalize(torch.randn((n // 16, 1, 3), generator=g), dim=-1)
t = torch.arange(16).float()[None, :, None] * 0.0135
x = (base + dirs * t).reshape(-1, 3).clamp(0, 1).to(dev).contiguous()
xr = torch.rand((x.shape[0], 3), generator=g).to(dev)
table = (torch.rand(fields.hash_n_entries() * 2, generator=g) * 2e-4 - 1e-4).to(dev)
def t_(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, pts in (("ray-ordered", x), ("random", xr)):
    print(name, "fwd %.2f ms" % t_(lambda: fields.hashgrid_forward(pts, table)), " fwd+jac %.2f ms" % t_(lambda: fields.hashgrid_forward(pts, table, with_jac=True)))
