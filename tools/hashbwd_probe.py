import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import fields
dev = "cuda:0"
n = 3_800_000
# samples along rays: 64 consecutive points close together (like real sample order)
g = torch.Generator().manual_seed(0)
base = torch.rand((n // 64, 1, 3), generator=g) * 0.5 + 0.25
dirs = torch.nn.functional.normalize(torch.randn((n // 64, 1, 3), generator=g), dim=-1)
t = torch.arange(64).float()[None, :, None] * 0.002
x = (base + dirs * t).reshape(-1, 3).clamp(0, 1).to(dev).contiguous()
n = x.shape[0]
table = torch.zeros(fields.hash_n_entries() * 2, device=dev)
def run(lo, hi, iters=3, method=None):
    gE = torch.zeros((n, 32), device=dev)
    gE[:, lo * 2:hi * 2] = torch.randn((n, (hi - lo) * 2), device=dev)
    fields.hashgrid_backward(x, gE, table, method=method)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fields.hashgrid_backward(x, gE, table, method=method)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for lo, hi in [(0, 16), (0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 8), (8, 12), (12, 16)]:
    print(f"levels {lo}..{hi - 1}: atomic {run(lo, hi, method='atomic'):.2f} ms   binned {run(lo, hi, method='binned'):.2f} ms")
