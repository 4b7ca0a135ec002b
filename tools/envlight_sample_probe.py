import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import pbr
H, W = 512, 1024
g = torch.Generator().manual_seed(0)
img = torch.rand((H, W, 3), generator=g) ** 6
env = pbr.EnvironmentLightTensor(img.cuda()); env.update_pdf()
k = 83_000_000
u = torch.rand((k, 3), device="cuda")
for _ in range(2): d = env.sample(k, u)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): d = env.sample(k, u)
e1.record(); torch.cuda.synchronize()
print("envlight sample", k, "samples:", round(e0.elapsed_time(e1) / 5, 3), "ms")
