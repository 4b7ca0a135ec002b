"""how long does a gloo all-reduce of a 50 MB CUDA tensor take with N ranks sharing one GPU?  (the N>1 control-flow smoke of
bench.py on 1-GPU boxes uses this transport; it is not a measurement of anything the real nccl/RCCL path does)"""
import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
torch.cuda.set_device(0)
t = torch.ones(12_600_000, device="cuda:0")
for i in range(3):
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    h = dist.all_reduce(t, async_op=True)
    h.wait(); torch.cuda.synchronize()
    if dist.get_rank() == 0:
        print(f"world {dist.get_world_size()}: all_reduce 50 MB (async handle) {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
