"""Experiment: does gloo's asynchronous all-reduce of a large CUDA tensor complete with two ranks on one GPU?
(tools/bench_train.py under IFL_REHEARSE_ONE_GPU=1 stalls in GradientAllReducer.finish.)"""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(50, exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
for n in (1024, 1 << 20, 8219744):
    for mode in ("sync", "async"):
        x = torch.full((n,), float(rank + 1), device=dev)
        torch.cuda.synchronize()
        t0 = time.time()
        if mode == "sync":
            dist.all_reduce(x)
        else:
            w = dist.all_reduce(x, async_op=True)
            w.wait()
        torch.cuda.synchronize()
        print(f"rank {rank} n={n} {mode}: {time.time() - t0:.3f} s, value {float(x[0])}", flush=True)
a = torch.full((8219744,), float(rank + 1), device=dev)
b = torch.full((6420298,), float(rank + 1), device=dev)
torch.cuda.synchronize()
t0 = time.time()
wa = dist.all_reduce(a, async_op=True)
wb = dist.all_reduce(b, async_op=True)
wa.wait(); a.div_(world)
wb.wait(); b.div_(world)
torch.cuda.synchronize()
print(f"rank {rank} two in flight: {time.time() - t0:.3f} s, values {float(a[0])} {float(b[0])}", flush=True)
# the same with the device busy when the collectives are issued (matrix products on the current and on a side stream)
m = torch.randn(8192, 8192, device=dev)
side = torch.cuda.Stream(dev)
for trial in range(3):
    a.fill_(float(rank + 1)); b.fill_(float(rank + 1))
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        mm = m @ m
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(20):
            mm2 = m @ m
        b.add_(mm2[0, 0] * 0)
    torch.cuda.current_stream().wait_stream(side)
    a.add_(mm[0, 0] * 0)
    wa = dist.all_reduce(a, async_op=True)
    wb = dist.all_reduce(b, async_op=True)
    t1 = time.time()
    wa.wait(); a.div_(world)
    wb.wait(); b.div_(world)
    torch.cuda.synchronize()
    print(f"rank {rank} busy device, trial {trial}: issued after {t1 - t0:.3f} s, done after {time.time() - t0:.3f} s, values {float(a[0])} {float(b[0])}", flush=True)
dist.destroy_process_group()
