"""GPU probe (not a test): cost of the per-step record exchange of bench.py's N > 1 path, piece by piece
(run under torchrun with one process)."""
import os, time
import numpy as np
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
W, world = 61, 1
allrec = torch.zeros((world, W), device=dev, dtype=torch.float64)
h_rec = torch.zeros(W, dtype=torch.float64).pin_memory()
h_all = torch.zeros((world, W), dtype=torch.float64).pin_memory()
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def full():
    allrec.zero_(); allrec[0].copy_(h_rec, non_blocking=True); dist.all_reduce(allrec)
    h_all.copy_(allrec, non_blocking=True); torch.cuda.current_stream(dev).synchronize()
def no_ar():
    allrec.zero_(); allrec[0].copy_(h_rec, non_blocking=True)
    h_all.copy_(allrec, non_blocking=True); torch.cuda.current_stream(dev).synchronize()
def only_ar():
    dist.all_reduce(allrec); torch.cuda.current_stream(dev).synchronize()
def only_sync():
    torch.cuda.current_stream(dev).synchronize()
def h2d():
    allrec[0].copy_(h_rec, non_blocking=True); torch.cuda.current_stream(dev).synchronize()
def d2h():
    h_all.copy_(allrec, non_blocking=True); torch.cuda.current_stream(dev).synchronize()
def gather():
    dist.all_gather_into_tensor(allrec, allrec[0]); torch.cuda.current_stream(dev).synchronize()
for name, fn in (("full", full), ("no_all_reduce", no_ar), ("all_reduce+sync", only_ar), ("sync", only_sync), ("h2d+sync", h2d), ("d2h+sync", d2h)):
    print(f"{name:18s} {t(fn):8.1f} us", flush=True)
# the same after 300 us of host idling (the solve of bench.py ends with a host synchronisation on another stream)
def spin(us):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e6 < us: pass
tot = 0.0
for _ in range(200):
    spin(300)
    t0 = time.perf_counter(); full(); tot += time.perf_counter() - t0
print(f"full after 300 us idle {tot / 200 * 1e6:8.1f} us", flush=True)
s2 = torch.cuda.Stream(device=dev)
x = torch.zeros(1 << 20, device=dev)
tot = 0.0
for _ in range(200):
    with torch.cuda.stream(s2):
        for _ in range(20): x.add_(1.0)
    s2.synchronize()
    t0 = time.perf_counter(); full(); tot += time.perf_counter() - t0
print(f"full after work on another stream {tot / 200 * 1e6:8.1f} us", flush=True)
dist.destroy_process_group()
