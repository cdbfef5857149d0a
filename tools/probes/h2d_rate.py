#!/usr/bin/env python3
"""tools/probes/h2d_rate.py -- what the box's PCIe link carries: pinned host -> device and device -> pinned host copies of 8 / 64 / 256 MiB (torch, best of 5)."""
import time
import torch
for mib in (8, 64, 256):
    n = mib << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory(); d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print("%s %4d MiB: %.1f GB/s" % (name, mib, n / best / 1e9))
# two copies in flight on two streams (what a double-buffered reader can keep up)
n = 64 << 20
hs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]; ds = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(2)]
ss = [torch.cuda.Stream() for _ in range(2)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(8):
    with torch.cuda.stream(ss[k & 1]):
        ds[k & 1].copy_(hs[k & 1], non_blocking=True)
torch.cuda.synchronize()
print("H2D 8 x 64 MiB on two streams: %.1f GB/s" % (8 * n / (time.perf_counter() - t0) / 1e9))
