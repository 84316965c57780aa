#!/usr/bin/env python3
"""Device copy rate (read + write bytes per second) by buffer size: what a stream of N bytes in + N bytes out
reaches on this box when the working set does (<= 100 MB) or does not (>= 512 MB) fit the 256 MiB Infinity Cache.
The step kernel's working set (~105 MB of state and outputs at 65 536 envs) is in the first regime."""
import torch
dev = torch.device("cuda", 0)
for mb in (16, 32, 50, 64, 100, 128, 256, 512):
    n = mb * 1024 * 1024 // 4
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    for _ in range(5):
        dst.copy_(src)
    reps = max(20, 4096 // mb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"copy {mb:4d} MB -> {mb:4d} MB: {us:8.2f} us per copy, {2 * n * 4 / us / 1e6:7.2f} TB/s (read+write)")
    # write-only (fill) and read-only (sum) of the same size
    e0.record()
    for _ in range(reps):
        dst.fill_(1.0)
    e1.record()
    torch.cuda.synchronize()
    usf = e0.elapsed_time(e1) * 1e3 / reps
    print(f"fill {mb:4d} MB: {usf:8.2f} us, {n * 4 / usf / 1e6:7.2f} TB/s (write only)")
    del src, dst
