#!/usr/bin/env python3
"""Practical HBM bandwidth of the box with plain torch kernels (ceiling for the bandwidth-bound kernels in DESIGN.md):
read-only reduction, copy (read + write) and fill (write-only) over a 2 GiB buffer."""
import time

import torch

n = 1 << 29   # 2 GiB of fp32
x = torch.empty(n, device="cuda").normal_()
y = torch.empty_like(x)


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


gb = n * 4 / 1e9
t = timeit(lambda: x.sum())
print(f"read   {gb / t / 1e3:.2f} TB/s")
t = timeit(lambda: y.copy_(x))
print(f"copy   {2 * gb / t / 1e3:.2f} TB/s (read + write)")
t = timeit(lambda: y.fill_(1.0))
print(f"fill   {gb / t / 1e3:.2f} TB/s")
