#!/usr/bin/env python3
"""Host-side time of one model(graph) call (Python + ctypes + ~23 kernel launches), measured while the GPU is the bottleneck:
the call returns as soon as its launches are queued, so the time spent inside it is the CPU's own."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from torch_m3gnet.data.synthetic import fcc_cu_graph  # noqa: E402

model = bench.default_model(torch.device("cuda"))
g = fcc_cu_graph(10, 10, 25).to("cuda")
for _ in range(5):
    model(g, forces=True, extras=False)
torch.cuda.synchronize()
n, inside = 100, 0.0
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter()
    model(g, forces=True, extras=False)
    inside += time.perf_counter() - a
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"10k-atom cell: {wall / n * 1e3:.3f} ms per step on the GPU, {inside / n * 1e3:.3f} ms of it inside the call on the CPU")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    model(g, forces=True, extras=False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
