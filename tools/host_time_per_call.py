#!/usr/bin/env python3
"""Host time of one engine call on a small cell (the interpreter + the library's launches, no wait for the device in between):
what a loop that waits for the device every step pays on top of the kernels.   python tools/host_time_per_call.py   (GPU box)"""
import ctypes as C
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from torch_m3gnet.data.synthetic import fcc_cu_graph  # noqa: E402

dev = torch.device("cuda")
model = bench.default_model(dev)
g = fcc_cu_graph(2, 2, 2).to(dev)
for _ in range(20):
    model(g, forces=True, extras=False)
torch.cuda.synchronize()
eng = model.engine
orig = eng.lib.m3g_energy_forces
t_c = [0.0]


def timed(*a):
    t0 = time.perf_counter()
    r = orig(*a)
    t_c[0] += time.perf_counter() - t0
    return r


reps = 300
# (1) host time per call with the device kept busy far behind: queue depth grows, nothing waits
t0 = time.perf_counter()
for _ in range(reps):
    model(g, forces=True, extras=False)
t_host = (time.perf_counter() - t0) / reps
torch.cuda.synchronize()
eng.lib.m3g_energy_forces = timed
t_c[0] = 0.0
for _ in range(reps):
    model(g, forces=True, extras=False)
torch.cuda.synchronize()
eng.lib.m3g_energy_forces = orig
t0 = time.perf_counter()
for _ in range(reps):
    eng._signature(dev)
t_sig = (time.perf_counter() - t0) / reps
print(f"32-atom cell: host time per model(graph) call {t_host * 1e6:.1f} us (no wait), of which inside m3g_energy_forces {t_c[0] / reps * 1e6:.1f} us, "
      f"parameter signature {t_sig * 1e6:.1f} us")
