#!/usr/bin/env python3
"""Sensitivity of the step time to the order of the atoms: the 10,000-atom Cu cell in lattice order (as bench.py builds it)
and with the atoms randomly permuted before the graph is built (table gathers then touch rows all over the node tables)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from torch_m3gnet.data.graph_gpu import batch_from_arrays  # noqa: E402
from torch_m3gnet.model.build import build_model  # noqa: E402


def timeit(model, g, n=20):
    for _ in range(3):
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.manual_seed(0)
model = build_model(5.0, 4.0, 3, 3, 95, 64, 3).cuda()
a, dims = 3.61, (10, 10, 25)
base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
gi = np.stack(np.meshgrid(*[np.arange(d) for d in dims], indexing="ij"), -1)
pos = (gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a
pos = pos + np.random.default_rng(0).uniform(-0.025, 0.025, pos.shape)
lat = np.diag([d * a for d in dims]).astype(float)
z = np.full(len(pos), 29)
for name, order in (("lattice order", np.arange(len(pos))), ("random order", np.random.default_rng(1).permutation(len(pos)))):
    g = batch_from_arrays([lat], [pos[order]], [z], 5.0, 4.0)
    ms = timeit(model, g)
    e = float(g["total_energy"][0])
    print(f"{name}: {ms:.3f} ms/step, E = {e:.6f}", flush=True)
