#!/usr/bin/env python3
"""MD-style iterations on small cells (reuse path, VerletGraph.evaluate) with and without hipGraph replay of the step:
python tools/time_md_small.py [n_cells_per_axis ...]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from torch_m3gnet.data.md import VerletGraph  # noqa: E402

dev = torch.device("cuda")
model = bench.default_model(dev)
a = 3.61
base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
for n in [int(v) for v in sys.argv[1:]] or [2, 3, 4, 6]:
    gi = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1)
    pos0 = torch.tensor((gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a, device=dev)
    lat = np.eye(3) * n * a
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    for replay in (0, 1):
        model.engine.set_option("graph_replay", replay)
        vg = VerletGraph([lat], [np.full(pos0.size(0), 29)], 5.0, 4.0, skin=0.5, device=dev)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            def it():
                pos = pos0 + (torch.rand(pos0.shape, generator=gen, device=dev, dtype=torch.float64) - 0.5) * 0.05
                vg.evaluate(model, pos, forces=True, extras=False)
            for _ in range(10):
                it()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                it()
            torch.cuda.synchronize()
        print(f"{pos0.size(0):5d} atoms, graph_replay={replay}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per MD iteration, paths {vg.stats}", flush=True)
    model.engine.set_option("graph_replay", 0)
