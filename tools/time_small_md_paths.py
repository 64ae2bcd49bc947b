#!/usr/bin/env python3
"""MD iteration on small Cu cells by path: reuse (evaluate, no wait), reuse with the wait (update), refill (forced), search (forced).
    python tools/time_small_md_paths.py [n_cells ...]      (GPU box; default 2 3 6 = 32 / 108 / 864 atoms)"""
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
base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
for n in [int(a) for a in sys.argv[1:]] or [2, 3, 6]:
    gi = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1)
    p0 = torch.tensor((gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * 3.61, device=dev)
    vg = VerletGraph([np.eye(3) * n * 3.61], [np.full(p0.size(0), 29)], 5.0, 4.0, skin=0.5, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)

    def jitter():
        return p0 + (torch.rand(p0.shape, generator=gen, device=dev, dtype=torch.float64) - 0.5) * 0.05

    def run(fn, reps=200, warm=30):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    g0 = vg.update(jitter())
    step = run(lambda: model(g0, forces=True, extras=False))
    res = {"step": step,
           "reuse, no wait (evaluate)": run(lambda: vg.evaluate(model, jitter(), forces=True, extras=False)),
           "reuse (update)": run(lambda: model(vg.update(jitter()), forces=True, extras=False)),
           "refill": run(lambda: model(vg.update(jitter(), force="refill"), forces=True, extras=False)),
           "search": run(lambda: model(vg.update(jitter(), force="search"), forces=True, extras=False), reps=60, warm=10),
           # the same through ONE library call per step (VerletGraph.step -> m3g_md_step)
           "step(): reuse": run(lambda: vg.step(model, jitter())),
           "step(): refill": run(lambda: vg.step(model, jitter(), force="refill")),
           "step(): search": run(lambda: vg.step(model, jitter(), force="search"), reps=60, warm=10)}
    print(f"{p0.size(0):5d} atoms: " + "  ".join(f"{k} {v:.3f} ms" for k, v in res.items()), flush=True)
