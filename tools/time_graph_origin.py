#!/usr/bin/env python3
"""Step time and per-stage device time of the 10,000-atom Cu cell for a graph built on the HOST (bench.py's) and the same cell
built on the GPU (graph_gpu.batch_from_arrays, what an MD loop uses): same atoms, same edge set -- do the two orderings cost the same?"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from torch_m3gnet.data.graph_gpu import batch_from_arrays  # noqa: E402
from torch_m3gnet.data.synthetic import fcc_cu_arrays, fcc_cu_graph  # noqa: E402
from torch_m3gnet.data import MaterialGraphKey as K  # noqa: E402


def timeit(model, g, n=40):
    for _ in range(10):
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


model = bench.default_model(torch.device("cuda"))
lat, pos, z = fcc_cu_arrays(10, 10, 25)
graphs = {"host-built": fcc_cu_graph(10, 10, 25).to("cuda"), "GPU-built": batch_from_arrays([lat], [pos], [z], 5.0, 4.0)}
for rep in range(2):
    for name, g in graphs.items():
        ms = timeit(model, g)
        per = bench.stage_times(model, lambda: model(g, forces=True, extras=False), 20)
        e = float(g[K.TOTAL_ENERGY][0])
        print(f"{name:10s} {ms:.4f} ms/step  E = {e:.6f}  " + "  ".join(f"{k} {m * c:.3f}" for k, (m, c) in per.items() if m * c > 0.02), flush=True)
ei_h, ei_g = graphs["host-built"][K.EDGE_INDEX], graphs["GPU-built"][K.EDGE_INDEX]
print("edge lists identical:", bool(torch.equal(ei_h.cpu(), ei_g.cpu())), " triplet lists identical:",
      bool(torch.equal(graphs["host-built"][K.TRIPLET_EDGE_INDEX].cpu(), graphs["GPU-built"][K.TRIPLET_EDGE_INDEX].cpu())))
