"""A/B timing of an engine option in one process: python tools/time_option.py <option> <v0> <v1> ...
Workload: the bench supercell, or M3G_CELLS="nx ny nz" fcc cells (e.g. "2 2 2" = BASELINE config 1, 32 atoms)."""
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402

model = bench.default_model(torch.device("cuda"))
cells = tuple(int(v) for v in os.environ.get("M3G_CELLS", "10 10 25").split())
from torch_m3gnet.data.synthetic import fcc_cu_graph  # noqa: E402

graph = fcc_cu_graph(*cells, seed=0).to("cuda")
if os.environ.get("M3G_PRECISION"):
    model.engine.set_precision(os.environ["M3G_PRECISION"])
opt, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
model(graph, forces=True, extras=False)
stream = torch.cuda.Stream() if os.environ.get("M3G_OWN_STREAM") else torch.cuda.current_stream()
torch.cuda.synchronize()
torch.cuda.set_stream(stream)
for rep in range(3):
    for v in vals:
        model.engine.set_option(opt, v)
        for _ in range(3):
            model(graph, forces=True, extras=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200 if cells[0] * cells[1] * cells[2] < 100 else 20
        for _ in range(n):
            model(graph, forces=True, extras=False)
        torch.cuda.synchronize()
        print(f"cells {cells} {opt}={v}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms/step", flush=True)
