"""Per-stage device time (HIP events in the library) of the bench workload for each value of an engine option:
python tools/stage_times.py <option> <v0> <v1> ...   (M3G_PRECISION selects the mode, default fp32)"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from torch_m3gnet.data.synthetic import fcc_cu_graph  # noqa: E402

model = bench.default_model(torch.device("cuda"))
graph = fcc_cu_graph(10, 10, 25, seed=0).to("cuda")
opt, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
for v in vals:
    model.engine.set_option(opt, v)
    for _ in range(3):
        model(graph, forces=True, extras=False)
    per = bench.stage_times(model, lambda: model(graph, forces=True, extras=False), 20)
    tot = sum(ms * cnt for ms, cnt in per.values())
    print(f"{opt}={v}: sum of stages {tot:.3f} ms/step  " + "  ".join(f"{k} {ms * cnt:.3f}" for k, (ms, cnt) in per.items() if ms * cnt > 0.02), flush=True)
