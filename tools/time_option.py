"""A/B timing of an engine option on the bench workload in one process: python tools/time_option.py <option> <v0> <v1> ..."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402

model = bench.default_model(torch.device("cuda"))
graph = bench.build_workload((10, 10, 25), 0, torch.device("cuda"))
opt, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
model(graph, forces=True, extras=False)
for rep in range(3):
    for v in vals:
        model.engine.set_option(opt, v)
        for _ in range(3):
            model(graph, forces=True, extras=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model(graph, forces=True, extras=False)
        torch.cuda.synchronize()
        print(f"{opt}={v}: {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms/step", flush=True)
