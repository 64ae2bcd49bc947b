"""One MD-style iteration on the bench workload = GPU neighbour list + triplets + topology build + one energy/force step, each
from fresh positions.  Prints wall clock per phase; run under `rocprofv3 --kernel-trace` and pass the results .db to
`tools/kernel_trace_summary.py` for the per-kernel split."""
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

torch.manual_seed(0)
model = build_model(5.0, 4.0, 3, 3, 95, 64, 3)
a = 3.61
base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
gi = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(25), indexing="ij"), -1)
pos0 = (gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a
lat = np.diag([10 * a, 10 * a, 25 * a]).astype(float)
Z = np.full(len(pos0), 29)
rng = np.random.default_rng(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tb = ts = 0.0
for it in range(iters + 3):
    pos = pos0 + rng.uniform(-0.025, 0.025, pos0.shape)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g = batch_from_arrays([lat], [pos], [Z], 5.0, 4.0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    model(g, forces=True, extras=False)     # first call on a new graph: topology build + step
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if it >= 3:
        tb += t1 - t0
        ts += t2 - t1
print(f"10,000 atoms, E={g['num_edges']} T={g['num_triplets']}: graph build (host arrays -> device graph) {tb / iters * 1e3:.2f} ms, "
      f"topology + step {ts / iters * 1e3:.2f} ms per iteration")
