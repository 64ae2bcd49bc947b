#!/usr/bin/env python3
"""MD-style iterations on the 10,000-atom Cu cell with device-resident positions (torch_m3gnet.data.md.VerletGraph), for a kernel
trace:   rocprofv3 --kernel-trace --stats -d gpurun_out/md -- python3 tools/profile_md_iteration.py [reuse|no_wait|refill|rebuild|step_reuse|step_refill|step_rebuild] [iterations] [precision]
Prints the wall time per iteration as well."""
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

mode = sys.argv[1] if len(sys.argv) > 1 else "reuse"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda")
model = bench.default_model(dev)
if len(sys.argv) > 3:
    model.engine.set_precision(sys.argv[3])
a = 3.61
base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
gi = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(25), indexing="ij"), -1)
pos0 = torch.tensor((gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a, device=dev)
lat = np.diag([10 * a, 10 * a, 25 * a]).astype(float)
vg = VerletGraph([lat], [np.full(pos0.size(0), 29)], 5.0, 4.0, skin=0.5, device=dev)
gen = torch.Generator(device=dev)
gen.manual_seed(0)


def iteration():
    pos = pos0 + (torch.rand(pos0.shape, generator=gen, device=dev, dtype=torch.float64) - 0.5) * 0.05
    if mode == "no_wait":
        vg.evaluate(model, pos, forces=True, extras=False)
    elif mode.startswith("step_"):   # one library call per step (VerletGraph.step -> m3g_md_step): step_reuse / step_refill / step_rebuild
        vg.step(model, pos, force={"step_rebuild": "search", "step_refill": "refill"}.get(mode))
    else:
        model(vg.update(pos, force={"rebuild": "search", "refill": "refill"}.get(mode)), forces=True, extras=False)


for _ in range(8):   # (evaluate() queues the step ahead of the verdict after four "unchanged" verdicts in a row)
    iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    iteration()
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration, paths {vg.stats}")
