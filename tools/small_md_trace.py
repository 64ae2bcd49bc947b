#!/usr/bin/env python3
"""A few MD iterations of one path on a small Cu cell, for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/smd -- python3 tools/small_md_trace.py <n_cells> <reuse|refill|search> [iterations]
    python tools/step_sequence.py /tmp/smd 2 k_verlet_update_small"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from torch_m3gnet.data.md import VerletGraph  # noqa: E402

n, mode = int(sys.argv[1]), sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dev = torch.device("cuda")
model = bench.default_model(dev)
base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
gi = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1)
p0 = torch.tensor((gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * 3.61, device=dev)
vg = VerletGraph([np.eye(3) * n * 3.61], [np.full(p0.size(0), 29)], 5.0, 4.0, skin=0.5, device=dev)
poss = [p0 + (torch.rand(p0.shape, device=dev, dtype=torch.float64) - 0.5) * 0.05 for _ in range(4)]
torch.cuda.synchronize()
for i in range(iters):
    model(vg.update(poss[i % 4], force=None if mode == "reuse" else mode), forces=True, extras=False)
torch.cuda.synchronize()
