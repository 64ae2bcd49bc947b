#!/usr/bin/env python3
"""A few steps of BASELINE config 5 (2,000 atoms in L = 31.1 A, cutoff 6 A, three-body cutoff 4 or 6 A) for a kernel trace:
rocprofv3 --kernel-trace --output-format csv -d /tmp/c5 -- python3 tools/config5_trace.py [tb_cutoff]; python tools/step_sequence.py /tmp/c5 2"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
from torch_m3gnet.data.graph_gpu import batch_from_arrays  # noqa: E402
from torch_m3gnet.data.synthetic import random_cell_arrays  # noqa: E402
from torch_m3gnet.model.build import build_model  # noqa: E402

tb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
lat, pos, z = random_cell_arrays(2000, 31.1, seed=0)
torch.manual_seed(0)
model = build_model(6.0, tb, 3, 3, 95, 64, 3).cuda()
g = batch_from_arrays([lat], [pos], [z], 6.0, tb)
for _ in range(10):
    model(g, forces=True, extras=False)
torch.cuda.synchronize()
