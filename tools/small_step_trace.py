#!/usr/bin/env python3
"""A few steps of a small fcc Cu cell (n x n x n conventional cells; n = 2 is BASELINE configs[0]) for a kernel trace:
rocprofv3 --kernel-trace --output-format csv -d /tmp/small -- python3 tools/small_step_trace.py [precision [n]] ;
python tools/step_sequence.py /tmp/small 2"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from torch_m3gnet.data.synthetic import fcc_cu_graph  # noqa: E402

model = bench.default_model(torch.device("cuda"))
if len(sys.argv) > 1:
    model.engine.set_precision(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
import os  # noqa: E402
if os.environ.get("M3G_SMALL_TILES"):   # threshold of the split-tile edge kernels (plan option "small_tiles")
    model.engine.set_option("small_tiles", int(os.environ["M3G_SMALL_TILES"]))
g = fcc_cu_graph(n, n, n).to("cuda")
for _ in range(30):
    model(g, forces=True, extras=False)
torch.cuda.synchronize()
