#!/usr/bin/env python3
"""Phase shares of the reverse edge-MLP kernel from its stamped diagnostic variant (GPU box)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from torch_m3gnet.data.synthetic import fcc_cu_graph  # noqa: E402
from torch_m3gnet import _lib  # noqa: E402
from torch_m3gnet.model.build import build_model  # noqa: E402

torch.manual_seed(0)
model = build_model(5.0, 4.0, 3, 3, 95, 64, 3)
g = fcc_cu_graph(10, 10, 25).to("cuda")
model(g)
eng = model.engine
eng.set_option("stamps", 2)
for _ in range(3):
    model(g, extras=False)
torch.cuda.synchronize()
buf = np.zeros(256 * 16 * 12, dtype=np.uint64)
_lib.check(eng.lib.m3g_debug_read_stamps(eng.plan, buf.ctypes.data))
s = buf.reshape(256, 16, 12).astype(np.float64)[:, :12, :]
names = {1: "tile loads + three-body recompute", 2: "table gather (+ load wait)", 3: "recompute both layers", 4: "gating derivatives",
         5: "layer-2^T chains + SiLU'", 6: "dp1 stores + layer-1^T chain", 7: "de store + three-body reverse + dm/dh"}
tot = s.sum(-1)
tiles_per_wave = 26250 / (256 * 12)
print("cycles per wave: mean %.0f min %.0f max %.0f; per tile %.0f" % (tot.mean(), tot.min(), tot.max(), tot.mean() / tiles_per_wave))
for i, n in names.items():
    print(f"{n:40s} {s[..., i].mean() / tiles_per_wave:9.0f} cyc/tile {100 * s[..., i].sum() / tot.sum():5.1f} %")
