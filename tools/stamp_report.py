#!/usr/bin/env python3
"""Phase shares of the forward MFMA edge kernel from the stamped diagnostic variant (GPU box)."""
import ctypes as C
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
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
model.engine.set_precision(prec)
model(g)
eng = model.engine
eng.set_option("stamps", 1)
for _ in range(3):
    model(g, extras=False)
torch.cuda.synchronize()
import os
W = int(os.environ.get("STAMP_WAVES", "16"))   # waves per workgroup of the build under test (M3G_WAVES_FWD)
buf = np.zeros(256 * 16 * 12, dtype=np.uint64)
_lib.check(eng.lib.m3g_debug_read_stamps(eng.plan, buf.ctypes.data))
s = buf.reshape(256, 16, 12)[:, :W].astype(np.float64)   # [workgroup][16 wave slots][12]: the first W slots are written
names = ["tile loads", "three-body MLP", "e: table gather", "e: both layers", "(unused)", "e: gating",
         "e2 residual+store", "n: table gather", "n: both layers", "(unused)", "n: gating", "message sums"]
print("precision", prec)
tot = s.sum(-1)
print("cycles per wave (mean / min / max over all waves):", tot.mean(), tot.min(), tot.max())
tiles_per_wave = 26250 / (256 * W)
for i, n in enumerate(names):
    print(f"{n:22s} {s[..., i].mean() / tiles_per_wave:10.0f} cyc/tile  {100 * s[..., i].sum() / tot.sum():5.1f} %")
wg = tot.max(1)          # a workgroup ends with its slowest wave
print("per-WG end (cycles): mean %.0f min %.0f max %.0f" % (wg.mean(), wg.min(), wg.max()))
for x in range(8):
    sel = wg[x::8]
    print(f"  XCD label {x}: mean {sel.mean():.0f} min {sel.min():.0f} max {sel.max():.0f}  waves/wg mean {tot[x::8].mean():.0f}")
print("wave totals within WG 0:", tot[0].astype(int).tolist())
print("wave totals within WG 9:", tot[9].astype(int).tolist())
