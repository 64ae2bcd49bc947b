#!/usr/bin/env python3
"""Phase cycle sums of the fused reverse kernel (f16x3) from its stamped diagnostic variant (GPU box).
STAMP_WAVES = waves per workgroup of the build under test (M3G_WAVES_REV_FUSED, default 8)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from torch_m3gnet import _lib  # noqa: E402
from torch_m3gnet.data.synthetic import fcc_cu_graph  # noqa: E402
from torch_m3gnet.model.build import build_model  # noqa: E402

torch.manual_seed(0)
model = build_model(5.0, 4.0, 3, 3, 95, 64, 3)
g = fcc_cu_graph(10, 10, 25).to("cuda")
model.engine.set_precision("f16x3")
model(g)
eng = model.engine
eng.set_option("stamps", 3)
STEPS = 3
for _ in range(STEPS):
    model(g, extras=False)
torch.cuda.synchronize()
W = int(os.environ.get("STAMP_WAVES", "8"))
buf = np.zeros(256 * 16 * 12, dtype=np.uint64)
_lib.check(eng.lib.m3g_debug_read_stamps(eng.plan, buf.ctypes.data))
s = buf.reshape(256, 16, 12)[:, :W].astype(np.float64)
names = ["n: inputs, tables, layer 1", "n: activations, layer 2", "n: gating derivatives", "n: transposed, dense half", "n: transposed, gate half",
         "mid: dL/de, e, three-body MLP", "e: tables, layer 1", "e: activations, layer 2", "e: gating derivatives", "e: transposed, dense half",
         "e: transposed, gate half", "tail: store, three-body reverse"]
tot = s.sum(-1)
tiles_per_wave = 26250 / (256 * W) * 3 * STEPS   # three launches per step
print(f"{W} waves per workgroup; cycles per tile and wave {tot.mean() / tiles_per_wave:.0f} (s_memtime ticks of 100 MHz x ... as read)")
for i, n in enumerate(names):
    print(f"{n:34s} {s[..., i].mean() / tiles_per_wave:9.1f} per tile  {100 * s[..., i].sum() / tot.sum():5.1f} %")
per_launch = tot / (3 * STEPS)
print("busy cycles per wave and launch: mean %.0f  min %.0f  max %.0f  (max / mean %.3f)" % (per_launch.mean(), per_launch.min(), per_launch.max(), per_launch.max() / per_launch.mean()))
wg = per_launch.max(1)
print("slowest wave of a workgroup: mean %.0f  min %.0f  max %.0f" % (wg.mean(), wg.min(), wg.max()))
for x in range(8):
    sel = per_launch[x::8]
    print(f"  XCD label {x}: wave mean {sel.mean():.0f}  min {sel.min():.0f}  max {sel.max():.0f}")
order = np.argsort(wg)
print("five fastest workgroups (index, cycles):", [(int(i), int(wg[i])) for i in order[:5]], " five slowest:", [(int(i), int(wg[i])) for i in order[-5:]])
