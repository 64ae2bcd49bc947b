#!/usr/bin/env python3
"""Diff the MFMA edge-block path against the VALU baseline kernels on a large system (GPU box)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from torch_m3gnet.data.synthetic import fcc_cu_graph  # noqa: E402
from torch_m3gnet.data import MaterialGraphKey as K  # noqa: E402
from torch_m3gnet.model.build import build_model  # noqa: E402

cells = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (10, 10, 25)
torch.manual_seed(0)
model = build_model(5.0, 4.0, 3, 3, 95, 64, 3)
for m in model.model:
    if type(m).__name__ == "ThreeBodyInteration":
        m.nsb.factors = m.nsb.documented_factors()
g = fcc_cu_graph(*cells).to("cuda")
res = {}
for kern in (0, 1):
    model.engine.set_option("edge_kernel", kern)
    out = model(g.clone())
    torch.cuda.synchronize()
    res[kern] = {k: out[k].double().cpu() for k in (K.TOTAL_ENERGY, K.SCALED_ATOMIC_ENERGIES, K.NODE_FEATURES, K.EDGE_ATTR, K.FORCES)}
a, b = res[0], res[1]
print("E", a[K.TOTAL_ENERGY].item(), b[K.TOTAL_ENERGY].item())
for k in (K.SCALED_ATOMIC_ENERGIES, K.NODE_FEATURES, K.EDGE_ATTR, K.FORCES):
    d = (a[k] - b[k]).abs()
    d = d.reshape(d.shape[0], -1).max(1).values
    bad = torch.nonzero(d > 1e-5 * a[k].abs().max()).flatten()
    print(k, "max abs diff", d.max().item(), "scale", a[k].abs().max().item(), "rows off:", bad.numel(), bad[:20].tolist())
    if k == K.EDGE_ATTR and bad.numel():
        tiles = torch.unique(bad // 32)
        print("  tiles off:", tiles.numel(), tiles[:40].tolist())
        print("  src atoms of bad edges:", torch.unique(g[K.EDGE_INDEX][0].cpu()[bad])[:20].tolist())
