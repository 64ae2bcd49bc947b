#!/usr/bin/env python3
"""Repeat the engine on the golden cases and on a 1,500-atom cell and report the largest run-to-run force deviation
(forces involve no atomics: any deviation beyond 0 is a hazard or a race).  Run on the GPU box."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from helpers import CASES, build_engine_model, engine_graph, fcc_cu_graph, load_oracle_case  # noqa: E402
from torch_m3gnet.data import MaterialGraphKey as K  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
worst = 0.0
for case, mode in CASES:
    params, cfg, consts, graph, expect = load_oracle_case(case, mode)
    model, _ = build_engine_model(case, mode)
    g = engine_graph(graph)
    ref = model(g)[K.FORCES].clone()
    dev = 0.0
    for _ in range(reps):
        f = model(g)[K.FORCES]
        dev = max(dev, float((f - ref).abs().max()))
    scale = float(ref.abs().max())
    worst = max(worst, dev / scale)
    print(f"{case}_{mode}: max run-to-run |dF| / max|F| = {dev / scale:.2e}", flush=True)
from torch_m3gnet.model.build import build_model  # noqa: E402

torch.manual_seed(0)
model = build_model(5.0, 4.0, 3, 3, 95, 64, 3).cuda()
g = fcc_cu_graph(5, 5, 15).to("cuda")
ref = model(g)[K.FORCES].clone()
dev = 0.0
for _ in range(reps):
    dev = max(dev, float((model(g)[K.FORCES] - ref).abs().max()))
scale = float(ref.abs().max())
worst = max(worst, dev / scale)
print(f"cu1500: max run-to-run |dF| / max|F| = {dev / scale:.2e}", flush=True)
print("WORST", worst)
