#!/usr/bin/env python3
"""Print per-stage relative errors of the HIP engine against golden vectors and the fp64 oracle
(diagnostic companion of tests/test_gpu_parity.py; run on the GPU box)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from helpers import CASES, build_engine_model, engine_graph, load_oracle_case, rel_err  # noqa: E402
from oracle import m3gnet_oracle as orc, staged  # noqa: E402
from torch_m3gnet.data import MaterialGraphKey as K  # noqa: E402

for case, mode in CASES:
    params, cfg, consts, graph, expect = load_oracle_case(case, mode)
    model, _ = build_engine_model(case, mode)
    g = model(engine_graph(graph))
    torch.cuda.synchronize()
    p64, cfg64, c64, graph64, _ = load_oracle_case(case, mode, dtype=torch.float64)
    o = orc.energy_forces(p64, cfg64, c64, graph64, legendre_backward="exact")
    rows = [
        ("dist/g", g[K.EDGE_DISTANCES], expect["out_edge_distances"]),
        ("ew/g", g[K.EDGE_WEIGHTS], expect["out_edge_weights"]),
        ("x/g", g[K.NODE_FEATURES], expect["out_x"]),
        ("e/g", g[K.EDGE_ATTR], expect["out_edge_attr"]),
        ("Ea/g", g[K.SCALED_ATOMIC_ENERGIES], expect["out_scaled_atomic_energies"]),
        ("E/g", g[K.TOTAL_ENERGY], expect["out_total_energy"]),
        ("F/g", g[K.FORCES], expect["out_forces"]),
        ("S/g", g[K.STRESSES], expect["out_stresses"]),
        ("E/o64", g[K.TOTAL_ENERGY], o["total_energy"]),
        ("F/o64", g[K.FORCES], o["forces"]),
        ("S/o64", g[K.STRESSES], o["stresses"]),
    ]
    for b in range(cfg.num_blocks):
        rows.append((f"m{b}/o64", g[K.MID_EDGE_FEATURES][b], o[f"mid_edge_features_{b}"]))
    ang = float((g[K.TRIPLET_ANGLES].cpu() - expect["out_triplet_angles"]).abs().max())
    print(f"{case}_{mode}: ang_abs={ang:.1e} " + " ".join(f"{n}={rel_err(a, b):.1e}" for n, a, b in rows), flush=True)
