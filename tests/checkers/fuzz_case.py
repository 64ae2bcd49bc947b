#!/usr/bin/env python3
"""One case of tests/checkers/fuzz_parity.py (same random stream) in the three precision modes against the fp32 and fp64 oracles:
    python tests/checkers/fuzz_case.py 84      (GPU box)"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from helpers import random_cell_graph
from oracle import m3gnet_oracle as orc
from test_gpu_properties import _oracle_inputs
from torch_m3gnet.data import MaterialGraphKey as K
from torch_m3gnet.data.material_graph import Batch
from torch_m3gnet.model.build import build_model
target = int(sys.argv[1])
rng = np.random.default_rng(2024)
for case in range(target + 1):
    l_max, n_max = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    if l_max * n_max > 16: n_max = 16 // l_max
    blocks = int(rng.integers(1, 5)); cutoff = float(rng.uniform(3.5, 6.0)); tb = float(rng.uniform(2.5, cutoff))
    torch.manual_seed(case)
    es, ls = float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.8, 1.5))
    ns = int(rng.integers(1, 5)); specs = []
    for s in range(ns):
        box = float(rng.uniform(4.5, 9.0)); n = int(rng.integers(1, max(2, min(40, int(box**3 / 14.0))))); specs.append((n, box, 1000 * case + s))
    if case < target: 
        # keep torch's RNG stream identical to the sweep: build_model consumes it
        continue
torch.manual_seed(target)
model = build_model(cutoff=cutoff, threebody_cutoff=tb, l_max=l_max, n_max=n_max, num_types=95, embedding_dim=64, num_blocks=blocks, energy_scale=es, length_scale=ls)
for m in model.model:
    if type(m).__name__ == "ThreeBodyInteration": m.nsb.factors = m.nsb.documented_factors()
graphs = [random_cell_graph(n, box, seed=sd, cutoff=cutoff, tb_cutoff=tb, dmin=1.4) for n, box, sd in specs]
print("specs", specs, "L R B", l_max, n_max, blocks)
b = Batch.from_data_list(graphs).to("cuda")
for prec in ("fp32", "f16x3", "bf16x3"):
    model.engine.set_precision(prec)
    g = model(b)
    p, cfg, c, og = _oracle_inputs(model, g)
    o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
    o64 = orc.energy_forces({k: v.double() if torch.is_floating_point(v) else v for k, v in p.items()}, cfg, c, {k: (v.double() if torch.is_tensor(v) and torch.is_floating_point(v) else v) for k, v in og.items()}, legendre_backward="exact") if prec == "fp32" else None
    print(prec, "gpu", g[K.TOTAL_ENERGY].cpu().tolist(), "oracle32", o["total_energy"].tolist())
    if o64 is not None: print("   oracle64", o64["total_energy"].tolist())
