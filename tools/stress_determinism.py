#!/usr/bin/env python3
"""300 repeated evaluations of the 10,000-atom workload and of a 64 x 64-atom random batch: every force array must equal
the first bit for bit (longer companion of tests/test_gpu_determinism.py; run on the GPU box)."""
import sys, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'torch-m3gnet_amd'), str(ROOT / 'tests')]
from torch_m3gnet.data.synthetic import fcc_cu_graph, random_cell_graph
from torch_m3gnet.data import MaterialGraphKey as K
from torch_m3gnet.data.material_graph import Batch
from torch_m3gnet.model.build import build_model
torch.manual_seed(0)
model = build_model(5.0, 4.0, 3, 3, 95, 64, 3).cuda()
for name, g in (("cu10k", fcc_cu_graph(10, 10, 25).to("cuda")),
                ("rand64x64", Batch.from_data_list([random_cell_graph(64, 9.1, seed=s) for s in range(64)]).to("cuda"))):
    ref = model(g)[K.FORCES].clone()
    bad = 0
    for i in range(300):
        f = model(g)[K.FORCES]
        if not torch.equal(f, ref):
            bad += 1
    print(name, "mismatching repeats:", bad, "of 300", flush=True)
