#!/usr/bin/env python3
"""BASELINE config 5 (high triplet density): 2,000 atoms uniform in L = 31.1 A (1.6 A rejection, seed 0), cutoff 6 A, with
three-body cutoff 4 A and 6 A.  Prints the step time and the three-body kernels' time, algorithmic HBM bytes/s and LDS read
rate per launch (HIP-event stage timers of the library).  Run once per build of the rows-per-workgroup sweep
(tools/sweep_config5.sh); one JSON line per three-body cutoff."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from torch_m3gnet.data.synthetic import random_cell_arrays  # noqa: E402
from torch_m3gnet.data.graph_gpu import batch_from_arrays  # noqa: E402
from torch_m3gnet.model.build import build_model  # noqa: E402
from torch_m3gnet.nn.modules import _Topology  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "base"
lat, pos, z = random_cell_arrays(2000, 31.1, seed=0)
for tb in (4.0, 6.0):
    torch.manual_seed(0)
    model = build_model(6.0, tb, 3, 3, 95, 64, 3).cuda()
    g = batch_from_arrays([lat], [pos], [z], 6.0, tb)
    import os
    for opt in filter(None, os.environ.get("M3G_ENGINE_OPTIONS", "").split(",")):   # e.g. threebody_moments=0
        model.engine.set_option(opt.split("=")[0], int(opt.split("=")[1]))
    for _ in range(3):
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    eng = model.engine
    eng.profile(True)
    for _ in range(n):
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    st = eng.profile_read()
    eng.profile(False)
    E, T, N = int(g["num_edges"]), int(g["num_triplets"]), int(g["num_nodes"])
    A = _Topology.of(g).n_active()
    C = 9
    rec = {"label": label, "threebody_cutoff": tb, "atoms": N, "edges": E, "active_edges": A, "triplets": T, "triplets_per_atom": T / N,
           "ms_per_step": ms, "atom_steps_per_s": N / ms * 1e3}
    for stage, hbm_bytes, lds_bytes in (
            # forward: per active row q (64) + u (12) + ids (8) + v gather (64) in, m (64) out; 1-byte partner id per triplet;
            # LDS: per triplet unit vector (12 B) + payload row (4 C) + id (1 B)
            ("threebody_fwd", A * (64 + 12 + 8 + 64 + 64) + T, T * (12 + 4 * C + 1)),
            # reverse: both halves -- rows q, q', dm, u, fc, v in, dg (64) + dd/du (16) out; two id bytes and two LDS visits per triplet
            ("threebody_rev", A * (64 + 64 + 64 + 12 + 8 + 64 + 64 + 16) + 2 * T, 2 * T * (12 + 4 * C + 1))):
        t_ms, cnt = st[stage]
        per = t_ms / cnt
        rec[stage] = {"ms_per_launch": per, "hbm_GBs": hbm_bytes / per / 1e6, "hbm_frac_of_8TBs": hbm_bytes / per / 1e6 / 8000.0,
                      "lds_read_TBs": lds_bytes / per / 1e9, "lds_frac_of_peak": lds_bytes / per / 1e9 / 78.6,
                      "triplets_per_us": T / per / 1e3}
    print(json.dumps(rec), flush=True)
