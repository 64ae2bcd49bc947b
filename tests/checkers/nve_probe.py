#!/usr/bin/env python3
"""Checker (imports the test helpers, which import the oracle): NVE energy drift of the 108-atom Cu cell on the LJ-fitted weights for several time steps / list strategies."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from helpers import build_engine_model
from torch_m3gnet.data import MaterialGraphKey as K
from torch_m3gnet.data.md import VerletGraph
from torch_m3gnet.data.graph_gpu import batch_from_arrays

case = sys.argv[1] if len(sys.argv) > 1 else "cu32fit"
model, _ = build_engine_model(case, "ref")
JIT = float(sys.argv[5]) if len(sys.argv) > 5 else 0.03
RC, R3 = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (5.0, 4.0)
if (RC, R3) != (5.0, 4.0):   # the same weights under other cutoffs (a different, equally valid potential)
    from torch_m3gnet.model.build import build_model
    from oracle import m3gnet_oracle as orc
    from helpers import GOLDEN, CASE_MODEL
    params, cfg, elemental = orc.load_model_npz(GOLDEN / f"{CASE_MODEL[case]}.npz")
    m2 = build_model(RC, R3, cfg.l_max, cfg.n_max, cfg.num_types, cfg.embedding_dim, cfg.num_blocks, elemental_energies=elemental,
                     energy_scale=cfg.energy_scale, length_scale=cfg.length_scale)
    m2.load_state_dict({k: v for k, v in params.items()})
    model = m2
a = float(sys.argv[2]) if len(sys.argv) > 2 else 3.61
base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
gi = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing="ij"), -1)
lat = np.eye(3) * 3 * a
mass, kB, acc_unit = 63.546, 8.617333e-5, 9.64853e-3
for dt, mode, temp in ((2.0, "evaluate", 50), (1.0, "evaluate", 50), (0.5, "evaluate", 50), (4.0, "evaluate", 50)):
    rng = np.random.default_rng(9)
    pos0 = (gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a + rng.normal(0, JIT, (108, 3))
    vg = VerletGraph([lat], [np.full(108, 29)], RC, R3, skin=0.3, device="cuda")
    pos = torch.tensor(pos0, device="cuda")
    vel = torch.tensor(rng.normal(0, np.sqrt(kB * temp / mass * acc_unit), (108, 3)), device="cuda")
    vel -= vel.mean(0, keepdim=True)
    def ef(p):
        if mode == "fresh":
            out = model(batch_from_arrays([lat], [p.cpu().numpy()], [np.full(108, 29)], RC, R3, device="cuda"), extras=False)
        else:
            out = vg.evaluate(model, p, extras=False)
        return out[K.TOTAL_ENERGY].double().sum(), out[K.FORCES].double().clone()
    e, f = ef(pos)
    tot, pots, fm, ne = [], [], [], set()
    n = int(300 / dt)
    for step in range(n):
        vel = vel + 0.5 * dt * acc_unit / mass * f
        pos = pos + dt * vel
        e, f = ef(pos)
        vel = vel + 0.5 * dt * acc_unit / mass * f
        ne.add((int(vg.graph[K.NUM_EDGES]), int(vg.graph[K.NUM_TRIPLETS])))
        tot.append(float(e + 0.5 * mass / acc_unit * (vel * vel).sum())); pots.append(float(e)); fm.append(float(f.abs().max()))
    tot, pots = np.array(tot), np.array(pots)
    print(f"dt {dt} {mode} T {temp}: swing {pots.max()-pots.min():.3f} eV, |drift|max {np.abs(tot-tot[0]).max():.3e}, end drift {tot[-1]-tot[0]:+.3e}, max|F| {max(fm):.2f}, paths {vg.stats}, distinct E {len(set(x[0] for x in ne))} T {len(set(x[1] for x in ne))}", flush=True)
