#!/usr/bin/env python3
"""Velocity-Verlet NVE dynamics on the MI355X engine with the positions resident on the device.

    python examples/md_nve.py [steps] [dt_fs] [temperature_K] [precision fp32|f16x3|bf16x3]

The model is the default M3GNet architecture with the LJ-fitted fixture weights (tests/golden/model_fitted_lj.npz: fitted with
the reference's own code to Lennard-Jones Cu) on a 4 x 4 x 4 fcc Cu cell (256 atoms).  Per step: `VerletGraph.evaluate` queues
the skin-list test and the energy / force evaluation behind it and reads the test's verdict afterwards; the neighbour / triplet
lists, the CSR topology and its certificate are rebuilt only when a pair has crossed a cutoff.  Prints the energy bookkeeping
and which list paths were taken.  (The reference model's energy jumps where a pair crosses the two-body cutoff -- its radial
basis does not vanish there -- so the total energy of a hot crystal drifts by those jumps, on the reference CPU path as well.)"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
from torch_m3gnet.data import MaterialGraphKey as K  # noqa: E402
from torch_m3gnet.data.md import VerletGraph  # noqa: E402
from torch_m3gnet.model.build import build_model_from_npz  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
temperature = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
model = build_model_from_npz(ROOT / "tests" / "golden" / "model_fitted_lj.npz")   # (weights as data; nothing of the test code runs here)
cutoff, threebody_cutoff = 5.0, 4.0
if len(sys.argv) > 4:
    model.engine.set_precision(sys.argv[4])

a, n = 3.61, 4
base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
gi = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1)
pos0 = (gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a
lat = np.eye(3) * n * a
n_atoms = len(pos0)
mass, kB, acc_unit = 63.546, 8.617333e-5, 9.64853e-3   # amu, eV/K, (eV/A/amu) -> A/fs^2
rng = np.random.default_rng(0)
dev = torch.device("cuda")
pos = torch.tensor(pos0 + rng.normal(0, 0.01, pos0.shape), device=dev)
vel = torch.tensor(rng.normal(0, np.sqrt(kB * temperature / mass * acc_unit), pos0.shape), device=dev)
vel -= vel.mean(0, keepdim=True)
vg = VerletGraph([lat], [np.full(n_atoms, 29)], cutoff, threebody_cutoff, skin=0.4, device=dev)


def energy_forces(p):
    out = vg.evaluate(model, p, extras=False)
    return out[K.TOTAL_ENERGY].double().sum(), out[K.FORCES].double().clone()


e_pot, f = energy_forces(pos)
e0 = None
torch.cuda.synchronize()
t0 = time.perf_counter()
for step in range(steps):
    vel = vel + 0.5 * dt * acc_unit / mass * f
    pos = pos + dt * vel
    e_pot, f = energy_forces(pos)
    vel = vel + 0.5 * dt * acc_unit / mass * f
    if step % max(1, steps // 10) == 0 or step == steps - 1:
        e_kin = float(0.5 * mass / acc_unit * (vel * vel).sum())
        e_tot = float(e_pot) + e_kin
        e0 = e_tot if e0 is None else e0
        print(f"step {step:5d}  E_pot {float(e_pot):12.5f} eV  E_kin {e_kin:9.5f} eV  E_tot - E_tot(0) {e_tot - e0:+.2e} eV  "
              f"T {2 * e_kin / (3 * n_atoms * kB):7.1f} K  edges {int(vg.graph[K.NUM_EDGES])}")
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{steps} steps of {n_atoms} atoms in {wall:.2f} s = {wall / steps * 1e3:.3f} ms per step ({model.engine.precision}); list paths {vg.stats}")
