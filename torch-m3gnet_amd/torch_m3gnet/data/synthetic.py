"""Synthetic workloads of the benchmark configurations (BASELINE.json `configs`): jittered fcc-Cu
supercells (configs 1 and 3) and random-species cubic cells under a minimum-distance rule (configs 2, 4, 5).  Used by
`bench.py`, the timing tools and the tests; host side, numpy only."""
from __future__ import annotations

import numpy as np

from .material_graph import Batch, MaterialGraph


def fcc_cu_arrays(nx, ny, nz, a=3.61, jitter=0.025, seed=0):
    """(lattice, cart_coords, Z) of an nx x ny x nz fcc Cu supercell, positions jittered uniformly by +-`jitter` A."""
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1)
    pos = (g.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a
    pos = pos + np.random.default_rng(seed).uniform(-jitter, jitter, pos.shape)
    lat = np.diag([nx * a, ny * a, nz * a]).astype(float)
    return lat, pos, np.full(len(pos), 29)


def fcc_cu_graph(nx, ny, nz, a=3.61, jitter=0.025, seed=0, cutoff=5.0, tb_cutoff=4.0):
    """The supercell as a one-structure Batch built on the host (config 1: 2 x 2 x 2, config 3: 10 x 10 x 25)."""
    lat, pos, z = fcc_cu_arrays(nx, ny, nz, a=a, jitter=jitter, seed=seed)
    return Batch.from_data_list([MaterialGraph.from_arrays(lat, pos, z, cutoff, tb_cutoff)])


def random_cell_arrays(n_atoms, box, seed, zmax=94, dmin=1.6):
    """(lattice, cart_coords, Z) of one cubic cell of side `box` with `n_atoms` atoms placed uniformly under a `dmin`
    minimum-image rejection rule, species uniform in 1..zmax (configs 2, 4, 5)."""
    rng = np.random.default_rng(seed)
    pos = np.zeros((0, 3))
    while len(pos) < n_atoms:
        p = rng.uniform(0, box, 3)
        dv = pos - p
        dv -= box * np.round(dv / box)
        if len(pos) == 0 or np.sqrt((dv**2).sum(1)).min() >= dmin:
            pos = np.vstack([pos, p])
    return np.eye(3) * box, pos, rng.integers(1, zmax + 1, n_atoms)


def random_cell_graph(n_atoms, box, seed, cutoff=5.0, tb_cutoff=4.0, zmax=94, dmin=1.6):
    lat, pos, z = random_cell_arrays(n_atoms, box, seed, zmax=zmax, dmin=dmin)
    return MaterialGraph.from_arrays(lat, pos, z, cutoff, tb_cutoff)
