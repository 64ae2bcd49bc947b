"""Shared test plumbing: fixtures -> oracle inputs and -> engine (torch_m3gnet) inputs."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from oracle import m3gnet_oracle as orc

GOLDEN = Path(__file__).resolve().parent / "golden"
CASE_MODEL = {"cu32": "model_default_seed0", "tio": "model_default_seed0", "alna": "model_small_seed0", "mix": "model_mix_seed3"}
CASES = [(c, m) for c in CASE_MODEL for m in ("ref", "doc")]


def load_oracle_case(case: str, mode: str, dtype=torch.float32):
    params, cfg, elemental = orc.load_model_npz(GOLDEN / f"{CASE_MODEL[case]}.npz", dtype=dtype)
    graph, expect = orc.load_case_npz(GOLDEN / f"case_{case}_{mode}.npz")
    consts = orc.make_constants(cfg, elemental, dtype=dtype)
    consts.factors = expect["const_factors"].to(dtype)  # captured constant (SURVEY finding 1)
    return params, cfg, consts, graph, expect


def build_engine_model(case: str, mode: str, device="cuda"):
    """torch_m3gnet (the product) model with the fixture's weights and captured constants."""
    from torch_m3gnet.model.build import build_model

    params, cfg, elemental = orc.load_model_npz(GOLDEN / f"{CASE_MODEL[case]}.npz")
    _, expect = orc.load_case_npz(GOLDEN / f"case_{case}_{mode}.npz")
    model = build_model(cfg.cutoff, cfg.threebody_cutoff, cfg.l_max, cfg.n_max, cfg.num_types, cfg.embedding_dim,
                        cfg.num_blocks, elemental_energies=elemental, energy_scale=cfg.energy_scale,
                        length_scale=cfg.length_scale)
    model.load_state_dict({k: v for k, v in params.items()})
    for mod in model.model:
        if type(mod).__name__ == "ThreeBodyInteration":
            mod.nsb.factors = expect["const_factors"].clone()
    return model, cfg


def engine_graph(graph: dict, device="cuda"):
    from torch_m3gnet.data.material_graph import Batch

    g = Batch.__new__(Batch)
    dict.__init__(g)
    for k, v in graph.items():
        g[k] = v.to(device)
    g["num_nodes"] = int(graph["pos"].size(0))
    return g


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b| (the metric SURVEY.md §8(d) prescribes for forces)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def fcc_cu_graph(nx, ny, nz, a=3.61, jitter=0.025, seed=0, cutoff=5.0, tb_cutoff=4.0):
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph

    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1)
    pos = (g.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a
    pos = pos + np.random.default_rng(seed).uniform(-jitter, jitter, pos.shape)
    lat = np.diag([nx * a, ny * a, nz * a]).astype(float)
    return Batch.from_data_list([MaterialGraph.from_arrays(lat, pos, np.full(len(pos), 29), cutoff, tb_cutoff)])


def random_cell_arrays(n_atoms, box, seed, zmax=94, dmin=1.6):
    """(lattice, cart_coords, Z) of one cubic cell of side `box` with `n_atoms` atoms placed uniformly under a `dmin`
    minimum-image rejection rule, species uniform in 1..zmax."""
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
    from torch_m3gnet.data.material_graph import MaterialGraph

    lat, pos, z = random_cell_arrays(n_atoms, box, seed, zmax=zmax, dmin=dmin)
    return MaterialGraph.from_arrays(lat, pos, z, cutoff, tb_cutoff)
