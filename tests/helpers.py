"""Shared test plumbing: fixtures -> oracle inputs and -> engine (torch_m3gnet) inputs."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from oracle import m3gnet_oracle as orc

GOLDEN = Path(__file__).resolve().parent / "golden"
CASE_MODEL = {"cu32": "model_default_seed0", "tio": "model_default_seed0", "alna": "model_small_seed0", "mix": "model_mix_seed3",
              # round 3: sheared / rotated / left-handed triclinic cell with atoms outside the home cell; the default model FITTED
              # (reference nn code, Adam) to Lennard-Jones energies and forces: |F| ~ 1 eV/A, activations no longer near-linear
              "tri": "model_default_seed0", "cu32fit": "model_fitted_lj", "mixfit": "model_fitted_lj",
              "cu32pair": "model_default_seed0"}   # (doc mode only: a triplet list of one pair)
CASES = [(c, m) for c in CASE_MODEL for m in ("ref", "doc") if c != "cu32pair"]
FITTED_CASES = [(c, m) for c, m in CASES if c.endswith("fit")]


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


def preactivation_stats(params, cfg, consts, graph):
    """Pre-activation statistics of every 64-wide Linear of the gated MLPs, in call order, from one oracle evaluation (checker):
    per layer (in_features, mean |p|, share of |p| > 2, share of |p| > 4, max |p|).  Per conv block the order is edge MLP
    {dense.0, gate.0, dense.2, gate.2}, node MLP likewise; the readout's layers come last."""
    stats = []
    orig = torch.nn.functional.linear

    def spy(x, w, b=None):
        y = orig(x, w, b)
        if w.shape[0] == cfg.embedding_dim and w.shape[1] in (cfg.embedding_dim, 3 * cfg.embedding_dim):
            a = y.detach().abs().double()
            stats.append((int(w.shape[1]), float(a.mean()), float((a > 2).double().mean()), float((a > 4).double().mean()), float(a.max())))
        return y

    torch.nn.functional.linear = spy
    try:
        orc.energy_forces(params, cfg, consts, graph, want_forces=False)
    finally:
        torch.nn.functional.linear = orig
    return stats


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b| (the metric SURVEY.md §8(d) prescribes for forces)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


# workload builders live in the package (bench.py uses them without importing this module, which imports the oracle)
from torch_m3gnet.data.synthetic import fcc_cu_graph, random_cell_arrays, random_cell_graph  # noqa: E402,F401
