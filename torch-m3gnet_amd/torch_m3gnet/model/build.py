"""`build_model`: same signature and module order as the reference (model/build.py:16-83), so the
returned object has the reference's `state_dict` key layout (`model.<index>.<attr>`)."""
from __future__ import annotations

import torch

from ..nn import modules as nn


def build_model(
    cutoff: float,
    threebody_cutoff: float,
    l_max: int,
    n_max: int,
    num_types: int,
    embedding_dim: int,
    num_blocks: int,
    elemental_energies: torch.Tensor | None = None,
    energy_scale: float = 1.0,  # eV
    length_scale: float = 1.0,  # Angstrom
    device: torch.device | None = None,
) -> nn.Gradient:
    if elemental_energies is None:
        elemental_energies = torch.zeros(num_types, device=device)
    rc, rc3 = cutoff / length_scale, threebody_cutoff / length_scale
    stages = [
        nn.ScaleLength(length_scale=length_scale),
        nn.AtomRef(elemental_energies, device=device),
        nn.DistanceAndAngle(),
        nn.AtomFeaturizer(num_types=num_types, embedding_dim=embedding_dim, device=device),
        nn.EdgeFeaturizer(degree=n_max, cutoff=rc, device=device),
        nn.EdgeAdjustor(degree=n_max, num_edge_features=embedding_dim, device=device),
    ]
    for _ in range(num_blocks):
        stages.append(nn.ThreeBodyInteration(cutoff=rc, threebody_cutoff=rc3, l_max=l_max, n_max=n_max,
                                             num_node_features=embedding_dim, num_edge_features=embedding_dim, device=device))
        stages.append(nn.M3GNetConv(degree=n_max, num_node_features=embedding_dim, num_edge_features=embedding_dim, device=device))
    stages.append(nn.AtomWiseReadout(in_features=embedding_dim, num_layers=3, scale=energy_scale, device=device))
    return nn.Gradient(torch.nn.Sequential(*stages))


def build_model_from_npz(path, cutoff: float | None = None, threebody_cutoff: float | None = None) -> nn.Gradient:
    """A model from an `.npz` archive that holds a `state_dict` (keys `model.<index>.<attr>`, as the reference's checkpoints name
    them), the `build_model` arguments as `__cfg_<name>` scalars and `__elemental_energies` -- the format of tests/golden/model_*.npz.
    `cutoff` / `threebody_cutoff` override the archive's (the same weights under other cutoffs are another, equally valid
    potential)."""
    import numpy as np

    z = np.load(path)
    cfg = {k[6:]: z[k].item() for k in z.files if k.startswith("__cfg_")}
    for k in ("l_max", "n_max", "num_types", "embedding_dim", "num_blocks"):
        cfg[k] = int(cfg[k])
    if cutoff is not None:
        cfg["cutoff"] = float(cutoff)
    if threebody_cutoff is not None:
        cfg["threebody_cutoff"] = float(threebody_cutoff)
    elemental = torch.tensor(z["__elemental_energies"], dtype=torch.float) if "__elemental_energies" in z.files else None
    model = build_model(elemental_energies=elemental, **cfg)
    model.load_state_dict({k: torch.tensor(z[k]) for k in z.files if not k.startswith("__")})
    return model
