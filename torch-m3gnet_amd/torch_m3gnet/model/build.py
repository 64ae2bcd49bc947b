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
