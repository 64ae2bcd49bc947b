"""MaterialGraph / batch container without torch_geometric or pymatgen.

Tensor schema and batching rules follow the reference (data/material_graph.py:14-130): a graph is a
mapping from `MaterialGraphKey` names to tensors; batching concatenates node/edge/triplet tensors,
offsets `edge_index` by the running atom count and `triplet_edge_index` by the running edge count
(`__inc__`, :122-130), stacks `lattice` along a new leading dimension (`__cat_dim__` None, :109-120)
and adds a `batch` vector.  Graph construction uses this package's own periodic neighbour list
(`neighbors.py`) instead of pymatgen; the reference's entry points keep their names and signatures:
`MaterialGraph.from_structure` (:132-166, any object with pymatgen's `lattice.matrix`, `cart_coords` and
`atomic_numbers` / per-site `specie.Z`), `get_all_neighbors_with_cell_shifts` (:168-193), `compute_threebody` (:196-254).
"""
from __future__ import annotations

from typing import Any, Iterable

import numpy as np
import torch

from . import MaterialGraphKey as K
from .neighbors import neighbor_list, threebody_index

_PER_GRAPH_STACK = (K.LATTICE, K.TOTAL_ENERGY, K.STRESSES)


class MaterialGraph(dict):
    """Keyed tensor container (dict with attribute access), the `graph` of `forward(graph) -> graph`."""

    def __init__(self, pos=None, atom_types=None, num_triplet_i=None, edge_index=None, edge_cell_shift=None,
                 num_triplet_ij=None, triplet_edge_index=None, lattice=None, **extra):
        super().__init__()
        self[K.POS] = pos
        self[K.ATOM_TYPES] = atom_types
        self[K.NUM_TRIPLET_I] = num_triplet_i
        self[K.EDGE_INDEX] = edge_index
        self[K.EDGE_CELL_SHIFT] = edge_cell_shift
        self[K.NUM_TRIPLET_IJ] = num_triplet_ij
        self[K.TRIPLET_EDGE_INDEX] = triplet_edge_index
        self[K.LATTICE] = lattice
        self[K.NUM_NODES] = int(pos.size(0)) if pos is not None else 0
        self[K.NUM_EDGES] = int(edge_index.size(1)) if edge_index is not None else 0
        self[K.NUM_TRIPLETS] = int(triplet_edge_index.size(1)) if triplet_edge_index is not None else 0
        self.update(extra)

    # attribute access like torch_geometric's Data
    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError as exc:
            raise AttributeError(name) from exc

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = value

    def _map(self, fn):
        out = type(self).__new__(type(self))
        dict.__init__(out)
        for k, v in self.items():
            out[k] = fn(v) if isinstance(v, torch.Tensor) else v
        return out

    def to(self, device, non_blocking: bool = False):
        return self._map(lambda t: t.to(device, non_blocking=non_blocking))

    def pin_memory(self):
        """Host tensors in page-locked memory, so that `.to(device, non_blocking=True)` is one asynchronous copy per tensor queued
        on the current stream (pinned host -> device staging)."""
        return self._map(lambda t: t.pin_memory() if t.device.type == "cpu" else t)

    def clone(self):
        return self._map(lambda t: t.clone())

    def cpu(self):
        return self.to("cpu")

    @classmethod
    def from_structure(cls, structure, cutoff: float, threebody_cutoff: float) -> "MaterialGraph":
        """The reference's constructor (data/material_graph.py:132-166) for any structure-like object: pymatgen's
        `Structure`, or anything exposing `lattice.matrix` [3,3], `cart_coords` [n,3] and the atomic numbers (see
        `_atomic_numbers`).  No pymatgen import: the neighbour search is this package's own."""
        lattice, pos, z = _structure_arrays(structure)
        return cls.from_arrays(lattice, pos, z, cutoff, threebody_cutoff)

    @classmethod
    def from_arrays(cls, lattice, cart_coords, atomic_numbers, cutoff: float, threebody_cutoff: float) -> "MaterialGraph":
        """Build a graph from a periodic cell (replaces `from_structure`, material_graph.py:132-166).
        `atomic_numbers` are Z (1-based); `atom_types` = Z - 1 as in the reference (:147)."""
        if threebody_cutoff > cutoff:
            raise ValueError("Three body cutoff raidus should be smaller than two body.")
        lattice = np.asarray(lattice, dtype=np.float64).reshape(3, 3)
        pos = np.asarray(cart_coords, dtype=np.float64).reshape(-1, 3)
        z = np.asarray(atomic_numbers).reshape(-1)
        ei, shift, dist = neighbor_list(lattice, pos, cutoff)
        # the reference thresholds fp32 distances (material_graph.py:191,224)
        tei, nti, ntij = threebody_index(len(pos), ei, dist.astype(np.float32), threebody_cutoff)
        return cls(
            pos=torch.tensor(pos, dtype=torch.float), atom_types=torch.tensor(z - 1, dtype=torch.long),
            num_triplet_i=torch.tensor(nti), edge_index=torch.tensor(ei, dtype=torch.long),
            edge_cell_shift=torch.tensor(shift, dtype=torch.int), num_triplet_ij=torch.tensor(ntij),
            triplet_edge_index=torch.tensor(tei, dtype=torch.long), lattice=torch.tensor(lattice, dtype=torch.float),
        )


def _atomic_numbers(structure) -> np.ndarray:
    """Z of every site: `structure.atomic_numbers` (pymatgen Structure / IStructure), else `site.specie.Z` per site as the
    reference reads them (data/material_graph.py:147)."""
    z = getattr(structure, "atomic_numbers", None)
    if z is None:
        z = [site.specie.Z for site in structure]
    return np.asarray(z, dtype=np.int64).reshape(-1)


def _structure_arrays(structure):
    lattice = np.array(structure.lattice.matrix, dtype=np.float64).reshape(3, 3)
    pos = np.asarray(structure.cart_coords, dtype=np.float64).reshape(-1, 3)
    return lattice, pos, _atomic_numbers(structure)


def get_all_neighbors_with_cell_shifts(structure, cutoff: float):
    """Full periodic neighbour list of a structure-like object: (edge_index [2,E] long, edge_cell_shift [E,3] int,
    distances [E] float) -- the reference's tuple (data/material_graph.py:168-193), in the canonical order of
    `neighbors.neighbor_list` (sorted by centre; pymatgen leaves the order inside a centre unspecified)."""
    lattice, pos, _ = _structure_arrays(structure)
    ei, shift, dist = neighbor_list(lattice, pos, cutoff)
    return torch.tensor(ei, dtype=torch.long), torch.tensor(shift, dtype=torch.int), torch.tensor(dist, dtype=torch.float)


def compute_threebody(num_nodes: int, edge_index: torch.Tensor, distances: torch.Tensor, threebody_cutoff: float):
    """(triplet_edge_index [2,T] long, num_triplet_i [N] long, num_triplet_ij [E] int) for a centre-sorted edge list: every
    ordered pair of distinct edges within `threebody_cutoff` that share a centre, in the reference's loop order
    (data/material_graph.py:196-254), without its O(T) Python loop."""
    ei = edge_index.detach().cpu().numpy()
    d = distances.detach().cpu().numpy()
    tei, nti, ntij = threebody_index(int(num_nodes), ei, d, threebody_cutoff)
    return torch.tensor(tei, dtype=torch.long), torch.tensor(nti), torch.tensor(ntij, dtype=torch.int)


class Batch(MaterialGraph):
    """Several graphs concatenated into one (what the reference obtains from PyG's Batch)."""

    @classmethod
    def from_data_list(cls, graphs: Iterable[MaterialGraph]) -> "Batch":
        graphs = list(graphs)
        if not graphs:
            raise ValueError("from_data_list needs at least one graph")
        out = cls.__new__(cls)
        dict.__init__(out)
        n_off = e_off = 0
        cols: dict[str, list] = {}
        batch = []
        for s, g in enumerate(graphs):
            n, e = int(g[K.NUM_NODES]), int(g[K.NUM_EDGES])
            for key, val in g.items():
                if not isinstance(val, torch.Tensor):
                    continue
                if key == K.EDGE_INDEX:
                    val = val + n_off
                elif key == K.TRIPLET_EDGE_INDEX:
                    val = val + e_off
                elif key == K.BATCH:
                    continue
                elif key in _PER_GRAPH_STACK:
                    val = val.unsqueeze(0) if (key != K.LATTICE or val.dim() == 2) else val
                cols.setdefault(key, []).append(val)
            batch.append(torch.full((n,), s, dtype=torch.long, device=g[K.POS].device))
            n_off += n
            e_off += e
        for key, vals in cols.items():
            if len(vals) != len(graphs):
                continue  # key not present in every graph
            dim = 1 if key in (K.EDGE_INDEX, K.TRIPLET_EDGE_INDEX) else 0
            out[key] = torch.cat(vals, dim=dim)
        out[K.BATCH] = torch.cat(batch)
        out[K.NUM_NODES] = n_off
        out[K.NUM_EDGES] = e_off
        out[K.NUM_TRIPLETS] = int(out[K.TRIPLET_EDGE_INDEX].size(1))
        out["num_graphs"] = len(graphs)
        return out


BatchMaterialGraph = Batch
