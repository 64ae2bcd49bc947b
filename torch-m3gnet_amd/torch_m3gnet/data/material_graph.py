"""MaterialGraph / batch container without torch_geometric or pymatgen.

Tensor schema and batching rules follow the reference (data/material_graph.py:14-130): a graph is a
mapping from `MaterialGraphKey` names to tensors; batching concatenates node/edge/triplet tensors,
offsets `edge_index` by the running atom count and `triplet_edge_index` by the running edge count
(`__inc__`, :122-130), stacks `lattice` along a new leading dimension (`__cat_dim__` None, :109-120)
and adds a `batch` vector.  Graph construction uses this package's own periodic neighbour list
(`neighbors.py`) instead of pymatgen.
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

    def clone(self):
        return self._map(lambda t: t.clone())

    def cpu(self):
        return self.to("cpu")

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
