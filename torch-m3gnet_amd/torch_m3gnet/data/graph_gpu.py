"""Graph construction on the GPU: periodic neighbour list + three-body indices (C ABI: m3g_neighbor_*,
m3g_threebody_*; csrc/m3g_graph_build.hip).  Replaces the reference's pymatgen search and its O(T) Python
triplet loop (data/material_graph.py:168-254) for whole batches of structures in one call."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np
import torch

from .. import _cuda, _lib
from . import MaterialGraphKey as K
from .material_graph import Batch


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_stream = _cuda.stream_ptr


def max_images(lattices: np.ndarray, cutoff: float) -> int:
    """Upper bound of (2rx+1)(2ry+1)(2rz+1) over the structures (same formula as the kernel)."""
    lat = np.asarray(lattices, dtype=np.float64).reshape(-1, 3, 3)
    vol = np.abs(np.linalg.det(lat))
    n = np.ones(len(lat), dtype=np.int64)
    for p in range(3):
        area = np.linalg.norm(np.cross(lat[:, (p + 1) % 3], lat[:, (p + 2) % 3]), axis=1)
        n *= 2 * np.ceil((cutoff + 1e-8) * area / vol).astype(np.int64) + 1
    return int(n.max()) if len(n) else 1


def neighbor_list_gpu(lattice: torch.Tensor, pos: torch.Tensor, batch: torch.Tensor, cutoff: float, host_lattice=None):
    """lattice [S,3,3] f64, pos [N,3] f64, batch [N] i64 (sorted), all on the GPU; host_lattice: the same lattices as a host
    array when the caller has them (saves the device read-back that sizes the image loop).
    Returns edge_index [2,E] i64, edge_cell_shift [E,3] i32, distances [E] f64 (device tensors)."""
    lib = _lib.load_library()
    dev = pos.device
    lattice = lattice.to(torch.float64).contiguous()
    pos = pos.to(torch.float64).contiguous()
    batch = batch.to(torch.int64).contiguous()
    N, S = int(pos.size(0)), int(lattice.size(0))
    M = max_images(lattice.cpu().numpy() if host_lattice is None else np.asarray(host_lattice, dtype=np.float64).reshape(-1, 3, 3), cutoff)
    nbytes = C.c_size_t()
    _lib.check(lib.m3g_neighbor_scratch_bytes(N, S, M, C.byref(nbytes)))
    scratch = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    n_edges = C.c_int64()
    with _cuda.on_device(dev):
        _lib.check(lib.m3g_neighbor_count(N, S, M, _ptr(pos), _ptr(lattice), _ptr(batch), float(cutoff), _ptr(scratch), nbytes.value,
                                          C.byref(n_edges), _stream()))
        E = int(n_edges.value)
        ei = torch.empty(2, E, dtype=torch.int64, device=dev)
        shift = torch.empty(E, 3, dtype=torch.int32, device=dev)
        dist = torch.empty(E, dtype=torch.float64, device=dev)
        _lib.check(lib.m3g_neighbor_fill(N, S, M, _ptr(batch), float(cutoff), _ptr(scratch), E, _ptr(ei), _ptr(shift), _ptr(dist), _stream()))
    # no wait here: `scratch` goes back to torch's caching allocator, which hands a block out again only to work queued on the
    # same stream behind the fill kernel
    return ei, shift, dist


def threebody_index_gpu(num_nodes: int, edge_index: torch.Tensor, distances: torch.Tensor, threebody_cutoff: float):
    """edge_index [2,E] i64 (sorted by centre), distances [E] (narrowed to fp32 like the reference).
    Returns triplet_edge_index [2,T] i64, num_triplet_i [N] i64, num_triplet_ij [E] i32 (device tensors)."""
    lib = _lib.load_library()
    dev = edge_index.device
    ei = edge_index.to(torch.int64).contiguous()
    d32 = distances.to(torch.float32).contiguous()
    N, E = int(num_nodes), int(ei.size(1))
    nbytes = C.c_size_t()
    _lib.check(lib.m3g_threebody_scratch_bytes(N, E, C.byref(nbytes)))
    scratch = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    n_t = C.c_int64()
    with _cuda.on_device(dev):
        _lib.check(lib.m3g_threebody_count(N, E, _ptr(ei), _ptr(d32), float(threebody_cutoff), _ptr(scratch), nbytes.value,
                                           C.byref(n_t), _stream()))
        T = int(n_t.value)
        tei = torch.empty(2, T, dtype=torch.int64, device=dev)
        nti = torch.empty(N, dtype=torch.int64, device=dev)
        ntij = torch.empty(E, dtype=torch.int32, device=dev)
        _lib.check(lib.m3g_threebody_fill(N, E, _ptr(ei), _ptr(scratch), T, _ptr(tei), _ptr(nti), _ptr(ntij), _stream()))
    return tei, nti, ntij   # (no wait: see neighbor_list_gpu)


def graph_indices_gpu(lattice: torch.Tensor, pos: torch.Tensor, batch: torch.Tensor, cutoff: float, threebody_cutoff: float,
                      host_lattice=None):
    """neighbor_list_gpu + threebody_index_gpu with ONE wait for the device: the neighbour count pass also counts the triplets
    (m3g_neighbor_count_triplets), so every tensor is sized at once and the fills are queued back to back.
    Returns (edge_index, edge_cell_shift, distances f64, triplet_edge_index, num_triplet_i, num_triplet_ij)."""
    lib = _lib.load_library()
    dev = pos.device
    lattice = lattice.to(torch.float64).contiguous()
    pos = pos.to(torch.float64).contiguous()
    batch = batch.to(torch.int64).contiguous()
    N, S = int(pos.size(0)), int(lattice.size(0))
    M = max_images(lattice.cpu().numpy() if host_lattice is None else np.asarray(host_lattice, dtype=np.float64).reshape(-1, 3, 3), cutoff)
    nb_bytes = C.c_size_t()
    _lib.check(lib.m3g_neighbor_scratch_bytes(N, S, M, C.byref(nb_bytes)))
    scratch = torch.empty(nb_bytes.value, dtype=torch.uint8, device=dev)
    n_edges, n_trip = C.c_int64(), C.c_int64()
    with _cuda.on_device(dev):
        _lib.check(lib.m3g_neighbor_count_triplets(N, S, M, _ptr(pos), _ptr(lattice), _ptr(batch), float(cutoff), float(threebody_cutoff),
                                                   _ptr(scratch), nb_bytes.value, C.byref(n_edges), C.byref(n_trip), _stream()))
        E, T = int(n_edges.value), int(n_trip.value)
        ei = torch.empty(2, E, dtype=torch.int64, device=dev)
        shift = torch.empty(E, 3, dtype=torch.int32, device=dev)
        dist = torch.empty(E, dtype=torch.float64, device=dev)
        tei = torch.empty(2, T, dtype=torch.int64, device=dev)
        nti = torch.empty(N, dtype=torch.int64, device=dev)
        ntij = torch.empty(E, dtype=torch.int32, device=dev)
        _lib.check(lib.m3g_neighbor_fill(N, S, M, _ptr(batch), float(cutoff), _ptr(scratch), E, _ptr(ei), _ptr(shift), _ptr(dist), _stream()))
        d32 = dist.to(torch.float32)
        tb_bytes = C.c_size_t()
        _lib.check(lib.m3g_threebody_scratch_bytes(N, E, C.byref(tb_bytes)))
        tb_scratch = torch.empty(tb_bytes.value, dtype=torch.uint8, device=dev)
        _lib.check(lib.m3g_threebody_build(N, E, _ptr(ei), _ptr(d32), float(threebody_cutoff), _ptr(tb_scratch), tb_bytes.value, T,
                                           _ptr(tei), _ptr(nti), _ptr(ntij), _stream()))
    return ei, shift, dist, tei, nti, ntij


def batch_from_arrays(lattices: Sequence, cart_coords: Sequence, atomic_numbers: Sequence, cutoff: float, threebody_cutoff: float,
                      device="cuda") -> Batch:
    """Build a whole batch of periodic structures on the GPU (lists of [3,3], [n_s,3], [n_s] arrays).
    Equivalent to Batch.from_data_list([MaterialGraph.from_arrays(...) ...]).to(device), without the host search."""
    if threebody_cutoff > cutoff:
        raise ValueError("Three body cutoff raidus should be smaller than two body.")
    lat = np.stack([np.asarray(L, dtype=np.float64).reshape(3, 3) for L in lattices])
    sizes = [len(p) for p in cart_coords]
    pos = np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1, 3) for p in cart_coords])
    z = np.concatenate([np.asarray(a).reshape(-1) for a in atomic_numbers])
    batch = np.repeat(np.arange(len(sizes)), sizes)
    dev = torch.device(device)
    lat_d = torch.tensor(lat, device=dev)
    pos_d = torch.tensor(pos, device=dev)
    batch_d = torch.tensor(batch, dtype=torch.int64, device=dev)
    ei, shift, dist, tei, nti, ntij = graph_indices_gpu(lat_d, pos_d, batch_d, cutoff, threebody_cutoff, host_lattice=lat)
    g = Batch.__new__(Batch)
    dict.__init__(g)
    g[K.POS] = pos_d.to(torch.float)
    g[K.ATOM_TYPES] = torch.tensor(z - 1, dtype=torch.long, device=dev)
    g[K.NUM_TRIPLET_I] = nti
    g[K.EDGE_INDEX] = ei
    g[K.EDGE_CELL_SHIFT] = shift
    g[K.NUM_TRIPLET_IJ] = ntij
    g[K.TRIPLET_EDGE_INDEX] = tei
    g[K.LATTICE] = lat_d.to(torch.float)
    g[K.BATCH] = batch_d
    g[K.NUM_NODES] = int(len(pos))
    g[K.NUM_EDGES] = int(ei.size(1))
    g[K.NUM_TRIPLETS] = int(tei.size(1))
    g["num_graphs"] = len(sizes)
    mark_canonical(g)
    return g


def mark_canonical(g) -> None:
    """Record that the index tensors of `g` are, as they stand, lists this library's own builders wrote (symmetric, complete triplet
    lists by construction): the engine's topology build may then skip its mirror / completeness checks
    (m3g_topology_build_canonical).  The mark is the identity and version of the three tensors -- any later in-place change, or a
    replaced tensor, voids it."""
    from ..nn.modules import _Topology

    dict.__setitem__(g, "_m3g_canonical_lists", _Topology.signature(g))


def batch_from_structures(structures: Sequence, cutoff: float, threebody_cutoff: float, device="cuda") -> Batch:
    """`Batch.from_data_list([MaterialGraph.from_structure(s, cutoff, threebody_cutoff) for s in structures]).to(device)` -- what a
    user of the reference writes (data/material_graph.py:132-166 per structure, then PyG collation) -- with the neighbour search,
    the triplet enumeration and the collation of the whole batch done on the GPU in one pass.  `structures`: pymatgen `Structure`s
    or anything with `lattice.matrix`, `cart_coords` and atomic numbers (material_graph._structure_arrays)."""
    from .material_graph import _structure_arrays

    lats, poss, zs = zip(*(_structure_arrays(s) for s in structures))
    return batch_from_arrays(lats, poss, zs, cutoff, threebody_cutoff, device=device)
