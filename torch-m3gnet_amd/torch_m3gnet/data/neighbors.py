"""Periodic neighbour list and three-body index construction (numpy, host side).

Stands in for the reference's pymatgen-based graph construction so that tests, the benchmark and
the golden-vector generator can build `MaterialGraph` inputs without third-party packages:

* `neighbor_list`   -- replaces `get_all_neighbors_with_cell_shifts`
                       (reference: src/torch_m3gnet/data/material_graph.py:168-193): full directed
                       list, sorted by centre atom, self-images included, integer cell shifts.
* `threebody_index` -- replaces `compute_threebody` (material_graph.py:196-254): every ordered pair
                       (e1, e2), e1 != e2, of edges with d <= threebody_cutoff sharing a centre.

This file is deliberately free of package-relative imports: tests/golden/generate_golden.py loads it
by path while the *reference* package occupies the name `torch_m3gnet`.
"""
from __future__ import annotations

import numpy as np

try:  # scipy is part of the image; keep a brute-force path for very small cells anyway
    from scipy.spatial import cKDTree
except Exception:  # pragma: no cover
    cKDTree = None


def _image_ranges(lattice: np.ndarray, cutoff: float) -> np.ndarray:
    """Number of periodic images needed along each lattice vector so every point within `cutoff`
    of the home cell is covered: ceil(cutoff / height_p), height_p = V / |a_q x a_r|."""
    vol = abs(np.linalg.det(lattice))
    reps = []
    for p in range(3):
        q, r = (p + 1) % 3, (p + 2) % 3
        area = np.linalg.norm(np.cross(lattice[q], lattice[r]))
        reps.append(int(np.ceil(cutoff * area / vol)))
    return np.array(reps, dtype=np.int64)


def neighbor_list(lattice, cart_coords, cutoff: float, tol: float = 1e-8):
    """Full periodic neighbour list.

    Returns (edge_index [2,E] int64 (row 0 = centre i, row 1 = neighbour j; sorted by i, then by the
    edge's cell shift (sx, sy, sz) lexicographically -- the shift that refers to the GIVEN coordinates, so
    the order does not depend on which atoms happen to sit outside the home cell (an unwrapped MD
    trajectory keeps its edge order while atoms cross cell faces) --, then by j: the canonical order
    shared with the GPU builder, csrc/m3g_graph_build.hip), edge_cell_shift [E,3] int32, distances [E] float64).
    A pair is kept when 1e-8 < d <= cutoff + tol (pymatgen semantics: self at zero distance is
    dropped, periodic self-images are kept)."""
    lattice = np.asarray(lattice, dtype=np.float64).reshape(3, 3)
    pos = np.asarray(cart_coords, dtype=np.float64).reshape(-1, 3)
    n = pos.shape[0]
    # wrap into the home cell and remember the integer offset so shifts refer to the given coords
    frac = pos @ np.linalg.inv(lattice)
    wrap = np.floor(frac).astype(np.int64)
    pos_w = (frac - wrap) @ lattice
    reps = _image_ranges(lattice, cutoff + tol)
    grids = [np.arange(-r, r + 1) for r in reps]
    images = np.stack(np.meshgrid(*grids, indexing="ij"), axis=-1).reshape(-1, 3)
    shifts_cart = images @ lattice
    # keep only image atoms that can be within cutoff of the home cell's bounding box
    lo = pos_w.min(axis=0) - (cutoff + tol)
    hi = pos_w.max(axis=0) + (cutoff + tol)
    all_pos, all_idx, all_img = [], [], []
    for img, sc in zip(images, shifts_cart):
        p = pos_w + sc
        keep = np.all((p >= lo) & (p <= hi), axis=1)
        if not keep.any():
            continue
        all_pos.append(p[keep])
        all_idx.append(np.nonzero(keep)[0])
        all_img.append(np.repeat(img[None, :], keep.sum(), axis=0))
    all_pos = np.concatenate(all_pos)
    all_idx = np.concatenate(all_idx)
    all_img = np.concatenate(all_img)

    if cKDTree is not None and n * all_pos.shape[0] > 4_000_000:
        tree = cKDTree(all_pos)
        hits = tree.query_ball_point(pos_w, cutoff + tol)
        src = np.repeat(np.arange(n), [len(h) for h in hits])
        col = np.concatenate([np.asarray(h, dtype=np.int64) for h in hits]) if len(src) else np.zeros(0, np.int64)
    else:
        d2 = ((pos_w[:, None, :] - all_pos[None, :, :]) ** 2).sum(-1)
        src, col = np.nonzero(d2 <= (cutoff + tol) ** 2)
    vec = all_pos[col] - pos_w[src]
    dist = np.sqrt((vec**2).sum(-1))
    keep = dist > 1e-8
    src, col, dist = src[keep], col[keep], dist[keep]
    dst = all_idx[col]
    img = all_img[col]
    # shift such that r = pos[dst] + shift @ lattice - pos[src] for the *given* (unwrapped) coords
    shift = img - wrap[dst] + wrap[src]
    order = np.lexsort((dst, shift[:, 2], shift[:, 1], shift[:, 0], src))
    src, dst, dist, shift = src[order], dst[order], dist[order], shift[order]
    edge_index = np.stack([src, dst]).astype(np.int64)
    return edge_index, shift.astype(np.int32), dist


def threebody_index(num_nodes: int, edge_index: np.ndarray, distances: np.ndarray, threebody_cutoff: float):
    """Vectorised equivalent of the reference's `compute_threebody` for centre-sorted edge lists.

    Returns (triplet_edge_index [2,T] int64, num_triplet_i [N] int64, num_triplet_ij [E] int32).
    Ordering matches the reference loop: for each centre i, for each valid edge e1 (in edge
    order), every other valid edge e2 of the same centre in edge order."""
    src = np.asarray(edge_index[0])
    if src.size and np.any(np.diff(src) < 0):
        raise ValueError("edge_index must be sorted by centre atom (row 0)")
    valid = np.nonzero(np.asarray(distances) <= threebody_cutoff)[0]
    vsrc = src[valid]
    deg = np.bincount(vsrc, minlength=num_nodes).astype(np.int64)
    num_triplet_i = deg * (deg - 1)
    row_start = np.concatenate([[0], np.cumsum(deg)])[:-1]
    # for each valid edge (position p in the valid list, centre c): partners are the deg[c] valid
    # edges of c except itself
    per_edge = deg[vsrc] - 1
    t_total = int(per_edge.sum())
    e1_pos = np.repeat(np.arange(valid.size), per_edge)
    # rank of the triplet within its e1 group: 0 .. deg-2
    grp_start = np.concatenate([[0], np.cumsum(per_edge)])[:-1]
    rank = np.arange(t_total) - np.repeat(grp_start, per_edge)
    local_e1 = e1_pos - row_start[vsrc[e1_pos]]
    local_e2 = rank + (rank >= local_e1)
    e2_pos = row_start[vsrc[e1_pos]] + local_e2
    tei = np.stack([valid[e1_pos], valid[e2_pos]]).astype(np.int64)
    num_triplet_ij = np.zeros(src.size, dtype=np.int32)
    num_triplet_ij[valid] = per_edge.astype(np.int32)
    return tei, num_triplet_i, num_triplet_ij
