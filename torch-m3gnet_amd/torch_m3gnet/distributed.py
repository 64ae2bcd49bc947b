"""Data-parallel evaluation of independent structures: one process per GPU, RCCL over xGMI.

The path shards by whole structures (no edge crosses structures: the reference's `__inc__` offsets keep
indices disjoint, data/material_graph.py:122-130, and batched == per-graph energies is one of its tests,
tests/test_model.py:59-78).  `ShardedBatch` partitions the structures of a job over the ranks ONCE, balanced
by triplet/edge cost, and keeps each rank's shard as one device-resident batch (built on the GPU by
`data/graph_gpu.batch_from_arrays`: no per-step host collation, and a rank only ever materialises the
structures it owns or prices).  A step is then: the fused engine on the local batch + one all-gather of the
per-structure energies (4 bytes per structure: latency-bound) and, on request, one of the forces.  There is no
data-path collective.  A single large cell does not shard (the reference has no spatial decomposition,
nn/gradient.py:26): "replicas only".
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .data import MaterialGraphKey as K
from .data.material_graph import Batch, MaterialGraph

StructureFn = Callable[[int], tuple]   # i -> (lattice [3,3], cart_coords [n,3], atomic_numbers [n])


def partition_structures(costs: Sequence[float], world_size: int) -> list[list[int]]:
    """Greedy longest-processing-time partition of structure indices by cost (triplet count):
    heaviest first onto the currently lightest rank.  Deterministic (ties by index)."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads = [0.0] * world_size
    shards: list[list[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += float(costs[i])
    for s in shards:
        s.sort()
    return shards


def structure_cost(graph: MaterialGraph) -> float:
    """Work estimate of one structure: triplets dominate the gather work, edges the dense work."""
    return float(graph[K.NUM_TRIPLETS]) + 32.0 * float(graph[K.NUM_EDGES])


def batch_structure_costs(batch: Batch) -> torch.Tensor:
    """`structure_cost` of every structure of a collated batch, from its index tensors alone ([S] float64, on the
    batch's device)."""
    b = batch[K.BATCH]
    s = int(batch[K.LATTICE].size(0))
    trip = torch.zeros(s, dtype=torch.float64, device=b.device).index_add_(0, b, batch[K.NUM_TRIPLET_I].to(torch.float64))
    edges = torch.bincount(b[batch[K.EDGE_INDEX][0]], minlength=s).to(torch.float64)
    return trip + 32.0 * edges


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _has_group(group) -> bool:
    """A process group exists (possibly of one rank): the collectives run through it, so a one-rank RCCL group exercises
    exactly the calls an eight-rank job makes."""
    return dist.is_available() and dist.is_initialized()


def _comm_device(device: torch.device, group) -> torch.device:
    """RCCL ("nccl") moves device buffers; the gloo rehearsal backend moves host buffers."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
        return device
    return torch.device("cpu")


def _build_batch(structures: Sequence[tuple], cutoff: float, threebody_cutoff: float, device: torch.device) -> Batch:
    lats, coords, zs = zip(*structures)
    if device.type == "cuda":
        from .data.graph_gpu import batch_from_arrays

        return batch_from_arrays(lats, coords, zs, cutoff, threebody_cutoff, device=device)
    return Batch.from_data_list([MaterialGraph.from_arrays(l, p, z, cutoff, threebody_cutoff) for l, p, z in structures])


class ShardedBatch:
    """The structures of one job, partitioned over the ranks of `group`; this rank's shard is `self.batch`.

    Attributes: `shards` (structure indices per rank, ascending), `mine`, `batch` (collated local shard on `device`,
    None when the rank owns nothing), `sizes` (atoms per structure, all structures), `costs`."""

    def __init__(self, shards, sizes, costs, batch, device, group):
        self.shards = shards
        self.sizes = [int(n) for n in sizes]
        self.costs = [float(c) for c in costs]
        self.batch = batch
        self.device = torch.device(device)
        self.group = group
        self.world, self.rank = _world(group)
        self.mine = shards[self.rank] if self.rank < len(shards) else []
        self.n_total = len(self.sizes)
        self._comm = _comm_device(self.device, group)
        self._n_max = max(1, max(len(s) for s in shards))
        # energies: position of structure i in the concatenation of the padded per-rank vectors
        perm = torch.empty(self.n_total, dtype=torch.long)
        for r, shard in enumerate(shards):
            for k, i in enumerate(shard):
                perm[i] = r * self._n_max + k
        self._perm = perm.to(self._comm)
        self._local_e = torch.zeros(self._n_max, dtype=torch.float, device=self._comm)
        self._gathered = torch.empty(self.world * self._n_max, dtype=torch.float, device=self._comm)
        # forces: rows of structure i inside the padded per-rank blocks
        self._f_max = max(1, max(sum(self.sizes[i] for i in s) for s in shards))
        self.local_offsets = {}
        off = 0
        for i in self.mine:
            self.local_offsets[i] = off
            off += self.sizes[i]
        self.n_local_atoms = off
        self._fperm = None

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_structures(cls, n_structures: int, structure_fn: StructureFn, cutoff: float, threebody_cutoff: float,
                        device="cuda", group=None, costs: Optional[Sequence[float]] = None) -> "ShardedBatch":
        """Partition `n_structures` structures described by `structure_fn(i) -> (lattice, coords, Z)`.

        Without `costs` the ranks price the structures cooperatively: rank r builds the graphs of structures
        r, r + W, ... (index work only), and the per-structure costs are all-gathered; then every rank builds the
        batch of the shard the greedy partition assigns to it.  `structure_fn` is only called for those indices."""
        device = torch.device(device)
        world, rank = _world(group)
        sizes = None
        priced = None
        if costs is None:
            slice_idx = list(range(rank, n_structures, world))
            per = (n_structures + world - 1) // world
            comm = _comm_device(device, group)
            mine_cost = torch.zeros(2 * per, dtype=torch.float64, device=comm)   # [costs | atom counts]
            if slice_idx:
                priced = _build_batch([structure_fn(i) for i in slice_idx], cutoff, threebody_cutoff, device)
                mine_cost[: len(slice_idx)] = batch_structure_costs(priced).to(comm)
                mine_cost[per: per + len(slice_idx)] = torch.bincount(priced[K.BATCH], minlength=len(slice_idx)).to(comm, torch.float64)
            if _has_group(group):
                allc = torch.empty(world * 2 * per, dtype=torch.float64, device=comm)
                dist.all_gather_into_tensor(allc, mine_cost, group=group)
            else:
                allc = mine_cost
            allc = allc.cpu().view(world, 2, per)
            costs = [float(allc[i % world, 0, i // world]) for i in range(n_structures)]
            sizes = [int(allc[i % world, 1, i // world]) for i in range(n_structures)]
        shards = partition_structures(costs, world)
        mine = shards[rank]
        if priced is not None and mine == list(range(rank, n_structures, world)):
            batch = priced          # (always the case at world size 1)
        elif mine:
            batch = _build_batch([structure_fn(i) for i in mine], cutoff, threebody_cutoff, device)
        else:
            batch = None
        if sizes is None:   # costs given: atom counts of the other ranks' structures come over the wire
            comm = _comm_device(device, group)
            n_max = max(1, max(len(s) for s in shards))
            loc = torch.zeros(n_max, dtype=torch.int64, device=comm)
            if mine:
                loc[: len(mine)] = torch.bincount(batch[K.BATCH], minlength=len(mine)).to(comm)
            if _has_group(group):
                alls = torch.empty(world * n_max, dtype=torch.int64, device=comm)
                dist.all_gather_into_tensor(alls, loc, group=group)
            else:
                alls = loc
            alls = alls.cpu()
            sizes = [0] * n_structures
            for r, shard in enumerate(shards):
                for k, i in enumerate(shard):
                    sizes[i] = int(alls[r * n_max + k])
        return cls(shards, sizes, costs, batch, device, group)

    @classmethod
    def from_graphs(cls, graphs: Sequence[MaterialGraph], device=None, group=None) -> "ShardedBatch":
        """Partition host-built graphs (the same list on every rank); the local shard is collated once and moved to
        `device`."""
        world, rank = _world(group)
        dev = torch.device(device) if device is not None else (graphs[0][K.POS].device if graphs else torch.device("cpu"))
        costs = [structure_cost(g) for g in graphs]
        shards = partition_structures(costs, world)
        mine = shards[rank]
        batch = Batch.from_data_list([graphs[i] for i in mine]).to(dev) if mine else None
        return cls(shards, [int(g[K.NUM_NODES]) for g in graphs], costs, batch, dev, group)

    # ------------------------------------------------------------------ the step
    def evaluate(self, evaluate: Callable[[Batch], Batch], gather_forces: bool = False):
        """One data-parallel step.  `evaluate` maps the local Batch to itself with TOTAL_ENERGY (and FORCES) filled --
        normally the model returned by `build_model`.  Returns (energies [n_structures] in input order, on every rank;
        forces): forces are the LOCAL shard's [n_local_atoms, 3] tensor (rows of structure i start at
        `local_offsets[i]`), or, with `gather_forces`, all atoms' forces [sum(sizes), 3] in input structure order."""
        out = None
        if self.batch is not None:
            out = evaluate(self.batch)
            self._local_e[: len(self.mine)] = out[K.TOTAL_ENERGY].to(self._comm, torch.float)
        if _has_group(self.group):
            dist.all_gather_into_tensor(self._gathered, self._local_e, group=self.group)
            energies = self._gathered[self._perm]
        else:
            energies = self._local_e[self._perm]
        forces = out.get(K.FORCES) if out is not None else None
        if gather_forces:
            forces = self._gather_forces(forces)
        return energies, forces

    # ------------------------------------------------------------------ one-process rehearsal of an N-rank job
    def assemble(self, per_rank_energies, per_rank_forces=None):
        """What the all-gathers of `evaluate` deliver, put together on ONE process: `per_rank_energies[r]` / `per_rank_forces[r]`
        are the local results of shard r (its structures in ascending order).  Uses the same padded layout and the same
        permutation tensors as the collective path, so an 8-way partition can be checked end to end on one GPU
        (tests/test_gpu_sharded.py::test_config4_full_partition_on_one_gpu).  Returns (energies, forces) in input order."""
        gathered = torch.zeros(len(self.shards) * self._n_max, dtype=torch.float, device=self._comm)
        for r, e in enumerate(per_rank_energies):
            gathered[r * self._n_max: r * self._n_max + len(self.shards[r])] = e.to(self._comm, torch.float)
        energies = gathered[self._perm]
        forces = None
        if per_rank_forces is not None:
            self._force_layout()
            fall = torch.zeros(len(self.shards) * self._f_max, 3, dtype=torch.float, device=self._comm)
            for r, f in enumerate(per_rank_forces):
                fall[r * self._f_max: r * self._f_max + f.size(0)] = f.to(self._comm, torch.float)
            forces = fall[self._fperm]
        return energies, forces

    def _force_layout(self):
        if self._fperm is None:
            starts = np.concatenate([[0], np.cumsum(self.sizes)])
            perm = torch.empty(int(starts[-1]), dtype=torch.long)
            for r, shard in enumerate(self.shards):
                off = r * self._f_max
                for i in shard:
                    perm[starts[i]: starts[i + 1]] = torch.arange(off, off + self.sizes[i])
                    off += self.sizes[i]
            self._fperm = perm.to(self._comm)
            self._fbuf = torch.zeros(self._f_max, 3, dtype=torch.float, device=self._comm)
            self._fall = torch.empty(self.world * self._f_max, 3, dtype=torch.float, device=self._comm)

    def _gather_forces(self, local_f):
        self._force_layout()
        if local_f is not None:
            self._fbuf[: self.n_local_atoms] = local_f.to(self._comm, torch.float)
        if _has_group(self.group):
            dist.all_gather_into_tensor(self._fall, self._fbuf, group=self.group)
            return self._fall[self._fperm]
        return self._fbuf[self._fperm]


def evaluate_sharded(graphs: Sequence[MaterialGraph], evaluate: Callable[[Batch], Batch], device=None,
                     group=None, gather_forces: bool = False):
    """One-shot convenience over `ShardedBatch.from_graphs` (for repeated steps keep the ShardedBatch and call its
    `evaluate`): returns (energies [len(graphs)] in input order on every rank, forces as {structure index:
    [n_i, 3] tensor} of the LOCAL shard, or of all structures when `gather_forces`)."""
    sb = ShardedBatch.from_graphs(graphs, device=device, group=group)
    energies, forces = sb.evaluate(evaluate, gather_forces=gather_forces)
    out = {}
    if gather_forces:
        starts = np.concatenate([[0], np.cumsum(sb.sizes)])
        out = {i: forces[starts[i]: starts[i + 1]] for i in range(sb.n_total)}
    elif forces is not None:
        out = {i: forces[off: off + sb.sizes[i]] for i, off in sb.local_offsets.items()}
    return energies, out
