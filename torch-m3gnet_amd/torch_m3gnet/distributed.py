"""Data-parallel evaluation of independent structures: one process per GPU, RCCL over xGMI.

The path shards by whole structures (no edge crosses structures: the reference's `__inc__` offsets keep
indices disjoint, data/material_graph.py:122-130, and batched == per-graph energies is one of its tests,
tests/test_model.py:59-78).  Each rank evaluates its shard with the fused engine; the only collective is
an all-gather of per-structure energies (a few bytes per structure -- latency-bound).  A single large
cell does not shard (the reference has no spatial decomposition, nn/gradient.py:26): "replicas only".
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.distributed as dist

from .data import MaterialGraphKey as K
from .data.material_graph import Batch, MaterialGraph


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


def evaluate_sharded(graphs: Sequence[MaterialGraph], evaluate: Callable[[Batch], Batch], device=None,
                     group=None, gather_forces: bool = False):
    """Evaluate `graphs` (the same list on every rank) data-parallel.

    `evaluate` maps a collated Batch (on `device`) to the same Batch with TOTAL_ENERGY (and FORCES) filled --
    normally the model returned by `build_model`.  Returns (energies [len(graphs)] in input order on every
    rank, forces of the LOCAL shard as {structure index: [n_i, 3] tensor}, or of all structures when
    `gather_forces`)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shards = partition_structures([structure_cost(g) for g in graphs], world)
    mine = shards[rank]
    n_max = max(len(s) for s in shards)
    dev = device if device is not None else (graphs[0][K.POS].device if graphs else torch.device("cpu"))
    local_e = torch.zeros(n_max, dtype=torch.float, device=dev)
    local_f: dict[int, torch.Tensor] = {}
    if mine:
        batch = Batch.from_data_list([graphs[i] for i in mine]).to(dev)
        out = evaluate(batch)
        local_e[: len(mine)] = out[K.TOTAL_ENERGY].to(torch.float)
        if K.FORCES in out and out[K.FORCES] is not None:
            off = 0
            for i in mine:
                n = int(graphs[i][K.NUM_NODES])
                local_f[i] = out[K.FORCES][off: off + n]
                off += n
    if world == 1:
        energies = torch.empty(len(graphs), dtype=torch.float, device=dev)
        energies[torch.tensor(mine, dtype=torch.long, device=dev)] = local_e[: len(mine)]
        return energies, local_f
    gathered = torch.empty(world * n_max, dtype=torch.float, device=dev)
    dist.all_gather_into_tensor(gathered, local_e, group=group)
    energies = torch.empty(len(graphs), dtype=torch.float, device=dev)
    for r, shard in enumerate(shards):
        if shard:
            energies[torch.tensor(shard, dtype=torch.long, device=dev)] = gathered[r * n_max: r * n_max + len(shard)]
    if gather_forces:
        sizes = [int(g[K.NUM_NODES]) for g in graphs]
        f_max = max(sum(sizes[i] for i in s) for s in shards)
        buf = torch.zeros(f_max, 3, dtype=torch.float, device=dev)
        off = 0
        for i in mine:
            buf[off: off + sizes[i]] = local_f[i]
            off += sizes[i]
        allf = torch.empty(world * f_max, 3, dtype=torch.float, device=dev)
        dist.all_gather_into_tensor(allf, buf, group=group)
        local_f = {}
        for r, shard in enumerate(shards):
            off = r * f_max
            for i in shard:
                local_f[i] = allf[off: off + sizes[i]]
                off += sizes[i]
    return energies, local_f
