"""world_size-2 gloo test of the data-parallel host logic (structure partition + energy all-gather).
The per-shard evaluation is stubbed with a deterministic host function: the HIP engine needs a GPU,
the sharding/gather/reorder plumbing does not."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _graphs():
    from helpers import random_cell_graph

    return [random_cell_graph(6 + 2 * (s % 3), 5.5, seed=s, cutoff=3.5, tb_cutoff=3.0) for s in range(7)]


def _stub_evaluate(batch):
    """energy_s = sum over atoms of (type + |pos|^2); forces = -2 pos (so shards are checkable)."""
    from torch_m3gnet.data import MaterialGraphKey as K

    per_atom = batch[K.ATOM_TYPES].float() + (batch[K.POS] ** 2).sum(1)
    n_s = int(batch[K.LATTICE].size(0))
    batch[K.TOTAL_ENERGY] = torch.zeros(n_s).index_add(0, batch[K.BATCH], per_atom)
    batch[K.FORCES] = -2 * batch[K.POS]
    return batch


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_m3gnet.distributed import evaluate_sharded

    graphs = _graphs()
    energies, forces = evaluate_sharded(graphs, _stub_evaluate, device=torch.device("cpu"), gather_forces=True)
    ret[rank] = (energies.numpy().copy(), {i: f.numpy().copy() for i, f in forces.items()})
    dist.destroy_process_group()


def test_partition_is_balanced_and_complete():
    from torch_m3gnet.distributed import partition_structures

    costs = [5, 100, 7, 60, 40, 3, 1, 90]
    shards = partition_structures(costs, 3)
    assert sorted(i for s in shards for i in s) == list(range(8))
    loads = [sum(costs[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(costs)
    assert partition_structures(costs, 3) == shards  # deterministic
    assert partition_structures([1.0], 4) == [[0], [], [], []]


def test_two_rank_gloo_energies_match_single_process():
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.material_graph import Batch

    graphs = _graphs()
    ref = _stub_evaluate(Batch.from_data_list(graphs))[K.TOTAL_ENERGY].numpy()
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    for rank in (0, 1):
        energies, forces = ret[rank]
        np.testing.assert_allclose(energies, ref, rtol=1e-6)
        assert sorted(forces) == list(range(len(graphs)))
        for i, g in enumerate(graphs):
            np.testing.assert_allclose(forces[i], -2 * g[K.POS].numpy(), rtol=1e-6)
