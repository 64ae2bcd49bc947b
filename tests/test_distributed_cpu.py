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


def _structure(i):
    from helpers import random_cell_arrays

    return random_cell_arrays(6 + 2 * (i % 3), 5.5, seed=i)


def _worker_structures(rank, world, port, ret):
    """ShardedBatch.from_structures: cooperative pricing (cost all-gather), shard build, repeated steps."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_m3gnet.distributed import ShardedBatch

    calls = []

    def fn(i):
        calls.append(i)
        return _structure(i)

    sb = ShardedBatch.from_structures(7, fn, 3.5, 3.0, device="cpu")
    steps = []
    for _ in range(2):   # the shard batch is built once and re-used
        e, f = sb.evaluate(_stub_evaluate, gather_forces=True)
        steps.append((e.numpy().copy(), f.numpy().copy()))
    e_loc, f_loc = sb.evaluate(_stub_evaluate)
    ret[rank] = dict(steps=steps, shards=sb.shards, costs=sb.costs, sizes=sb.sizes, calls=sorted(set(calls)),
                     local=(f_loc.numpy().copy(), dict(sb.local_offsets)))
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


def test_two_rank_gloo_sharded_batch_from_structures():
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph
    from torch_m3gnet.distributed import partition_structures, structure_cost

    graphs = [MaterialGraph.from_arrays(*_structure(i), 3.5, 3.0) for i in range(7)]
    ref_batch = _stub_evaluate(Batch.from_data_list(graphs))
    ref_e, ref_f = ref_batch[K.TOTAL_ENERGY].numpy(), ref_batch[K.FORCES].numpy()
    costs = [structure_cost(g) for g in graphs]
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_structures, args=(2, port, ret), nprocs=2, join=True)
    for rank in (0, 1):
        r = ret[rank]
        assert r["costs"] == costs and r["sizes"] == [int(g[K.NUM_NODES]) for g in graphs]
        assert r["shards"] == partition_structures(costs, 2)
        # a rank only ever builds the structures it prices (rank, rank + 2, ...) or owns
        assert set(r["calls"]) <= set(range(rank, 7, 2)) | set(r["shards"][rank])
        for e, f in r["steps"]:
            np.testing.assert_allclose(e, ref_e, rtol=1e-6)
            np.testing.assert_allclose(f, ref_f, rtol=1e-6)
        f_loc, offs = r["local"]
        starts = np.concatenate([[0], np.cumsum(r["sizes"])])
        for i, off in offs.items():
            np.testing.assert_allclose(f_loc[off: off + r["sizes"][i]], ref_f[starts[i]: starts[i + 1]], rtol=1e-6)


def test_sharded_batch_single_process_matches_plain_batch():
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph
    from torch_m3gnet.distributed import ShardedBatch

    sb = ShardedBatch.from_structures(5, _structure, 3.5, 3.0, device="cpu")
    assert sb.shards == [[0, 1, 2, 3, 4]] and sb.world == 1
    ref = _stub_evaluate(Batch.from_data_list([MaterialGraph.from_arrays(*_structure(i), 3.5, 3.0) for i in range(5)]))
    e, f = sb.evaluate(_stub_evaluate, gather_forces=True)
    np.testing.assert_allclose(e.numpy(), ref[K.TOTAL_ENERGY].numpy(), rtol=1e-6)
    np.testing.assert_allclose(f.numpy(), ref[K.FORCES].numpy(), rtol=1e-6)


def test_launcher_supervision_terminates_peers_of_a_failed_rank():
    """bench.py's `supervise`: a child that exits non-zero ends the job at once -- the survivors are terminated and the
    failing rank's code is returned (a rank dying in RCCL init would otherwise leave rank 0 waiting in the rendezvous)."""
    import importlib.util
    import subprocess
    import sys
    import time
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("m3g_bench", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sleeper = [sys.executable, "-c", "import time; time.sleep(120)"]
    t0 = time.time()
    procs = [subprocess.Popen(sleeper), subprocess.Popen([sys.executable, "-c", "import time, sys; time.sleep(0.5); sys.exit(3)"]),
             subprocess.Popen(sleeper)]
    code, rank = bench.supervise(procs)
    assert (code, rank) == (3, 1) and time.time() - t0 < 30
    assert all(p.poll() is not None for p in procs)           # nobody left behind
    ok = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(3)]
    assert bench.supervise(ok) == (0, None)
    killed = [subprocess.Popen(sleeper), subprocess.Popen([sys.executable, "-c", "import os, signal; os.kill(os.getpid(), signal.SIGKILL)"])]
    code, rank = bench.supervise(killed)
    assert rank == 1 and code == 9 and killed[0].poll() is not None


def test_assemble_matches_the_collective_layout():
    """`ShardedBatch.assemble` (one-process rehearsal of an N-rank job) reproduces what the all-gathers deliver."""
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph
    from torch_m3gnet.distributed import ShardedBatch, partition_structures, structure_cost

    graphs = [MaterialGraph.from_arrays(*_structure(i), 3.5, 3.0) for i in range(7)]
    ref = _stub_evaluate(Batch.from_data_list(graphs))
    costs = [structure_cost(g) for g in graphs]
    shards = partition_structures(costs, 3)
    sb = ShardedBatch(shards, [int(g[K.NUM_NODES]) for g in graphs], costs, None, "cpu", None)
    outs = [_stub_evaluate(Batch.from_data_list([graphs[i] for i in s])) for s in shards]
    e, f = sb.assemble([o[K.TOTAL_ENERGY] for o in outs], [o[K.FORCES] for o in outs])
    np.testing.assert_allclose(e.numpy(), ref[K.TOTAL_ENERGY].numpy(), rtol=1e-6)
    np.testing.assert_allclose(f.numpy(), ref[K.FORCES].numpy(), rtol=1e-6)
