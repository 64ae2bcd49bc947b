"""GPU: BASELINE config 4 (4,096 independent 64-atom structures over 8 GPUs) on the real engine -- one GPU's share
(512 cells, seeds 0..511) through the sharding code (`ShardedBatch`, world size 1: every structure is local, the
partition / pricing / reorder plumbing and the device-side collation all run), against the CPU oracle on a 32-cell
subset and through size-independent properties on the rest.  The N > 1 collectives are covered on CPU by
tests/test_distributed_cpu.py (gloo, world size 2)."""
import numpy as np
import pytest
import torch

from helpers import random_cell_arrays, rel_err
from test_gpu_properties import _default_model, _oracle_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cell(i):
    return random_cell_arrays(64, 9.1, seed=i)


def test_config4_share_through_sharded_batch():
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.distributed import ShardedBatch

    n_cells = 512
    model = _default_model()
    built = []

    def fn(i):
        built.append(i)
        return _cell(i)

    sb = ShardedBatch.from_structures(n_cells, fn, 5.0, 4.0, device=DEV)
    assert sorted(built) == list(range(n_cells))            # each structure generated exactly once
    assert sb.shards == [list(range(n_cells))] and sb.sizes == [64] * n_cells
    assert int(sb.batch[K.NUM_NODES]) == 64 * n_cells
    # costs = triplets + 32 edges per structure, from the device-built index tensors
    assert abs(sum(sb.costs) - (float(sb.batch[K.NUM_TRIPLETS]) + 32.0 * float(sb.batch[K.NUM_EDGES]))) < 1e-6 * sum(sb.costs)
    energies, forces = sb.evaluate(lambda b: model(b, forces=True, extras=False), gather_forces=True)
    assert energies.shape == (n_cells,) and forces.shape == (64 * n_cells, 3)
    assert torch.isfinite(energies).all() and torch.isfinite(forces).all()
    # (1) 32-cell subset against the CPU oracle (checker): the same cells batched on their own
    sub = batch_from_arrays(*zip(*[_cell(i) for i in range(32)]), 5.0, 4.0, device=DEV)
    out = model(sub)
    torch.set_num_threads(8)
    p, cfg, c, og = _oracle_inputs(model, out)
    o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
    e_sub = energies[:32].cpu()
    assert float(((e_sub - o["total_energy"]).abs() / o["total_energy"].abs()).max()) < 1e-5
    assert rel_err(forces[: 32 * 64], o["forces"]) < 1e-4
    # (2) the rest: batched == the same structures evaluated in two halves (reference tests/test_model.py:59-78)
    halves = []
    for lo, hi in ((0, 256), (256, 512)):
        h = model(batch_from_arrays(*zip(*[_cell(i) for i in range(lo, hi)]), 5.0, 4.0, device=DEV), extras=False)
        halves.append((h[K.TOTAL_ENERGY].clone(), h[K.FORCES].clone()))
    torch.testing.assert_close(energies.to(DEV), torch.cat([h[0] for h in halves]), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(forces.to(DEV), torch.cat([h[1] for h in halves]), rtol=1e-5, atol=1e-7)
    # (3) no net force on any structure
    f = forces.double().view(n_cells, 64, 3)
    assert float(f.sum(1).abs().max()) < 1e-3 * float(f.abs().max())
    # (4) a second step re-uses the resident shard and reproduces the first bit for bit (no atomics on the force path)
    e2, f2 = sb.evaluate(lambda b: model(b, forces=True, extras=False), gather_forces=True)
    assert torch.equal(f2, forces)
    torch.testing.assert_close(e2, energies, rtol=1e-6, atol=1e-6)   # per-structure sums use float atomics


def test_sharded_batch_with_given_costs_and_local_forces():
    """Costs supplied by the caller (no pricing pass); local forces addressed through `local_offsets`."""
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.distributed import ShardedBatch

    model = _default_model()
    cells = [random_cell_arrays(8 + 3 * (i % 4), 6.5, seed=100 + i) for i in range(9)]
    sb = ShardedBatch.from_structures(9, lambda i: cells[i], 5.0, 4.0, device=DEV, costs=[float(len(c[1])) ** 2 for c in cells])
    assert sb.sizes == [len(c[1]) for c in cells]
    energies, f_loc = sb.evaluate(lambda b: model(b, forces=True, extras=False))
    for i in (0, 4, 8):
        one = model(batch_from_arrays(*zip(cells[i]), 5.0, 4.0, device=DEV), extras=False)
        torch.testing.assert_close(energies[i].to(DEV), one[K.TOTAL_ENERGY][0], rtol=1e-6, atol=1e-6)
        off = sb.local_offsets[i]
        torch.testing.assert_close(f_loc[off: off + sb.sizes[i]], one[K.FORCES], rtol=1e-5, atol=1e-7)
