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
    assert torch.equal(e2, energies)   # (per-structure sums are fixed-order reductions too: no atomics on the default path)


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


def _run_child(cmd, env=None, timeout=600):
    import os
    import subprocess
    import sys

    full_env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    full_env.update(env or {})
    return subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=timeout, env=full_env)


def test_one_rank_nccl_group_runs_the_rccl_branch():
    """The N > 1 path on RCCL, executed at first contact with one rank: `init_process_group("nccl", device_id=...)`, the
    cost / size / energy / force `all_gather_into_tensor` calls on DEVICE buffers (`_comm_device` = cuda), all-reduce and
    barrier -- in a fresh child process, checked against the un-sharded call (tests/checkers/nccl_one_rank.py)."""
    import json
    from pathlib import Path

    proc = _run_child([str(Path(__file__).resolve().parent / "checkers" / "nccl_one_rank.py")])
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = json.loads(proc.stdout.strip().splitlines()[-1])
    assert rec["ok"] and rec["backend"] == "nccl" and rec["energy_rel_err"] < 1e-6 and rec["force_rel_err"] < 1e-6


def test_bench_nccl_branch_with_one_rank():
    """bench.py's own RCCL code (Job: init with device_id, barrier, all-reduce of the elapsed time, the all-gather of the replicas'
    energies into a device buffer after the timed region) with a forced one-rank group, on a small cell."""
    import json
    from pathlib import Path

    bench = Path(__file__).resolve().parent.parent / "bench.py"
    proc = _run_child([str(bench), "--gpus", "1", "--steps", "3", "--warmup", "1", "--cells", "4", "4", "4", "--no-cpu-baseline", "--no-secondary"],
                      env={"M3G_BENCH_FORCE_DIST": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                           "MASTER_PORT": "29547"})
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = json.loads(proc.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["config"]["atoms_per_gpu"] == 256
    assert rec["config"]["replica_energy_rel_spread"] == 0.0


def test_self_launched_job_fails_fast_when_a_rank_dies():
    """`bench.py --gpus 2` (gloo rehearsal, two ranks sharing the GPU) with rank 1 dying before the rendezvous: the launcher
    must notice, terminate rank 0 (which is waiting in the rendezvous) and return non-zero within seconds -- not sit until a
    store time-out."""
    import time
    from pathlib import Path

    bench = Path(__file__).resolve().parent.parent / "bench.py"
    t0 = time.time()
    proc = _run_child([str(bench), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cells", "4", "4", "4", "--no-cpu-baseline", "--no-secondary"],
                      env={"M3G_BENCH_BACKEND": "gloo", "M3G_BENCH_TEST_DIE_RANK": "1"}, timeout=300)
    dt = time.time() - t0
    assert proc.returncode == 7, (proc.returncode, proc.stderr[-2000:])
    assert "rank 1 exited with code 7" in proc.stderr
    assert proc.stdout.strip() == ""          # no result line from a failed job
    assert dt < 120, dt                        # (import torch on a fresh box can take a minute; the wait itself is seconds)


def test_config4_full_partition_on_one_gpu():
    """BASELINE config 4 as the 8-GPU job will form it: 4,096 cells priced, `partition_structures(costs, 8)`, the 8 shards
    evaluated one after another on this GPU, results reassembled through ShardedBatch's own permutation tensors
    (`assemble`: the layout the all-gathers deliver) and compared with the cells evaluated in input order."""
    import os

    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.distributed import ShardedBatch, batch_structure_costs, partition_structures

    n_cells, world, chunk = 4096, 8, 512
    model = _default_model()
    cells = [_cell(i) for i in range(n_cells)]
    step = lambda b: model(b, forces=True, extras=False)   # noqa: E731
    costs, sizes, ref_e, ref_f = [], [], [], []
    for lo in range(0, n_cells, chunk):   # input order: prices every cell and gives the reference results
        b = batch_from_arrays(*zip(*cells[lo: lo + chunk]), 5.0, 4.0, device=DEV)
        costs += batch_structure_costs(b).tolist()
        sizes += torch.bincount(b[K.BATCH], minlength=chunk).tolist()
        out = step(b)
        ref_e.append(out[K.TOTAL_ENERGY].clone())
        ref_f.append(out[K.FORCES].clone())
    ref_e, ref_f = torch.cat(ref_e), torch.cat(ref_f)
    shards = partition_structures(costs, world)
    assert sorted(i for s in shards for i in s) == list(range(n_cells))
    loads = [sum(costs[i] for i in s) for s in shards]
    imbalance = max(loads) / (sum(loads) / world) - 1.0
    assert imbalance < 0.01, imbalance          # greedy LPT over 4,096 similar cells is essentially perfect
    sb = ShardedBatch(shards, sizes, costs, None, DEV, None)
    per_e, per_f = [], []
    for shard in shards:
        out = step(batch_from_arrays(*zip(*[cells[i] for i in shard]), 5.0, 4.0, device=DEV))
        per_e.append(out[K.TOTAL_ENERGY].clone())
        per_f.append(out[K.FORCES].clone())
    energies, forces = sb.assemble(per_e, per_f)
    torch.testing.assert_close(energies.to(DEV), ref_e, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(forces.to(DEV), ref_f, rtol=1e-5, atol=1e-7)
    line = (f"config 4, 4096 cells over 8 ranks: partition imbalance {imbalance:.2e} (max load / mean load - 1), "
            f"shard sizes {[len(s) for s in shards]}, atoms per shard {[sum(sizes[i] for i in s) for s in shards]}")
    print(line)
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/config4_partition.txt", "w") as fh:
            fh.write(line + "\n")
