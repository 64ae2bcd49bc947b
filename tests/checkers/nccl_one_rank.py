"""Child process of tests/test_gpu_sharded.py::test_one_rank_nccl_group_runs_the_rccl_branch: a ONE-rank "nccl" (= RCCL) process group
on the real GPU, a small job through `ShardedBatch.from_structures(...)` + `evaluate(gather_forces=True)`, checked against the
un-sharded call.  Run in a fresh process because a process group (and RCCL's device state) is per process.  Prints one JSON line."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))

import datetime  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device, timeout=datetime.timedelta(seconds=120))
    from torch_m3gnet.config import ModelConfig
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.data.synthetic import random_cell_arrays
    from torch_m3gnet.distributed import ShardedBatch, _comm_device

    torch.manual_seed(0)
    model = ModelConfig().build().to(device)
    cells = [random_cell_arrays(10 + 3 * (i % 5), 7.0, seed=300 + i) for i in range(24)]
    assert _comm_device(device, None).type == "cuda"          # RCCL moves device buffers
    sb = ShardedBatch.from_structures(len(cells), lambda i: cells[i], 5.0, 4.0, device=device)   # pricing all-gather on RCCL
    assert sb._comm.type == "cuda" and sb.world == 1 and sb.shards == [list(range(len(cells)))]
    step = lambda b: model(b, forces=True, extras=False)   # noqa: E731
    energies, forces = sb.evaluate(step, gather_forces=True)   # energy + force all-gathers on RCCL (device buffers)
    e2, f_loc = sb.evaluate(step)
    torch.cuda.synchronize()
    assert energies.is_cuda and forces.is_cuda
    # costs given: the atom counts travel through the int64 all-gather
    sb2 = ShardedBatch.from_structures(len(cells), lambda i: cells[i], 5.0, 4.0, device=device, costs=[float(len(c[1])) for c in cells])
    assert sb2.sizes == [len(c[1]) for c in cells]
    plain = model(batch_from_arrays(*zip(*cells), 5.0, 4.0, device=device), extras=False)
    e_err = float((energies - plain[K.TOTAL_ENERGY]).abs().max() / plain[K.TOTAL_ENERGY].abs().max())
    f_err = float((forces - plain[K.FORCES]).abs().max() / plain[K.FORCES].abs().max())
    # the collectives bench.py issues: barrier, all-reduce (max) of a float64 device scalar, all-gather of one energy per rank
    t = torch.tensor([1.25], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    ok = e_err < 1e-6 and f_err < 1e-6 and torch.equal(e2, energies) and float(t.item()) == 1.25
    print(json.dumps({"ok": bool(ok), "backend": dist.get_backend(), "energy_rel_err": e_err, "force_rel_err": f_err,
                      "structures": len(cells), "atoms": int(sum(sb.sizes))}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
