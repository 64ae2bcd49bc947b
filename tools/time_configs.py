"""Step time of the engine on BASELINE.json's other configurations (parity-test cases, not bench lines):
config 2 = 256 random 64-atom cells in one batch, config 4 = one GPU's share (512 cells) of 4,096 64-atom structures,
config 5 = 2,000 atoms with cutoff 6 / three-body cutoff 4 and 6; plus a 100,000-atom Cu supercell (10x config 3, graph
built on the GPU) whose energy per atom and force statistics must match the 10,000-atom cell's."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from torch_m3gnet.data.synthetic import random_cell_graph  # noqa: E402
from torch_m3gnet.data.material_graph import Batch  # noqa: E402
from torch_m3gnet.model.build import build_model  # noqa: E402


def timeit(model, g, n=20):
    for _ in range(3):
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def large_supercell(model):
    """100,000-atom fcc Cu (25 x 25 x 40 cells): the 10,000-atom cell of config 3 repeated; unjittered, so every atom is
    equivalent and energy per atom / zero forces are size-independent properties."""
    from torch_m3gnet.data.graph_gpu import batch_from_arrays

    a = 3.61
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    out = {}
    for dims in ((10, 10, 25), (25, 25, 40)):
        gi = np.stack(np.meshgrid(*[np.arange(d) for d in dims], indexing="ij"), -1)
        pos = (gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a
        lat = np.diag([d * a for d in dims]).astype(float)
        t0 = time.perf_counter()
        g = batch_from_arrays([lat], [pos], [np.full(len(pos), 29)], 5.0, 4.0)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        ms = timeit(model, g, n=10)
        n = len(pos)
        e = float(g["total_energy"][0]) / n
        fmax = float(g["forces"].abs().max())
        out[n] = e
        print(f"Cu {dims}: {n} atoms, E={g['num_edges']} T={g['num_triplets']} (GPU graph build {t_build * 1e3:.0f} ms): "
              f"{ms:.3f} ms/step = {n / ms * 1e3 / 1e6:.2f} M atom-steps/s, E/atom {e:.7f}, max|F| {fmax:.2e}, "
              f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
    small, big = out[10_000], out[100_000]
    assert abs(small - big) <= 1e-5 * abs(small), (small, big)


def main():
    torch.manual_seed(0)
    model = build_model(5.0, 4.0, 3, 3, 95, 64, 3).cuda()
    g = Batch.from_data_list([random_cell_graph(64, 9.1, seed=s) for s in range(256)]).to("cuda")
    ms = timeit(model, g)
    n = int(g["pos"].size(0))
    print(f"config 2: {n} atoms in 256 cells, E={g['num_edges']} T={g['num_triplets']}: {ms:.3f} ms/step = {n / ms * 1e3 / 1e6:.2f} M atom-steps/s", flush=True)
    rng = np.random.default_rng(0)
    cells = [random_cell_graph(64, 9.1, seed=1000 + s) for s in range(64)]
    g = Batch.from_data_list([cells[i % 64].clone() for i in range(512)]).to("cuda")
    ms = timeit(model, g)
    n = int(g["pos"].size(0))
    print(f"config 4 (one GPU's 512 of 4,096 cells): {n} atoms, E={g['num_edges']} T={g['num_triplets']}: {ms:.3f} ms/step = {n / ms * 1e3 / 1e6:.2f} M atom-steps/s", flush=True)
    large_supercell(model)
    for tb in (4.0, 6.0):
        torch.manual_seed(0)
        model = build_model(6.0, tb, 3, 3, 95, 64, 3).cuda()
        g = Batch.from_data_list([random_cell_graph(2000, 31.1, seed=0, cutoff=6.0, tb_cutoff=tb)]).to("cuda")
        ms = timeit(model, g)
        n = int(g["pos"].size(0))
        print(f"config 5 (r3={tb}): {n} atoms, E={g['num_edges']} T={g['num_triplets']}: {ms:.3f} ms/step = {n / ms * 1e3 / 1e6:.2f} M atom-steps/s", flush=True)


if __name__ == "__main__":
    main()
