"""Time graph construction (neighbour list + triplets) for the bench workload: host numpy builder vs GPU builder."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "torch-m3gnet_amd"))
from torch_m3gnet.data.graph_gpu import neighbor_list_gpu, threebody_index_gpu  # noqa: E402
from torch_m3gnet.data.neighbors import neighbor_list, threebody_index  # noqa: E402


def cu(nx, ny, nz, a=3.61):
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1)
    pos = (g.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a
    pos = pos + np.random.default_rng(0).uniform(-0.025, 0.025, pos.shape)
    return np.diag([nx * a, ny * a, nz * a]).astype(float), pos


def main():
    torch.set_num_threads(16)
    for dims, nrep in (((2, 2, 2), 1), ((2, 2, 2), 256), ((10, 10, 25), 1)):
        lat, pos = cu(*dims)
        lats = np.stack([lat] * nrep)
        poss = np.concatenate([pos] * nrep)
        batch = np.repeat(np.arange(nrep), len(pos))
        t0 = time.perf_counter()
        for _ in range(nrep if nrep < 4 else 4):
            ei, sh, d = neighbor_list(lat, pos, 5.0)
            tei, _, _ = threebody_index(len(pos), ei, d.astype(np.float32), 4.0)
        host = (time.perf_counter() - t0) / (nrep if nrep < 4 else 4) * nrep
        L = torch.tensor(lats, device="cuda"); P = torch.tensor(poss, device="cuda"); B = torch.tensor(batch, device="cuda")
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gei, gsh, gd = neighbor_list_gpu(L, P, B, 5.0)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            gtei, _, _ = threebody_index_gpu(len(poss), gei, gd, 4.0)
            torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{dims} x{nrep}: atoms {len(poss)} edges {gei.size(1)} triplets {gtei.size(1)}  host numpy {host*1e3:.1f} ms | "
              f"GPU neighbours {(t1-t0)*1e3:.2f} ms + triplets {(t2-t1)*1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
