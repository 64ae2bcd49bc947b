"""Random batches along a four-step random walk through BOTH trajectory drivers (run on the GPU box): VerletGraph.step (m3g_md_step:
one library call per step on capacity buffers) against model(VerletGraph.update(pos)) -- random lattices (cubic to strongly sheared,
2-25 A), 1-120 atoms per cell, cutoffs 2.5-6 A with a model built for each, 1-4 structures, atoms up to half a cell outside the home
cell; the walk takes the first search, a second one (moves beyond skin / 2), a refill and a reuse.  Energies, forces, stresses and
the lists must agree bit for bit on every step.   Usage: python tests/checkers/fuzz_md_step.py [cases] [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from fuzz_graph_build import random_cell  # noqa: E402
from torch_m3gnet.data import MaterialGraphKey as K  # noqa: E402
from torch_m3gnet.data.md import VerletGraph  # noqa: E402
from torch_m3gnet.model.build import build_model  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    done = 0
    paths = {"reuse": 0, "refill": 0, "search": 0}
    while done < cases:
        cutoff = float(rng.uniform(2.5, 6.0))
        tb = float(rng.uniform(0.5, 1.0) * cutoff)
        cells = [random_cell(rng) for _ in range(int(rng.integers(1, 5)))]
        if min(abs(np.linalg.det(l)) for l, _ in cells) < 8.0:
            continue
        lats, poss = [l for l, _ in cells], [p for _, p in cells]
        sizes = [len(p) for p in poss]
        zs = [rng.integers(1, 90, n) for n in sizes]
        torch.manual_seed(done)
        model = build_model(cutoff, tb, 3, 3, 95, 64, int(rng.integers(1, 4))).cuda()
        a, b = (VerletGraph(lats, zs, cutoff, tb, skin=0.35, device="cuda") for _ in range(2))
        pos = np.concatenate(poss)
        big = False
        for step in range(4):
            pos = pos + rng.normal(0.0, (0.0, 0.25, 0.02, 1e-9)[step], pos.shape)
            p = torch.tensor(pos, device="cuda")
            g = b.update(p)
            if int(g[K.NUM_TRIPLETS]) > 3_000_000:
                big = True
                break
            want = model(g, extras=False)
            got = a.step(model, p)
            lists = a.step_lists()
            ok = all(torch.equal(got[k], want[k]) for k in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES)) and \
                all(torch.equal(lists[k], g[k]) for k in (K.EDGE_INDEX, K.EDGE_CELL_SHIFT, K.TRIPLET_EDGE_INDEX, K.NUM_TRIPLET_I, K.NUM_TRIPLET_IJ))
            if not ok or a.stats != b.stats:
                print(f"case {done} step {step} FAILED: cutoff {cutoff:.3f} tb {tb:.3f} sizes {sizes} stats {a.stats} {b.stats}", flush=True)
                raise SystemExit(1)
        if big:
            continue
        for k in paths:
            paths[k] += a.stats[k]
        done += 1
    print(f"{cases} random batches x 4 steps: VerletGraph.step == model(VerletGraph.update) bit for bit (energies, forces, stresses, lists); paths {paths}")


if __name__ == "__main__":
    main()
