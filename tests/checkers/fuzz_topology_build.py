"""Random batches through BOTH topology builds (run on the GPU box): m3g_topology_build_hints (general) and
m3g_topology_build_canonical (six launches, csrc/m3g_topology.hip k_canon_*) on the lists of the GPU graph builder -- random
lattices (cubic to strongly sheared, 2-25 A), 1-120 atoms per cell, cutoffs 2.5-9 A, 1-6 structures, three-body cutoff from half
the cutoff up to the cutoff itself.  The data part of the two topology buffers must agree byte for byte and carry the same
certificate; the canonical call must have taken its own path unless a row exceeds the in-edge kernel's 512-edge stage.
Usage: python tests/checkers/fuzz_topology_build.py [cases] [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from fuzz_graph_build import random_cell  # noqa: E402
from test_gpu_graph_build import _topology_buffers  # noqa: E402
from torch_m3gnet.data import MaterialGraphKey as K  # noqa: E402
from torch_m3gnet.data.graph_gpu import batch_from_arrays  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    done = fast = edges = trips = 0
    for c in range(cases):
        cutoff = float(rng.uniform(2.5, 9.0))
        tb = float(rng.uniform(0.5, 1.0) * cutoff) if rng.random() < 0.8 else cutoff
        cells = [random_cell(rng) for _ in range(int(rng.integers(1, 7)))]
        if min(abs(np.linalg.det(l)) for l, _ in cells) < 4.0:
            continue
        g = batch_from_arrays([l for l, _ in cells], [p for _, p in cells], [np.full(len(p), 14) for _, p in cells], cutoff, tb, device="cuda")
        if int(g[K.NUM_TRIPLETS]) == 0 or int(g[K.NUM_TRIPLETS]) > 6_000_000:
            continue
        (a, ha, fa), (b, hb, fb), path = _topology_buffers(g)
        longest = int(torch.bincount(g[K.EDGE_INDEX][0]).max())
        ok = fa == 0 and fb == 0 and ha == hb and torch.equal(a, b) and (path == 1 or longest > 512)
        if not ok:
            print(f"case {c} FAILED: cutoff {cutoff:.3f} tb {tb:.3f} sizes {[len(p) for _, p in cells]} flags {fa} {fb} hints {ha:#x} {hb:#x} "
                  f"path {path} longest row {longest} equal {torch.equal(a, b)}", flush=True)
            raise SystemExit(1)
        done += 1
        fast += path
        edges += int(g[K.NUM_EDGES])
        trips += int(g[K.NUM_TRIPLETS])
    print(f"{done} batches with triplets of {cases} cases: buffers identical ({edges} edges, {trips} triplets in total); "
          f"the canonical build took its six-launch path on {fast} of them (the rest: rows of more than 512 edges)")


if __name__ == "__main__":
    main()
