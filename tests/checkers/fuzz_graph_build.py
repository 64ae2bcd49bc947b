"""Random cells through the GPU graph builder (m3g_neighbor_*, m3g_threebody_*) against the host numpy builder, element by
element (index tensors identical, fp64 distances to 1e-12): random lattices (cubic to strongly sheared, 2-25 A), 1-120 atoms,
cutoffs 2.5-9 A, batches of 1-6 structures.  Usage: python tests/checkers/fuzz_graph_build.py [cases] [seed]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from test_gpu_graph_build import _assert_same, _gpu, _host  # noqa: E402


def random_cell(rng):
    kind = rng.integers(0, 4)
    a = rng.uniform(2.0, 25.0, 3) if kind else np.full(3, rng.uniform(2.0, 25.0))
    lat = np.diag(a)
    if kind >= 2:
        lat = lat + rng.uniform(-0.45, 0.45, (3, 3)) * a.min()
    if kind == 3 and rng.random() < 0.5:
        lat = lat[[1, 0, 2]]   # left-handed
    vol = abs(np.linalg.det(lat))
    n = int(min(120, max(1, rng.integers(1, 8) if vol < 60 else vol * rng.uniform(0.002, 0.05))))
    pos = rng.uniform(-0.5, 1.5, (n, 3)) @ lat
    return lat, pos


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    edges = trips = 0
    for c in range(cases):
        cutoff = float(rng.uniform(2.5, 9.0))
        tb = float(rng.uniform(0.5, 1.0) * cutoff) if rng.random() < 0.8 else cutoff
        cells = [random_cell(rng) for _ in range(int(rng.integers(1, 7)))]
        vols = [abs(np.linalg.det(l)) for l, _ in cells]
        if min(vols) < 4.0:
            continue
        # the host builder works one structure at a time; concatenate with the batch offsets (material_graph.py:122-130)
        hosts = [_host(l, p, cutoff, tb) for l, p in cells]
        if sum(h[3].shape[1] for h in hosts) > 6_000_000:
            continue
        n_off = np.cumsum([0] + [len(p) for _, p in cells])
        e_off = np.cumsum([0] + [h[0].shape[1] for h in hosts])
        host = [np.concatenate([h[0] + n_off[i] for i, h in enumerate(hosts)], axis=1),
                np.concatenate([h[1] for h in hosts]), np.concatenate([h[2] for h in hosts]),
                np.concatenate([h[3] + e_off[i] for i, h in enumerate(hosts)], axis=1),
                np.concatenate([h[4] for h in hosts]), np.concatenate([h[5] for h in hosts])]
        gpu = _gpu([l for l, _ in cells], [p for _, p in cells], cutoff, tb)
        try:
            _assert_same(host, gpu)
        except AssertionError:
            print(f"case {c} FAILED: cutoff {cutoff:.3f} tb {tb:.3f} sizes {[len(p) for _, p in cells]}", flush=True)
            raise
        edges += host[0].shape[1]
        trips += host[3].shape[1]
    print(f"{cases} cases: identical ({edges} edges, {trips} triplets in total)")


if __name__ == "__main__":
    main()
