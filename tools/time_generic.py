"""The any-size path (csrc/m3g_generic.hip, edge_kernel = 2) at BASELINE config-3 size, timed:
    python tools/time_generic.py [atoms-cells nx ny nz]
  * default model (D = 64) on the any-size path against the same model on the MFMA path (fp32 mode), same graph;
  * embedding_dim = 128 (and 96): only the any-size path runs these widths (the MFMA kernels hold D <= 64 in LDS).
Reports ms/step and time per useful FLOP relative to the D = 64 MFMA step (useful FLOPs scale with D^2 for the dense layers:
SURVEY.md 8(d) -- 2 B (E (2 MLPs x (3 D^2 ... factorised: D^2 + D^2) ...))), so "per FLOP" = ms / (D / 64)^2."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
from torch_m3gnet.data.synthetic import fcc_cu_graph  # noqa: E402
from torch_m3gnet.model.build import build_model  # noqa: E402

cells = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (10, 10, 25)
graph = fcc_cu_graph(*cells, seed=0).to("cuda")
n_atoms = int(graph["pos"].size(0))


def timed(model, reps):
    for _ in range(2):
        model(graph.clone(), forces=True, extras=False)
    g = graph.clone()
    model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


base = None
for dim, kernel, reps in ((64, 1, 20), (64, 2, 3), (96, 2, 3), (128, 2, 3)):
    torch.manual_seed(0)
    model = build_model(5.0, 4.0, 3, 3, 95, dim, 3).to("cuda")
    if not (dim > 64):
        model.engine.set_option("edge_kernel", kernel)
    if kernel == 1:
        model.engine.set_precision("fp32")   # the any-size path computes in exact fp32: compare like with like
    ms = timed(model, reps)
    if base is None:
        base = ms
    per_flop = ms / (dim / 64) ** 2
    print(f"{n_atoms} atoms, embedding_dim {dim:3d}, {'MFMA kernels (fp32 mode)' if kernel == 1 else 'any-size path           '}: "
          f"{ms:9.3f} ms/step = {n_atoms / ms * 1e3 / 1e6:6.3f} M atom-steps/s   per useful FLOP vs the D = 64 MFMA step: {per_flop / base:6.1f} x", flush=True)
    del model
    torch.cuda.empty_cache()
