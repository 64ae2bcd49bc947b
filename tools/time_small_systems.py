"""Step latency of small systems (fcc Cu supercells of n x n x n conventional cells, cutoff 5 / 4): a small system's step is
the serial latency of its ~36 kernels, not throughput.  Usage: python tools/time_small_systems.py [fp32|bf16x3] [n ...]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
from torch_m3gnet.data.synthetic import fcc_cu_graph  # noqa: E402
from torch_m3gnet.model.build import build_model  # noqa: E402

torch.manual_seed(0)
model = build_model(5.0, 4.0, 3, 3, 95, 64, 3)
model.engine.set_precision(sys.argv[1] if len(sys.argv) > 1 else "fp32")
import os  # noqa: E402
if os.environ.get("M3G_SMALL_TILES"):   # threshold of the split-tile edge kernels (plan option "small_tiles"; 0 = persistent kernels only)
    model.engine.set_option("small_tiles", int(os.environ["M3G_SMALL_TILES"]))
if os.environ.get("M3G_SMALL_TILES_FWD"):   # ... of the forward kernel alone
    model.engine.set_option("small_tiles_fwd", int(os.environ["M3G_SMALL_TILES_FWD"]))
for opt in filter(None, os.environ.get("M3G_ENGINE_OPTIONS", "").split(",")):   # e.g. split_tail=0
    model.engine.set_option(opt.split("=")[0], int(opt.split("=")[1]))
def _cells(tok):   # "6" -> (6, 6, 6); "6,7,8" -> (6, 7, 8)
    v = [int(x) for x in str(tok).split(",")]
    return tuple(v) if len(v) == 3 else (v[0],) * 3


for cells in ([_cells(v) for v in sys.argv[2:]] or [(n,) * 3 for n in (2, 3, 4, 6, 8, 10)]):
    g = fcc_cu_graph(*cells).to("cuda")
    for _ in range(40):   # (the host-side graph build in front of this leaves the GPU idle: 5 warm-up steps were not enough for its clocks)
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        model(g, forces=True, extras=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    atoms = 4 * cells[0] * cells[1] * cells[2]
    print(f"{atoms:6d} atoms  {int(g['edge_index'].shape[1]):8d} edges: {ms:.3f} ms/step = {atoms / ms * 1e3 / 1e6:.3f} M atom-steps/s", flush=True)
