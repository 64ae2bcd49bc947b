"""Energy-only calls (forces=False: forward pass only, nothing saved for a reverse pass) against full calls on the bench cell."""
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
g = fcc_cu_graph(10, 10, 25).to("cuda")
for prec in ("fp32", "bf16x3"):
    model.engine.set_precision(prec)
    for forces in (True, False):
        for _ in range(3):
            model(g, forces=forces, extras=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model(g, forces=forces, extras=False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{prec:7s} forces={forces!s:5s}: {ms:.3f} ms/step = {10000 / ms * 1e3 / 1e6:.2f} M atom-steps/s", flush=True)
