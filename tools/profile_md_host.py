#!/usr/bin/env python3
"""Host-side (Python) profile of the refill iteration of tools/profile_md_iteration.py: where the interpreter spends the time
during which the GPU waits for launches.   python tools/profile_md_host.py [iterations]   (GPU box)"""
import cProfile
import pstats
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402
from torch_m3gnet.data.md import VerletGraph  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda")
model = bench.default_model(dev)
a = 3.61
base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
gi = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(25), indexing="ij"), -1)
pos0 = torch.tensor((gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a, device=dev)
lat = np.diag([10 * a, 10 * a, 25 * a]).astype(float)
vg = VerletGraph([lat], [np.full(pos0.size(0), 29)], 5.0, 4.0, skin=0.5, device=dev)
poss = [pos0 + (torch.rand(pos0.shape, device=dev, dtype=torch.float64) - 0.5) * 0.05 for _ in range(8)]


def run(n):
    for i in range(n):
        model(vg.update(poss[i % 8], force="refill"), forces=True, extras=False)
    torch.cuda.synchronize()


run(5)
pr = cProfile.Profile()
pr.enable()
run(iters)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"per iteration (us) over {iters} refill iterations; tottime = inside the function itself")
rows = []
for (fn, line, name), (cc, nc, tt, ct, _) in st.stats.items():
    rows.append((tt / iters * 1e6, ct / iters * 1e6, nc / iters, f"{Path(fn).name}:{line} {name}"))
rows.sort(reverse=True)
for tt, ct, nc, label in rows[:45]:
    print(f"{tt:8.1f} tot {ct:8.1f} cum {nc:6.1f} calls  {label}")
