"""Topology build and hints certificate, timed on the bench graph (ms, mean of 20)."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
from torch_m3gnet.data.synthetic import fcc_cu_graph
from torch_m3gnet.nn.modules import _Topology
g = fcc_cu_graph(10, 10, 25, seed=0).to("cuda")
for _ in range(3):
    t = _Topology(g); t.query_hints()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    t = _Topology(g)
torch.cuda.synchronize()
b = (time.perf_counter() - t0) / n * 1e3
t0 = time.perf_counter()
for _ in range(n):
    t = _Topology(g); t.query_hints()
torch.cuda.synchronize()
bh = (time.perf_counter() - t0) / n * 1e3
print(f"topology build {b:.3f} ms, build + hints {bh:.3f} ms (certificate {bh - b:.3f} ms), hints word {t.query_hints()}")
