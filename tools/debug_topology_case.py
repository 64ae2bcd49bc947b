import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path('/root/repo') if Path('/root/repo/tests').exists() else Path.cwd()
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests", ROOT / "tests/checkers"):
    sys.path.insert(0, str(p))
from fuzz_graph_build import random_cell
from test_gpu_graph_build import _topology_buffers
from torch_m3gnet.data import MaterialGraphKey as K
from torch_m3gnet.data.graph_gpu import batch_from_arrays
target = int(sys.argv[1])
rng = np.random.default_rng(0)
for c in range(target + 1):
    cutoff = float(rng.uniform(2.5, 9.0))
    tb = float(rng.uniform(0.5, 1.0) * cutoff) if rng.random() < 0.8 else cutoff
    cells = [random_cell(rng) for _ in range(int(rng.integers(1, 7)))]
    if min(abs(np.linalg.det(l)) for l, _ in cells) < 4.0:
        continue
    g = batch_from_arrays([l for l, _ in cells], [p for _, p in cells], [np.full(len(p), 14) for _, p in cells], cutoff, tb, device="cuda")
    if c < target: continue
    N, E, T, S = int(g[K.NUM_NODES]), int(g[K.NUM_EDGES]), int(g[K.NUM_TRIPLETS]), len(cells)
    (a, ha, fa), (b, hb, fb), path = _topology_buffers(g)
    al = lambda x: (x + 255) // 256 * 256
    W = E // 128 + 2
    layout = [("src", (E+1)*4), ("dst", (E+1)*4), ("row_ptr", (N+1)*4), ("in_ptr", (N+1)*4), ("in_edge", (E+1)*4), ("in_pair", 2*(E+1)*4), ("in_pos", (E+1)*4),
              ("t1_ptr", (E+1)*4), ("t1_e2", (T+1)*4), ("t2_ptr", (E+1)*4), ("t2_e1", (T+1)*4), ("act_list", (E+1)*4), ("act_scan", (E+2)*4), ("act_id", (E+1)*4),
              ("arow_ptr", (N+2)*4), ("act_dst", (E+1)*4), ("tb_win", 6*W*4), ("tb_fast", 2*W*4), ("t1_e2c", (T+1)*4), ("t2_e1c", (T+1)*4), ("t1_b", T+16), ("t2_b", T+16),
              ("batch", (N+1)*4), ("struct_ptr", (S+2)*4), ("flags", 64)]
    print("N E T S", N, E, T, S, "bytes", a.numel())
    off = 0
    for name, nb in layout:
        seg_a, seg_b = a[off:off+nb], b[off:off+nb]
        if not torch.equal(seg_a, seg_b):
            bad = torch.nonzero(seg_a != seg_b).flatten()
            print(name, "differs at", bad.numel(), "bytes; first byte", int(bad[0]), "-> element", int(bad[0]) // (1 if name.endswith('_b') else 4))
            if not name.endswith('_b'):
                ia = seg_a[: nb // 4 * 4].view(torch.int32); ib = seg_b[: nb // 4 * 4].view(torch.int32)
                idx = torch.nonzero(ia != ib).flatten()[:8]
                print("   idx", idx.tolist(), "general", ia[idx].tolist(), "canonical", ib[idx].tolist())
        off += al(nb)
    print("total accounted", off)
    off = 0
    for name, nb in layout:
        if name in ("flags", "tb_win", "tb_fast"):
            ia = a[off:off+nb].view(torch.int32); ib = b[off:off+nb].view(torch.int32)
            print(name, "general", ia[:12].tolist(), "canonical", ib[:12].tolist())
        off += al(nb)
    print("hints", hex(ha), hex(hb), "path", path)
