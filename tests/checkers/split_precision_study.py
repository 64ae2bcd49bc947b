#!/usr/bin/env python3
"""CPU numerics study: would a split-precision matrix path hold the parity budget?

Emulates the edge-block matrix products (the MFMA work) with operands split into low-precision parts and fp32
accumulation, inside the staged oracle (oracle/staged.py MM hook), and reports energy / force errors against the
fp64 oracle on the golden fixtures.  Schemes:
  fp32      plain fp32 (what the shipped kernels do)
  bf16x3    a = ah + al (bf16, bf16): ah*bh + ah*bl + al*bh
  f16x3s    a = ah + al/2^11 (f16, f16 with the low part scaled by 2^11): ah*bh + (ah*bl' + al'*bh)/2^11
  f16x1     single f16 (for scale)
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
for p in (ROOT, ROOT / "tests"):
    sys.path.insert(0, str(p))
from helpers import CASES, load_oracle_case, rel_err  # noqa: E402
from oracle import m3gnet_oracle as orc, staged  # noqa: E402


def split(x, dt, scale):
    hi = x.to(dt).float()
    lo = ((x - hi) * scale).to(dt).float()
    return hi, lo


def make_mm(kind):
    if kind == "fp32":
        return lambda a, b: a @ b
    if kind == "f16x1":
        return lambda a, b: a.half().float() @ b.half().float()
    dt, scale = (torch.bfloat16, 1.0) if kind == "bf16x3" else (torch.float16, 2048.0)

    def mm(a, b):
        ah, al = split(a, dt, scale)
        bh, bl = split(b, dt, scale)
        return ah @ bh + (ah @ bl + al @ bh) / scale
    return mm


torch.set_num_threads(8)
for case, mode in CASES:
    p64, cfg64, c64, g64, _ = load_oracle_case(case, mode, dtype=torch.float64)
    ref = orc.energy_forces(p64, cfg64, c64, g64, legendre_backward="exact")
    p, cfg, c, g, _ = load_oracle_case(case, mode)
    row = []
    for kind in ("fp32", "f16x3s", "bf16x3", "f16x1"):
        staged.MM = make_mm(kind)
        out = staged.forward_backward(p, cfg, c, g)
        e = float(((out["total_energy"].double() - ref["total_energy"]).abs() / ref["total_energy"].abs()).max())
        f = rel_err(out["forces"], ref["forces"])
        row.append(f"{kind}: E {e:.1e} F {f:.1e}")
    print(f"{case}_{mode:3s} " + " | ".join(row), flush=True)
staged.MM = lambda a, b: a @ b
