#!/usr/bin/env python3
"""CPU numerics study: would a split-precision matrix path hold the parity budget?

Emulates the edge-block matrix products (the MFMA work) with operands split into low-precision parts and fp32
accumulation, inside the staged oracle (oracle/staged.py MM hook), and reports energy / force errors against the
fp64 oracle on the golden fixtures.  Schemes:
  fp32      plain fp32 (what the shipped kernels do)
  bf16x3    a = ah + al (bf16, bf16): ah*bh + ah*bl + al*bh
  f16x3     the engine's default mode (csrc/m3g_mfma_common.h): every activation / gradient row scaled by the power of two that
            puts its largest entry into [2^12, 2^13), the weights by one power of two, both split in two f16 parts:
            (ah*bh + ah*bl + al*bh) / (scales)
  f16x3s    round 1's variant: unscaled high part, low part scaled by 2^11 (no range protection for small gradients)
  f16x1     single f16 (for scale)
Second part: K = 64 dot products of synthetic operands (O(1) activations, 1e-6 gradients, rows spanning six decades, saturated
activations), error relative to sum |w||x|, each scheme against fp64.
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
for p in (ROOT, ROOT / "tests", ROOT / "torch-m3gnet_amd"):
    sys.path.insert(0, str(p))
from helpers import CASES, load_oracle_case, rel_err  # noqa: E402
from oracle import m3gnet_oracle as orc, staged  # noqa: E402


def split(x, dt, scale):
    hi = x.to(dt).float()
    lo = ((x - hi) * scale).to(dt).float()
    return hi, lo


def pow2_scale(x, dim):
    m = x.abs().amax(dim=dim, keepdim=True).clamp_min(1e-30)
    return torch.pow(2.0, 12 - torch.floor(torch.log2(m)))


def mm_f16x3(a, b):
    """a [rows, K] activations / gradients (one scale per row = per edge), b [K, N] weights (one scale for the matrix)."""
    sa, sb = pow2_scale(a, 1), pow2_scale(b.reshape(1, -1), 1)
    ah, al = split(a * sa, torch.float16, 1.0)
    bh, bl = split(b * sb, torch.float16, 1.0)
    return (ah @ bh + ah @ bl + al @ bh) / (sa * sb)


def make_mm(kind):
    if kind == "fp32":
        return lambda a, b: a @ b
    if kind == "f16x3":
        return mm_f16x3
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
    for kind in ("fp32", "f16x3", "f16x3s", "bf16x3", "f16x1"):
        staged.MM = make_mm(kind)
        out = staged.forward_backward(p, cfg, c, g)
        e = float(((out["total_energy"].double() - ref["total_energy"]).abs() / ref["total_energy"].abs()).max())
        f = rel_err(out["forces"], ref["forces"])
        row.append(f"{kind}: E {e:.1e} F {f:.1e}")
    print(f"{case}_{mode:3s} " + " | ".join(row), flush=True)
staged.MM = lambda a, b: a @ b


# ---- part 2: dot-product level ------------------------------------------------------------------------------------------------------
def dots(w, x, kind):
    if kind == "fp32":
        acc = torch.zeros(x.shape[0], w.shape[0])
        for k in range(w.shape[1]):
            acc = acc + x[:, k:k + 1] * w[:, k][None, :]
        return acc
    return make_mm(kind)(x, w.T.contiguous())


torch.manual_seed(0)
print("\nK = 64 dot products, max and rms error / sum |w||x| against fp64")
for name, gen in (("O(1) activations, w ~ N(0, 1/8)", lambda: (torch.randn(64, 64) / 8, torch.randn(256, 64))),
                  ("gradients of 1e-6", lambda: (torch.randn(64, 64) / 8, torch.randn(256, 64) * 1e-6)),
                  ("rows spanning six decades", lambda: (torch.randn(64, 64) / 8, torch.randn(256, 64) * torch.pow(10, -6 * torch.rand(256, 64)))),
                  ("saturated activations (|x| up to 30)", lambda: (torch.randn(64, 64) / 2, torch.randn(256, 64) * 10))):
    w, x = gen()
    ref = x.double() @ w.double().T
    scale = x.double().abs() @ w.double().abs().T
    row = []
    for kind in ("fp32", "f16x3", "f16x3s", "bf16x3"):
        err = (dots(w, x, kind).double() - ref).abs() / scale
        row.append(f"{kind}: max {float(err.max()):.1e} rms {float(err.pow(2).mean().sqrt()):.1e}")
    print(f"  {name:38s} " + " | ".join(row), flush=True)
