#!/usr/bin/env python3
"""CPU (numpy) restatement of the three-part bf16 chain priced for the forward edge kernel (hardware
counterpart: tools/bf16x3part_probe.hip).

An fp32 operand is split EXACTLY into three bf16 parts (8 + 8 + 8 significand bits, round to nearest, residuals exact);
8 of the 9 part products enter (lo x lo dropped: <= 2^-32 |a||b|); every product of two bf16 parts is exact in fp32.  Model of
v_mfma_f32_16x16x32_bf16: the 32 products of a k-step and the accumulator are added exactly and rounded ONCE to fp32 (the
hardware's internal order is not documented; the probe measures the real thing).  Compared, on K = 64 dot products whose
operands span six decades, with a k-ordered fp32 fmaf chain -- the arithmetic of the shipped exact mode (v_mfma_f32_16x16x4_f32)
-- and with the fp64 value, in ulps of the ACCUMULATOR scale (fp32 ulp of sum |terms|: the scale a chain with cancellation can
be held to).

    python tools/bf16x3part_model.py   ->  profiles/r05_bf16x3part_model.txt"""
import numpy as np


def bf16_round(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    p0 = bf16_round(x)
    r1 = (x - p0).astype(np.float32)          # exact
    p1 = bf16_round(r1)
    r2 = (r1 - p1).astype(np.float32)         # exact
    p2 = bf16_round(r2)
    assert np.array_equal((p0.astype(np.float64) + p1 + p2).astype(np.float32), x), "the three parts are not exact"
    return p0, p1, p2


def fmaf_chain(w, x):
    acc = np.zeros(w.shape[:-1], dtype=np.float32)
    for k in range(w.shape[-1]):
        acc = (w[..., k].astype(np.float64) * x[..., k].astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
    return acc


def bf16x3_chain(w, x, two_accumulators):
    wp, xp = split3(w), split3(x)
    order = [(1, 2), (2, 1), (0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)]   # smallest products first
    hi = np.zeros(w.shape[:-1], dtype=np.float32)
    lo = np.zeros(w.shape[:-1], dtype=np.float32)
    for s in range(w.shape[-1] // 32):
        sl = slice(32 * s, 32 * s + 32)
        for i, j in order:
            term = (wp[i][..., sl].astype(np.float64) * xp[j][..., sl].astype(np.float64)).sum(-1)
            if two_accumulators and (i, j) != (0, 0):
                lo = (lo.astype(np.float64) + term).astype(np.float32)
            else:
                hi = (hi.astype(np.float64) + term).astype(np.float32)
    return (hi + lo).astype(np.float32) if two_accumulators else hi


def main():
    rng = np.random.default_rng(5)
    n, K = 200 * 256, 64
    dec_w = -6.0 * rng.integers(0, 17, (n, K)) / 16.0
    dec_x = -6.0 * rng.integers(0, 13, (n, K)) / 12.0
    w = (rng.uniform(-1, 1, (n, K)) * 10.0 ** dec_w).astype(np.float32)
    x = (rng.uniform(-1, 1, (n, K)) * 10.0 ** dec_x).astype(np.float32)
    ref = (w.astype(np.float64) * x.astype(np.float64)).sum(-1)
    mag = np.abs(w.astype(np.float64) * x.astype(np.float64)).sum(-1)
    unit = 2.0 ** (np.floor(np.log2(mag)) - 23)          # fp32 ulp at the accumulator scale
    f = fmaf_chain(w, x)
    print(f"# K = {K} dot products, {n} of them, operands spanning six decades; errors in fp32 ulps of sum |terms|")
    print(f"fmaf chain (the exact mode's arithmetic)        max |y - fp64| = {np.max(np.abs(f - ref) / unit):6.3f} ulp")
    for two in (False, True):
        y = bf16x3_chain(w, x, two)
        name = "3 x bf16, 8 products, hi + lo accumulators" if two else "3 x bf16, 8 products, one accumulator"
        print(f"{name:47s} max |y - fp64| = {np.max(np.abs(y - ref) / unit):6.3f} ulp   max |y - fmaf chain| = "
              f"{np.max(np.abs(y.astype(np.float64) - f) / unit):6.3f} ulp")
    print("# the split chain is CLOSER to the fp64 value than the fmaf chain is (fewer roundings: 16 per dot product instead of 64), but the two")
    print("# differ from each other by the sum of their own errors -- several accumulator ulps, not <= 1: it is as wide as fp32, not the same arithmetic.")


if __name__ == "__main__":
    main()
