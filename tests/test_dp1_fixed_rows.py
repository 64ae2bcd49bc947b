"""The 24-bit fixed-point hand-over rows of the f16x3 mode (csrc/m3g_mfma_common.h: pack24_fixed / unpack24_fixed), restated in
numpy with the same fp32 operations and byte selections: a row quarter (64 columns) is stored as k = round(x s 2^9) on the
power-of-two scale s that puts its largest |x| into [2^12, 2^13) (edge_scale), three bytes per value, and read back as k / (s 2^9).
Checked here: the byte layout round-trips, the error bound the kernels rely on (2^-22 of the quarter's largest value, i.e. the
size of one split product's error in that mode), exactness for zeros / the largest value's sign symmetry, and the range the
magic-number rounding needs (|k| < 2^22).  The GPU path itself is covered by every f16x3 parity test (forces flow through these
rows) and by test_fused_and_split_reverse_kernels_agree (fixed-point rows against fp32 rows)."""
import numpy as np
import pytest

MAGIC = np.float32(12582912.0)   # 1.5 * 2^23


def edge_scale(x):
    """s = 2^(13 - e) with max|x| in [2^(e-1), 2^e), exponents below -100 treated as -100 (m3g_mfma_common.h: edge_scale)."""
    m = np.float32(np.max(np.abs(x)))
    u = max(int(np.float32(m).view(np.uint32)), 0x0D000000) & 0x7F800000
    s = np.uint32(0x85000000 - u).view(np.float32)
    inv = np.uint32(u - 0x06000000).view(np.float32)
    return np.float32(s), np.float32(inv)


def pack24_fixed(v4, s9):
    y = (v4.astype(np.float32) * np.float32(s9) + MAGIC).astype(np.float32)   # one fma per value on the GPU: the product is exact
    b = y.view(np.uint32)
    by = [[(int(w) >> (8 * i)) & 0xFF for i in range(4)] for w in b]
    a, bb, c, d = by
    words = [(a[0], a[1], a[2], bb[0]), (bb[1], bb[2], c[0], c[1]), (c[2], d[0], d[1], d[2])]
    return [w[0] | w[1] << 8 | w[2] << 16 | w[3] << 24 for w in words]


def unpack24_fixed(w, inv9):
    by = [[(x >> (8 * i)) & 0xFF for i in range(4)] for x in w]
    vals = [(by[0][0], by[0][1], by[0][2]), (by[0][3], by[1][0], by[1][1]), (by[1][2], by[1][3], by[2][0]), (by[2][1], by[2][2], by[2][3])]
    out = []
    for lo, mid, hi in vals:
        bits = np.uint32(lo | mid << 8 | hi << 16 | 0x4B << 24)
        out.append((bits.view(np.float32) - MAGIC) * np.float32(inv9))
    return np.array(out, dtype=np.float32)


def roundtrip(x):
    s, inv = edge_scale(x)
    s9, inv9 = np.float32(s * np.float32(512.0)), np.float32(inv * np.float32(1.0 / 512.0))
    out = np.empty_like(x)
    for g in range(0, len(x), 4):
        out[g:g + 4] = unpack24_fixed(pack24_fixed(x[g:g + 4], s9), inv9)
    return out, s


@pytest.mark.parametrize("scale", [1e-9, 1e-4, 1.0, 37.0, 1e6])
def test_error_is_within_2_to_minus_22_of_the_largest_value(scale):
    rng = np.random.default_rng(int(abs(np.log10(scale)) * 7) + 1)
    for _ in range(50):
        x = (rng.standard_normal(64) * scale * 10.0 ** rng.uniform(-6, 0, 64)).astype(np.float32)
        y, s = roundtrip(x)
        m = np.max(np.abs(x))
        assert np.max(np.abs(y.astype(np.float64) - x.astype(np.float64))) <= 2.0 ** -22 * m
        assert 2.0 ** 12 <= m * float(s) < 2.0 ** 13
        assert np.max(np.abs(np.rint(x.astype(np.float64) * float(s) * 512.0))) < 2 ** 22   # the rounding trick's range


def test_zeros_signs_and_exact_values():
    x = np.zeros(64, dtype=np.float32)
    y, _ = roundtrip(x)
    assert np.array_equal(y, x)
    x[3], x[17], x[40] = 3.0, -3.0, 0.75
    y, _ = roundtrip(x)
    assert np.array_equal(y, x)            # values on the 2^-9 grid of the scaled quarter are exact
    x = np.float32(1.2345678) * np.array([(-1) ** i for i in range(64)], dtype=np.float32)
    y, _ = roundtrip(x)
    assert np.array_equal(y[0::2], -y[1::2])


def test_small_entries_keep_an_absolute_not_a_relative_error():
    """An entry 2^-20 of the largest one is stored to within 2^-22 of the LARGEST (a quarter of its own size): what the node
    reverse needs, because every column of a row meets the same outputs of W1b^T."""
    x = np.zeros(64, dtype=np.float32)
    x[0], x[1] = 1.0, np.float32(2.0 ** -20 * 1.37)
    y, _ = roundtrip(x)
    assert y[0] == 1.0 and abs(float(y[1]) - float(x[1])) <= 2.0 ** -22
