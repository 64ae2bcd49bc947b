#!/usr/bin/env python3
"""Exhaustive bank check of the dual-use fp32 LDS image (csrc/m3g_dual_f32.h): index(row, col) = row*64 + (col ^ f(row & 15)).
ds_read_b32 serves 32 lanes per LDS cycle over 32 dword banks (MI355X_MICROARCH.md, LDS); both access patterns of the
v_mfma_f32_16x16x4_f32 A operand -- by rows (lane (m, q) reads W[ob*16 + m][blk*16 + 4q + r]) and by columns (lane (m, q)
reads W[blk*16 + 4q + r][ob*16 + m]) -- must hit 32 distinct banks per 32-lane group.  Also checks the index map is a bijection."""


def f(m):
    return (m & 3) | (((m >> 2) & 1) << 4) | (((m >> 3) & 1) << 3)


def index(row, col):
    return row * 64 + (col ^ f(row & 15))


def main():
    ok = True
    for rows in (64, 128):
        for ob in range(rows // 16):
            for blk in range(4):
                for r in range(4):
                    for grp in range(2):
                        by_rows, by_cols = [], []
                        for lane in range(32 * grp, 32 * grp + 32):
                            m, q = lane & 15, lane >> 4
                            by_rows.append(index(ob * 16 + m, blk * 16 + 4 * q + r) % 32)
                            by_cols.append(index((ob % (rows // 16)) * 16 + 4 * q + r, blk * 16 + m) % 32)
                        if len(set(by_rows)) != 32 or len(set(by_cols)) != 32:
                            ok = False
                            print("conflict", rows, ob, blk, r, grp)
        cells = {index(r, c) for r in range(rows) for c in range(64)}
        ok &= len(cells) == rows * 64 and max(cells) == rows * 64 - 1
    print("dual-use fp32 image: conflict-free and bijective" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
