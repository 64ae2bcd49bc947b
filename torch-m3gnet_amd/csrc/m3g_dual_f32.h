// Dual-use fp32 LDS weight image for v_mfma_f32_16x16x4_f32 chains: ONE copy of a [ROWS][64] matrix W serves as the A operand
// of Y^T = W X^T (lane (m, q) reads W[ob*16 + m][k], k = blk*16 + 4q + r) and of dX^T = W^T dY^T (lane (m, q) reads
// W[k][ob*16 + m]).  fp32 elements are individually addressable, so "dual use" is only a matter of banks:
//     index(row, col) = row*64 + (col ^ f(row & 15)),   f(m) = (m & 3) | ((m >> 2) & 1) << 4 | ((m >> 3) & 1) << 3   [dwords]
// ds_read_b32 serves 32 lanes per cycle over 32 banks (MI355X_MICROARCH.md, LDS).  Row use: the 32 lanes (m, q in {0,1}) of a
// group read 16 rows x 2 columns that differ in bit 2; f maps the 16 rows onto 16 distinct values of bits {0,1,3,4} -> 32
// banks.  Transposed use: the lanes read 2 x 4-row groups (q) x 16 adjacent columns (m); f of row 4q + r is r | (q & 1) << 4 |
// (q >> 1) << 3, so q moves bit 4 while m covers bits 0-3 -> 32 banks.  (Checked exhaustively: tools/dual_f32_bank_check.py.)
#pragma once
#include "m3g_mfma_common.h"

namespace m3g {

__host__ __device__ inline int dual32_f(int m) { return (m & 3) | (((m >> 2) & 1) << 4) | (((m >> 3) & 1) << 3); }
__host__ __device__ inline int dual32_index(int row, int col) { return row * 64 + (col ^ dual32_f(row & 15)); }

#if defined(M3G_F32_STAGGER_PRIO)
#define M3G_DUAL32_PRIO(p) do { if (p) { const int cls = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)); if (cls == 0) __builtin_amdgcn_s_setprio(3); else if (cls == 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); } else __builtin_amdgcn_s_setprio(0); } while (0)
#elif !defined(M3G_NO_F32_CHAIN_PRIO)
#define M3G_DUAL32_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define M3G_DUAL32_PRIO(p) ((void)0)
#endif

// acc[AOFF + ob] += W[(RB0 + ob)*16 .. +16][0..64) . x[XOFF .. XOFF + 4)     (x: 64 input features in accumulator layout)
template <int OB, int XOFF = 0, int AOFF = 0, int RB0 = 0, int NX, int NA>
__device__ __forceinline__ void chain_dual32(const float* img, const f32x4 (&x)[NX], f32x4 (&acc)[NA], int lane) {
  static_assert(XOFF + 4 <= NX && AOFF + OB <= NA, "chain_dual32 operand out of range");
  const int m = lane & 15, q = lane >> 4;
  const int base = m * 64 + ((4 * q) ^ dual32_f(m));
  M3G_DUAL32_PRIO(1);
  static_for<4>([&]<int blk>() {
    static_for<4>([&]<int r>() {
      const float b = x[XOFF + blk][r];
      const float* p = img + (base ^ (blk * 16 + r));
      static_for<OB>([&]<int ob>() { acc[AOFF + ob] = mfma16(p[(RB0 + ob) * 1024], b, acc[AOFF + ob]); });
    });
  });
  M3G_DUAL32_PRIO(0);
}

// transposed: acc[AOFF + ob] (64 input-feature gradients, ob = 0..3) += sum_o W[o][ob*16 ..] d[o]; d holds NB 16-row blocks of
// output-feature gradients starting at block DOFF, i.e. rows (KB0 + blk)*16 .. of W
template <int NB, int DOFF = 0, int AOFF = 0, int KB0 = 0, int ND, int NA>
__device__ __forceinline__ void chain_dual32_t(const float* img, const f32x4 (&d)[ND], f32x4 (&acc)[NA], int lane) {
  static_assert(DOFF + NB <= ND && AOFF + 4 <= NA, "chain_dual32_t operand out of range");
  const int m = lane & 15, q = lane >> 4;
  const int base = q * 256 + (m ^ (((q & 1) << 4) | ((q >> 1) << 3)));
  M3G_DUAL32_PRIO(1);
  static_for<NB>([&]<int blk>() {
    static_for<4>([&]<int r>() {
      const float b = d[DOFF + blk][r];
      static_for<4>([&]<int ob>() {
        const float* p = img + (KB0 + blk) * 1024 + r * 64 + (base ^ ((ob * 16) ^ r));
        acc[AOFF + ob] = mfma16(*p, b, acc[AOFF + ob]);
      });
    });
  });
  M3G_DUAL32_PRIO(0);
}

// host: img receives rows*64 floats; get(row, col) with col < 64
template <class F>
inline void pack_dual32_image(float* img, int rows, F get) {
  for (int row = 0; row < rows; ++row)
    for (int col = 0; col < 64; ++col) img[dual32_index(row, col)] = get(row, col);
}

}  // namespace m3g
