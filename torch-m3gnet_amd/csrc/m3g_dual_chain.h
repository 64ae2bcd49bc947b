// bf16x3 MFMA chains over the dual-use LDS weight image (layout and bank analysis: m3g_dual_image.h).
//   chain_dual    acc[ob] += W[rows ob*16..][:] . x          (A operand by row reads, 2 x ds_read_b64 per fragment part)
//   chain_dual_t  acc[ob] += W[:][cols ob*16..]^T . d        (A operand by 2 x ds_read_b64_tr_b16 per fragment part)
// Both consume accumulator-layout operands (block b, register r of lane (m, q) = feature b*16 + 4q + r of edge m).
#pragma once
#include <cstring>

#include "m3g_dual_image.h"
#include "m3g_mfma_common.h"

// Wave priority around the MFMA chains: with two waves per SIMD the arbiter otherwise lets the other wave's VALU stream
// delay the chain's MFMA issue; raised priority keeps the matrix pipe fed while that VALU work fills the gaps
// (measured on the fused reverse kernel: 0.973 -> 0.925 ms per step).
#ifndef M3G_NO_CHAIN_PRIO
#define M3G_CHAIN_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define M3G_CHAIN_PRIO(p) ((void)0)
#endif

namespace m3g {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
using lds_s16x4_ptr = __attribute__((address_space(3))) s16x4*;

__device__ __forceinline__ bf16x8 join_halves(s16x4 a, s16x4 b) {
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// rows [RB0*16, (RB0+OB)*16) of a ROWS-row image; x holds 64 input features in blocks XOFF .. XOFF+3
template <int OB, int KS, int ROWS, int XOFF = 0, int AOFF = 0, int RB0 = 0, int NX, int NA>
__device__ __forceinline__ void chain_dual(const float* img, const f32x4 (&x)[NX], f32x4 (&acc)[NA], int lane) {
  static_assert(KS == 2, "the image holds 64 input features");
  static_assert(XOFF + 2 * KS <= NX && AOFF + OB <= NA && (RB0 + OB) * 16 <= ROWS, "chain_dual operand out of range");
  const int m = lane & 15, q = lane >> 4, sw = dual_swz(m);
  const char* base = reinterpret_cast<const char*>(img) + (m >> 3) * 1024 + (m & 7) * 64;
  M3G_CHAIN_PRIO(1);
  static_for<KS>([&]<int s>() {
    bf16x8 bh, bl;
    split8(x[XOFF + 2 * s], x[XOFF + 2 * s + 1], bh, bl);
    const char* u = base + (((s * 4 + q) ^ sw) << 3);
    {  // term-major order: the three products of one accumulator sit OB MFMAs apart instead of back to back (a third
       // fewer s_nop hazard fills in the fused reverse kernel, ~1 % of its time; 12 more live registers)
      constexpr int plane = 512, lo = ROWS * 128;
      bf16x8 ah[OB];
      static_for<OB>([&]<int ob>() {
        constexpr int roff = (RB0 + ob) * 2048;
        ah[ob] = join_halves(*(const s16x4*)(u + roff), *(const s16x4*)(u + roff + plane));
      });
      static_for<OB>([&]<int ob>() { acc[AOFF + ob] = mfma_bf16(ah[ob], bh, acc[AOFF + ob]); });
      static_for<OB>([&]<int ob>() { acc[AOFF + ob] = mfma_bf16(ah[ob], bl, acc[AOFF + ob]); });
      static_for<OB>([&]<int ob>() {
        constexpr int roff = (RB0 + ob) * 2048;
        const bf16x8 al = join_halves(*(const s16x4*)(u + roff + lo), *(const s16x4*)(u + roff + lo + plane));
        acc[AOFF + ob] = mfma_bf16(al, bh, acc[AOFF + ob]);
      });
    }
  });
  M3G_CHAIN_PRIO(0);
}

// transposed: d holds ROWS/16 blocks of output-feature gradients starting at block DOFF (KS = ROWS/32 k-steps);
// acc[AOFF .. AOFF+3] receive the 64 input-feature gradients.  EXEC must be all ones (ds_read_b64_tr_b16).
template <int OB, int KS, int ROWS, int DOFF = 0, int AOFF = 0, int KB0 = 0, int ND, int NA>
__device__ __forceinline__ void chain_dual_t(const float* img, const f32x4 (&d)[ND], f32x4 (&acc)[NA], int lane) {
  static_assert(OB == 4, "64 input features");
  static_assert(DOFF + 2 * KS <= ND && AOFF + OB <= NA && (KB0 + 2 * KS) * 16 <= ROWS, "chain_dual_t operand out of range");
  const int q = lane >> 4, qp = (lane & 15) >> 2, p = lane & 3;
  const int row_lo = 4 * q + qp, sw = dual_swz(row_lo);
  const char* base = reinterpret_cast<const char*>(img) + (row_lo >> 3) * 1024 + (row_lo & 7) * 64;
  M3G_CHAIN_PRIO(1);
  static_for<KS>([&]<int s>() {
    bf16x8 bh, bl;
    split8(d[DOFF + 2 * s], d[DOFF + 2 * s + 1], bh, bl);
    {
      constexpr int plane = 512, lo = ROWS * 128;
      constexpr int r0 = (KB0 + 2 * s) * 2048, r1 = (KB0 + 2 * s + 1) * 2048;
      bf16x8 ah[OB];
      static_for<OB>([&]<int ob>() {
        const char* u = base + (ob & 1) * plane + ((((ob >> 1) * 4 + p) ^ sw) << 3);
        ah[ob] = join_halves(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(u + r0)),
                             __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(u + r1)));
      });
      static_for<OB>([&]<int ob>() { acc[AOFF + ob] = mfma_bf16(ah[ob], bh, acc[AOFF + ob]); });
      static_for<OB>([&]<int ob>() { acc[AOFF + ob] = mfma_bf16(ah[ob], bl, acc[AOFF + ob]); });
      static_for<OB>([&]<int ob>() {
        const char* u = base + (ob & 1) * plane + ((((ob >> 1) * 4 + p) ^ sw) << 3);
        const bf16x8 al = join_halves(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(u + r0 + lo)),
                                      __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(u + r1 + lo)));
        acc[AOFF + ob] = mfma_bf16(al, bh, acc[AOFF + ob]);
      });
    }
  });
  M3G_CHAIN_PRIO(0);
}

// ---- f16x3 mode: the same dual-use image holding fp16 parts of the SCALED weights (m3g_mfma_common.h); acc[AOFF + ob] += (W x)
// in TRUE units -- the scaled sum of one row block at a time is folded in with one fma per value (x's per-edge scale is taken
// over the chain's own input blocks) ------------------------------------------------------------------------------------------
__device__ __forceinline__ f16x8 join_halves_h(s16x4 a, s16x4 b) {
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, v);
}
// A operands of one 16-row block (both parts, KS k-steps) -- requested one block AHEAD of the MFMAs that consume them: the LDS
// round trip (64+ cycles, more with eight waves reading) is longer than the two or three MFMAs the compiler's own schedule puts
// between a ds_read and its use, and with two waves per SIMD those waits are exposed (M3G_CHAIN_PREFETCH).
template <int KS>
struct DualA { f16x8 h[KS], l[KS]; };
#ifndef M3G_NO_CHAIN_PREFETCH
#define M3G_CHAIN_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define M3G_CHAIN_FENCE() ((void)0)
#endif
template <int OB, int KS, int ROWS, int XOFF = 0, int AOFF = 0, int RB0 = 0, int NX, int NA>
__device__ __forceinline__ void chain_dual_h(const float* img, const f32x4 (&x)[NX], f32x4 (&acc)[NA], int lane, float w_inv) {
  static_assert(KS == 2, "the image holds 64 input features");
  static_assert(XOFF + 2 * KS <= NX && AOFF + OB <= NA && (RB0 + OB) * 16 <= ROWS, "chain_dual_h operand out of range");
  const int m = lane & 15, q = lane >> 4, sw = dual_swz(m);
  const char* base = reinterpret_cast<const char*>(img) + (m >> 3) * 1024 + (m & 7) * 64;
  constexpr int plane = 512, lo = ROWS * 128;
  auto fetch = [&]<int ob>() {
    constexpr int roff = (RB0 + ob) * 2048;
    DualA<KS> a;
    static_for<KS>([&]<int s>() {
      const char* u = base + (((s * 4 + q) ^ sw) << 3);
      a.h[s] = join_halves_h(*(const s16x4*)(u + roff), *(const s16x4*)(u + roff + plane));
#ifdef M3G_DIAG_NO_AL   // timing diagnostic only (wrong results): no LDS reads of the low-part image
      a.l[s] = a.h[s];
#else
      a.l[s] = join_halves_h(*(const s16x4*)(u + roff + lo), *(const s16x4*)(u + roff + lo + plane));
#endif
    });
    return a;
  };
  // the first block's operands travel while the vector ALU finds the scale and splits x
  DualA<KS> cur = fetch.template operator()<0>();
  M3G_CHAIN_FENCE();
  const EdgeScale sc = edge_scale<2 * KS, XOFF>(x);
  const HalfB<KS> b = split_h<KS, XOFF>(x, sc.s);
  const float inv = sc.inv * w_inv;
  M3G_CHAIN_PRIO(1);
  static_for<OB>([&]<int ob>() {
    DualA<KS> nxt = cur;
    if constexpr (ob + 1 < OB) nxt = fetch.template operator()<ob + 1>();
    M3G_CHAIN_FENCE();
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    static_for<KS>([&]<int s>() {
      t = mfma_f16(cur.h[s], b.hi[s], t);
#ifndef M3G_DIAG_H1
      t = mfma_f16(cur.h[s], b.lo[s], t);
      t = mfma_f16(cur.l[s], b.hi[s], t);
#elif defined(M3G_DIAG_H1_KEEP)   // ... with the low-part reads and splits kept alive: the MFMAs alone
      asm volatile("" ::"v"(cur.l[s]), "v"(b.lo[s]));
#endif
    });
    acc[AOFF + ob] = t * inv + acc[AOFF + ob];   // (vector form: the compiler emits two v_pk_fma_f32)
    M3G_CHAIN_FENCE();
    cur = nxt;
  });
  M3G_CHAIN_PRIO(0);
}
// `used` (optional): receives the per-edge scale the chain found for d, for a caller that stores d on the same scale (pack24_fixed)
template <int OB, int KS, int ROWS, int DOFF = 0, int AOFF = 0, int KB0 = 0, int ND, int NA>
__device__ __forceinline__ void chain_dual_t_h(const float* img, const f32x4 (&d)[ND], f32x4 (&acc)[NA], int lane, float w_inv,
                                               EdgeScale* used = nullptr) {
  static_assert(OB == 4, "64 input features");
  static_assert(DOFF + 2 * KS <= ND && AOFF + OB <= NA && (KB0 + 2 * KS) * 16 <= ROWS, "chain_dual_t_h operand out of range");
  const int q = lane >> 4, qp = (lane & 15) >> 2, p = lane & 3;
  const int row_lo = 4 * q + qp, sw = dual_swz(row_lo);
  const char* base = reinterpret_cast<const char*>(img) + (row_lo >> 3) * 1024 + (row_lo & 7) * 64;
  constexpr int plane = 512, lo = ROWS * 128;
  auto fetch = [&]<int ob>() {
    const char* u = base + (ob & 1) * plane + ((((ob >> 1) * 4 + p) ^ sw) << 3);
    DualA<KS> a;
    static_for<KS>([&]<int s>() {
      constexpr int r0 = (KB0 + 2 * s) * 2048, r1 = (KB0 + 2 * s + 1) * 2048;
      a.h[s] = join_halves_h(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(u + r0)),
                             __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(u + r1)));
#ifdef M3G_DIAG_NO_AL   // timing diagnostic only (wrong results): no LDS reads of the low-part image
      a.l[s] = a.h[s];
#else
      a.l[s] = join_halves_h(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(u + r0 + lo)),
                             __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(u + r1 + lo)));
#endif
    });
    return a;
  };
  // the first block's operands travel while the vector ALU finds the scale and splits d
  DualA<KS> cur = fetch.template operator()<0>();
  M3G_CHAIN_FENCE();
  const EdgeScale sc = edge_scale<2 * KS, DOFF>(d);
  if (used) *used = sc;
  const HalfB<KS> b = split_h<KS, DOFF>(d, sc.s);
  const float inv = sc.inv * w_inv;
  M3G_CHAIN_PRIO(1);
  static_for<OB>([&]<int ob>() {
    DualA<KS> nxt = cur;
    if constexpr (ob + 1 < OB) nxt = fetch.template operator()<ob + 1>();
    M3G_CHAIN_FENCE();
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    static_for<KS>([&]<int s>() {
      t = mfma_f16(cur.h[s], b.hi[s], t);
#ifndef M3G_DIAG_H1
      t = mfma_f16(cur.h[s], b.lo[s], t);
      t = mfma_f16(cur.l[s], b.hi[s], t);
#elif defined(M3G_DIAG_H1_KEEP)   // ... with the low-part reads and splits kept alive: the MFMAs alone
      asm volatile("" ::"v"(cur.l[s]), "v"(b.lo[s]));
#endif
    });
    acc[AOFF + ob] = t * inv + acc[AOFF + ob];   // (vector form: the compiler emits two v_pk_fma_f32)
    M3G_CHAIN_FENCE();
    cur = nxt;
  });
  M3G_CHAIN_PRIO(0);
}

// host: img receives ROWS*64 floats (hi part, then lo part); get(row, col) with col < 64
template <class F>
inline void pack_dual_image_h(float* img, int rows, float scale, F get) {
  auto to_h = [](float w) { const _Float16 h = (_Float16)w; uint16_t u; memcpy(&u, &h, 2); return u; };
  auto to_f = [](uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; };
  uint16_t* hi = reinterpret_cast<uint16_t*>(img);
  uint16_t* lo = hi + (size_t)rows * 64;
  for (int row = 0; row < rows; ++row)
    for (int col = 0; col < 64; ++col) {
      const float w = get(row, col) * scale;
      const uint16_t h = to_h(w);
      const size_t idx = (size_t)(dual_unit_byte(rows, row, col >> 2) >> 1) + (col & 3);
      hi[idx] = h;
      lo[idx] = to_h(w - to_f(h));
    }
}
template <class F>
inline void pack_dual_image(float* img, int rows, F get) {
  auto rne = [](float w) {
    uint32_t u;
    memcpy(&u, &w, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  };
  auto tof = [](uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
  };
  uint16_t* hi = reinterpret_cast<uint16_t*>(img);
  uint16_t* lo = hi + (size_t)rows * 64;
  for (int row = 0; row < rows; ++row)
    for (int col = 0; col < 64; ++col) {
      const float w = get(row, col);
      const uint16_t h = rne(w);
      const size_t idx = (size_t)(dual_unit_byte(rows, row, col >> 2) >> 1) + (col & 3);
      hi[idx] = h;
      lo[idx] = rne(w - tof(h));
    }
}

}  // namespace m3g
