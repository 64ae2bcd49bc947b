// Shared device helpers of the MFMA edge kernels: vector types, compile-time loops, activation derivatives and the
// bf16x3 operand split.
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

namespace m3g {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kDp1Groups = 64;   // a dp1 row is 256 columns = 64 groups of 4 (one accumulator register quad each)

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f.template operator()<I>(), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

#ifdef M3G_DIAG_CHEAP_ACT   // timing diagnostic only (wrong results): activations without transcendentals
__device__ __forceinline__ float fsigmoid(float p) { return p * 0.25f + 0.5f; }
#else
__device__ __forceinline__ float fsigmoid(float p) { return __builtin_amdgcn_rcpf(1.f + __expf(-p)); }
#endif
__device__ __forceinline__ float fsilu(float p) { return p * fsigmoid(p); }
// SiLU(p) * sigmoid(g) = p / ((1 + e^-p)(1 + e^-g)): one reciprocal for the gated product instead of two sigmoids
// (each factor is >= 1, an overflowing exponential gives inf -> rcp 0, the correct limit)
__device__ __forceinline__ float fgated(float p, float g) {
#ifdef M3G_DIAG_CHEAP_ACT
  return p * (g * 0.25f + 0.5f);
#else
  return p * __builtin_amdgcn_rcpf((1.f + __expf(-p)) * (1.f + __expf(-g)));
#endif
}
__device__ __forceinline__ float fdsilu(float p) {
  float s = fsigmoid(p);
  return s * (1.f + p * (1.f - s));
}

// SiLU and SiLU' of a value PAIR on packed fp32 instructions (v_pk_mul / v_pk_add / v_pk_fma do two values per issue slot; the
// compiler packs this chain only in part): act = p sg, der = sg + act (1 - sg) = sg (1 + p (1 - sg)); five packed instructions and
// four transcendentals per pair instead of nine and four.  1 - sg is formed as 1 + (-sg), not as e^-p sg: the latter is inf * 0
// for p < -88.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void silu_pair(f32x2 p, f32x2& act, f32x2& der) {
#ifdef M3G_DIAG_CHEAP_ACT
  const f32x2 sg = p * 0.25f + 0.5f;
#else
  const f32x2 t = p * -1.4426950408889634f;
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.f;
  const f32x2 sg = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
#endif
  act = p * sg;
  der = act * (1.f - sg) + sg;
}

__device__ __forceinline__ f32x2 sigmoid_pair(f32x2 g) {
#ifdef M3G_DIAG_CHEAP_ACT
  return g * 0.25f + 0.5f;
#else
  const f32x2 t = g * -1.4426950408889634f;
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.f;
  return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
#endif
}
// SiLU of a value pair / SiLU(p) * sigmoid(g) of two value pairs, the same way (fsilu, fgated)
__device__ __forceinline__ f32x2 silu_pair(f32x2 p) {
#ifdef M3G_DIAG_CHEAP_ACT
  return p * (p * 0.25f + 0.5f);
#else
  const f32x2 t = p * -1.4426950408889634f;
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.f;
  return p * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
#endif
}
__device__ __forceinline__ f32x2 gated_pair(f32x2 p, f32x2 g) {
#ifdef M3G_DIAG_CHEAP_ACT
  return p * (g * 0.25f + 0.5f);
#else
  const f32x2 tp = p * -1.4426950408889634f, tg = g * -1.4426950408889634f;
  const f32x2 d = (f32x2{__builtin_amdgcn_exp2f(tp[0]), __builtin_amdgcn_exp2f(tp[1])} + 1.f) *
                  (f32x2{__builtin_amdgcn_exp2f(tg[0]), __builtin_amdgcn_exp2f(tg[1])} + 1.f);
  return p * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
#endif
}

// ---- the dense chains run on v_mfma_f32_16x16x32_bf16 with split operands ("bf16x3") ----------------------------
// a = a_hi + a_lo (both bf16; the residual a - a_hi is formed exactly in fp32), a.b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi,
// accumulated in fp32: 3 MFMAs at 16x the fp32-MFMA rate.  bf16 keeps the fp32 exponent range, which the tiny gradient
// operands of the reverse pass need (f16 would flush them).  Parity effect (tests/checkers/split_precision_study.py, same
// arithmetic emulated in the oracle): force error 1.1e-5 of max|F| vs 7e-6 for plain fp32 -- budget 1e-4.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// B operand of one k-step (32 features = accumulator blocks a, b): element j < 4 from a, j >= 4 from b
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
  static_for<4>([&]<int j>() {
    hi[j] = (__bf16)a[j];
    hi[4 + j] = (__bf16)b[j];
  });
  static_for<4>([&]<int j>() {
    lo[j] = (__bf16)(a[j] - (float)hi[j]);
    lo[4 + j] = (__bf16)(b[j] - (float)hi[4 + j]);
  });
}

// ---- f16x3 mode: the dense chains on v_mfma_f32_16x16x32_f16 with operands SCALED by powers of two and split in two fp16 parts --
// x s = h + l, h = fp16(x s) (round to nearest), l = fp16(x s - h): h carries 11 significant bits, l the next 11-12, so h + l is
// x s to within 2^-23 relative -- an fp32 value to about its own rounding -- PROVIDED both parts are normal fp16 numbers.  That
// is what the scales are for: every B operand (a tile's activations or gradients) is multiplied per EDGE by the power of two that
// puts the largest of the edge's K inputs into [2^12, 2^13) (edge_scale: gradients of 1e-9 and saturated activations of 30 alike),
// every weight matrix by one power of two for the whole model (plan->w_scale_inv).  Entries within 2^-14 of their column's
// largest keep the full 22-23 bits; smaller ones carry an absolute error below 2^-37 of the largest, irrelevant in a dot product.
// Products of parts are exact in fp32 (11 x 11 bits); a b ~ h_a h_b + h_a l_b + l_a h_b drops l_a l_b <= 2^-22 a b, accumulated in
// fp32: measured (tests/checkers/split_precision_study.py) rms error 1.8 x that of an fp32 fmaf chain, 17 x below bf16x3's.
// Powers of two commute with fp32 rounding, so results do not depend on the scales as long as nothing leaves the fp16 range.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// scaled split of one value pair: hi = fp16(s x), lo = fp16(s x - hi), both rounded to nearest (the residual has at most 12
// significant bits, so hi + lo is s x to within 2^-24 relative: half an fp32 ulp).  Five vector instructions per PAIR, the scaling
// included: v_fma_mixlo_f16 / v_fma_mixhi_f16 form fp16(s a) and fp16(s b) in the two halves of one register straight from the
// fp32 inputs, v_fma_mix_f32 forms s a - hi with the fp16 half as its addend (op_sel picks the half), v_cvt_pk_f16_f32 packs the
// residuals.  The compiler finds the mix forms too but builds the packed hi with two extra multiplies and a v_cvt_pk on top (8
// per pair).  Measured issue costs (tools/valu_issue_probe.hip, profiles/r03_valu_issue_probe.txt): the half-register writers
// v_fma_mixlo/hi_f16 cost 1.67 plain instructions each (like v_exp_f32), so forming the low parts with two more of them (4
// instructions per pair, one rounding) is SLOWER than this form (24.7 against 22.0 cycles per pair) -- tried and reverted.
// Hazards inside the block (the recogniser does not look into inline assembly): a half-register write (v_fma_mixlo/hi) needs one
// wait state before a vector instruction reads that register (dst_sel / op_sel forwarding, gfx940+): the reader of the low half
// comes two instructions after its writer, the reader of the high half two after its.
#if !defined(M3G_SPLIT_H_MIX) && !defined(M3G_SPLIT_H_PLAIN)
// default form: the scaling as one packed multiply, the high parts by v_cvt_pk_f16_f32 -- no half-register writers at all, five
// plain-rate instructions per pair, bit-identical parts (the power-of-two product is exact); same-box A/B against the mix form
// below: forward 0.452 -> 0.447, fused reverse 0.955 -> 0.943 ms per step
typedef float f32x2_split __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_h(float a, float b, float s, f16x2& hi, f16x2& lo) {
  const f32x2_split t = f32x2_split{a, b} * s;
  unsigned h, l;
  float r;
  // read-only inputs and separate outputs: with the products as tied in/out operands the register allocator could not place
  // `h` in its slot of the four-register B operand without copying a product out of the way first (one v_mov_b32 for two of
  // three pairs in the listings of round 3)
  asm("v_cvt_pk_f16_f32 %0, %3, %4\n\t"
      "s_nop 0\n\t"
      "v_fma_mix_f32 %1, %3, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %2, %4, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_cvt_pk_f16_f32 %1, %1, %2"
      : "=&v"(h), "=&v"(l), "=&v"(r)
      : "v"(t[0]), "v"(t[1]));
  hi = __builtin_bit_cast(f16x2, h);
  lo = __builtin_bit_cast(f16x2, l);
}
#elif defined(M3G_SPLIT_H_MIX)
__device__ __forceinline__ void split_pair_h(float a, float b, float s, f16x2& hi, f16x2& lo) {
  unsigned h;
  float ra, rb;
  asm("v_fma_mixlo_f16 %0, %3, %4, 0\n\t"
      "v_fma_mixhi_f16 %0, %3, %5, 0\n\t"
      "v_fma_mix_f32 %1, %3, %4, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %2, %3, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(h), "=&v"(ra), "=v"(rb)
      : "v"(s), "v"(a), "v"(b));
  hi = __builtin_bit_cast(f16x2, h);
  lo = f16x2{(_Float16)ra, (_Float16)rb};   // v_cvt_pk_f16_f32: round to nearest, like the high part
}
#else
__device__ __forceinline__ void split_pair_h(float a, float b, float s, f16x2& hi, f16x2& lo) {
  const float as = a * s, bs = b * s;
  hi = f16x2{(_Float16)as, (_Float16)bs};
  const float ra = __builtin_fmaf((float)hi[0], -1.0f, as), rb = __builtin_fmaf((float)hi[1], -1.0f, bs);
  lo = f16x2{(_Float16)ra, (_Float16)rb};
}
#endif
// B operand of one k-step (accumulator blocks a, b scaled by s): element j < 4 from a, j >= 4 from b (as split8)
__device__ __forceinline__ void split8h(const f32x4& a, const f32x4& b, float s, f16x8& hi, f16x8& lo) {
  f16x2 h[4], l[4];
  split_pair_h(a[0], a[1], s, h[0], l[0]);
  split_pair_h(a[2], a[3], s, h[1], l[1]);
  split_pair_h(b[0], b[1], s, h[2], l[2]);
  split_pair_h(b[2], b[3], s, h[3], l[3]);
  hi = f16x8{h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], h[3][0], h[3][1]};
  lo = f16x8{l[0][0], l[0][1], l[1][0], l[1][1], l[2][0], l[2][1], l[3][0], l[3][1]};
}

// max over the four lane quarters (lanes l, l^16, l^32, l^48) of a NON-NEGATIVE float given as its bit pattern, result in every
// lane (v_permlane16/32_swap are VALU moves on gfx950).  Non-negative floats order like their bit patterns, and an integer
// maximum needs no operand canonicalisation: the float form compiled to three v_max_f32 per exchange (two of them x = max(x, x)).
__device__ __forceinline__ unsigned max_lane_quarters_bits(unsigned u) {
  auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned v = max((unsigned)r16[0], (unsigned)r16[1]);
  auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return max((unsigned)r32[0], (unsigned)r32[1]);
}
// power of two that puts the largest |x| of this lane's EDGE (over blocks XOFF .. XOFF + NB, all four lane quarters) into
// [2^12, 2^13), and its inverse, both formed from the exponent field of that maximum (m in [2^(e-1), 2^e): s = 2^(13-e), inv =
// 2^(e-13)).  Exponents below -100 are treated as -100 (an all-zero column gets s = 2^113: 0 stays 0, nothing overflows).
struct EdgeScale { float s, inv; };
template <int NB, int XOFF = 0, int NX>
__device__ __forceinline__ EdgeScale edge_scale(const f32x4 (&x)[NX]) {
  static_assert(XOFF + NB <= NX, "edge_scale operand out of range");
#ifdef M3G_DIAG_NO_SCALE   // timing diagnostic only (wrong results): what finding the per-edge scales costs
  return EdgeScale{1024.f, 1.f / 1024.f};
#endif
  float m = 0.f;
  static_for<NB>([&]<int b>() { static_for<4>([&]<int r>() { m = fmaxf(m, fabsf(x[XOFF + b][r])); }); });
  unsigned u = max_lane_quarters_bits(__builtin_bit_cast(unsigned, m));
  u = max(u, 0x0D000000u) & 0x7F800000u;             // biased exponent field E of max(m, 2^-101); m in [2^(E-127), 2^(E-126))
  return EdgeScale{__builtin_bit_cast(float, 0x85000000u - u),    // 2^(139 - E) = 2^(13 - e), e = E - 126
                   __builtin_bit_cast(float, u - 0x06000000u)};   // 2^(E - 139)
}
// the scaled, split B operands of a whole chain (KS 32-wide k-steps = blocks XOFF .. XOFF + 2 KS of x)
template <int KS>
struct HalfB { f16x8 hi[KS], lo[KS]; };
template <int KS, int XOFF = 0, int NX>
__device__ __forceinline__ HalfB<KS> split_h(const f32x4 (&x)[NX], float s) {
  static_assert(XOFF + 2 * KS <= NX, "split_h operand out of range");
  HalfB<KS> b;
  static_for<KS>([&]<int k>() { split8h(x[XOFF + 2 * k], x[XOFF + 2 * k + 1], s, b.hi[k], b.lo[k]); });
  return b;
}

// ---- 24-bit rows for the dp1 hand-over (fused reverse -> node reverse) -------------------------------------------
// The x_j half of the node reverse gathers one dp1 row per incoming edge, the largest stream of the reverse pass.  The
// rows are stored with 16 significand bits (sign, exponent and the top 15 mantissa bits, rounded: relative error
// <= 2^-17, the size of one split product's error in the bf16x3 chains; the fp32 mode keeps fp32 rows) as 3 bytes per value: 4 values -> 3 dwords, 768 B per row instead of
// 1 KB.  The per-centre sums of the same rows (x_i half) are formed in registers from the unrounded values.
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));   // in memory a group starts at any dword
__device__ __forceinline__ u32x3 pack24(const f32x4& v) {
  // (elements copied to scalars first: __builtin_bit_cast of a vector-element lvalue reads element 0)
  const float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
  const unsigned a = __builtin_bit_cast(unsigned, v0) + 0x80u, b = __builtin_bit_cast(unsigned, v1) + 0x80u,
                 c = __builtin_bit_cast(unsigned, v2) + 0x80u, d = __builtin_bit_cast(unsigned, v3) + 0x80u;
  // v_perm_b32: selector bytes 0-3 take bytes of the second operand, 4-7 of the first
  return u32x3{__builtin_amdgcn_perm(b, a, 0x05030201u), __builtin_amdgcn_perm(c, b, 0x06050302u),
               __builtin_amdgcn_perm(d, c, 0x07060503u)};
}
__device__ __forceinline__ f32x4 unpack24(const u32x3& w) {
  return f32x4{__builtin_bit_cast(float, w[0] << 8), __builtin_bit_cast(float, __builtin_amdgcn_perm(w[1], w[0], 0x0504030cu)),
               __builtin_bit_cast(float, __builtin_amdgcn_perm(w[2], w[1], 0x0403020cu)),
               __builtin_bit_cast(float, w[2] & 0xffffff00u)};
}
constexpr int kDp1PackedDwords = 3 * kDp1Groups;   // dwords per packed row

// f16x3 mode: the same 768-byte rows as 24-bit FIXED-POINT values relative to the power-of-two scale the W1c^T chain gives the
// row's 64-column quarter anyway (edge_scale: the quarter's largest |value| times s lies in [2^12, 2^13)):  k = round(x s 2^9), |k| <
// 2^22, absolute error 2^-10 in scaled units = at most 2^-22 of the quarter's largest value -- what one split product of this mode
// carries, and every column of a row meets the same 64 outputs of W1b^T in the node reverse, so an error relative to the largest
// column is the relevant one.  One fma per value forms k in the mantissa of 1.5 * 2^23 + k (round to nearest even), the low three
// bytes are the value; the four inverse scales of a row (times 2^-9) go to a separate [E][4] array.
constexpr float kFix24Magic = 12582912.f;   // 1.5 * 2^23: ulp 1 on [2^23, 2^24)
__device__ __forceinline__ u32x3 pack24_fixed(const f32x4& v, float s9) {
  const float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
  const unsigned a = __builtin_bit_cast(unsigned, __builtin_fmaf(v0, s9, kFix24Magic)), b = __builtin_bit_cast(unsigned, __builtin_fmaf(v1, s9, kFix24Magic)),
                 c = __builtin_bit_cast(unsigned, __builtin_fmaf(v2, s9, kFix24Magic)), d = __builtin_bit_cast(unsigned, __builtin_fmaf(v3, s9, kFix24Magic));
  // bytes (a0 a1 a2 b0) (b1 b2 c0 c1) (c2 d0 d1 d2)
  return u32x3{__builtin_amdgcn_perm(b, a, 0x04020100u), __builtin_amdgcn_perm(c, b, 0x05040201u), __builtin_amdgcn_perm(d, c, 0x06050402u)};
}
__device__ __forceinline__ f32x4 unpack24_fixed(const u32x3& w, float inv9) {
  const unsigned top = 0x4B4B4B4Bu;   // exponent byte of [2^23, 2^24)
  const unsigned a = __builtin_amdgcn_perm(top, w[0], 0x04020100u), b = __builtin_amdgcn_perm(w[1], w[0], 0x0c050403u) | 0x4B000000u,
                 c = __builtin_amdgcn_perm(w[2], w[1], 0x0c040302u) | 0x4B000000u, d = __builtin_amdgcn_perm(top, w[2], 0x04030201u);
  return f32x4{(__builtin_bit_cast(float, a) - kFix24Magic) * inv9, (__builtin_bit_cast(float, b) - kFix24Magic) * inv9,
               (__builtin_bit_cast(float, c) - kFix24Magic) * inv9, (__builtin_bit_cast(float, d) - kFix24Magic) * inv9};
}

template <int N>
__device__ __forceinline__ void zero(f32x4 (&v)[N]) {
  static_for<N>([&]<int i>() { v[i] = f32x4{0.f, 0.f, 0.f, 0.f}; });
}

}  // namespace m3g
