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

template <int N>
__device__ __forceinline__ void zero(f32x4 (&v)[N]) {
  static_for<N>([&]<int i>() { v[i] = f32x4{0.f, 0.f, 0.f, 0.f}; });
}

}  // namespace m3g
