// Issue cost of the vector instructions the edge kernels are made of, on one wave alone on its SIMD (gfx950).
//   independent: 8 destination registers in rotation, sources fixed  -> cycles per instruction = issue cost
//   dependent:   every instruction reads the previous one's result   -> cycles per instruction = latency seen by a chain
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 tools/valu_issue_probe.hip -o tools/bin/valu_issue_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int kIters = 512;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define OPS8 [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [d4] "+v"(d[4]), [d5] "+v"(d[5]), [d6] "+v"(d[6]), [d7] "+v"(d[7])
#define PK8 [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]), [p4] "+v"(p[4]), [p5] "+v"(p[5]), [p6] "+v"(p[6]), [p7] "+v"(p[7])

// independent forms (destination d<i>, sources a, b)
#define I_FMA(i) "v_fma_f32 %[d" #i "], %[a], %[b], %[a]\n\t"
#define I_MUL(i) "v_mul_f32 %[d" #i "], %[a], %[b]\n\t"
#define I_MIX32(i) "v_fma_mix_f32 %[d" #i "], %[a], %[b], -%[h] op_sel_hi:[0,0,1]\n\t"
#define I_MIXLO(i) "v_fma_mixlo_f16 %[d" #i "], %[a], %[b], 0\n\t"
#define I_MIXHI(i) "v_fma_mixhi_f16 %[d" #i "], %[a], %[b], 0\n\t"
#define I_MIXLOH(i) "v_fma_mixlo_f16 %[d" #i "], %[a], %[b], -%[h] op_sel_hi:[0,0,1]\n\t"
#define I_CVTF16(i) "v_cvt_pk_f16_f32 %[d" #i "], %[a], %[b]\n\t"
#define I_CVTBF16(i) "v_cvt_pk_bf16_f32 %[d" #i "], %[a], %[b]\n\t"
#define I_CVT1F16(i) "v_cvt_f16_f32 %[d" #i "], %[a]\n\t"
#define I_EXP(i) "v_exp_f32 %[d" #i "], %[a]\n\t"
#define I_RCP(i) "v_rcp_f32 %[d" #i "], %[a]\n\t"
#define I_MAX3(i) "v_max3_f32 %[d" #i "], |%[a]|, |%[b]|, %[a]\n\t"
#define I_LDEXP(i) "v_ldexp_f32 %[d" #i "], %[a], %[ci]\n\t"
#define I_FREXP(i) "v_frexp_exp_i32_f32 %[d" #i "], %[a]\n\t"
#define I_CNDMASK(i) "v_cndmask_b32 %[d" #i "], %[a], %[b], vcc\n\t"
#define I_DPPFMAC(i) "v_fmac_f32_dpp %[d" #i "], %[a], %[b] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define I_DPPMOV(i) "v_mov_b32_dpp %[d" #i "], %[a] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_PERM16(i) "v_permlane16_swap_b32 %[d" #i "], %[d" #i "]\n\t"
#define I_PERM32(i) "v_permlane32_swap_b32 %[d" #i "], %[d" #i "]\n\t"
#define I_PKFMA(i) "v_pk_fma_f32 %[p" #i "], %[pa], %[pa], %[pa]\n\t"
#define I_PKMUL(i) "v_pk_mul_f32 %[p" #i "], %[pa], %[pa]\n\t"
#define I_PKADD(i) "v_pk_add_f32 %[p" #i "], %[pa], %[pa]\n\t"
#define I_ADD64(i) "v_lshl_add_u64 %[p" #i "], %[pa], 2, %[pa]\n\t"
// dependent forms (each reads d0 and writes d0)
#define D_FMA(i) "v_fma_f32 %[d0], %[d0], %[b], %[a]\n\t"
#define D_EXP(i) "v_exp_f32 %[d0], %[d0]\n\t"
#define D_RCP(i) "v_rcp_f32 %[d0], %[d0]\n\t"
#define D_MAX3(i) "v_max3_f32 %[d0], |%[d0]|, |%[b]|, %[a]\n\t"
#define D_MIXLO(i) "v_fma_mixlo_f16 %[d0], %[d0], %[b], 0\n\t"
#define D_DPPFMAC(i) "v_fmac_f32_dpp %[d0], %[d0], %[b] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define D_EXPADD(i) "v_exp_f32 %[d0], %[d0]\n\tv_add_f32 %[d0], 1.0, %[d0]\n\t"   /* two instructions per repetition */
// the scaled split of a value pair: old form (2 mixlo/hi + 2 mix_f32 + cvt_pk), new form (4 mixlo/hi), plain form (2 mul + cvt + 2 mix + cvt)
#define S_OLD(i)                                                                                                                  \
  "v_fma_mixlo_f16 %[d" #i "], %[a], %[b], 0\n\tv_fma_mixhi_f16 %[d" #i "], %[a], %[c], 0\n\t"                                    \
  "v_fma_mix_f32 %[t0], %[a], %[b], -%[d" #i "] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %[t1], %[a], %[c], -%[d" #i "] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t" \
  "v_cvt_pk_f16_f32 %[t0], %[t0], %[t1]\n\t"
#define S_NEW(i)                                                                                                                  \
  "v_fma_mixlo_f16 %[d" #i "], %[a], %[b], 0\n\tv_fma_mixhi_f16 %[d" #i "], %[a], %[c], 0\n\t"                                    \
  "v_fma_mixlo_f16 %[t0], %[a], %[b], -%[d" #i "] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %[t0], %[a], %[c], -%[d" #i "] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
#define S_PLAIN(i)                                                                                                                \
  "v_mul_f32 %[t0], %[a], %[b]\n\tv_mul_f32 %[t1], %[a], %[c]\n\tv_cvt_pk_f16_f32 %[d" #i "], %[t0], %[t1]\n\t"                    \
  "v_fma_mix_f32 %[t0], %[t0], 1.0, -%[d" #i "] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %[t1], %[t1], 1.0, -%[d" #i "] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t" \
  "s_nop 0\n\tv_cvt_pk_f16_f32 %[t0], %[t0], %[t1]\n\t"

#define KERNEL(NAME, BODY)                                                                                                        \
  __global__ void __launch_bounds__(64) NAME(float* out, unsigned long long* cycles, float seed) {                                \
    float d[8], a = seed, b = seed * 1.0001f, c = seed * 0.999f, t0 = 0.f, t1 = 0.f;                                             \
    typedef float f32x2 __attribute__((ext_vector_type(2)));                                                                      \
    f32x2 p[8], pa = {seed, seed};                                                                                                \
    unsigned h = 0x3c003c00u;                                                                                                     \
    int ci = 1;                                                                                                                   \
    for (int i = 0; i < 8; ++i) { d[i] = seed + i; p[i] = f32x2{seed, seed}; }                                                    \
    const unsigned long long t_begin = __builtin_readcyclecounter();                                                              \
    for (int it = 0; it < kIters; ++it) {                                                                                         \
      asm volatile(REP8(BODY) REP8(BODY) : OPS8, PK8, [t0] "+v"(t0), [t1] "+v"(t1) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [h] "v"(h), [ci] "v"(ci), [pa] "v"(pa) : "vcc"); \
    }                                                                                                                             \
    const unsigned long long t_end = __builtin_readcyclecounter();                                                                \
    float s = t0 + t1;                                                                                                            \
    for (int i = 0; i < 8; ++i) s += d[i] + p[i][0] + p[i][1];                                                                    \
    out[threadIdx.x] = s;                                                                                                         \
    if (threadIdx.x == 0) cycles[0] = t_end - t_begin;                                                                            \
  }

#define ALL(X)                                                                                                                    \
  X(I_FMA) X(I_MUL) X(I_MIX32) X(I_MIXLO) X(I_MIXHI) X(I_MIXLOH) X(I_CVTF16) X(I_CVTBF16) X(I_CVT1F16) X(I_EXP) X(I_RCP) X(I_MAX3) \
  X(I_LDEXP) X(I_FREXP) X(I_CNDMASK) X(I_DPPFMAC) X(I_DPPMOV) X(I_PERM16) X(I_PERM32) X(I_PKFMA) X(I_PKMUL) X(I_PKADD) X(I_ADD64)  \
  X(D_FMA) X(D_EXP) X(D_RCP) X(D_MAX3) X(D_MIXLO) X(D_DPPFMAC) X(D_EXPADD) X(S_OLD) X(S_NEW) X(S_PLAIN)

#define DEF(B) KERNEL(k_##B, B)
ALL(DEF)

// MFMA chains: NACC independent accumulators in rotation (1 = fully dependent)
template <int NACC>
__global__ void __launch_bounds__(64) k_mfma(float* out, unsigned long long* cycles, float seed) {
  f32x4 acc[4] = {{seed, 0, 0, 0}, {seed, 0, 0, 0}, {seed, 0, 0, 0}, {seed, 0, 0, 0}};
  f16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(seed + i); y[i] = (_Float16)(seed - i); }
  const unsigned long long t_begin = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[r % NACC], 0, 0, 0);
  }
  const unsigned long long t_end = __builtin_readcyclecounter();
  out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (threadIdx.x == 0) cycles[0] = t_end - t_begin;
}
// MFMA result consumed by a vector instruction, then fed back: MFMA -> v_fma -> MFMA ...
__global__ void __launch_bounds__(64) k_mfma_valu(float* out, unsigned long long* cycles, float seed) {
  f32x4 acc = {seed, 0, 0, 0};
  f16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(seed + i); y[i] = (_Float16)(seed - i); }
  const unsigned long long t_begin = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
      acc[0] = __builtin_fmaf(acc[0], seed, seed);
    }
  }
  const unsigned long long t_end = __builtin_readcyclecounter();
  out[threadIdx.x] = acc[0] + acc[1];
  if (threadIdx.x == 0) cycles[0] = t_end - t_begin;
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 64 * sizeof(float));
  hipMalloc(&cyc, sizeof(unsigned long long));
  // the cycle counter (s_memrealtime) ticks at 100 MHz: convert with the measured time of a known stream (v_fma_f32: 4 cycles)
  auto run = [&](auto kern) {
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f);
      hipDeviceSynchronize();
      unsigned long long c;
      hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
      if (c < best) best = c;
    }
    return (double)best;
  };
  const double fma_ticks = run(k_I_FMA) / (kIters * 16.0);
  printf("# tools/valu_issue_probe.hip on MI355X: one wave alone on its SIMD; cost relative to an independent v_fma_f32 (= 4 cycles)\n");
  printf("# I_* independent (8 destinations in rotation), D_* dependent chain, S_* one scaled fp16 split of a value pair (whole sequence)\n");
#define REPORT(B) printf("%-12s %6.2f cycles per repetition\n", #B, 4.0 * run(k_##B) / (kIters * 16.0) / fma_ticks);
  ALL(REPORT)
  printf("%-12s %6.2f cycles per MFMA (4 accumulators in rotation)\n", "mfma_f16 x4", 4.0 * run(k_mfma<4>) / (kIters * 16.0) / fma_ticks);
  printf("%-12s %6.2f cycles per MFMA (2 accumulators in rotation)\n", "mfma_f16 x2", 4.0 * run(k_mfma<2>) / (kIters * 16.0) / fma_ticks);
  printf("%-12s %6.2f cycles per MFMA (dependent chain)\n", "mfma_f16 x1", 4.0 * run(k_mfma<1>) / (kIters * 16.0) / fma_ticks);
  printf("%-12s %6.2f cycles per MFMA + dependent v_fma_f32 pair\n", "mfma->valu", 4.0 * run(k_mfma_valu) / (kIters * 16.0) / fma_ticks);
  return 0;
}
