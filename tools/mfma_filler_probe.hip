// Which instructions of a SIMD co-execute with the fp32 MFMA (v_mfma_f32_16x16x4_f32, 32 cycles per SIMD)?
// Part 1 -- same wave: one or two waves per SIMD run a pinned inline-assembly stream { MFMA ; K independent fillers } with 4
//   accumulators in rotation; filler = v_fma_f32 / v_exp_f32 / v_mov_b32 / v_add_u32 / v_mov_b32 dpp / v_pk_fma_f32 /
//   ds_read_b32.  The same streams on the bf16 MFMAs (v_mfma_f32_32x32x16_bf16, 32 cycles; v_mfma_f32_16x16x32_bf16, 16) for
//   comparison.  Prints shader cycles per MFMA for K = 0 .. 6: a filler that hides costs 0, one that does not adds its issue time.
// Part 2 -- different waves: wave class 0 of every SIMD runs MFMAs only, class 1 (the SIMD's second wave) vector instructions
//   only, a fixed amount of work each; prints when each class finishes alone and together (co-execution: together = max of
//   the two; a shared datapath: together = sum).
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 tools/mfma_filler_probe.hip -o tools/bin/mfma_filler_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kIters = 2000;

#define OP_FMA(f) "v_fma_f32 %[" #f "], %[" #f "], %[c], %[c]\n\t"
#define OP_EXP(f) "v_exp_f32 %[" #f "], %[" #f "]\n\t"
#define OP_MOV(f) "v_mov_b32 %[" #f "], %[c]\n\t"
#define OP_IADD(f) "v_add_u32 %[" #f "], %[" #f "], %[ci]\n\t"
#define OP_DPP(f) "v_mov_b32_dpp %[" #f "], %[c] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define OP_PK(f) "v_pk_fma_f32 %[p" #f "], %[p" #f "], %[pc], %[pc]\n\t"
#define OP_LDS(f) "ds_read_b32 %[" #f "], %[addr]\n\t"
#define FILL0(OP)
#define FILL1(OP) OP(f0)
#define FILL2(OP) FILL1(OP) OP(f1)
#define FILL3(OP) FILL2(OP) OP(f2)
#define FILL4(OP) FILL3(OP) OP(f3)
#define FILL5(OP) FILL4(OP) OP(f4)
#define FILL6(OP) FILL5(OP) OP(f5)

#define OPERANDS                                                                                                                     \
  [f0] "+v"(f[0]), [f1] "+v"(f[1]), [f2] "+v"(f[2]), [f3] "+v"(f[3]), [f4] "+v"(f[4]), [f5] "+v"(f[5]), [pf0] "+v"(pf[0]),            \
      [pf1] "+v"(pf[1]), [pf2] "+v"(pf[2]), [pf3] "+v"(pf[3]), [pf4] "+v"(pf[4]), [pf5] "+v"(pf[5])
#define INPUTS [c] "v"(c), [ci] "v"(ci), [pc] "v"(pc), [addr] "v"(addr)

// MK 0: f32 16x16x4 (acc VGPR), 1: bf16 32x32x16, 2: bf16 16x16x32, 3: f32 32x32x2 (64 cycles per SIMD)
#define STREAM(MK, FILL)                                                                                                             \
  if constexpr (MK == 0) {                                                                                                           \
    asm volatile("v_mfma_f32_16x16x4_f32 %[a0], %[x], %[y], %[a0]\n\t" FILL "v_mfma_f32_16x16x4_f32 %[a1], %[x], %[y], %[a1]\n\t" FILL \
                 "v_mfma_f32_16x16x4_f32 %[a2], %[x], %[y], %[a2]\n\t" FILL "v_mfma_f32_16x16x4_f32 %[a3], %[x], %[y], %[a3]\n\t" FILL \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), OPERANDS                                 \
                 : [x] "v"(x), [y] "v"(y), INPUTS);                                                                                     \
  } else if constexpr (MK == 1) {                                                                                                    \
    asm volatile("v_mfma_f32_32x32x16_bf16 %[a0], %[x], %[y], %[a0]\n\t" FILL "v_mfma_f32_32x32x16_bf16 %[a1], %[x], %[y], %[a1]\n\t" FILL \
                 "v_mfma_f32_32x32x16_bf16 %[a0], %[x], %[y], %[a0]\n\t" FILL "v_mfma_f32_32x32x16_bf16 %[a1], %[x], %[y], %[a1]\n\t" FILL \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
                 : [a0] "+v"(big[0]), [a1] "+v"(big[1]), OPERANDS                                                                       \
                 : [x] "v"(xb), [y] "v"(yb), INPUTS);                                                                                   \
  } else if constexpr (MK == 3) {                                                                                                    \
    asm volatile("v_mfma_f32_32x32x2_f32 %[a0], %[x], %[y], %[a0]\n\t" FILL "v_mfma_f32_32x32x2_f32 %[a1], %[x], %[y], %[a1]\n\t" FILL   \
                 "v_mfma_f32_32x32x2_f32 %[a0], %[x], %[y], %[a0]\n\t" FILL "v_mfma_f32_32x32x2_f32 %[a1], %[x], %[y], %[a1]\n\t" FILL   \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
                 : [a0] "+v"(big[0]), [a1] "+v"(big[1]), OPERANDS                                                                       \
                 : [x] "v"(x), [y] "v"(y), INPUTS);                                                                                     \
  } else {                                                                                                                           \
    asm volatile("v_mfma_f32_16x16x32_bf16 %[a0], %[x], %[y], %[a0]\n\t" FILL "v_mfma_f32_16x16x32_bf16 %[a1], %[x], %[y], %[a1]\n\t" FILL \
                 "v_mfma_f32_16x16x32_bf16 %[a2], %[x], %[y], %[a2]\n\t" FILL "v_mfma_f32_16x16x32_bf16 %[a3], %[x], %[y], %[a3]\n\t" FILL \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), OPERANDS                                 \
                 : [x] "v"(xb), [y] "v"(yb), INPUTS);                                                                                   \
  }

#define BY_K(MK, OP)                                   \
  if constexpr (K == 0) { STREAM(MK, FILL0(OP)) }      \
  else if constexpr (K == 1) { STREAM(MK, FILL1(OP)) } \
  else if constexpr (K == 2) { STREAM(MK, FILL2(OP)) } \
  else if constexpr (K == 3) { STREAM(MK, FILL3(OP)) } \
  else if constexpr (K == 4) { STREAM(MK, FILL4(OP)) } \
  else if constexpr (K == 5) { STREAM(MK, FILL5(OP)) } \
  else { STREAM(MK, FILL6(OP)) }

// FT: 0 v_fma_f32, 1 v_exp_f32, 2 v_mov_b32, 3 v_add_u32, 4 v_mov_b32 dpp, 5 v_pk_fma_f32, 6 ds_read_b32
template <int MK, int FT, int K>
__global__ void __launch_bounds__(512) probe(float* out, unsigned long long* cycles, float seed) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = seed;
  lds[threadIdx.x + 512] = seed;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4];
  f32x16 big[2];
  float f[6];
  f32x2 pf[6];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, 0.f, seed, 0.f};
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) big[i][j] = seed * j;
  for (int i = 0; i < 6; ++i) { f[i] = seed * 1e-3f * (i + lane); pf[i] = f32x2{f[i], -f[i]}; }
  const float x = 1e-3f * lane, y = 1e-3f, c = 0.5f;
  const f32x2 pc = {0.5f, 0.25f};
  const int ci = 3, addr = lane * 4;
  bf16x8 xb, yb;
  for (int j = 0; j < 8; ++j) { xb[j] = (__bf16)(1e-3f * (lane + j)); yb[j] = (__bf16)1e-3f; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
    if constexpr (FT == 0) { BY_K(MK, OP_FMA) }
    else if constexpr (FT == 1) { BY_K(MK, OP_EXP) }
    else if constexpr (FT == 2) { BY_K(MK, OP_MOV) }
    else if constexpr (FT == 3) { BY_K(MK, OP_IADD) }
    else if constexpr (FT == 4) { BY_K(MK, OP_DPP) }
    else if constexpr (FT == 5) { BY_K(MK, OP_PK) }
    else { BY_K(MK, OP_LDS) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float res = 0.f;
  for (int i = 0; i < 4; ++i) res += acc[i][0] + acc[i][3];
  for (int i = 0; i < 2; ++i) res += big[i][0] + big[i][15];
  for (int i = 0; i < 6; ++i) res += f[i] + pf[i][0] + pf[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
  if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MK, int FT, int K>
static void run_one(float* out, unsigned long long* cyc, int waves) {
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MK, FT, K>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
  (void)hipDeviceSynchronize();
  unsigned long long h[8], mx = 0;
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
  printf(" %6.1f", (double)mx / ((double)kIters * 4 * (waves / 4)));
}

template <int MK, int FT>
static void run(const char* mfma, const char* filler, float* out, unsigned long long* cyc) {
  for (int waves : {4, 8}) {
    printf("%-26s + K x %-14s %d wave(s)/SIMD, K = 0..6: ", mfma, filler, waves / 4);
    run_one<MK, FT, 0>(out, cyc, waves); run_one<MK, FT, 1>(out, cyc, waves); run_one<MK, FT, 2>(out, cyc, waves);
    run_one<MK, FT, 3>(out, cyc, waves); run_one<MK, FT, 4>(out, cyc, waves); run_one<MK, FT, 5>(out, cyc, waves);
    run_one<MK, FT, 6>(out, cyc, waves);
    printf("   cycles per MFMA per SIMD\n");
  }
}

// ---- part 2: MFMA-only waves beside vector-only waves on the same SIMD -------------------------------------------------------
// VT: 0 v_fma_f32, 1 v_exp_f32.  ROLE bit 0: class-0 waves (waves 0-3, one per SIMD) run NM MFMAs; bit 1: class-1 waves (4-7) run NV
// vector instructions (8 independent registers).
template <int MK, int VT, int ROLES>
__global__ void __launch_bounds__(512) roles(float* out, unsigned long long* cycles, float seed, int n_mfma, int n_vec) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4];
  float f[8];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, 0.f, seed, 0.f};
  for (int i = 0; i < 8; ++i) f[i] = seed * 1e-3f * (i + lane);
  const float x = 1e-3f * lane, y = 1e-3f, c = 0.5f;
  bf16x8 xb, yb;
  for (int j = 0; j < 8; ++j) { xb[j] = (__bf16)(1e-3f * (lane + j)); yb[j] = (__bf16)1e-3f; }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (ROLES & 1) {
      for (int it = 0; it < n_mfma / 4; ++it) {
        if constexpr (MK == 0)
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\tv_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n\t"
                       "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n\tv_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n\t"
                       : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(x), "v"(y));
        else
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t"
                       "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %5, %3\n\t"
                       : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(xb), "v"(yb));
      }
    }
  } else if (ROLES & 2) {
    for (int it = 0; it < n_vec / 8; ++it) {
      if constexpr (VT == 0)
        asm volatile("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
                     "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8\n\t"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c));
      else
        asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                     "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float res = 0.f;
  for (int i = 0; i < 4; ++i) res += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) res += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
  if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MK, int VT>
static void run_roles(const char* mfma, const char* vec, float* out, unsigned long long* cyc) {
  const int n_mfma = MK == 0 ? 8000 : 16000, n_vec = 48000;
  double t[3][2];
  auto launch = [&](auto kernel, int idx) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kernel, dim3(256), dim3(512), 0, 0, out, cyc, 1.0f, n_mfma, n_vec);
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    t[idx][0] = (double)h[0];
    t[idx][1] = (double)h[4];
  };
  launch(roles<MK, VT, 1>, 0);
  launch(roles<MK, VT, 2>, 1);
  launch(roles<MK, VT, 3>, 2);
  printf("%-26s wave (%d MFMAs) beside a %-10s wave (%d instructions) on each SIMD: alone %7.0f / %7.0f cycles, together %7.0f / %7.0f"
         "   (sum of the two alone %7.0f)\n", mfma, n_mfma, vec, n_vec, t[0][0], t[1][1], t[2][0], t[2][1], t[0][0] + t[1][1]);
}

int main() {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  (void)hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long));
  run<0, 0>("v_mfma_f32_16x16x4_f32", "v_fma_f32", out, cyc);
  run<0, 1>("v_mfma_f32_16x16x4_f32", "v_exp_f32", out, cyc);
  run<0, 2>("v_mfma_f32_16x16x4_f32", "v_mov_b32", out, cyc);
  run<0, 3>("v_mfma_f32_16x16x4_f32", "v_add_u32", out, cyc);
  run<0, 4>("v_mfma_f32_16x16x4_f32", "v_mov_b32 dpp", out, cyc);
  run<0, 5>("v_mfma_f32_16x16x4_f32", "v_pk_fma_f32", out, cyc);
  run<0, 6>("v_mfma_f32_16x16x4_f32", "ds_read_b32", out, cyc);
  run<3, 0>("v_mfma_f32_32x32x2_f32", "v_fma_f32", out, cyc);
  run<3, 1>("v_mfma_f32_32x32x2_f32", "v_exp_f32", out, cyc);
  run<1, 0>("v_mfma_f32_32x32x16_bf16", "v_fma_f32", out, cyc);
  run<1, 1>("v_mfma_f32_32x32x16_bf16", "v_exp_f32", out, cyc);
  run<2, 0>("v_mfma_f32_16x16x32_bf16", "v_fma_f32", out, cyc);
  run<2, 1>("v_mfma_f32_16x16x32_bf16", "v_exp_f32", out, cyc);
  run<2, 5>("v_mfma_f32_16x16x32_bf16", "v_pk_fma_f32", out, cyc);
  run_roles<0, 0>("v_mfma_f32_16x16x4_f32", "v_fma_f32", out, cyc);
  run_roles<0, 1>("v_mfma_f32_16x16x4_f32", "v_exp_f32", out, cyc);
  run_roles<2, 0>("v_mfma_f32_16x16x32_bf16", "v_fma_f32", out, cyc);
  run_roles<2, 1>("v_mfma_f32_16x16x32_bf16", "v_exp_f32", out, cyc);
  return 0;
}
