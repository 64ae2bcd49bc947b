// Do vector-ALU instructions issue in the shadow of a matrix instruction of ANOTHER wave on the same SIMD?
// One 512-thread workgroup = two waves per SIMD.  Waves 0-3 run a stream of independent MFMAs (kind M: 0 = fp32
// v_mfma_f32_16x16x4_f32, 1 = bf16 v_mfma_f32_16x16x32_bf16), waves 4-7 a stream of independent vector instructions
// (kind V: 0 = v_fma_f32, 1 = v_exp_f32); each role is also timed alone.  Reports shader cycles (s_memtime) of the
// matrix waves and of the vector waves: if the two roles overlap, the pair takes max(alone, alone), otherwise the sum.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_coexec_probe.hip -o tools/bin/mfma_valu_coexec_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kIters = 4000;

// PRIO: 0 both roles at priority 0; 1: matrix waves at s_setprio 1; 2: vector waves at s_setprio 1
template <int M, int V, int ROLES, int PRIO = 0>   // ROLES bit 0: matrix waves run, bit 1: vector waves run
__global__ void __launch_bounds__(512) probe(float* out, unsigned long long* cycles, float seed) {
  const int wave = threadIdx.x >> 6;
  const bool matrix = wave < 4;
  if (PRIO == 1 && matrix) __builtin_amdgcn_s_setprio(1);
  if (PRIO == 2 && !matrix) __builtin_amdgcn_s_setprio(1);
  unsigned long long t0 = 0, t1 = 0;
  float res = 0.f;
  if (matrix && (ROLES & 1)) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    const float a = seed + threadIdx.x, b = seed * 0.5f;
    bf16x8 ah, bh;
    for (int j = 0; j < 8; ++j) { ah[j] = (__bf16)(seed + j); bh[j] = (__bf16)(seed - j); }
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (M == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
      }
    }
    t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) res += acc[i][0] + acc[i][3];
  } else if (!matrix && (ROLES & 2)) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed + i + threadIdx.x;
    const float m = seed * 0.25f + 1.f;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (V == 0) x[i] = __builtin_fmaf(x[i], m, 1.0f);
        else x[i] = __builtin_amdgcn_exp2f(x[i]);
      }
    }
    t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 16; ++i) res += x[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int M, int V, int ROLES, int PRIO = 0>
static void run(const char* label, float* out, unsigned long long* cyc) {
  hipLaunchKernelGGL((probe<M, V, ROLES, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, 1.0f);   // warm-up
  hipLaunchKernelGGL((probe<M, V, ROLES, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, 1.0f);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double mat = (double)h[0] / (kIters * 8), vec = (double)h[4] / (kIters * 16);
  printf("%-44s matrix waves %7.2f cycles/MFMA   vector waves %6.2f cycles/instruction\n", label, (ROLES & 1) ? mat : 0.0, (ROLES & 2) ? vec : 0.0);
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long));
  run<0, 0, 1>("f32 MFMA alone", out, cyc);
  run<0, 0, 2>("v_fma_f32 alone", out, cyc);
  run<0, 1, 2>("v_exp_f32 alone", out, cyc);
  run<0, 0, 3>("f32 MFMA + v_fma_f32 on the same SIMD", out, cyc);
  run<0, 1, 3>("f32 MFMA + v_exp_f32 on the same SIMD", out, cyc);
  run<0, 0, 3, 1>("f32 MFMA (prio 1) + v_fma_f32 (prio 0)", out, cyc);
  run<0, 1, 3, 1>("f32 MFMA (prio 1) + v_exp_f32 (prio 0)", out, cyc);
  run<0, 0, 3, 2>("f32 MFMA (prio 0) + v_fma_f32 (prio 1)", out, cyc);
  run<1, 0, 1>("bf16 MFMA alone", out, cyc);
  run<1, 0, 3>("bf16 MFMA + v_fma_f32 on the same SIMD", out, cyc);
  run<1, 1, 3>("bf16 MFMA + v_exp_f32 on the same SIMD", out, cyc);
  return 0;
}
