// Issue-rate probe for a few gfx950 vector instructions: one wave per SIMD-sized workgroup runs a long unrolled stream of
// independent instructions of one kind and reports shader cycles (s_memtime) per instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/bin/valu_rate_probe && tools/bin/valu_rate_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kIters = 2000;
constexpr int kLanes = 16;   // independent accumulators per iteration

template <int KIND>
__global__ void __launch_bounds__(64) probe(float* out, unsigned long long* cycles, float seed) {
  float a[kLanes];
  for (int i = 0; i < kLanes; ++i) a[i] = seed + i + threadIdx.x;
  const float m = seed * 0.5f + 1.f;
  bf16x2 one;
  one[0] = (__bf16)(-1.0f);
  one[1] = (__bf16)(0.0f);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < kLanes; ++i) {
      if (KIND == 0) {
        a[i] = __builtin_fmaf(a[i], m, 1.0f);
      } else if (KIND == 1) {
        a[i] = __builtin_amdgcn_exp2f(a[i]);
      } else if (KIND == 2) {
        bf16x2 p = __builtin_bit_cast(bf16x2, __builtin_bit_cast(unsigned, a[i]));
        a[i] = __builtin_amdgcn_fdot2_f32_bf16(p, one, a[i], false);
      } else if (KIND == 3) {
        bf16x2 p;
        p[0] = (__bf16)a[i];
        p[1] = (__bf16)m;
        a[i] = __builtin_bit_cast(float, p);
      } else if (KIND == 4) {
        a[i] = __builtin_bit_cast(float, __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, a[i]), __builtin_bit_cast(unsigned, m), 0x07060302u));
      } else if (KIND == 5) {
        a[i] = __builtin_amdgcn_rcpf(a[i]);
      }
    }
    if (KIND == 6) {
#pragma unroll
      for (int i = 0; i < kLanes; i += 2) {
        f32x2 v = {a[i], a[i + 1]};
        v = __builtin_elementwise_fma(v, f32x2{m, m}, f32x2{1.f, 1.f});
        a[i] = v[0];
        a[i + 1] = v[1];
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < kLanes; ++i) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_iter) {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 64 * sizeof(float));
  (void)hipMalloc(&cyc, sizeof(unsigned long long));
  hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(64), 0, 0, out, cyc, 0.25f);
  hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(64), 0, 0, out, cyc, 0.25f);
  (void)hipDeviceSynchronize();
  unsigned long long h = 0;
  (void)hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-22s %8.2f counter ticks per instruction (%d per iteration)\n", name, (double)h / ((double)kIters * per_iter), per_iter);
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  run<0>("v_fma_f32", kLanes);
  run<6>("v_pk_fma_f32", kLanes / 2);
  run<1>("v_exp_f32", kLanes);
  run<5>("v_rcp_f32", kLanes);
  run<2>("v_dot2_f32_bf16", kLanes);
  run<3>("v_cvt_pk_bf16_f32", kLanes);
  run<4>("v_perm_b32", kLanes);
  return 0;
}
