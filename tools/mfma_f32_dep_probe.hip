// Aggregate issue rate of v_mfma_f32_16x16x4_f32 on one SIMD as a function of (waves per SIMD, independent accumulators
// per wave, LDS-fed A operand or not).  Prints shader cycles per MFMA per SIMD (32 = the pipe's rate).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f32_dep_probe.hip -o tools/bin/mfma_f32_dep_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kIters = 2000;

// MODE 0: A operand in a register; 1: register + one independent v_fma_f32 per MFMA in the same wave; 2: A from LDS at fixed
// offsets (one opaque copy of the lane id per 8 MFMAs keeps the reads inside the loop); 3: A from LDS with one address
// computation (VALU) per read
template <int ACC, int MODE>
__global__ void __launch_bounds__(1024) probe(float* out, unsigned long long* cycles, float seed) {
  __shared__ float img[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) img[i] = seed + i;
  __syncthreads();
  f32x4 acc[ACC];
  for (int i = 0; i < ACC; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  const int lane = threadIdx.x & 63;
  float a = seed + lane;
  const float b = seed * 0.5f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  float y[8];
  for (int j = 0; j < 8; ++j) y[j] = seed + j;
  for (int it = 0; it < kIters; ++it) {
    int lv = lane;
    asm volatile("" : "+v"(lv));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 3) a = img[((it + j) & 63) * 64 + lane];
      if (MODE == 2) a = img[j * 64 + lv];
      if (MODE == 1) y[j] = __builtin_fmaf(y[j], b, 1.0f);
      acc[j % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j % ACC], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float res = 0.f;
  for (int j = 0; j < 8; ++j) res += y[j];
  for (int i = 0; i < ACC; ++i) res += acc[i][0] + acc[i][2];
  out[blockIdx.x * 1024 + threadIdx.x] = res;
  if (lane == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int ACC, int MODE>
static void run(int waves, float* out, unsigned long long* cyc) {
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<ACC, MODE>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
  (void)hipDeviceSynchronize();
  unsigned long long h[16];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long mx = 0;
  for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
  const int per_simd = waves / 4 > 0 ? waves / 4 : 1;
  const char* names[4] = {"reg", "reg + v_fma per MFMA", "LDS fixed offsets", "LDS + address VALU"};
  printf("waves/SIMD %d  accumulators/wave %d  A from %-22s: %6.2f cycles per MFMA per SIMD\n", per_simd, ACC, names[MODE],
         (double)mx / (kIters * 8.0 * per_simd));
}

int main() {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
  (void)hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
  for (int waves : {4, 8, 16}) {
    run<8, 0>(waves, out, cyc);
    run<8, 1>(waves, out, cyc);
    run<8, 2>(waves, out, cyc);
    run<8, 3>(waves, out, cyc);
    run<2, 2>(waves, out, cyc);
  }
  return 0;
}
