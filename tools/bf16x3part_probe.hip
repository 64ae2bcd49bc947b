// Probe for VERDICT r04 item 6: would an fp32-WIDE split of the dense chains pay in the FORWARD edge kernel?
//   operands as THREE bf16 parts (8 + 8 + 8 significand bits: exact, no scale search -- bf16 has fp32's exponent range),
//   8 of the 9 part products per k-step on v_mfma_f32_16x16x32_bf16 (lo x lo dropped: <= 2^-32 |a||b|), fp32 accumulate.
// Part 1 (numerics, on the hardware): Y[16,16] = W[16,64] X[64,16] with rows spanning six decades, computed as
//   (a) a k-ordered fp32 fmaf chain on the vector ALU, (b) v_mfma_f32_16x16x4_f32 (the shipped exact mode), (c) the three-part
//   bf16 chain with ONE accumulator, (d) the same with the 7 small products in a second accumulator added at the end;
//   printed: max |difference| to an fp64 reference in units of the result's fp32 ulp, and to (a).
// Part 2 (time): one "unit" = 4 output blocks x K = 64 (what W2d or W2g of a conv MLP is; a forward tile is 8 such units + 24
//   small MFMAs) in a dependent loop  chain -> SiLU -> chain ...  with the A operands in LDS and the B operands split on the fly:
//   shader cycles per unit at 1, 2 and 4 waves per SIMD for the fp32 chain and the three-part chain, and the kernels' VGPR use.
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 tools/bf16x3part_probe.hip -o tools/bin/bf16x3part_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// x = p0 + p1 + p2 exactly (round-to-nearest parts: 8 significand bits each, the residuals are exact in fp32)
__device__ __host__ inline void split3(float x, __bf16& p0, __bf16& p1, __bf16& p2) {
  p0 = (__bf16)x;
  const float r1 = x - (float)p0;
  p1 = (__bf16)r1;
  const float r2 = r1 - (float)p1;
  p2 = (__bf16)r2;
}

// ------------------------------------------------------------------------------------------------ part 1: numerics
// W [16][64] row-major, X [64][16] row-major, out [4 variants][16][16]
__global__ void __launch_bounds__(64) k_numerics(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ out) {
  const int lane = threadIdx.x, m = lane & 15, q = lane >> 4;
  // (a) fmaf chain: lane (m, q) computes Y[4q + r][m], k ascending
  for (int r = 0; r < 4; ++r) {
    float acc = 0.f;
    for (int k = 0; k < 64; ++k) acc = __builtin_fmaf(W[(4 * q + r) * 64 + k], X[k * 16 + m], acc);
    out[0 * 256 + (4 * q + r) * 16 + m] = acc;
  }
  // (b) fp32 MFMA: 16 k-steps of 4; A lane (i = m, kq = q): W[m][4 s + q]; B lane (j = m, kq = q): X[4 s + q][m]
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < 16; ++s) acc = mfma16(W[m * 64 + 4 * s + q], X[(4 * s + q) * 16 + m], acc);
    for (int r = 0; r < 4; ++r) out[1 * 256 + (4 * q + r) * 16 + m] = acc[r];
  }
  // (c), (d) three bf16 parts, 2 k-steps of 32: A lane (m, q): W[m][32 s + 8 q + j]; B lane (m, q): X[32 s + 8 q + j][m]
  {
    f32x4 one = {0.f, 0.f, 0.f, 0.f}, hi = one, lo = one;
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3], b[3];
      for (int j = 0; j < 8; ++j) {
        __bf16 p0, p1, p2;
        split3(W[m * 64 + 32 * s + 8 * q + j], p0, p1, p2);
        a[0][j] = p0; a[1][j] = p1; a[2][j] = p2;
        split3(X[(32 * s + 8 * q + j) * 16 + m], p0, p1, p2);
        b[0][j] = p0; b[1][j] = p1; b[2][j] = p2;
      }
      // smallest products first
      one = mfma_bf16(a[1], b[2], one); one = mfma_bf16(a[2], b[1], one);
      one = mfma_bf16(a[0], b[2], one); one = mfma_bf16(a[2], b[0], one); one = mfma_bf16(a[1], b[1], one);
      one = mfma_bf16(a[0], b[1], one); one = mfma_bf16(a[1], b[0], one);
      one = mfma_bf16(a[0], b[0], one);
      lo = mfma_bf16(a[1], b[2], lo); lo = mfma_bf16(a[2], b[1], lo);
      lo = mfma_bf16(a[0], b[2], lo); lo = mfma_bf16(a[2], b[0], lo); lo = mfma_bf16(a[1], b[1], lo);
      lo = mfma_bf16(a[0], b[1], lo); lo = mfma_bf16(a[1], b[0], lo);
      hi = mfma_bf16(a[0], b[0], hi);
    }
    for (int r = 0; r < 4; ++r) {
      out[2 * 256 + (4 * q + r) * 16 + m] = one[r];
      out[3 * 256 + (4 * q + r) * 16 + m] = hi[r] + lo[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------ part 2: time
constexpr int kIters = 400;
__device__ __forceinline__ float fsilu(float p) { return p * __builtin_amdgcn_rcpf(1.f + __expf(-p)); }

// MODE 0: fp32 chain (A image [4 ob][16 k-steps][64] floats); MODE 1: three bf16 parts, 8 products (A images [3][4 ob][2 k-steps][64] x 16 B)
template <int MODE>
__global__ void __launch_bounds__(1024) k_time(const float* __restrict__ img, float* out, unsigned long long* cycles, float seed) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = MODE == 0 ? 4 * 16 * 64 : 3 * 4 * 2 * 64 * 4;
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = img[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 x[4];
  for (int b = 0; b < 4; ++b) x[b] = f32x4{seed * (lane + b), seed, -seed * b, seed * 0.5f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
    f32x4 acc[4];
    for (int ob = 0; ob < 4; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == 0) {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float b = x[blk][r];
#pragma unroll
          for (int ob = 0; ob < 4; ++ob) acc[ob] = mfma16(lds[(ob * 16 + blk * 4 + r) * 64 + lane], b, acc[ob]);
        }
    } else {
      const bf16x8* A = reinterpret_cast<const bf16x8*>(lds);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 b[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          __bf16 p0, p1, p2;
          split3(x[2 * s][j], p0, p1, p2);
          b[0][j] = p0; b[1][j] = p1; b[2][j] = p2;
          split3(x[2 * s + 1][j], p0, p1, p2);
          b[0][4 + j] = p0; b[1][4 + j] = p1; b[2][4 + j] = p2;
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
          const bf16x8 a0 = A[(0 * 8 + ob * 2 + s) * 64 + lane], a1 = A[(1 * 8 + ob * 2 + s) * 64 + lane], a2 = A[(2 * 8 + ob * 2 + s) * 64 + lane];
          f32x4 c = acc[ob];
          c = mfma_bf16(a1, b[2], c); c = mfma_bf16(a2, b[1], c);
          c = mfma_bf16(a0, b[2], c); c = mfma_bf16(a2, b[0], c); c = mfma_bf16(a1, b[1], c);
          c = mfma_bf16(a0, b[1], c); c = mfma_bf16(a1, b[0], c);
          c = mfma_bf16(a0, b[0], c);
          acc[ob] = c;
        }
      }
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[ob][r] = fsilu(acc[ob][r]);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float res = 0.f;
  for (int b = 0; b < 4; ++b) res += x[b][0] + x[b][1] + x[b][2] + x[b][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
  if (lane == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

static double ulp32(double v) {
  float f = (float)std::fabs(v);
  if (f == 0.f) return std::ldexp(1.0, -149);
  int e;
  std::frexp(f, &e);
  return std::ldexp(1.0, e - 24);
}

int main() {
  // ---- part 1
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::vector<float> W(16 * 64), X(64 * 16), out(4 * 256);
  float* dW; float* dX; float* dO;
  (void)hipMalloc(&dW, W.size() * 4); (void)hipMalloc(&dX, X.size() * 4); (void)hipMalloc(&dO, out.size() * 4);
  const char* names[4] = {"fmaf chain (vector ALU)", "v_mfma_f32_16x16x4_f32", "3 x bf16, 8 products, one accumulator", "3 x bf16, 8 products, hi + lo accumulators"};
  double worst_ref[4] = {0, 0, 0, 0}, worst_fma[4] = {0, 0, 0, 0};
  for (int trial = 0; trial < 200; ++trial) {
    // rows of W and columns of X spanning six decades, with cancellation (signed values)
    for (int i = 0; i < 16; ++i)
      for (int k = 0; k < 64; ++k) W[i * 64 + k] = u(rng) * std::pow(10.f, -6.f * ((i * 7 + k * 3 + trial) % 17) / 16.f);
    for (int k = 0; k < 64; ++k)
      for (int j = 0; j < 16; ++j) X[k * 16 + j] = u(rng) * std::pow(10.f, -6.f * ((j * 5 + k + trial) % 13) / 12.f);
    (void)hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_numerics, dim3(1), dim3(64), 0, 0, dW, dX, dO);
    (void)hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0.0, mag = 0.0;
        for (int k = 0; k < 64; ++k) { ref += (double)W[i * 64 + k] * (double)X[k * 16 + j]; mag += std::fabs((double)W[i * 64 + k] * (double)X[k * 16 + j]); }
        // errors in ulps of the ACCUMULATOR's scale (sum of |terms|): what an fp32 chain can be held to under cancellation
        const double unit = ulp32(mag);
        for (int v = 0; v < 4; ++v) {
          worst_ref[v] = std::fmax(worst_ref[v], std::fabs(out[v * 256 + i * 16 + j] - ref) / unit);
          worst_fma[v] = std::fmax(worst_fma[v], std::fabs((double)out[v * 256 + i * 16 + j] - (double)out[i * 16 + j]) / unit);
        }
      }
  }
  printf("# part 1: K = 64 dot products, 200 x 256 results, operands spanning six decades; errors in ulps of the accumulator scale (fp32 ulp of sum |terms|)\n");
  for (int v = 0; v < 4; ++v) printf("%-46s max |y - fp64| = %6.3f ulp   max |y - fmaf chain| = %6.3f ulp\n", names[v], worst_ref[v], worst_fma[v]);

  // ---- part 2
  std::vector<float> img(4 * 16 * 64 > 3 * 8 * 64 * 4 ? 4 * 16 * 64 : 3 * 8 * 64 * 4);
  for (size_t i = 0; i < img.size(); ++i) img[i] = 0.02f * u(rng);
  std::vector<float> img1(3 * 8 * 64 * 4);
  {  // bf16 images of small finite values
    __bf16* p = reinterpret_cast<__bf16*>(img1.data());
    for (size_t i = 0; i < img1.size() * 2; ++i) p[i] = (__bf16)(0.02f * u(rng));
  }
  float* dI; float* dI1; float* dOut; unsigned long long* dC;
  (void)hipMalloc(&dI, img.size() * 4); (void)hipMalloc(&dI1, img1.size() * 4); (void)hipMalloc(&dOut, 256 * 1024 * 4); (void)hipMalloc(&dC, 256 * 16 * 8);
  (void)hipMemcpy(dI, img.data(), img.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dI1, img1.data(), img1.size() * 4, hipMemcpyHostToDevice);
  printf("# part 2: one unit = 4 output blocks x K = 64 + SiLU of the 16 results, dependent loop; shader cycles per unit PER SIMD (one MFMA pipe)\n");
  for (int mode = 0; mode < 2; ++mode) {
    hipFuncAttributes fa;
    if (mode == 0) (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_time<0>));
    else (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_time<1>));
    for (int waves : {4, 8, 16}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(256), dim3(64 * waves), 4 * 16 * 64 * 4, 0, dI, dOut, dC, 1e-3f);
        else hipLaunchKernelGGL(k_time<1>, dim3(256), dim3(64 * waves), 3 * 8 * 64 * 16, 0, dI1, dOut, dC, 1e-3f);
      }
      (void)hipDeviceSynchronize();
      unsigned long long h[16], mx = 0;
      (void)hipMemcpy(h, dC, sizeof(h), hipMemcpyDeviceToHost);
      for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
      // `waves / 4` waves share a SIMD: cycles per unit per SIMD = elapsed / (iterations x waves per SIMD)
      printf("%-40s %d wave(s)/SIMD: %7.1f cycles per unit   (%d VGPRs)\n", mode == 0 ? "fp32 chain (64 x v_mfma_f32_16x16x4_f32)" : "3 x bf16 parts (64 x v_mfma_f32_16x16x32_bf16)",
             waves / 4, (double)mx / ((double)kIters * (waves / 4)), fa.numRegs);
    }
  }
  printf("# floor: fp32 64 MFMAs x 32 cycles = 2048; bf16 64 MFMAs x 16 cycles = 1024 (+ the three-part split of 16 values per lane and unit on the vector ALU, which co-executes)\n");
  return 0;
}
