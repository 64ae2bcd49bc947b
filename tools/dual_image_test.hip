// Standalone check of the dual-use LDS weight image (csrc/m3g_dual_image.h): Y = W X and Z = W^T D through bf16x3
// MFMA chains reading ONE image by rows (ds_read_b64) and transposed (ds_read_b64_tr_b16), against fp64 on the host.
// Build: hipcc -O3 -std=c++20 --offload-arch=gfx950 tools/dual_image_test.hip -o gpurun_out/dual_image_test
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../torch-m3gnet_amd/csrc/m3g_dual_chain.h"

using namespace m3g;

template <int ROWS>
__global__ void __launch_bounds__(64) k_test(const float* img, const float* X /*[16][64]*/, const float* D /*[16][ROWS]*/,
                                             float* Y /*[16][ROWS]*/, float* Z /*[16][64]*/) {
  __shared__ __attribute__((aligned(16))) float lds[ROWS * 64];
  for (int i = threadIdx.x; i < ROWS * 64; i += 64) lds[i] = img[i];
  __syncthreads();
  const int lane = threadIdx.x, m = lane & 15, q = lane >> 4;
  constexpr int OB = ROWS / 16;
  f32x4 x[4], y[OB], d[OB], z[4];
  for (int b = 0; b < 4; ++b)
    for (int r = 0; r < 4; ++r) x[b][r] = X[m * 64 + b * 16 + 4 * q + r];
  for (int b = 0; b < OB; ++b)
    for (int r = 0; r < 4; ++r) { d[b][r] = D[m * ROWS + b * 16 + 4 * q + r]; y[b][r] = 0.f; }
  for (int b = 0; b < 4; ++b) z[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  chain_dual<OB, 2, ROWS>(lds, x, y, lane);
  chain_dual_t<4, OB / 2, ROWS>(lds, d, z, lane);
  for (int b = 0; b < OB; ++b)
    for (int r = 0; r < 4; ++r) Y[m * ROWS + b * 16 + 4 * q + r] = y[b][r];
  for (int b = 0; b < 4; ++b)
    for (int r = 0; r < 4; ++r) Z[m * 64 + b * 16 + 4 * q + r] = z[b][r];
}

template <int ROWS>
static int run() {
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> W(ROWS * 64), X(16 * 64), D(16 * ROWS), img(ROWS * 64), Y(16 * ROWS), Z(16 * 64);
  for (auto& v : W) v = nd(rng) * 0.2f;
  for (auto& v : X) v = nd(rng);
  for (auto& v : D) v = nd(rng) * 1e-6f;   // gradient-sized operands
  pack_dual_image(img.data(), ROWS, [&](int row, int col) { return W[row * 64 + col]; });
  float *dimg, *dX, *dD, *dY, *dZ;
  (void)hipMalloc(&dimg, img.size() * 4); (void)hipMalloc(&dX, X.size() * 4); (void)hipMalloc(&dD, D.size() * 4);
  (void)hipMalloc(&dY, Y.size() * 4); (void)hipMalloc(&dZ, Z.size() * 4);
  (void)hipMemcpy(dimg, img.data(), img.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dD, D.data(), D.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_test<ROWS>, dim3(1), dim3(64), 0, 0, dimg, dX, dD, dY, dZ);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  (void)hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(Z.data(), dZ, Z.size() * 4, hipMemcpyDeviceToHost);
  double ey = 0, my = 0, ez = 0, mz = 0;
  for (int e = 0; e < 16; ++e) {
    for (int o = 0; o < ROWS; ++o) {
      double r = 0;
      for (int k = 0; k < 64; ++k) r += (double)W[o * 64 + k] * X[e * 64 + k];
      ey = std::max(ey, std::fabs(r - Y[e * ROWS + o])); my = std::max(my, std::fabs(r));
    }
    for (int k = 0; k < 64; ++k) {
      double r = 0;
      for (int o = 0; o < ROWS; ++o) r += (double)W[o * 64 + k] * D[e * ROWS + o];
      ez = std::max(ez, std::fabs(r - Z[e * 64 + k])); mz = std::max(mz, std::fabs(r));
    }
  }
  printf("ROWS=%d  rows: err %.3e of %.3e (rel %.2e)   transposed: err %.3e of %.3e (rel %.2e)\n", ROWS, ey, my, ey / my, ez, mz, ez / mz);
  return (ey / my < 3e-5 && ez / mz < 3e-5) ? 0 : 1;
}

int main() {
  int rc = run<64>() | run<128>();
  printf(rc ? "FAIL\n" : "PASS\n");
  return rc;
}
