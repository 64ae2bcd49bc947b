// HBM bandwidth probe with hand-written streaming kernels (16-byte accesses, 8 independent loads in flight per thread,
// grid sized to fill the chip): read-only sum, copy, and the nontemporal variants.  Complements tools/hbm_bw_probe.py.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_bw_probe.hip -o tools/bin/hbm_bw_probe && tools/bin/hbm_bw_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ in, size_t n, float* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n; i += 8 * stride) {
    f4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = NT ? __builtin_nontemporal_load(in + i + j * stride) : in[i + j * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += t[j];
  }
  for (; i < n; i += stride) acc += in[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) out[0] = acc[0];
}

template <bool NT>
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n; i += 8 * stride) {
    f4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = NT ? __builtin_nontemporal_load(in + i + j * stride) : in[i + j * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (NT) __builtin_nontemporal_store(t[j], out + i + j * stride);
      else out[i + j * stride] = t[j];
    }
  }
  for (; i < n; i += stride) out[i] = in[i];
}

template <class F>
static double time_ms(F f, int reps = 20) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const size_t bytes = (size_t)2 << 30, n = bytes / 16;
  f4 *in, *out;
  float* sink;
  (void)hipMalloc(&in, bytes);
  (void)hipMalloc(&out, bytes);
  (void)hipMalloc(&sink, 4);
  (void)hipMemset(in, 0, bytes);
  for (int grid : {1024, 2048, 4096, 8192}) {
    const double r = time_ms([&] { hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, in, n, sink); });
    const double rn = time_ms([&] { hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, in, n, sink); });
    const double c = time_ms([&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, in, out, n); });
    const double cn = time_ms([&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, in, out, n); });
    printf("grid %5d: read %.2f TB/s, read nt %.2f, copy %.2f TB/s (read + write), copy nt %.2f\n", grid, bytes / r / 1e9, bytes / rn / 1e9,
           2.0 * bytes / c / 1e9, 2.0 * bytes / cn / 1e9);
  }
  return 0;
}
