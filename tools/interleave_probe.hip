// Can ONE instruction stream keep the fp32 matrix pipe busy while it does the vector work of another tile?
// (round-2 finding, tools/phase_overlap_probe.hip: between WAVES the MFMA / vector phases overlap only by chance -- 41 cycles
// per MFMA in the edge kernels against the pipe's 32.)  Here every wave carries TWO tiles, A and B, half a period apart:
//     step 1: chain(A)  ||  activations(B)        step 2: chain(B)  ||  activations(A)
// chain = 128 LDS-fed v_mfma_f32_16x16x4_f32 (8 accumulator blocks x 16 k-steps: one layer of a gated MLP on a 16-edge tile),
// activations = 32 gated activations (2 v_exp + 1 v_rcp + ~6 VALU each) that turn the OTHER tile's accumulators into its next
// chain's B operand.  Modes:
//     0  phases in sequence inside the stream (what one wave of the round-2 kernels does)
//     1  both phases in one scheduling region, the compiler's own order
//     2  both phases in one region, pinned by sched_group_barrier: 1 MFMA : VPM vector instructions (the solver gives up on a
//        region of this size: the listing shows the phases in sequence)
//     5 + NG   quad image [k-step][ob group][lane][4 ob]: one ds_read_b128 feeds 4 MFMAs, A operands requested two r-steps ahead;
//        one fenced region per r-step = 8 MFMAs + 1/16 of the other tile's vector work (NG gated activations),
//        sched_group_barrier 1 MFMA : VPM vector instructions inside the region (VPM 0: the compiler's order)
// A operand: AB = 4 one ds_read_b32 per MFMA (round-2 image layout), AB = 16 one ds_read_b128 per 4 MFMAs ([ob][kblk][lane][4]).
// Prints shader cycles per MFMA per SIMD (32 = the pipe never idles) and the wall-clock TFLOP/s of the launch.
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/interleave_probe.hip -o tools/bin/interleave_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kTiles = 100;   // tile PAIRS per wave

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f.template operator()<I>(), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float fgated(float p, float g) { return p * __builtin_amdgcn_rcpf((1.f + __expf(-p)) * (1.f + __expf(-g))); }

template <int AB>
__device__ __forceinline__ void chain(const float* img, const f32x4 (&hid)[4], f32x4 (&acc)[8], int lane) {
  static_for<8>([&]<int ob>() { acc[ob] = f32x4{0.01f, 0.02f, 0.03f, 0.04f}; });
  static_for<4>([&]<int blk>() {
    if constexpr (AB == 16) {
      f32x4 a[8];
      static_for<8>([&]<int ob>() { a[ob] = *(const f32x4*)(img + ((ob * 4 + blk) * 64 + lane) * 4); });
      static_for<4>([&]<int r>() {
        static_for<8>([&]<int ob>() { acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][r], hid[blk][r], acc[ob], 0, 0, 0); });
      });
    } else {
      static_for<4>([&]<int r>() {
        static_for<8>([&]<int ob>() {
          acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(img[(ob * 16 + blk * 4 + r) * 64 + lane], hid[blk][r], acc[ob], 0, 0, 0);
        });
      });
    }
  });
}

__device__ __forceinline__ void activations(const f32x4 (&acc)[8], f32x4 (&hid)[4]) {
  static_for<4>([&]<int blk>() {
    static_for<4>([&]<int r>() { hid[blk][r] = fgated(acc[blk][r], acc[4 + blk][r]) + 0.5f * fgated(acc[4 + blk][r], acc[blk][r]); });
  });
}

// MODE 5+NG: quad image [kstep][og][lane][4 ob], A operand prefetched TWO r-steps ahead, vector slice per r-step, group barriers
template <int VPM, int NG>
__device__ __forceinline__ void chain_q(const float* img, const f32x4 (&hid)[4], f32x4 (&acc)[8], const f32x4 (&accv)[8], f32x4 (&hidv)[4], int lane) {
  static_for<8>([&]<int ob>() { acc[ob] = f32x4{0.01f, 0.02f, 0.03f, 0.04f}; });
  f32x4 a[3][2];
  const float* p = img + lane * 4;
  static_for<2>([&]<int k>() { a[k][0] = *(const f32x4*)(p + k * 512); a[k][1] = *(const f32x4*)(p + k * 512 + 256); });
  static_for<16>([&]<int ks>() {
    constexpr int blk = ks >> 2, r = ks & 3, cur = ks % 3, nxt = (ks + 2) % 3;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ks < 14) {
      a[nxt][0] = *(const f32x4*)(p + (ks + 2) * 512);
      a[nxt][1] = *(const f32x4*)(p + (ks + 2) * 512 + 256);
    }
    static_for<8>([&]<int ob>() { acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][ob >> 2][ob & 3], hid[blk][r], acc[ob], 0, 0, 0); });
    if constexpr (NG >= 1) hidv[blk][r] = fgated(accv[blk][r], accv[4 + blk][r]);
    if constexpr (NG >= 2) hidv[blk][r] += 0.5f * fgated(accv[4 + blk][r], accv[blk][r]);
    if constexpr (NG >= 3) hidv[blk][r] += 0.25f * fgated(accv[4 + blk][r] + 1.f, accv[blk][r]);
    if constexpr (VPM > 0) {
      static_for<8>([&]<int i>() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (i < 2 && ks < 14) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, VPM, 0);
      });
    }
  });
  __builtin_amdgcn_sched_barrier(0);
}

template <int MODE, int AB, int VPM>
__device__ __forceinline__ void half_step(const float* img, const f32x4 (&hid_c)[4], f32x4 (&acc_c)[8], const f32x4 (&acc_v)[8],
                                          f32x4 (&hid_v)[4], int lane) {
  if constexpr (MODE >= 5) { chain_q<VPM, MODE - 5>(img, hid_c, acc_c, acc_v, hid_v, lane); }
  else if constexpr (MODE == 0) {
    chain<AB>(img, hid_c, acc_c, lane);
    __builtin_amdgcn_sched_barrier(0);
    activations(acc_v, hid_v);
    __builtin_amdgcn_sched_barrier(0);
  } else {
    __builtin_amdgcn_sched_barrier(0);
    chain<AB>(img, hid_c, acc_c, lane);
    activations(acc_v, hid_v);
    if constexpr (MODE == 2) {
      // 128 MFMAs; vector work: 32 gated x ~10 instructions = ~330 (VALU + transcendental)
      static_for<128>([&]<int i>() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
        if constexpr (AB == 16) { if constexpr (i % 4 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }   // 1 DS read per 4
        else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, VPM, 0);                     // VPM VALU / transcendental
      });
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int MODE, int AB, int VPM, int WPS>
__global__ void __launch_bounds__(256 * WPS) probe(float* out, unsigned long long* cycles, const float* w, float seed) {
  __shared__ __attribute__((aligned(16))) float img[8 * 16 * 64];
  for (int i = threadIdx.x; i < 8 * 16 * 64; i += blockDim.x) img[i] = w[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 accA[8], accB[8], hidA[4], hidB[4];
  static_for<8>([&]<int i>() {
    accA[i] = f32x4{seed, 0.5f * seed, -seed, 0.1f * i} + 0.001f * lane;
    accB[i] = f32x4{-seed, 0.3f * seed, seed, 0.2f * i} - 0.001f * lane;
  });
  static_for<4>([&]<int i>() { hidA[i] = f32x4{0.1f, 0.2f, 0.3f, 0.4f} * seed + 0.002f * lane; hidB[i] = hidA[i] * 0.5f; });
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < kTiles; ++t) {
    int lv = lane;
    asm volatile("" : "+v"(lv));   // keeps the loop-invariant LDS reads inside the loop
    half_step<MODE, AB, VPM>(img, hidA, accA, accB, hidB, lv);   // chain(A) || activations(B)
    half_step<MODE, AB, VPM>(img, hidB, accB, accA, hidA, lv);   // chain(B) || activations(A)
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float res = 0.f;
  static_for<8>([&]<int i>() { res += accA[i][0] + accA[i][1] + accA[i][2] + accA[i][3] + accB[i][0] + accB[i][1] + accB[i][2] + accB[i][3]; });
  static_for<4>([&]<int i>() { res += hidA[i][0] + hidB[i][3]; });
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
  if (lane == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE, int AB, int VPM, int WPS>
static void run(float* out, unsigned long long* cyc, const float* w) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, AB, VPM, WPS>), dim3(256), dim3(256 * WPS), 0, 0, out, cyc, w, 1.0f);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long h[16], mx = 0;
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 4 * WPS; ++i) mx = h[i] > mx ? h[i] : mx;
  float o[4];
  (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
  const double mfmas = (double)kTiles * 256;
  const double flops = 256.0 * 4 * WPS * mfmas * 2048.0;
  printf("mode %d  A-read %2d B  %d vector/MFMA  %d wave(s)/SIMD: %6.2f cycles per MFMA per SIMD   %7.1f us  %6.1f TFLOP/s   (check %g)\n", MODE, AB,
         VPM, WPS, (double)mx / (mfmas * WPS), ms * 1e3, flops / (ms * 1e-3) / 1e12, (double)o[1]);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
}

int main() {
  float *out, *w;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
  (void)hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
  (void)hipMalloc(&w, 8 * 16 * 64 * sizeof(float));
  static float hw[8 * 16 * 64];
  unsigned s = 12345u;
  for (int i = 0; i < 8 * 16 * 64; ++i) { s = s * 1664525u + 1013904223u; hw[i] = ((int)(s >> 8) % 2001 - 1000) * 1.2e-4f; }
  (void)hipMemcpy(w, hw, sizeof(hw), hipMemcpyHostToDevice);
  run<0, 4, 0, 1>(out, cyc, w);
  run<0, 4, 0, 2>(out, cyc, w);
  run<0, 4, 0, 4>(out, cyc, w);
  run<0, 16, 0, 1>(out, cyc, w);
  run<0, 16, 0, 2>(out, cyc, w);
  run<1, 16, 0, 1>(out, cyc, w);
  run<1, 16, 0, 2>(out, cyc, w);
  run<5, 16, 0, 1>(out, cyc, w);   // chain only, prefetched quads, fenced r-steps
  run<5, 16, 0, 2>(out, cyc, w);
  run<6, 16, 2, 1>(out, cyc, w);   // 16 gated activations per chain
  run<6, 16, 2, 2>(out, cyc, w);
  run<7, 16, 2, 1>(out, cyc, w);   // 32
  run<7, 16, 3, 1>(out, cyc, w);
  run<7, 16, 4, 1>(out, cyc, w);
  run<7, 16, 0, 1>(out, cyc, w);   // fenced r-steps, no group barriers
  run<7, 16, 2, 2>(out, cyc, w);
  run<7, 16, 3, 2>(out, cyc, w);
  run<7, 16, 4, 2>(out, cyc, w);
  run<8, 16, 3, 1>(out, cyc, w);   // 48
  run<8, 16, 4, 1>(out, cyc, w);
  run<8, 16, 5, 1>(out, cyc, w);
  run<8, 16, 4, 2>(out, cyc, w);
  run<8, 16, 5, 2>(out, cyc, w);
  return 0;
}
