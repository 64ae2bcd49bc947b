// How well do the MFMA phases of one wave overlap the vector / memory phases of the other waves on its SIMD?
// Every wave loops over "tiles" shaped like the fp32 edge kernels' (m3g_edge_mfma.hip): a chain of CHAIN LDS-fed
// v_mfma_f32_16x16x4_f32 (8 accumulators), then a vector phase of NVAL gated activations (2 v_exp + 1 v_rcp + ~6 VALU each),
// optionally (MEM) preceded by a dependent 16-byte gather per lane from a 64 MB table.  Prints the SIMD's cycles per MFMA
// (32 = the matrix pipe never idles) for 1, 2 and 4 waves per SIMD.
//   PRIO 1: s_setprio 1 around the chain (what the kernels do), 2: s_setprio 3 - (wave / 4), a fixed order among the SIMD's waves;
//   STAG 1: wave w starts after w/4 quarter-periods of dummy work;
//   LOCK K > 0: a FIFO semaphore per SIMD in LDS admits at most K of its waves to their chains at a time.
//   hipcc -O3 --offload-arch=gfx950 tools/phase_overlap_probe.hip -o tools/bin/phase_overlap_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kTiles = 200;

template <int CHAIN, int NVAL, int PRIO, int STAG, int MEM, int LOCK, int ROLE = 0, int AREG = 0, int NOTRANS = 0, int EARLY = 0>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) probe(float* out, unsigned long long* cycles, const float* table, float seed) {
  __shared__ float img[128 * 64];
  __shared__ int sem[8];   // [simd]: tickets handed out, [4 + simd]: chains finished
  __shared__ unsigned long long rel_time[4];   // when the SIMD's semaphore was last released
  unsigned long long t_gap = 0;
  if (threadIdx.x < 8) sem[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < 128 * 64; i += blockDim.x) img[i] = seed * 1e-3f + 1e-6f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  float x[NVAL > 0 ? NVAL : 1];
  for (int i = 0; i < NVAL; ++i) x[i] = seed * 0.1f + 0.01f * i + 0.001f * lane;
  if (STAG) {   // dummy vector work: wave class c = wave / 4 waits c quarter-periods
    float d = seed;
    const int n = (wave >> 2) * (CHAIN * 8 / 4 + NVAL * 9 / 4);
    for (int i = 0; i < n; ++i) d = __builtin_fmaf(d, 1.0001f, 0.5f);
    x[0] += d * 1e-30f;
  }
  unsigned idx = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u;
  unsigned long long t_acq = 0, t_chain = 0, ta, tb, tc;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < kTiles; ++t) {
    int lv = lane;
    asm volatile("" : "+v"(lv));
    if (MEM) {
      idx = idx * 1664525u + 1013904223u;
      const f32x4 g = *(const f32x4*)(table + ((idx >> 8) & 0xffffff) / 4 * 4);   // 64 MB table: misses L2 mostly
      acc[0] += g;
    }
    const bool do_chain = ROLE == 0 || (ROLE > 0 && (wave >> 2) < ROLE), do_valu = ROLE == 0 || ROLE < 0 || (wave >> 2) >= ROLE;   // wave-uniform
    if (do_chain) {
    if (LOCK) ta = __builtin_readcyclecounter();
    if (LOCK) {
      int tk = 0;
      if (lane == 0) tk = atomicAdd(&sem[wave & 3], 1);
      tk = __builtin_amdgcn_readfirstlane(tk);
      while (__builtin_amdgcn_readfirstlane(((volatile int*)sem)[4 + (wave & 3)]) + LOCK <= tk) __builtin_amdgcn_s_sleep(1);
    }
    if (LOCK) { tb = __builtin_readcyclecounter(); if (t > 0) t_gap += tb - ((volatile unsigned long long*)rel_time)[wave & 3]; }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2) { const int c = __builtin_amdgcn_readfirstlane(wave >> 2); if (c == 0) __builtin_amdgcn_s_setprio(3); else if (c == 1) __builtin_amdgcn_s_setprio(2); else if (c == 2) __builtin_amdgcn_s_setprio(1); }
#pragma unroll
    for (int k = 0; k < CHAIN / 8; ++k) {
      if (LOCK && EARLY > 0 && k == CHAIN / 8 - EARLY && lane == 0) { ((volatile unsigned long long*)rel_time)[wave & 3] = __builtin_readcyclecounter(); atomicAdd(&sem[4 + (wave & 3)], 1); }
      const float b = acc[(k + 3) & 7][k & 3] * 1e-20f + x[0];
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(AREG ? x[0] : img[((k & 15) * 8 + ob) * 64 + lv], b, acc[ob], 0, 0, 0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (LOCK && EARLY == 0 && lane == 0) { ((volatile unsigned long long*)rel_time)[wave & 3] = __builtin_readcyclecounter(); atomicAdd(&sem[4 + (wave & 3)], 1); }
    if (LOCK) { tc = __builtin_readcyclecounter(); t_acq += tb - ta; t_chain += tc - tb; }
    }
    if (do_valu)
#pragma unroll
    for (int i = 0; i < NVAL; ++i) {
      const float p = x[i] + acc[i & 7][i & 3] * 1e-20f, g = x[(i + 1) % NVAL];
      if (NOTRANS) x[i] = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(p, g, 0.3f), __builtin_fmaf(p, 0.5f, g), 0.1f), __builtin_fmaf(g, g, p), __builtin_fmaf(p, p, 0.2f)) * 1e-3f + 0.3f;
      else x[i] = p * __builtin_amdgcn_rcpf((1.f + __expf(-p)) * (1.f + __expf(-g))) + 0.3f;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float res = 0.f;
  for (int i = 0; i < NVAL; ++i) res += x[i];
  for (int i = 0; i < 8; ++i) res += acc[i][0] + acc[i][2];
  out[blockIdx.x * 1024 + threadIdx.x] = res;
  if (lane == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
  if (LOCK && lane == 0 && blockIdx.x == 0) { cycles[256 * 16 + wave * 2] = t_acq; cycles[256 * 16 + wave * 2 + 1] = t_gap; }
}

template <int CHAIN, int NVAL, int PRIO, int STAG, int MEM, int LOCK = 0, int EARLY = 0>
static void run(float* out, unsigned long long* cyc, const float* table) {
  printf("chain %3d MFMAs, %3d activations, prio %d, stagger %d, gather %d, lock %d (release %d k-steps early):", CHAIN, NVAL, PRIO, STAG, MEM, LOCK, EARLY);
  for (int waves : {4, 8, 16}) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<CHAIN, NVAL, PRIO, STAG, MEM, LOCK, 0, 0, 0, EARLY>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, table, 1.0f);
    (void)hipDeviceSynchronize();
    unsigned long long h[16], mx = 0;
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("  %d/SIMD %6.2f", waves / 4, (double)mx / ((double)kTiles * CHAIN * (waves / 4)));
    if (LOCK) {
      unsigned long long ph[32];
      (void)hipMemcpy(ph, cyc + 256 * 16, sizeof(ph), hipMemcpyDeviceToHost);
      printf(" [wave 0 per tile: acquire %5.0f, from last release to start %5.0f]", (double)ph[0] / kTiles, (double)ph[1] / kTiles);
    }
  }
  printf("   cycles per MFMA per SIMD\n");
}

// fixed roles, 4 waves per SIMD: NCH of them run chains only, the others the activation phase only
template <int NCH, int PRIO, int AREG = 0, int NOTRANS = 0>
static void run_roles(float* out, unsigned long long* cyc, const float* table) {
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<128, 32, PRIO, 0, 0, 0, (NCH == 0 ? -1 : NCH), AREG, NOTRANS>), dim3(256), dim3(1024), 0, 0, out, cyc, table, 1.0f);
  (void)hipDeviceSynchronize();
  unsigned long long h[16];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("roles (A from %s, %s): %d chain-only + %d activation-only waves per SIMD, prio %d: chain waves %6.2f cycles per MFMA per SIMD, activation waves %6.2f cycles per activation per SIMD\n",
         AREG ? "reg" : "LDS", NOTRANS ? "fma only" : "exp/rcp", NCH, 4 - NCH, PRIO, NCH > 0 ? (double)h[0] / ((double)kTiles * 128 * NCH) : 0.0, NCH < 4 ? (double)h[15] / ((double)kTiles * 32 * (4 - NCH)) : 0.0);
}

int main() {
  float *out, *table;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
  (void)hipMalloc(&cyc, (256 * 16 + 32) * sizeof(unsigned long long));
  (void)hipMalloc(&table, 64u << 20);
  (void)hipMemset(table, 0, 64u << 20);
  run<128, 0, 0, 0, 0>(out, cyc, table);
  run<128, 32, 0, 0, 0>(out, cyc, table);
  run<128, 32, 1, 0, 0>(out, cyc, table);
  run<128, 32, 1, 1, 0>(out, cyc, table);
  run<128, 32, 0, 1, 0>(out, cyc, table);
  run<128, 64, 1, 0, 0>(out, cyc, table);
  run<128, 32, 1, 0, 1>(out, cyc, table);
  run<128, 32, 0, 0, 1>(out, cyc, table);
  run<128, 32, 1, 1, 1>(out, cyc, table);
  run<128, 32, 2, 0, 0>(out, cyc, table);
  run<128, 32, 2, 0, 1>(out, cyc, table);
  run<128, 32, 1, 0, 0, 1>(out, cyc, table);
  run<128, 32, 1, 0, 0, 2>(out, cyc, table);
  run<128, 32, 1, 0, 0, 3>(out, cyc, table);
  run<128, 32, 1, 0, 1, 1>(out, cyc, table);
  run<128, 32, 1, 0, 1, 2>(out, cyc, table);
  run<128, 32, 1, 0, 1, 3>(out, cyc, table);
  run<128, 32, 0, 0, 1, 2>(out, cyc, table);
  run<128, 32, 1, 0, 0, 1, 2>(out, cyc, table);
  run<128, 32, 1, 0, 0, 1, 4>(out, cyc, table);
  run<128, 32, 1, 0, 0, 1, 6>(out, cyc, table);
  run<128, 32, 0, 0, 0, 1, 4>(out, cyc, table);
  run<128, 32, 1, 0, 1, 1, 4>(out, cyc, table);
  run<128, 32, 0, 0, 1, 1, 4>(out, cyc, table);
  run_roles<1, 1>(out, cyc, table);
  run_roles<2, 1>(out, cyc, table);
  run_roles<3, 1>(out, cyc, table);
  run_roles<4, 1>(out, cyc, table);
  run_roles<1, 1, 1>(out, cyc, table);
  run_roles<2, 1, 1>(out, cyc, table);
  run_roles<1, 1, 0, 1>(out, cyc, table);
  run_roles<2, 1, 0, 1>(out, cyc, table);
  run_roles<1, 1, 1, 1>(out, cyc, table);
  run_roles<2, 1, 1, 1>(out, cyc, table);
  run_roles<0, 1, 0, 0>(out, cyc, table);
  run_roles<0, 1, 0, 1>(out, cyc, table);
  return 0;
}
