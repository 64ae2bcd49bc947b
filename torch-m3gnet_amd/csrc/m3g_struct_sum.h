// Per-structure sums in a fixed order (shared by the stand-alone kernels of m3g_geometry.hip and by the last workgroup of the
// fused small-system launches in m3g_geometry.hip / m3g_node_mfma.hip).
#pragma once
#include "m3g_internal.h"

namespace m3g {

// Per-structure sums without atomics: the atoms of a structure are contiguous when `batch` is sorted (as the reference's
// batching produces it), so one workgroup per structure adds its atoms in a fixed order -- strided private sums, then a
// fixed LDS tree: energies and stresses are then bit-reproducible like the forces.  An unsorted `batch` (flags[3] != 0, which
// the reference's scatter_sum accepts) has no contiguous range: the structure's workgroup then walks ALL atoms and keeps its
// own -- O(N S) index tests on a path nobody benchmarks, in exchange for the same fixed order, no float atomics anywhere and
// no second set of kernels that is launched only to return (round 2 launched both kinds every step).
constexpr int kStructThreads = 1024;   // one large cell is ONE workgroup: 1,024 threads keep its strided loop at ~10 trips for 10k atoms
// `part`: LDS [kStructThreads][W].  REAL threads of the calling workgroup stand in for the 1,024 "virtual" threads the order of
// the sum is defined on (strided private sums, then a halving tree): every caller -- the stand-alone kernels with 1,024 threads,
// the last workgroup of a fused launch with 1,024 or 256 -- forms bit-identical totals.
template <int W, int REAL, class F>
__device__ __forceinline__ void struct_reduce(int s, const int32_t* __restrict__ struct_ptr, const int32_t* __restrict__ flags, int64_t n_atoms,
                                              const int32_t* __restrict__ batch, float (&total)[W], float* part, F per_atom) {
  static_assert(kStructThreads % REAL == 0, "REAL must divide the virtual thread count");
  const bool sorted = flags[3] == 0;   // uniform
  const int a0 = sorted ? struct_ptr[s] : 0, a1 = sorted ? struct_ptr[s + 1] : (int)n_atoms;
  auto strided_sum = [&](int vt, float (&acc)[W]) {
#pragma unroll
    for (int k = 0; k < W; ++k) acc[k] = 0.f;
    for (int a = a0 + vt; a < a1; a += kStructThreads) {
      if (!sorted && batch[a] != s) continue;
      float v[W];
      per_atom(a, s, v);
#pragma unroll
      for (int k = 0; k < W; ++k) acc[k] += v[k];
    }
  };
  if constexpr (REAL == 256) {
    // The same sum, evaluated with fewer barriers (the last workgroup of a fused launch runs this at the launch's tail): thread t
    // holds the virtual threads t + 256 j.  Tree levels 512 and 256 pair (t, t + 512), (t + 256, t + 768) and then (t, t + 256): all
    // in this thread's registers.  Levels 128 and 64 cross waves (two LDS hand-overs), levels 32 .. 1 are lane shifts in wave 0:
    // lane v adds lane v + off exactly as part[v] += part[v + off] does.
    float v0[W], v1[W], v2[W], v3[W];
    strided_sum((int)threadIdx.x, v0);
    strided_sum((int)threadIdx.x + 256, v1);
    strided_sum((int)threadIdx.x + 512, v2);
    strided_sum((int)threadIdx.x + 768, v3);
#pragma unroll
    for (int k = 0; k < W; ++k) { v0[k] += v2[k]; v1[k] += v3[k]; v0[k] += v1[k]; }
    const int t = (int)threadIdx.x;
    if (t >= 128)
#pragma unroll
      for (int k = 0; k < W; ++k) part[(t - 128) * W + k] = v0[k];
    __syncthreads();
    if (t < 128)
#pragma unroll
      for (int k = 0; k < W; ++k) v0[k] += part[t * W + k];
    __syncthreads();
    if (t >= 64 && t < 128)
#pragma unroll
      for (int k = 0; k < W; ++k) part[(t - 64) * W + k] = v0[k];
    __syncthreads();
    if (t < 64) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        float x = v0[k] + part[t * W + k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);   // lanes >= off add garbage-free values nobody reads
        if (t == 0) part[kStructThreads * W - W + k] = x;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < W; ++k) total[k] = part[kStructThreads * W - W + k];   // every thread holds the sums
    __syncthreads();                                                          // (`part` is free for the next structure)
    return;
  }
  if constexpr (REAL == kStructThreads) {
    // One real thread per virtual thread (the stand-alone kernels; one large cell is ONE workgroup, so this is a chain of dependent
    // waits on a single CU).  Same sums in the same order, with fewer waits: the strided loop requests four atoms' values before it
    // adds them (in order), tree levels 512 .. 64 cross waves through LDS (one barrier each), levels 32 .. 1 are lane shifts in
    // wave 0 -- lane v adds lane v + off exactly as part[v] += part[v + off] does.
    const int vt = (int)threadIdx.x;
    float acc[W];
#pragma unroll
    for (int k = 0; k < W; ++k) acc[k] = 0.f;
    if (sorted) {
      for (int a = a0 + vt; a < a1; a += 4 * kStructThreads) {
        float v[4][W];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int aj = a + j * kStructThreads;
          per_atom(aj < a1 ? aj : a, s, v[j]);   // (an index past the range re-reads atom a: its values are not added)
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (a + j * kStructThreads < a1)
#pragma unroll
            for (int k = 0; k < W; ++k) acc[k] += v[j][k];
      }
    } else {
      strided_sum(vt, acc);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) part[vt * W + k] = acc[k];
    __syncthreads();
    for (int off = kStructThreads / 2; off >= 64; off >>= 1) {
      if (vt < off)
#pragma unroll
        for (int k = 0; k < W; ++k) part[vt * W + k] += part[(vt + off) * W + k];
      __syncthreads();
    }
    if (vt < 64) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        float x = part[vt * W + k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);   // lanes >= off add values nobody reads
        if (vt == 0) part[k] = x;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < W; ++k) total[k] = part[k];   // every thread holds the sums
    __syncthreads();                                  // (`part` is free for the next structure)
    return;
  }
  for (int vt = (int)threadIdx.x; vt < kStructThreads; vt += REAL) {
    float acc[W];
    strided_sum(vt, acc);
#pragma unroll
    for (int k = 0; k < W; ++k) part[vt * W + k] = acc[k];
  }
  __syncthreads();
  for (int off = kStructThreads / 2; off > 0; off >>= 1) {
    for (int vt = (int)threadIdx.x; vt < off; vt += REAL)
#pragma unroll
      for (int k = 0; k < W; ++k) part[vt * W + k] += part[(vt + off) * W + k];
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < W; ++k) total[k] = part[k];   // every thread holds the sums
  __syncthreads();                                  // (`part` is free for the next structure)
}
// scaled_total[s] = sum of the structure's scaled atomic energies (nn/readout.py:49-53), total[s] = energy_scale * that (:55-57)
template <int REAL>
__device__ __forceinline__ void struct_energy(int s, const int32_t* __restrict__ struct_ptr, const int32_t* __restrict__ flags, int64_t n_atoms,
                                              const int32_t* __restrict__ batch, const float* ea, float energy_scale,
                                              float* __restrict__ scaled_total, float* __restrict__ total, float* part) {
  float tot[1];
  struct_reduce<1, REAL>(s, struct_ptr, flags, n_atoms, batch, tot, part, [&](int a, int, float* v) { v[0] = ea[a]; });
  if (threadIdx.x == 0) {
    scaled_total[s] = tot[0];
    total[s] = energy_scale * tot[0];
  }
}

}  // namespace m3g
