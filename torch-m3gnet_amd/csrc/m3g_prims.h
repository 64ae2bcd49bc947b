// Device-wide exclusive scan and stable LSD radix sort, written for the graph side of the library (neighbour search, list /
// topology construction -- the replacements of data/material_graph.py:168-254).  They replace the hipCUB calls of rounds 1-5: the
// arrays here are index arrays of at most a few million entries and every result is an integer, so the primitives are small --
//   exclusive_scan   tiles of 2,048 items per workgroup (8 per thread, blocked), tile totals scanned recursively, offsets added:
//                    one launch up to 2,048 items, three up to 4 M, five beyond; in place (in == out) allowed;
//   radix_sort_*     least-significant-digit passes of 8 bits: per-workgroup digit histograms, one scan of the
//                    [digit][workgroup] table, then a stable scatter (a key's rank among the equal digits of its 256-key tile
//                    from eight wave ballots; the four waves' counts cross through LDS).  Stable, deterministic, any length.
// Everything is queued on the caller's stream; temporary storage comes from the caller (sizes from the *_tmp_bytes functions).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace m3g {
namespace prims {

constexpr int kScanThreads = 256, kScanItems = 8, kScanTile = kScanThreads * kScanItems;
inline size_t align256(size_t n) { return (n + 255) / 256 * 256; }

// exclusive scan of one tile; sums != nullptr: the tile's total goes to sums[blockIdx.x]
template <class T>
__global__ void __launch_bounds__(kScanThreads) k_scan_tile(const T* in, T* out, int64_t n, T* sums) {
  __shared__ T wave_tot[kScanThreads / 64];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  T x[kScanItems];
  T run = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const T v = base + j < n ? in[base + j] : T(0);
    x[j] = run;
    run += v;
  }
  // inclusive scan of the threads' totals inside the wave, then across the four waves
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  T inc = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const T up = __shfl_up(inc, off);
    if (lane >= off) inc += up;
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  T before = inc - run;
  T total = 0;
#pragma unroll
  for (int k = 0; k < kScanThreads / 64; ++k) {
    const T t = wave_tot[k];
    if (k < wv) before += t;
    total += t;
  }
#pragma unroll
  for (int j = 0; j < kScanItems; ++j)
    if (base + j < n) out[base + j] = x[j] + before;
  if (sums && threadIdx.x == 0) sums[blockIdx.x] = total;
}
template <class T>
__global__ void __launch_bounds__(kScanThreads) k_scan_add(T* out, int64_t n, const T* __restrict__ offsets) {
  if (blockIdx.x == 0) return;   // (the first tile's offset is zero)
  const T off = offsets[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j)
    if (base + j < n) out[base + j] += off;
}
template <class T>
inline size_t scan_tmp_bytes(int64_t n) {
  size_t bytes = 0;
  for (int64_t m = (n + kScanTile - 1) / kScanTile; m > 1; m = (m + kScanTile - 1) / kScanTile) bytes += align256((size_t)m * sizeof(T));
  return bytes + 256;
}
// out[i] = in[0] + ... + in[i-1]; in == out allowed
template <class T>
inline hipError_t exclusive_scan(const T* in, T* out, int64_t n, void* tmp, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int64_t tiles = (n + kScanTile - 1) / kScanTile;
  T* sums = tiles > 1 ? (T*)tmp : nullptr;
  hipLaunchKernelGGL(k_scan_tile<T>, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, in, out, n, sums);
  if (tiles > 1) {
    const hipError_t e = exclusive_scan<T>(sums, sums, tiles, (char*)tmp + align256((size_t)tiles * sizeof(T)), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_scan_add<T>, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, out, n, sums);
  }
  return hipGetLastError();
}

// ---- stable LSD radix sort --------------------------------------------------------------------------------------------------
constexpr int kSortThreads = 256, kSortBits = 8, kSortDigits = 1 << kSortBits;
constexpr int kSortTilesPerBlock = 16;                       // 4,096 keys per workgroup and pass
constexpr int kSortChunk = kSortThreads * kSortTilesPerBlock;

template <class K>
__device__ __forceinline__ int sort_digit(K key, int shift) { return (int)((key >> shift) & (K)(kSortDigits - 1)); }

template <class K>
__global__ void __launch_bounds__(kSortThreads) k_sort_hist(const K* __restrict__ keys, int64_t n, int shift, int32_t* __restrict__ hist,
                                                            int64_t blocks) {
  __shared__ int32_t h[kSortDigits];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortChunk;
  for (int t = 0; t < kSortTilesPerBlock; ++t) {
    const int64_t i = base + (int64_t)t * kSortThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[sort_digit(keys[i], shift)], 1);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * blocks + blockIdx.x] = h[threadIdx.x];   // [digit][workgroup]: its exclusive scan is every (digit, workgroup)'s first slot
}
template <class K, class V>
__global__ void __launch_bounds__(kSortThreads) k_sort_scatter(const K* __restrict__ keys_in, K* __restrict__ keys_out, const V* __restrict__ vals_in,
                                                               V* __restrict__ vals_out, int64_t n, int shift, const int32_t* __restrict__ offsets,
                                                               int64_t blocks) {
  __shared__ int32_t slot[kSortDigits];                    // next free slot of each digit for this workgroup
  __shared__ int32_t cnt[kSortThreads / 64][kSortDigits];  // per wave: keys of each digit in the current tile
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  slot[threadIdx.x] = offsets[(int64_t)threadIdx.x * blocks + blockIdx.x];
#pragma unroll
  for (int k = 0; k < kSortThreads / 64; ++k) cnt[k][threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortChunk;
  for (int t = 0; t < kSortTilesPerBlock; ++t) {
    const int64_t i = base + (int64_t)t * kSortThreads + threadIdx.x;
    if (base + (int64_t)t * kSortThreads >= n) break;   // uniform
    const bool live = i < n;
    const K key = live ? keys_in[i] : K(0);
    const int d = live ? sort_digit(key, shift) : 0;
    // lanes of this wave holding the same digit (live lanes only): eight ballots
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < kSortBits; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    const int rank = __builtin_popcountll(peers & ((1ull << lane) - 1ull));
    if (live && rank == 0) cnt[wv][d] = __builtin_popcountll(peers);
    __syncthreads();
    if (live) {
      int at = slot[d] + rank;
#pragma unroll
      for (int k = 0; k < kSortThreads / 64; ++k) at += k < wv ? cnt[k][d] : 0;
      keys_out[at] = key;
      if (vals_in) vals_out[at] = vals_in[i];
    }
    __syncthreads();
    {   // thread = digit: advance the digit's slot, clear the tile's counts
      int add = 0;
#pragma unroll
      for (int k = 0; k < kSortThreads / 64; ++k) { add += cnt[k][threadIdx.x]; cnt[k][threadIdx.x] = 0; }
      slot[threadIdx.x] += add;
    }
    __syncthreads();
  }
}
inline int64_t sort_blocks(int64_t n) { return (n + kSortChunk - 1) / kSortChunk; }
// temporary storage of a sort of n keys: the [256][workgroups] histogram table and its scan's own scratch
inline size_t sort_tmp_bytes(int64_t n) {
  const int64_t cells = sort_blocks(n) * kSortDigits;
  return align256((size_t)cells * sizeof(int32_t)) + scan_tmp_bytes<int32_t>(cells);
}
// Sorts by the key bits [begin_bit, end_bit).  keys_a / vals_a hold the input; keys_b / vals_b are buffers of the same size.  The
// passes ping-pong between them; the return value says where the result is: 0 = in (keys_a, vals_a), 1 = in (keys_b, vals_b),
// < 0 = a HIP error.  vals_a == nullptr: keys only.  n < 2^31.
template <class K, class V>
inline int radix_sort(K* keys_a, K* keys_b, V* vals_a, V* vals_b, int64_t n, int begin_bit, int end_bit, void* tmp, hipStream_t s) {
  if (n <= 0) return 0;
  const int64_t blocks = sort_blocks(n), cells = blocks * kSortDigits;
  int32_t* hist = (int32_t*)tmp;
  void* scan_tmp = (char*)tmp + align256((size_t)cells * sizeof(int32_t));
  int where = 0;
  for (int shift = begin_bit; shift < end_bit; shift += kSortBits) {
    K* kin = where ? keys_b : keys_a;
    K* kout = where ? keys_a : keys_b;
    V* vin = vals_a ? (where ? vals_b : vals_a) : nullptr;
    V* vout = vals_a ? (where ? vals_a : vals_b) : nullptr;
    hipLaunchKernelGGL(k_sort_hist<K>, dim3((unsigned)blocks), dim3(kSortThreads), 0, s, (const K*)kin, n, shift, hist, blocks);
    if (exclusive_scan<int32_t>(hist, hist, cells, scan_tmp, s) != hipSuccess) return -1;
    hipLaunchKernelGGL((k_sort_scatter<K, V>), dim3((unsigned)blocks), dim3(kSortThreads), 0, s, (const K*)kin, kout, (const V*)vin, vout, n, shift,
                       (const int32_t*)hist, blocks);
    where ^= 1;
  }
  return hipGetLastError() == hipSuccess ? where : -1;
}

}  // namespace prims
}  // namespace m3g
