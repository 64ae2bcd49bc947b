// Topology build: MaterialGraph index tensors (int64, reference layout data/material_graph.py:14-107)
// -> int32 receiver-sorted CSR lists used by every kernel.
//   * edges by centre (row_ptr)            -- input must already be centre-sorted, as the reference's
//                                             builder produces it (material_graph.py:182-187)
//   * edges by neighbour (in_ptr/in_edge)  -- reverse-pass gathers without atomics
//   * triplets by first edge (t1) and by second edge (t2), each list in canonical (sorted) order so
//     results do not depend on the order of triplet_edge_index (reference property test
//     tests/test_model.py:21-38).
// Index-only integer work: HBM-bound radix sorts (hipCUB) + binary searches; not on the per-step path
// while the neighbour list is unchanged.
#include <cstdlib>

#include "m3g_internal.h"
#include "m3g_prims.h"

namespace m3g {

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

size_t topo_sort_tmp_bytes(int64_t E, int64_t T) {
  size_t m = (size_t)std::max<int64_t>(std::max<int64_t>(E, T), 1);
  // scratch of the radix sorts (general lists only) and of the active-edge scan (m3g_prims.h); + E + 1: the certificate's per-row
  // flags live behind the two key arrays (launch_hint_kernels)
  const size_t sort = prims::sort_tmp_bytes((int64_t)m), scan = prims::scan_tmp_bytes<int32_t>(E + 1);
  return align_up(std::max(std::max(sort, scan), (size_t)E + 1)) + 2 * align_up(m * sizeof(uint64_t));
}

Topo topo_carve(int64_t N, int64_t E, int64_t T, int64_t S, void* base) {
  Topo t{};
  t.N = N; t.E = E; t.T = T; t.S = S;
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = p ? (void*)(p + off) : nullptr; off += align_up(bytes); return r; };
  t.src = (int32_t*)take(sizeof(int32_t) * (E + 1));
  t.dst = (int32_t*)take(sizeof(int32_t) * (E + 1));
  t.row_ptr = (int32_t*)take(sizeof(int32_t) * (N + 1));
  t.in_ptr = (int32_t*)take(sizeof(int32_t) * (N + 1));
  t.in_edge = (int32_t*)take(sizeof(int32_t) * (E + 1));
  t.in_pair = (int32_t*)take(sizeof(int32_t) * 2 * (E + 1));
  t.in_pos = (int32_t*)take(sizeof(int32_t) * (E + 1));
  t.t1_ptr = (int32_t*)take(sizeof(int32_t) * (E + 1));
  t.t1_e2 = (int32_t*)take(sizeof(int32_t) * (T + 1));
  t.t2_ptr = (int32_t*)take(sizeof(int32_t) * (E + 1));
  t.t2_e1 = (int32_t*)take(sizeof(int32_t) * (T + 1));
  t.act_list = (int32_t*)take(sizeof(int32_t) * (E + 1));
  t.act_scan = (int32_t*)take(sizeof(int32_t) * (E + 2));
  t.act_id = (int32_t*)take(sizeof(int32_t) * (E + 1));
  t.arow_ptr = (int32_t*)take(sizeof(int32_t) * (N + 2));
  t.act_dst = (int32_t*)take(sizeof(int32_t) * (E + 1));
  t.tb_win = (int32_t*)take(sizeof(int32_t) * 6 * (E / kTbRows + 2));
  t.tb_fast = (int32_t*)take(sizeof(int32_t) * 2 * (E / kTbRows + 2));
  t.t1_e2c = (int32_t*)take(sizeof(int32_t) * (T + 1));
  t.t2_e1c = (int32_t*)take(sizeof(int32_t) * (T + 1));
  t.t1_b = (uint8_t*)take((size_t)T + 16);
  t.t2_b = (uint8_t*)take((size_t)T + 16);
  t.batch = (int32_t*)take(sizeof(int32_t) * (N + 1));
  t.struct_ptr = (int32_t*)take(sizeof(int32_t) * (S + 2));
  t.flags = (int32_t*)take(sizeof(int32_t) * kTopoFlags);
  t.n_act = t.flags ? t.flags + 2 : nullptr;
  t.sort_tmp_bytes = topo_sort_tmp_bytes(E, T);
  t.sort_tmp = take(t.sort_tmp_bytes);
  t.total_bytes = off;
  return t;
}

__global__ void k_convert_edges(int64_t N, int64_t E, const int64_t* __restrict__ ei, int32_t* src, int32_t* dst,
                                int32_t* edge_ids, int32_t* flags) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t i = ei[e], j = ei[E + e];
  int bad = 0;
  if (i < 0 || i >= N || j < 0 || j >= N) { bad |= 2; i = 0; j = 0; }
  if (e > 0 && ei[e - 1] > i) bad |= 1;
  src[e] = (int32_t)i;
  dst[e] = (int32_t)j;
  edge_ids[e] = (int32_t)e;
  if (bad) atomicOr(flags, bad);
}

__global__ void k_iota32(int64_t n, int32_t* out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)i;
}
__global__ void k_convert_batch(int64_t N, int64_t S, const int64_t* __restrict__ batch, int32_t* out, int32_t* flags) {
  int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (a >= N) return;
  int64_t b = batch[a];
  if (b < 0 || b >= S) { atomicOr(flags, 2); b = 0; }
  if (a > 0 && batch[a - 1] > batch[a]) atomicOr(flags + 3, 1);   // not sorted: per-structure sums fall back to atomics
  out[a] = (int32_t)b;
}

// `order` (which == 0 only): order[0] != 0 when the (e1, e2) keys are not already in ascending order
// low_out != nullptr: the keys' low words (the partner of each slot, for a list that is already in key order) in the same pass
__global__ void k_convert_triplets(int64_t E, int64_t T, const int64_t* __restrict__ tei, const int32_t* __restrict__ src,
                                   uint64_t* keys, int which, int32_t* flags, int32_t* order, int32_t* __restrict__ low_out = nullptr) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= T) return;
  int64_t e1 = tei[t], e2 = tei[T + t];
  int bad = 0;
  if (e1 < 0 || e1 >= E || e2 < 0 || e2 >= E) { bad |= 2; e1 = 0; e2 = 0; }
  else if (src[e1] != src[e2]) bad |= 4;
  keys[t] = which == 0 ? (((uint64_t)e1 << 32) | (uint64_t)e2) : (((uint64_t)e2 << 32) | (uint64_t)e1);
  if (low_out) low_out[t] = (int32_t)(which == 0 ? e2 : e1);
  if (bad) atomicOr(flags, bad);
  if (which == 0 && t > 0) {
    const int64_t p1 = tei[t - 1], p2 = tei[T + t - 1];
    if (p1 > e1 || (p1 == e1 && p2 > e2)) atomicOr(order, 1);
  }
}
// order[0] |= 2 when some triplet (e1, e2) has no mirror (e2, e1) in the SORTED key list: the list is one-sided.  The mirror can
// only sit in the row of e2, rows[e2] .. rows[e2 + 1] (a handful of slots), so the search stays inside that row.
__global__ void k_check_symmetric(int64_t T, const uint64_t* __restrict__ sorted_keys, const int32_t* __restrict__ rows, int32_t* order) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= T) return;
  const uint64_t k = sorted_keys[t], want = (k << 32) | (k >> 32);
  const int64_t e2 = (int64_t)(k & 0xffffffffu);
  int64_t lo = rows[e2], hi = rows[e2 + 1];
  const int64_t end = hi;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_keys[mid] < want) lo = mid + 1; else hi = mid;
  }
  if (lo >= end || sorted_keys[lo] != want) atomicOr(order, 2);
}

// Incoming-edge lists without a sort, for SYMMETRIC edge lists (every i -> j has its j -> i, with multiplicity: what any full
// neighbour list is): atom j's incoming edges are the mirrors of its outgoing ones, so in_ptr == row_ptr, and one wave per atom
// finds, for its k-th outgoing edge j -> i, the matching entry of row i (the d-th entry with neighbour j for the d-th outgoing
// edge to the same i), then writes the found edge ids in ascending order -- exactly what the stable sort on the neighbour index
// produces.  A mirror that does not exist, or a row longer than the stage, raises flags[0] bit 3 and the caller sorts instead.
constexpr int kInStage = 512;   // outgoing edges per atom handled here
__device__ __forceinline__ void in_edges_symmetric_body(int64_t N, int64_t j, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ dst,
                                                        int32_t* in_ptr, int32_t* in_edge, int32_t* flags, int32_t* s_e, int32_t* s_d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (j > N) return;
  if (lane == 0) in_ptr[j] = row_ptr[j];
  if (j == N) return;
  const int r0 = row_ptr[j], r1 = row_ptr[j + 1], n = r1 - r0;
  if (n == 0) return;
  if (n > kInStage) {
    if (lane == 0) atomicOr(flags, 8);
    for (int k = lane; k < n; k += 64) in_edge[r0 + k] = 0;   // defined until the sort replaces the list
    return;
  }
  int32_t *se = s_e + wave * kInStage, *sd = s_d + wave * kInStage;
  for (int k = lane; k < n; k += 64) sd[k] = dst[r0 + k];   // the row's neighbours, staged once (the duplicate count reads them k times)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  bool missing = false;
  for (int k = lane; k < n; k += 64) {
    const int i = sd[k];
    int dup = 0;
    for (int f = 0; f < k; ++f) dup += sd[f] == i ? 1 : 0;
    int found = -1;
    // row i in batches of eight independent loads (one load per candidate made this a chain of ~20 dependent round trips per lane)
    for (int q = row_ptr[i], q1 = row_ptr[i + 1]; q < q1 && found < 0; q += 8) {
      int v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = q + u < q1 ? dst[q + u] : -1;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (found < 0 && v[u] == (int)j && dup-- == 0) found = q + u;
    }
    missing = missing || found < 0;
    se[k] = found;
  }
  if (__any(missing)) {
    if (lane == 0) atomicOr(flags, 8);
    for (int k = lane; k < n; k += 64) in_edge[r0 + k] = 0;
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int k = lane; k < n; k += 64) {
    const int e = se[k];
    int rank = 0;
    for (int f = 0; f < n; ++f) rank += se[f] < e ? 1 : 0;
    in_edge[r0 + rank] = e;
  }
}
__global__ void __launch_bounds__(256) k_in_edges_symmetric(int64_t N, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ dst,
                                                            int32_t* in_ptr, int32_t* in_edge, int32_t* flags) {
  __shared__ int32_t s_e[4 * kInStage], s_d[4 * kInStage];
  const int64_t j = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);   // wave-uniform
  in_edges_symmetric_body(N, j, row_ptr, dst, in_ptr, in_edge, flags, s_e, s_d);
}

// ptr[r] = first position whose key (high word of keys64, or keys32[pos]) >= r, for r = 0..rows
__global__ void k_lower_bound64(int64_t rows, int64_t n, const uint64_t* __restrict__ keys, int32_t* ptr) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r > rows) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)(keys[mid] >> 32) < r) lo = mid + 1; else hi = mid;
  }
  ptr[r] = (int32_t)lo;
}
__global__ void k_lower_bound32(int64_t rows, int64_t n, const int32_t* __restrict__ keys, int32_t* ptr) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r > rows) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < r) lo = mid + 1; else hi = mid;
  }
  ptr[r] = (int32_t)lo;
}
__global__ void k_low_word(int64_t n, const uint64_t* __restrict__ keys, int32_t* out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)(keys[i] & 0xffffffffu);
}

// active-edge compaction: flag -> exclusive scan -> scatter; compacted row pointers and partner lists.
// An edge is active when it appears in EITHER column of triplet_edge_index: the reference's gather + scatter_sum
// (nn/interaction.py:188-217) accepts any list of (e1, e2) pairs, also one-sided or filtered ones in which an edge is
// only ever a partner e2 -- it then has no aggregate of its own but still needs its row (q, u, dL/dg) in the compacted arrays.
__global__ void k_active_flags(int64_t E, const int32_t* __restrict__ t1_ptr, const int32_t* __restrict__ t2_ptr, int32_t* flag) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e <= E) flag[e] = e < E && (t1_ptr[e + 1] > t1_ptr[e] || t2_ptr[e + 1] > t2_ptr[e]) ? 1 : 0;
}
__global__ void k_active_scatter(int64_t N, int64_t E, const int32_t* __restrict__ t1_ptr, const int32_t* __restrict__ t2_ptr,
                                 const int32_t* __restrict__ scan,
                                 const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ dst, int32_t* act_list,
                                 int32_t* act_dst, int32_t* act_id, int32_t* arow_ptr, int32_t* n_act) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e < E) {
    const bool active = t1_ptr[e + 1] > t1_ptr[e] || t2_ptr[e + 1] > t2_ptr[e];
    act_id[e] = active ? scan[e] : -1;
    if (active) { act_list[scan[e]] = (int32_t)e; act_dst[scan[e]] = dst[e]; }
  }
  if (e <= N) arow_ptr[e] = scan[row_ptr[e]];
  if (e == 0) *n_act = scan[E];
}
// window of compacted rows (all active edges of the centres touched) per three-body workgroup
__global__ void k_tb_windows(int64_t blocks, const int32_t* __restrict__ n_act, const int32_t* __restrict__ act_list,
                             const int32_t* __restrict__ src, const int32_t* __restrict__ arow_ptr,
                             const int32_t* __restrict__ t1_ptr, const int32_t* __restrict__ t2_ptr, int32_t* win) {
  int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= blocks) return;
  const int A = *n_act;
  const int64_t rb = b * kTbRows;
  int w[6] = {0, 0, 0, 0, 0, 0};
  if (rb < A) {
    const int64_t rlast = rb + kTbRows - 1 < A ? rb + kTbRows - 1 : A - 1;
    const int ef = act_list[rb], el = act_list[rlast];
    w[0] = arow_ptr[src[ef]];
    w[1] = arow_ptr[src[el] + 1];
    w[2] = t1_ptr[ef]; w[3] = t1_ptr[el + 1];
    w[4] = t2_ptr[ef]; w[5] = t2_ptr[el + 1];
  }
  for (int k = 0; k < 6; ++k) win[6 * b + k] = w[k];
}
// Are the partner lists of compacted row r complete -- every other active edge of its centre exactly once, as first AND as
// second edge?  (Ascending order inside a row is what the sorted lists have; it makes "exactly once" a local test.)
__global__ void k_tb_row_complete(const int32_t* __restrict__ n_act, const int32_t* __restrict__ act_list, const int32_t* __restrict__ src,
                                  const int32_t* __restrict__ arow_ptr, const int32_t* __restrict__ t1_ptr, const int32_t* __restrict__ t1_e2c,
                                  const int32_t* __restrict__ t2_ptr, const int32_t* __restrict__ t2_e1c, uint8_t* ok, int32_t* stats) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r < 3) stats[r] = 0;   // k_tb_fast, the next kernel, accumulates into them
  if (r >= *n_act) return;
  const int e = act_list[r], c = src[e];
  const int s0 = arow_ptr[c], s1 = arow_ptr[c + 1];
  auto complete = [&](const int32_t* ptr, const int32_t* other) {
    const int b = ptr[e], end = ptr[e + 1];
    if (end - b != s1 - s0 - 1) return false;
    int prev = -1;
    for (int t = b; t < end; ++t) {
      const int o = other[t];
      if (o <= prev || o < s0 || o >= s1 || o == (int)r) return false;
      prev = o;
    }
    return true;
  };
  ok[r] = complete(t1_ptr, t1_e2c) && complete(t2_ptr, t2_e1c) ? 1 : 0;
}
// per three-body workgroup: may it use the moment path (Topo::tb_fast)?  stats: [0] workgroups that may not, [1] largest window,
// [2] most atoms per window (Topo::flags[4..6])
__global__ void __launch_bounds__(256) k_tb_fast(int64_t blocks, const int32_t* __restrict__ n_act, const int32_t* __restrict__ act_list,
                                                 const int32_t* __restrict__ src, const int32_t* __restrict__ win, const uint8_t* __restrict__ ok,
                                                 int32_t* fast, int32_t* stats) {
  // one wave per window: its lanes test the rows, a ballot combines them
  const int64_t b = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= blocks) return;
  int na = 0, a0 = 0;
  const int A = *n_act;
  if (b * kTbRows < A) {
    const int lo = win[6 * b], hi = win[6 * b + 1];
    // This kernel runs in the optimistic pass, BEFORE the malformed-graph flags are read back: an edge list not sorted by centre
    // leaves row_ptr / arow_ptr non-monotonic and k_tb_windows then writes windows with hi <= lo (or beyond A).  Such a window
    // is never used (the build raises), but it must not be dereferenced here: act_list[hi - 1] would index padding.
    if (lo < 0 || hi <= lo || hi > A) {
      if (lane == 0) { fast[2 * b] = 0; fast[2 * b + 1] = 0; }
      return;
    }
    a0 = src[act_list[lo]];
    na = src[act_list[hi - 1]] - a0 + 1;
    bool mine = true;
    if (ok) for (int r = lo + lane; r < hi; r += 64) mine = mine && ok[r] != 0;   // (ok == nullptr: lists complete by construction)
    const bool all = na <= kTbFastAtoms && hi - lo <= kTbCap && __all(mine);
    if (!all) na = 0;
  }
  if (lane == 0) {   // (the statistics of all windows are formed by k_tb_stats: thousands of atomics on three words serialise, 33 us)
    fast[2 * b] = na;
    fast[2 * b + 1] = a0;
  }
}
// one workgroup: stats[0] = windows that hold rows but may not use the moment path, [1] = largest such window (rows), [2] = most atoms
__global__ void __launch_bounds__(1024) k_tb_stats(int64_t blocks, const int32_t* __restrict__ n_act, const int32_t* __restrict__ win,
                                                   const int32_t* __restrict__ fast, int32_t* stats) {
  __shared__ int s_bad[16], s_rows[16], s_atoms[16];
  const int A = *n_act;
  int bad = 0, rows = 0, atoms = 0;
  for (int64_t b = threadIdx.x; b < blocks; b += blockDim.x) {
    if (b * kTbRows >= A) continue;
    const int na = fast[2 * b];
    if (na > 0) { rows = max(rows, win[6 * b + 1] - win[6 * b]); atoms = max(atoms, na); }
    else ++bad;
  }
  for (int o = 32; o > 0; o >>= 1) { bad += __shfl_xor(bad, o); rows = max(rows, __shfl_xor(rows, o)); atoms = max(atoms, __shfl_xor(atoms, o)); }
  if ((threadIdx.x & 63) == 0) { s_bad[threadIdx.x >> 6] = bad; s_rows[threadIdx.x >> 6] = rows; s_atoms[threadIdx.x >> 6] = atoms; }
  __syncthreads();
  if (threadIdx.x == 0) {
    bad = 0; rows = 0; atoms = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { bad += s_bad[k]; rows = max(rows, s_rows[k]); atoms = max(atoms, s_atoms[k]); }
    stats[0] = bad; stats[1] = rows; stats[2] = atoms;
  }
}
__global__ void k_set_word(int32_t* dst, int32_t v) { *dst = v; }
// (idx may hold anything while a build runs ahead of its own verdict -- an edge list that turns out not to be symmetric leaves
// rows of in_edge unwritten until the sort replaces them --, so the lookup is bounded: never an out-of-range read)
__global__ void k_pair_with_lookup(int64_t n, const int32_t* __restrict__ idx, const int32_t* __restrict__ table, int32_t* out, int32_t* pos) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t e = idx[i];
  const bool ok = e >= 0 && (int64_t)e < n;
  out[2 * i] = ok ? e : 0;
  out[2 * i + 1] = ok ? table[e] : -1;
  if (ok) pos[e] = (int32_t)i;   // inverse of the by-neighbour list (a permutation of the edges once the build has passed its checks)
}
// byte-sized partner ids: for triplet slot t of list (t_ptr, t_other_c), the row it belongs to fixes the workgroup and so the
// staged window [lo, lo + n); the partner is stored relative to lo, 255 when it falls outside.  The row is the high word of the
// slot's sorted key when the keys are still at hand (KEYS), otherwise a binary search in t_ptr.
template <bool KEYS>
__global__ void k_partner_bytes(int64_t E, int64_t T, const int32_t* __restrict__ t_ptr, const uint64_t* __restrict__ sorted_keys,
                                const int32_t* __restrict__ act_id, const int32_t* __restrict__ win, const int32_t* __restrict__ t_other_c,
                                uint8_t* out) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= T) return;
  int64_t lo = 0;
  if (KEYS) {
    lo = (int64_t)(sorted_keys[t] >> 32);
  } else {
    int64_t hi = E;   // last e with t_ptr[e] <= t
    while (lo < hi) { int64_t mid = (lo + hi + 1) >> 1; if (t_ptr[mid] <= t) lo = mid; else hi = mid - 1; }
  }
  const int r = act_id[lo];
  const int blk = (r < 0 ? 0 : r) / kTbRows;
  const int wlo = win[6 * blk], whi = win[6 * blk + 1];
  const int n = (whi - wlo) < kTbCap ? (whi - wlo) : kTbCap;
  const int d = t_other_c[t] - wlo;
  out[t] = (uint8_t)((r >= 0 && d >= 0 && d < n && d < 255) ? d : 255);
}
__global__ void k_compact_partners(int64_t T, const int32_t* __restrict__ scan, const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                   int32_t* ac, int32_t* bc) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= T) return;
  ac[t] = scan[a[t]];
  bc[t] = scan[b[t]];
}

// ---- the canonical build in six launches (m3g_topology_build_canonical) ---------------------------------------------------------
// For the lists this library's own builders write -- edges sorted by centre and symmetric, triplets = every ordered pair of a
// centre's edges inside the three-body cutoff, sorted by (e1, e2) -- every array of Topo follows from per-atom rows: the rows of the
// triplet list are found by binary search in triplet_edge_index[0] itself, the active edges of a centre are consecutive in the
// compacted numbering (one 1-workgroup scan over the ATOMS replaces the device scan over the edges), partners as second edge are
// the partners as first edge (every mirrored array is written by the kernel that forms the original).  The arrays are the ones
// the general build writes, bit for bit (tests/test_gpu_graph_build.py compares the buffers); what the general build checks --
// index ranges, row order, edge-list symmetry, triplet order and centres -- is checked here too, and any failed check sends the
// whole build down the general path.  Every kernel after the first leaves at once when the first has flagged the edge list, so
// none walks rows that are not rows; the triplet list is checked in the last kernel (its one pass over the triplets), and the kernels
// before it only read it through bounded searches and a streamed pass that ignores what does not belong to the atom's row.
__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* __restrict__ a, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// the same by a whole wave: 64 probes per round, so ~log64(n) dependent loads instead of log2(n) (4 against 22 on the 10k-atom
// cell's triplet list; the per-lane search made k_canon_triplets a 47-us chain of round trips).  Wave-uniform result.
__device__ __forceinline__ int64_t wave_lower_bound_i64(const int64_t* __restrict__ a, int64_t n, int64_t key, int lane) {
  int64_t lo = 0, hi = n;   // the answer lies in [lo, hi]
  while (hi > lo) {
    const int64_t step = (hi - lo + 63) / 64;
    const int64_t p = lo + (int64_t)(lane + 1) * step - 1;
    const bool ge = p < hi ? a[p] >= key : true;
    const unsigned long long m = __ballot(ge);
    if (m == 0) { lo = hi; break; }
    const int f = __ffsll((long long)m) - 1;
    const int64_t new_hi = lo + (int64_t)(f + 1) * step - 1;
    lo = lo + (int64_t)f * step;
    hi = new_hi < hi ? new_hi : hi;
  }
  return lo;
}
// two keys (key0 <= key1) in the same rounds: the loads of both searches are in flight together
__device__ __forceinline__ void wave_lower_bound2_i64(const int64_t* __restrict__ a, int64_t n, int64_t key0, int64_t key1, int lane,
                                                      int64_t* out0, int64_t* out1) {
  int64_t lo0 = 0, hi0 = n, lo1 = 0, hi1 = n;
  while (hi0 > lo0 || hi1 > lo1) {
    const int64_t step0 = (hi0 - lo0 + 63) / 64, step1 = (hi1 - lo1 + 63) / 64;
    const int64_t p0 = lo0 + (int64_t)(lane + 1) * step0 - 1, p1 = lo1 + (int64_t)(lane + 1) * step1 - 1;
    const bool in0 = hi0 > lo0 && p0 < hi0, in1 = hi1 > lo1 && p1 < hi1;
    const int64_t v0 = in0 ? a[p0] : 0, v1 = in1 ? a[p1] : 0;   // (both loads issued before either is used)
    if (hi0 > lo0) {
      const unsigned long long m = __ballot(in0 ? v0 >= key0 : true);
      if (m == 0) lo0 = hi0;
      else {
        const int f = __ffsll((long long)m) - 1;
        const int64_t nh = lo0 + (int64_t)(f + 1) * step0 - 1;
        lo0 = lo0 + (int64_t)f * step0;
        hi0 = nh < hi0 ? nh : hi0;
      }
    }
    if (hi1 > lo1) {
      const unsigned long long m = __ballot(in1 ? v1 >= key1 : true);
      if (m == 0) lo1 = hi1;
      else {
        const int f = __ffsll((long long)m) - 1;
        const int64_t nh = lo1 + (int64_t)(f + 1) * step1 - 1;
        lo1 = lo1 + (int64_t)f * step1;
        hi1 = nh < hi1 ? nh : hi1;
      }
    }
  }
  *out0 = lo0;
  *out1 = lo1;
}
constexpr int kCanonStage = 512;   // edges per atom whose triplet rows are resolved in LDS (longer rows: one binary search per edge)
// roles by thread index: edges (k_convert_edges), atoms (k_convert_batch), rows (lower bounds in the int64 lists themselves), zero
// fill of the window tables
__global__ void __launch_bounds__(256) k_canon_edges(int64_t N, int64_t E, int64_t S, const int64_t* __restrict__ ei, const int64_t* __restrict__ batch,
                                                     int32_t* src, int32_t* dst, int32_t* out_batch, int32_t* row_ptr, int32_t* struct_ptr, int32_t* tb_win,
                                                     int64_t n_win, int32_t* tb_fast, int64_t n_fast, int32_t* flags) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < E) {
    int64_t a = ei[i], b = ei[E + i];
    int bad = 0;
    if (a < 0 || a >= N || b < 0 || b >= N) { bad |= 2; a = 0; b = 0; }
    if (i > 0 && ei[i - 1] > a) bad |= 1;
    src[i] = (int32_t)a;
    dst[i] = (int32_t)b;
    if (bad) atomicOr(flags, bad);
  }
  if (i < N) {
    int64_t b = batch[i];
    if (b < 0 || b >= S) { atomicOr(flags, 2); b = 0; }
    if (i > 0 && batch[i - 1] > batch[i]) atomicOr(flags + 3, 1);
    out_batch[i] = (int32_t)b;
  }
  if (i <= N) row_ptr[i] = (int32_t)lower_bound_i64(ei, E, i);
  if (i <= S) struct_ptr[i] = (int32_t)lower_bound_i64(batch, N, i);
  if (i < n_win) tb_win[i] = 0;
  if (i < n_fast) tb_fast[i] = 0;
}
// one wave per atom, two roles by workgroup range.  [0, a_blocks): the rows of the triplet list for the atom's edges, and the
// number of its active edges (cnt[atom], scanned by k_canon_scan into arow_ptr).  [a_blocks, 2 a_blocks): its incoming-edge list from
// the mirrors of its own row (k_in_edges_symmetric's body; the two roles share nothing but row_ptr).
__global__ void __launch_bounds__(256) k_canon_rows(int64_t N, int64_t E, int64_t T, int64_t a_blocks, const int64_t* __restrict__ tei,
                                                    const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ dst, int32_t* flags,
                                                    int32_t* t1_ptr, int32_t* t2_ptr, int32_t* cnt, int32_t* in_ptr, int32_t* in_edge) {
  __shared__ int32_t s_lds[8 * kInStage];   // in-edge role: two stages of kInStage per wave; row role: one of kCanonStage per wave
  static_assert(kCanonStage <= 2 * kInStage, "k_canon_rows: the row role's stage must fit the in-edge role's");
  if (flags[0] & 7) return;   // (bits raised by k_canon_edges, complete at this launch boundary)
  if ((int64_t)blockIdx.x >= a_blocks) {
    const int64_t ja = ((int64_t)blockIdx.x - a_blocks) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    in_edges_symmetric_body(N, ja, row_ptr, dst, in_ptr, in_edge, flags, s_lds, s_lds + 4 * kInStage);
    return;
  }
  int* s_start = s_lds;
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // wave-uniform
  if (j > N) return;
  if (j == N) {
    if (lane == 0) { t1_ptr[E] = (int32_t)T; t2_ptr[E] = (int32_t)T; cnt[N] = 0; }
    return;
  }
  const int r0 = row_ptr[j], r1 = row_ptr[j + 1], n = r1 - r0;
  int n_active = 0;
  if (n > kCanonStage) {
    for (int base = r0; base < r1; base += 64) {
      const int e = base + lane;
      const bool valid = e < r1;
      int p = valid ? (int)lower_bound_i64(tei, T, e) : 0;
      int pn = __shfl_down(p, 1);
      if (valid && (lane == 63 || e + 1 >= r1)) pn = (int)lower_bound_i64(tei, T, (int64_t)e + 1);
      if (valid) { t1_ptr[e] = p; t2_ptr[e] = p; }
      n_active += __popcll(__ballot(valid && pn > p));
    }
    if (lane == 0) cnt[j] = n_active;
    return;
  }
  if (n <= 0) {
    if (lane == 0) cnt[j] = 0;
    return;
  }
  // the atom's triplets are one contiguous block [tlo, thi) of the sorted list: two searches by the whole wave, then the block is
  // streamed once and every edge that heads a run of equal e1 learns where its row starts; an edge without triplets takes the start
  // of the next edge that has some (the end of the block when there is none), which is what a lower bound per edge returns
  int64_t tlo, thi;
  wave_lower_bound2_i64(tei, T, r0, r1, lane, &tlo, &thi);
  int* st = s_start + (threadIdx.x >> 6) * kCanonStage;
  for (int k = lane; k < n; k += 64) st[k] = -1;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int64_t t = tlo + lane; t < thi; t += 64) {
    const int64_t e1 = tei[t], prev = t > tlo ? tei[t - 1] : -1;
    if (e1 != prev && e1 >= r0 && e1 < r1) st[e1 - r0] = (int)t;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  int carry = (int)thi;
  for (int base = r0 + ((n - 1) / 64) * 64; base >= r0; base -= 64) {
    const int e = base + lane;
    const bool valid = e < r1;
    const int v = valid ? st[e - r0] : -1;
    const unsigned long long m = __ballot(v >= 0);
    const unsigned long long after = m >> lane;
    const int got = __shfl(v, after ? lane + (__ffsll((long long)after) - 1) : lane);
    const int val = after ? got : carry;
    if (valid) { t1_ptr[e] = val; t2_ptr[e] = val; }
    n_active += __popcll(m);
    carry = __shfl(val, 0);
  }
  if (lane == 0) cnt[j] = n_active;
}
// one workgroup: exclusive scan of the n counts in place (a contiguous chunk per thread), total -> *total
__global__ void __launch_bounds__(1024) k_canon_scan(int64_t n, int32_t* data, int32_t* total, const int32_t* __restrict__ flags) {
  __shared__ int s_wave[16];
  if (flags[0] & 7) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t chunk = (n + 1023) / 1024;
  const int64_t b = threadIdx.x * chunk, e = b + chunk < n ? b + chunk : n;
  int sum = 0;
  for (int64_t i = b; i < e; ++i) sum += data[i];
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += s_wave[w];
  int run = before + inc - sum;
  for (int64_t i = b; i < e; ++i) { const int v = data[i]; data[i] = run; run += v; }
  if (threadIdx.x == 1023) *total = before + inc;
}
// one wave per atom: compacted ids of its edges (k_active_scatter's arrays) and the window bounds its rows own (k_tb_windows)
__global__ void __launch_bounds__(256) k_canon_active(int64_t N, int64_t E, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ dst,
                                                      const int32_t* __restrict__ t1_ptr, const int32_t* __restrict__ arow_ptr,
                                                      const int32_t* __restrict__ flags, int32_t* act_scan, int32_t* act_id, int32_t* act_list,
                                                      int32_t* act_dst, int32_t* win) {
  if (flags[0] & 7) return;
  const int lane = threadIdx.x & 63;
  const int64_t j = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j > N) return;
  const int A = arow_ptr[N];
  if (j == N) {
    if (lane == 0) act_scan[E] = A;
    return;
  }
  const int r0 = row_ptr[j], r1 = row_ptr[j + 1], c0 = arow_ptr[j], c1 = arow_ptr[j + 1];
  int run = 0;
  for (int base = r0; base < r1; base += 64) {
    const int e = base + lane;
    const bool valid = e < r1;
    int ta = 0, tb = 0;
    if (valid) { ta = t1_ptr[e]; tb = t1_ptr[e + 1]; }
    const bool act = valid && tb > ta;
    const unsigned long long mask = __ballot(act);
    const int cid = c0 + run + __popcll(mask & ((1ull << lane) - 1ull));
    if (valid) {
      act_scan[e] = cid;
      act_id[e] = act ? cid : -1;
      if (act) {
        act_list[cid] = e;
        act_dst[cid] = dst[e];
        const int b = cid / kTbRows;
        if (cid % kTbRows == 0) { win[6 * b] = c0; win[6 * b + 2] = ta; win[6 * b + 4] = ta; }
        if (cid % kTbRows == kTbRows - 1 || cid == A - 1) { win[6 * b + 1] = c1; win[6 * b + 3] = tb; win[6 * b + 5] = tb; }
      }
    }
    run += __popcll(mask);
  }
}
// roles by workgroup: [0, e_blocks) in-edge pairs (k_pair_with_lookup); [e_blocks, e_blocks + t_blocks) compacted partners and their
// window bytes in both roles (k_compact_partners, k_partner_bytes); the last workgroup: which windows may use the moment path, and the
// statistics of the certificate (k_tb_fast for lists complete by construction, k_tb_stats)
__global__ void __launch_bounds__(256) k_canon_finish(int64_t E, int64_t T, int64_t e_blocks, int64_t t_blocks, int64_t windows,
                                                      const int64_t* __restrict__ tei, const int32_t* __restrict__ in_edge,
                                                      const int32_t* __restrict__ act_id, const int32_t* __restrict__ act_scan,
                                                      const int32_t* __restrict__ act_list, const int32_t* __restrict__ src,
                                                      const int32_t* __restrict__ win, int32_t* flags, int32_t* in_pair,
                                                      int32_t* in_pos, int32_t* t1_e2, int32_t* t2_e1, int32_t* t1_e2c, int32_t* t2_e1c, uint8_t* t1_b,
                                                      uint8_t* t2_b, int32_t* fast, int32_t* stats) {
  // (bit 3: the in-edge role found the edge list one-sided -- the general build sorts.  The triplet checks below raise their bits
  // while other workgroups of this launch run: whatever those write is discarded with the rest when the host reads the verdict)
  if (__builtin_nontemporal_load(flags) & 15) return;
  const int64_t blk = blockIdx.x;
  if (blk < e_blocks) {
    const int64_t i = blk * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const int32_t e = in_edge[i];
    const bool ok = e >= 0 && (int64_t)e < E;
    in_pair[2 * i] = ok ? e : 0;
    in_pair[2 * i + 1] = ok ? act_id[e] : -1;
    if (ok) in_pos[e] = (int32_t)i;
    return;
  }
  if (blk < e_blocks + t_blocks) {
    const int64_t t = (blk - e_blocks) * blockDim.x + threadIdx.x;
    if (t >= T) return;
    // the one pass over the triplet list: range / centre / order checks (k_convert_triplets), partner lists in both roles, then the
    // compacted partners and their window bytes
    int64_t e1 = tei[t], e2 = tei[T + t];
    int bad = 0;
    if (e1 < 0 || e1 >= E || e2 < 0 || e2 >= E) { bad |= 2; e1 = 0; e2 = 0; }
    else if (src[e1] != src[e2]) bad |= 4;
    if (bad) atomicOr(flags, bad);
    if (t > 0) {
      const int64_t p1 = tei[t - 1], p2 = tei[T + t - 1];
      if (p1 > tei[t] || (p1 == tei[t] && p2 > tei[T + t])) atomicOr(flags + 1, 1);
    }
    t1_e2[t] = (int32_t)e2;
    t2_e1[t] = (int32_t)e2;
    const int c2 = act_scan[e2];
    t1_e2c[t] = c2;
    t2_e1c[t] = c2;
    const int r = act_id[e1];
    const int b = (r < 0 ? 0 : r) / kTbRows;
    const int wlo = win[6 * b], whi = win[6 * b + 1];
    const int n = (whi - wlo) < kTbCap ? (whi - wlo) : kTbCap;
    const int d = c2 - wlo;
    const uint8_t byte = (uint8_t)((r >= 0 && d >= 0 && d < n && d < 255) ? d : 255);
    t1_b[t] = byte;
    t2_b[t] = byte;
    return;
  }
  __shared__ int s_bad[4], s_rows[4], s_atoms[4];
  const int A = flags[2];
  int bad = 0, rows = 0, atoms = 0;
  for (int64_t b = threadIdx.x; b < windows; b += blockDim.x) {
    int na = 0, a0 = 0;
    if (b * kTbRows < A) {
      const int lo = win[6 * b], hi = win[6 * b + 1];
      if (lo >= 0 && hi > lo && hi <= A) {
        a0 = src[act_list[lo]];
        na = src[act_list[hi - 1]] - a0 + 1;
        if (!(na <= kTbFastAtoms && hi - lo <= kTbCap)) na = 0;
        if (na > 0) { rows = max(rows, hi - lo); atoms = max(atoms, na); }
        else ++bad;
      } else {
        a0 = 0; ++bad;
      }
    }
    fast[2 * b] = na;
    fast[2 * b + 1] = a0;
  }
  for (int o = 32; o > 0; o >>= 1) { bad += __shfl_xor(bad, o); rows = max(rows, __shfl_xor(rows, o)); atoms = max(atoms, __shfl_xor(atoms, o)); }
  if ((threadIdx.x & 63) == 0) { s_bad[threadIdx.x >> 6] = bad; s_rows[threadIdx.x >> 6] = rows; s_atoms[threadIdx.x >> 6] = atoms; }
  __syncthreads();
  if (threadIdx.x == 0) {
    bad = 0; rows = 0; atoms = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { bad += s_bad[k]; rows = max(rows, s_rows[k]); atoms = max(atoms, s_atoms[k]); }
    stats[0] = bad; stats[1] = rows; stats[2] = atoms;
    // the certificate's word stays with the buffer (flags[7] = stats[3]; hints_word() on the host forms the same word from the
    // verdict): every check has passed when this line is reached, so no launch is needed after the host has read the verdict
    stats[3] = (bad == 0 && rows > 0) ? (M3G_TOPO_TB_COMPLETE | ((rows & 0xff) << 8) | ((atoms & 0xff) << 16)) : 0;
  }
}

static inline int bits_for(int64_t n) {
  int b = 1;
  while ((int64_t(1) << b) < n) ++b;
  return b;
}

}  // namespace m3g

using namespace m3g;

extern "C" int m3g_topology_bytes(int64_t N, int64_t E, int64_t T, int64_t S, size_t* bytes) {
  if (!bytes || N < 0 || E < 0 || T < 0 || S < 0) { set_error("m3g_topology_bytes: bad argument"); return M3G_ERR_VALUE; }
  if (N >= (int64_t(1) << 31) || E >= (int64_t(1) << 31) || T >= (int64_t(1) << 31)) {
    set_error("graph too large for int32 indices");
    return M3G_ERR_UNSUPPORTED;
  }
  *bytes = topo_carve(N, E, T, S, nullptr).total_bytes;
  return M3G_OK;
}

// The certificate kernels of m3g_topology_hints (row flags in the sort scratch of the buffer, which nothing reads after the build);
// false when the scratch is too small for the row flags (never for buffers sized by m3g_topology_bytes)
static bool launch_hint_kernels(const Topo& t, hipStream_t s, bool trusted = false) {
  const int64_t E = t.E, T = t.T;
  size_t m = (size_t)std::max<int64_t>(std::max<int64_t>(E, T), 1);
  char* tmp = (char*)t.sort_tmp;
  uint8_t* row_ok = (uint8_t*)(tmp + 2 * align_up(m * sizeof(uint64_t)));
  if (t.sort_tmp_bytes < 2 * align_up(m * sizeof(uint64_t)) + (size_t)E + 1) return false;
  const int TPB = 256;
  auto grid = [&](int64_t n) { return dim3((unsigned)((n + TPB - 1) / TPB)); };
  // trusted (m3g_topology_build_canonical): the lists come from this library's own builders, whose triplet lists hold every ordered
  // pair of a centre's active edges by construction -- no per-row completeness test, the windows only have to fit
  if (!trusted)
    hipLaunchKernelGGL(k_tb_row_complete, grid(E), dim3(TPB), 0, s, t.n_act, t.act_list, t.src, t.arow_ptr, t.t1_ptr, t.t1_e2c, t.t2_ptr, t.t2_e1c, row_ok,
                       t.flags + 4);
  hipLaunchKernelGGL(k_tb_fast, grid((E / kTbRows + 1) * 64), dim3(TPB), 0, s, E / kTbRows + 1, t.n_act, t.act_list, t.src, t.tb_win,
                     trusted ? (const uint8_t*)nullptr : row_ok, t.tb_fast, t.flags + 4);
  hipLaunchKernelGGL(k_tb_stats, dim3(1), dim3(1024), 0, s, E / kTbRows + 1, t.n_act, t.tb_win, t.tb_fast, t.flags + 4);
  return true;
}
static int32_t hints_word(const int32_t (&h)[3]) {
  // the window sizes travel in one byte each
  static_assert(kTbCap <= 255 && kTbFastAtoms <= 255, "m3g_topology_hints packs the largest window (rows, atoms) in 8 bits each");
  return (h[0] == 0 && h[1] > 0) ? (M3G_TOPO_TB_COMPLETE | ((h[1] & 0xff) << 8) | ((h[2] & 0xff) << 16)) : 0;
}

// host_hints != NULL: also form the certificate of m3g_topology_hints and return its word.  For the lists the graph builders emit
// (triplets sorted by (e1, e2) and symmetric, edge list symmetric) the WHOLE build -- rows, partner lists, active-edge compaction,
// windows, certificate -- is queued on that assumption and ONE read-back (malformed-graph bits, order / symmetry verdicts, the
// certificate's statistics) confirms it; any other list redoes the affected parts with the radix sorts (the round-3 path).
static int topology_build(int64_t N, int64_t E, int64_t T, int64_t S, const int64_t* edge_index,
                          const int64_t* triplet_edge_index, const int64_t* batch, void* topo_buf,
                          size_t topo_bytes, int32_t* host_flags, int32_t* host_hints, void* stream_, bool trusted) {
  hipStream_t s = (hipStream_t)stream_;
  size_t need = 0;
  int rc = m3g_topology_bytes(N, E, T, S, &need);
  if (rc) return rc;
  if (!topo_buf || topo_bytes < need) { set_error("topology buffer too small: %zu < %zu", topo_bytes, need); return M3G_ERR_SIZE; }
  if (host_hints) *host_hints = 0;
  Topo t = topo_carve(N, E, T, S, topo_buf);
  const int TPB = 256;
  auto grid = [&](int64_t n) { return dim3((unsigned)((n + TPB - 1) / TPB)); };
  M3G_HIP_CHECK(hipMemsetAsync(t.flags, 0, kTopoFlags * sizeof(int32_t), s));   // incl. the certified hints word [7] and the sticky error [8]
  // no window may use the three-body moment path until the certificate has been formed for THIS buffer (stale rows of an earlier
  // topology in the same memory must not survive a rebuild)
  M3G_HIP_CHECK(hipMemsetAsync(t.tb_fast, 0, sizeof(int32_t) * 2 * (E / kTbRows + 2), s));

  size_t m = (size_t)std::max<int64_t>(std::max<int64_t>(E, T), 1);
  char* tmp = (char*)t.sort_tmp;
  uint64_t* keysA = (uint64_t*)tmp;
  uint64_t* keysB = (uint64_t*)(tmp + align_up(m * sizeof(uint64_t)));
  void* cub_tmp = tmp + 2 * align_up(m * sizeof(uint64_t));

  if (N > 0) hipLaunchKernelGGL(k_convert_batch, grid(N), dim3(TPB), 0, s, N, S, batch, t.batch, t.flags);
  // incoming-edge lists: edges ordered by (neighbour atom, edge id).  A symmetric edge list (any full neighbour list) gets them
  // from the mirrors of each atom's own row in ONE kernel (k_in_edges_symmetric; it raises flags[0] bit 3 when the list is not
  // symmetric); otherwise -- and for graphs without triplets, which have no host read-back to learn the verdict from -- a STABLE
  // radix sort of the edge ids on the neighbour index (bits_for(N) key bits).
  int32_t* dst_sorted = (int32_t*)keysA;
  int32_t* edge_ids = (int32_t*)keysB;
  auto sort_in_edges = [&]() -> int {
    // (keys, values) ping-pong between the halves of the two key arrays: [dst copy | spare] and [edge ids | spare]
    int32_t *ka = dst_sorted, *kb = dst_sorted + E, *va = edge_ids, *vb = edge_ids + E;   // (each key array holds max(E, T) 8-byte words)
    hipLaunchKernelGGL(k_iota32, grid(E), dim3(TPB), 0, s, E, va);
    M3G_HIP_CHECK(hipMemcpyAsync(ka, t.dst, sizeof(int32_t) * E, hipMemcpyDeviceToDevice, s));
    const int where = prims::radix_sort<int32_t, int32_t>(ka, kb, va, vb, E, 0, bits_for(N + 1), cub_tmp, s);
    if (where < 0) { set_error("radix sort of the incoming edges failed: %s", hipGetErrorString(hipGetLastError())); return M3G_ERR_HIP; }
    M3G_HIP_CHECK(hipMemcpyAsync(t.in_edge, where ? vb : va, sizeof(int32_t) * E, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_lower_bound32, grid(N + 1), dim3(TPB), 0, s, N, E, where ? kb : ka, t.in_ptr);
    return M3G_OK;
  };
  if (E > 0) hipLaunchKernelGGL(k_convert_edges, grid(E), dim3(TPB), 0, s, N, E, edge_index, t.src, t.dst, edge_ids, t.flags);
  hipLaunchKernelGGL(k_lower_bound32, grid(N + 1), dim3(TPB), 0, s, N, E, t.src, t.row_ptr);
  hipLaunchKernelGGL(k_lower_bound32, grid(S + 1), dim3(TPB), 0, s, S, N, t.batch, t.struct_ptr);
  const bool try_mirrors = E > 0 && T > 0;
  if (try_mirrors) hipLaunchKernelGGL(k_in_edges_symmetric, grid((N + 1) * 64), dim3(TPB), 0, s, N, t.row_ptr, t.dst, t.in_ptr, t.in_edge, t.flags);
  else if (E > 0) { int rc2 = sort_in_edges(); if (rc2) return rc2; }
  else hipLaunchKernelGGL(k_lower_bound32, grid(N + 1), dim3(TPB), 0, s, N, E, t.dst, t.in_ptr);

  // everything downstream of the triplet rows: active-edge compaction (act_scan doubles as the flag array before the scan),
  // windows, compacted partner lists, byte-sized partner ids
  auto downstream = [&](bool symmetric, const uint64_t* t1_keys) -> int {
    hipLaunchKernelGGL(k_active_flags, grid(E + 1), dim3(TPB), 0, s, E, t.t1_ptr, t.t2_ptr, t.act_scan);
    M3G_HIP_CHECK(prims::exclusive_scan<int32_t>(t.act_scan, t.act_scan, E + 1, cub_tmp, s));
    hipLaunchKernelGGL(k_active_scatter, grid(std::max(E, N) + 1), dim3(TPB), 0, s, N, E, t.t1_ptr, t.t2_ptr, t.act_scan, t.row_ptr, t.dst, t.act_list,
                       t.act_dst, t.act_id, t.arow_ptr, t.n_act);
    hipLaunchKernelGGL(k_tb_windows, grid(E / kTbRows + 1), dim3(TPB), 0, s, E / kTbRows + 1, t.n_act, t.act_list, t.src, t.arow_ptr, t.t1_ptr, t.t2_ptr,
                       t.tb_win);
    if (E > 0) hipLaunchKernelGGL(k_pair_with_lookup, grid(E), dim3(TPB), 0, s, E, t.in_edge, t.act_id, t.in_pair, t.in_pos);
    if (T > 0) {
      hipLaunchKernelGGL(k_compact_partners, grid(T), dim3(TPB), 0, s, T, t.act_scan, t.t1_e2, t.t2_e1, t.t1_e2c, t.t2_e1c);
      if (symmetric) {   // one list serves both roles
        hipLaunchKernelGGL(k_partner_bytes<true>, grid(T), dim3(TPB), 0, s, E, T, t.t1_ptr, t1_keys, t.act_id, t.tb_win, t.t1_e2c, t.t1_b);
        M3G_HIP_CHECK(hipMemcpyAsync(t.t2_b, t.t1_b, (size_t)T, hipMemcpyDeviceToDevice, s));
      } else {
        hipLaunchKernelGGL(k_partner_bytes<false>, grid(T), dim3(TPB), 0, s, E, T, t.t1_ptr, nullptr, t.act_id, t.tb_win, t.t1_e2c, t.t1_b);
        hipLaunchKernelGGL(k_partner_bytes<false>, grid(T), dim3(TPB), 0, s, E, T, t.t2_ptr, nullptr, t.act_id, t.tb_win, t.t2_e1c, t.t2_b);
      }
    }
    return M3G_OK;
  };
  auto mirror_lists = [&]() -> int {   // symmetric triplet list: the partners of e as second edge are its partners as first edge
    M3G_HIP_CHECK(hipMemcpyAsync(t.t2_ptr, t.t1_ptr, sizeof(int32_t) * (E + 1), hipMemcpyDeviceToDevice, s));
    M3G_HIP_CHECK(hipMemcpyAsync(t.t2_e1, t.t1_e2, sizeof(int32_t) * T, hipMemcpyDeviceToDevice, s));
    return M3G_OK;
  };

  // Triplet lists grouped by first edge (t1) and by second edge (t2), each in canonical (sorted) order.  The list the graph
  // builders emit (compute_threebody's order, data/material_graph.py:239-248) is already sorted by (e1, e2) and symmetric
  // (every ordered pair of a centre's edges): then t1 needs no sort and t2 IS t1.  Both properties are checked on the device.
  int32_t* order = t.flags + 1;   // flags[1] (otherwise unused): bit 0 "not sorted", bit 1 "not symmetric"
  bool flags_read = false;
  int32_t h[2] = {0, 0};               // flags[0] (malformed graph), flags[1] (order)
  if (T > 0) {
    // optimistic pass: rows and partners as if the list were sorted and symmetric, the mirror check on it, and -- on the same
    // assumption -- everything downstream and the certificate; ONE host read-back decides
    hipLaunchKernelGGL(k_convert_triplets, grid(T), dim3(TPB), 0, s, E, T, triplet_edge_index, t.src, keysA, 0, t.flags, order, t.t1_e2);
    hipLaunchKernelGGL(k_lower_bound64, grid(E + 1), dim3(TPB), 0, s, E, T, keysA, t.t1_ptr);
    if (!trusted) hipLaunchKernelGGL(k_check_symmetric, grid(T), dim3(TPB), 0, s, T, keysA, t.t1_ptr, order);
    { int r = mirror_lists(); if (r) return r; }
    { int r = downstream(true, keysA); if (r) return r; }
    const bool hinted = host_hints && E > 0 && launch_hint_kernels(t, s, trusted);
    int32_t hs[3] = {0, 0, 0}, h7[7] = {0, 0, 0, 0, 0, 0, 0};
    // flags[0..1] and the hint words flags[4..6] in ONE copy (a second small copy to pageable memory costs ~30 us of host time)
    M3G_HIP_CHECK(hipMemcpyAsync(h7, t.flags, (hinted ? 7 : 2) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    M3G_HIP_CHECK(hipStreamSynchronize(s));
    h[0] = h7[0]; h[1] = h7[1]; hs[0] = h7[4]; hs[1] = h7[5]; hs[2] = h7[6];
    flags_read = true;   // every kernel that can flag a malformed graph has run
    const bool assumed_ok = !(h[1] & 3) && !(try_mirrors && (h[0] & 8));
    if (assumed_ok) {
      if (hinted) {
        *host_hints = hints_word(hs);
        // the same word stays with the buffer (flags[7]): the moment kernels run only when the word the caller hands to
        // m3g_energy_forces is the one certified for THIS topology buffer
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, s, t.flags + 7, *host_hints);
      }
    } else {
      // ---- the exception: redo what the assumption got wrong, with the radix sorts (round 3's path) ----
      if (try_mirrors && (h[0] & 8)) {   // not a symmetric edge list (or very long rows): the stable sort after all
        int rc2 = sort_in_edges();       // (clobbers keysA / keysB: the triplet keys are formed again below)
        if (rc2) return rc2;
      }
      M3G_HIP_CHECK(hipMemsetAsync(order, 0, sizeof(int32_t), s));
      hipLaunchKernelGGL(k_convert_triplets, grid(T), dim3(TPB), 0, s, E, T, triplet_edge_index, t.src, keysA, 0, t.flags, order);
      uint64_t* sorted = keysA;
      if (h[1] & 1) {      // not sorted: radix sort, then rows, partners and the mirror check again
        const int where = prims::radix_sort<uint64_t, int32_t>(keysA, keysB, nullptr, nullptr, T, 0, 32 + bits_for(E + 1), cub_tmp, s);
        if (where < 0) { set_error("radix sort of the triplet keys failed: %s", hipGetErrorString(hipGetLastError())); return M3G_ERR_HIP; }
        sorted = where ? keysB : keysA;
      }
      hipLaunchKernelGGL(k_low_word, grid(T), dim3(TPB), 0, s, T, sorted, t.t1_e2);
      hipLaunchKernelGGL(k_lower_bound64, grid(E + 1), dim3(TPB), 0, s, E, T, sorted, t.t1_ptr);
      bool symmetric = !(h[1] & 2);
      if (h[1] & 1) {      // the mirror check of the optimistic pass ran on unsorted keys: its verdict means nothing
        M3G_HIP_CHECK(hipMemsetAsync(order, 0, sizeof(int32_t), s));
        hipLaunchKernelGGL(k_check_symmetric, grid(T), dim3(TPB), 0, s, T, sorted, t.t1_ptr, order);
        int32_t o2 = 0;
        M3G_HIP_CHECK(hipMemcpyAsync(&o2, order, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        M3G_HIP_CHECK(hipStreamSynchronize(s));
        symmetric = !(o2 & 2);
      }
      const uint64_t* t1_keys = nullptr;
      if (symmetric) {
        int r = mirror_lists();
        if (r) return r;
        t1_keys = sorted;
      } else {
        uint64_t* other = sorted == keysA ? keysB : keysA;   // the t1 keys are no longer needed
        hipLaunchKernelGGL(k_convert_triplets, grid(T), dim3(TPB), 0, s, E, T, triplet_edge_index, t.src, other, 1, t.flags, order);
        uint64_t* spare = other == keysA ? keysB : keysA;
        const int where = prims::radix_sort<uint64_t, int32_t>(other, spare, nullptr, nullptr, T, 0, 32 + bits_for(E + 1), cub_tmp, s);
        if (where < 0) { set_error("radix sort of the triplet keys failed: %s", hipGetErrorString(hipGetLastError())); return M3G_ERR_HIP; }
        uint64_t* out = where ? spare : other;
        hipLaunchKernelGGL(k_low_word, grid(T), dim3(TPB), 0, s, T, out, t.t2_e1);
        hipLaunchKernelGGL(k_lower_bound64, grid(E + 1), dim3(TPB), 0, s, E, T, out, t.t2_ptr);
      }
      int r = downstream(symmetric, t1_keys);
      if (r) return r;
      // the optimistic certificate described other lists: none is valid for this buffer until m3g_topology_hints forms it
      M3G_HIP_CHECK(hipMemsetAsync(t.tb_fast, 0, sizeof(int32_t) * 2 * (E / kTbRows + 2), s));
      M3G_HIP_CHECK(hipMemsetAsync(t.flags + 4, 0, 4 * sizeof(int32_t), s));
      if (host_hints) {
        M3G_HIP_CHECK(hipGetLastError());
        int rh = m3g_topology_hints(N, E, T, S, topo_buf, host_hints, stream_);
        if (rh) return rh;
      }
    }
    M3G_HIP_CHECK(hipMemsetAsync(order, 0, sizeof(int32_t), s));
    h[0] &= ~8;
  } else {
    M3G_HIP_CHECK(hipMemsetAsync(t.t1_ptr, 0, sizeof(int32_t) * (E + 1), s));
    M3G_HIP_CHECK(hipMemsetAsync(t.t2_ptr, 0, sizeof(int32_t) * (E + 1), s));
    int r = downstream(false, nullptr);
    if (r) return r;
  }
  M3G_HIP_CHECK(hipGetLastError());
  if (host_flags) {
    if (!flags_read) {   // no triplets: nothing above waited for the device
      M3G_HIP_CHECK(hipMemcpyAsync(h, t.flags, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      M3G_HIP_CHECK(hipStreamSynchronize(s));
    }
    host_flags[0] = h[0];
  }
  return M3G_OK;
}

extern "C" int m3g_topology_build_hints(int64_t N, int64_t E, int64_t T, int64_t S, const int64_t* edge_index,
                                        const int64_t* triplet_edge_index, const int64_t* batch, void* topo_buf,
                                        size_t topo_bytes, int32_t* host_flags, int32_t* host_hints, void* stream_) {
  return topology_build(N, E, T, S, edge_index, triplet_edge_index, batch, topo_buf, topo_bytes, host_flags, host_hints, stream_, false);
}
// For lists this library's own builders have just written (m3g_neighbor_fill / m3g_verlet_fill* + m3g_threebody_*): symmetric
// triplet lists that hold every ordered pair of a centre's edges inside the three-body cutoff BY CONSTRUCTION, so the mirror check of
// the triplet list and the per-row completeness test of the certificate are skipped (two kernels over all triplets, 38 us of the
// 0.25-ms build on the 10k-atom cell).  Everything else -- index ranges, row order, edge-list symmetry -- is still checked.
// Six launches (a memset and five kernels) and the copy of the verdict words (flags[0..6]) to `verdict`, PINNED host memory: queued, not waited for
constexpr int64_t kCanonMaxAtoms = 131072;   // k_canon_scan is one workgroup
static thread_local int32_t g_last_canonical_path = 0;
static bool canonical_fast_applies(int64_t N, int64_t E, int64_t T, int64_t S, const void* ei, const void* tei, const void* batch) {
  static const bool fast_off = [] { const char* v = getenv("M3G_CANON_FAST"); return v && v[0] == '0'; }();   // A/B measurements
  return !fast_off && N > 0 && N <= kCanonMaxAtoms && E > 0 && T > 0 && S > 0 && ei && tei && batch;
}
static int canonical_fast_launch(int64_t N, int64_t E, int64_t T, int64_t S, const int64_t* edge_index, const int64_t* triplet_edge_index,
                                 const int64_t* batch, const Topo& t, int32_t* verdict, hipStream_t s) {
  const int TPB = 256;
  auto blocks = [&](int64_t n) { return (n + TPB - 1) / TPB; };
  const int64_t windows = E / kTbRows + 1, n_win = 6 * windows, n_fast = 2 * (E / kTbRows + 2);
  M3G_HIP_CHECK(hipMemsetAsync(t.flags, 0, kTopoFlags * sizeof(int32_t), s));
  const int64_t n1 = std::max(std::max(E, N + 1), std::max(S + 1, n_win));
  hipLaunchKernelGGL(k_canon_edges, dim3((unsigned)blocks(n1)), dim3(TPB), 0, s, N, E, S, edge_index, batch, t.src, t.dst, t.batch, t.row_ptr,
                     t.struct_ptr, t.tb_win, n_win, t.tb_fast, n_fast, t.flags);
  const int64_t t_blocks = blocks(T), a_blocks = blocks((N + 1) * 64), e_blocks = blocks(E);
  hipLaunchKernelGGL(k_canon_rows, dim3((unsigned)(2 * a_blocks)), dim3(TPB), 0, s, N, E, T, a_blocks, triplet_edge_index, t.row_ptr, t.dst, t.flags,
                     t.t1_ptr, t.t2_ptr, t.arow_ptr, t.in_ptr, t.in_edge);
  hipLaunchKernelGGL(k_canon_scan, dim3(1), dim3(1024), 0, s, N + 1, t.arow_ptr, t.n_act, t.flags);
  hipLaunchKernelGGL(k_canon_active, dim3((unsigned)a_blocks), dim3(TPB), 0, s, N, E, t.row_ptr, t.dst, t.t1_ptr, t.arow_ptr, t.flags, t.act_scan, t.act_id,
                     t.act_list, t.act_dst, t.tb_win);
  hipLaunchKernelGGL(k_canon_finish, dim3((unsigned)(e_blocks + t_blocks + 1)), dim3(TPB), 0, s, E, T, e_blocks, t_blocks, windows, triplet_edge_index,
                     t.in_edge, t.act_id, t.act_scan, t.act_list, t.src, t.tb_win, t.flags, t.in_pair, t.in_pos, t.t1_e2, t.t2_e1, t.t1_e2c, t.t2_e1c,
                     t.t1_b, t.t2_b, t.tb_fast, t.flags + 4);
  M3G_HIP_CHECK(hipMemcpyAsync(verdict, t.flags, 7 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}
// after the stream has been waited for: true = every check passed, the certificate's word is on the buffer and in *host_hints
static int canonical_fast_settle(const Topo& t, const int32_t* verdict, int32_t* host_flags, int32_t* host_hints, hipStream_t s, bool* done) {
  *done = false;
  if ((verdict[0] & 15) || (verdict[1] & 1)) return M3G_OK;   // malformed, not symmetric or not sorted: the general path decides what it is
  const int32_t hs[3] = {verdict[4], verdict[5], verdict[6]};
  *host_hints = hints_word(hs);   // (k_canon_finish has left the same word on the buffer, flags[7])
  (void)s;
  if (host_flags) host_flags[0] = verdict[0];
  *done = true;
  return M3G_OK;
}
static int canonical_check_buffer(int64_t N, int64_t E, int64_t T, int64_t S, void* topo_buf, size_t topo_bytes) {
  size_t need = 0;
  int rc = m3g_topology_bytes(N, E, T, S, &need);
  if (rc) return rc;
  if (!topo_buf || topo_bytes < need) { set_error("topology buffer too small: %zu < %zu", topo_bytes, need); return M3G_ERR_SIZE; }
  return M3G_OK;
}

extern "C" int m3g_topology_build_canonical(int64_t N, int64_t E, int64_t T, int64_t S, const int64_t* edge_index,
                                            const int64_t* triplet_edge_index, const int64_t* batch, void* topo_buf,
                                            size_t topo_bytes, int32_t* host_flags, int32_t* host_hints, void* stream_) {
  g_last_canonical_path = 0;
  if (host_hints && canonical_fast_applies(N, E, T, S, edge_index, triplet_edge_index, batch)) {
    int rc = canonical_check_buffer(N, E, T, S, topo_buf, topo_bytes);
    if (rc) return rc;
    *host_hints = 0;
    // the verdict lands in pinned host memory (one 64-byte block per host thread, kept for the life of the process): a copy to
    // pageable memory goes through the runtime's staging path and cost ~45 us of host time between the copy and the return
    static thread_local int32_t* pinned = nullptr;
    if (!pinned) M3G_HIP_CHECK(hipHostMalloc((void**)&pinned, 16 * sizeof(int32_t), hipHostMallocDefault));
    hipStream_t s = (hipStream_t)stream_;
    const Topo t = topo_carve(N, E, T, S, topo_buf);
    rc = canonical_fast_launch(N, E, T, S, edge_index, triplet_edge_index, batch, t, pinned, s);
    if (rc) return rc;
    M3G_HIP_CHECK(hipStreamSynchronize(s));
    bool done = false;
    rc = canonical_fast_settle(t, pinned, host_flags, host_hints, s, &done);
    g_last_canonical_path = done ? 1 : 0;
    if (rc || done) return rc;
  }
  return topology_build(N, E, T, S, edge_index, triplet_edge_index, batch, topo_buf, topo_bytes, host_flags, host_hints, stream_, true);
}

// The same build in two calls, so that the host prepares its next call (m3g_energy_forces: workspace, outputs) while the device builds:
// _begin queues the launches and the copy of the verdict to `pinned_verdict` (>= 8 int32 of PINNED host memory the caller keeps
// untouched until _end; word 7 says whether anything was queued), _end waits for the stream and certifies the buffer -- or, when a
// check failed or _begin did not apply (no triplets, very many atoms), runs m3g_topology_build_canonical's general path there and then.
extern "C" int m3g_topology_build_canonical_begin(int64_t N, int64_t E, int64_t T, int64_t S, const int64_t* edge_index,
                                                  const int64_t* triplet_edge_index, const int64_t* batch, void* topo_buf,
                                                  size_t topo_bytes, int32_t* pinned_verdict, void* stream_) {
  if (!pinned_verdict) { set_error("m3g_topology_build_canonical_begin: null verdict buffer"); return M3G_ERR_VALUE; }
  pinned_verdict[7] = 0;
  int rc = canonical_check_buffer(N, E, T, S, topo_buf, topo_bytes);
  if (rc) return rc;
  if (!canonical_fast_applies(N, E, T, S, edge_index, triplet_edge_index, batch)) return M3G_OK;
  rc = canonical_fast_launch(N, E, T, S, edge_index, triplet_edge_index, batch, topo_carve(N, E, T, S, topo_buf), pinned_verdict, (hipStream_t)stream_);
  if (rc) return rc;
  pinned_verdict[7] = 1;
  return M3G_OK;
}
extern "C" int m3g_topology_build_canonical_end(int64_t N, int64_t E, int64_t T, int64_t S, const int64_t* edge_index,
                                                const int64_t* triplet_edge_index, const int64_t* batch, void* topo_buf,
                                                size_t topo_bytes, const int32_t* pinned_verdict, int32_t* host_flags, int32_t* host_hints,
                                                void* stream_) {
  if (!pinned_verdict || !host_hints) { set_error("m3g_topology_build_canonical_end: null argument"); return M3G_ERR_VALUE; }
  g_last_canonical_path = 0;
  *host_hints = 0;
  if (pinned_verdict[7] == 1) {
    hipStream_t s = (hipStream_t)stream_;
    M3G_HIP_CHECK(hipStreamSynchronize(s));
    bool done = false;
    int rc = canonical_fast_settle(topo_carve(N, E, T, S, topo_buf), pinned_verdict, host_flags, host_hints, s, &done);
    g_last_canonical_path = done ? 1 : 0;
    if (rc || done) return rc;
  }
  return topology_build(N, E, T, S, edge_index, triplet_edge_index, batch, topo_buf, topo_bytes, host_flags, host_hints, stream_, true);
}

extern "C" int m3g_topology_build(int64_t N, int64_t E, int64_t T, int64_t S, const int64_t* edge_index,
                                  const int64_t* triplet_edge_index, const int64_t* batch, void* topo_buf,
                                  size_t topo_bytes, int32_t* host_flags, void* stream_) {
  return m3g_topology_build_hints(N, E, T, S, edge_index, triplet_edge_index, batch, topo_buf, topo_bytes, host_flags, nullptr, stream_);
}

extern "C" int m3g_topology_hints(int64_t N, int64_t E, int64_t T, int64_t S, const void* topo_buf, int32_t* host_hints, void* stream_) {
  if (!topo_buf || !host_hints) { set_error("m3g_topology_hints: null argument"); return M3G_ERR_VALUE; }
  *host_hints = 0;
  if (T <= 0 || E <= 0) return M3G_OK;   // no triplets: the three-body kernels never run
  hipStream_t s = (hipStream_t)stream_;
  Topo t = topo_carve(N, E, T, S, const_cast<void*>(topo_buf));
  // The certificate for the three-body moment kernels, formed on demand (a topology that is used once does not pay for it): per
  // compacted row, are its partner lists complete; per 128-row window, are all its rows, and how large do the windows get.
  if (!launch_hint_kernels(t, s)) return M3G_OK;
  int32_t h[3] = {0, 0, 0};
  M3G_HIP_CHECK(hipMemcpyAsync(h, t.flags + 4, sizeof(h), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipStreamSynchronize(s));
  *host_hints = hints_word(h);
  // the same word stays with the buffer (flags[7]): the moment kernels run only when the word the caller hands to m3g_energy_forces
  // is the one certified for THIS topology buffer -- a stale word, or one copied from another buffer, flags an error instead
  hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, s, t.flags + 7, *host_hints);
  return M3G_OK;
}

extern "C" int m3g_topology_data_bytes(int64_t N, int64_t E, int64_t T, int64_t S, size_t* bytes) {
  size_t total = 0;
  int rc = m3g_topology_bytes(N, E, T, S, &total);
  if (rc) return rc;
  const Topo t = topo_carve(N, E, T, S, nullptr);
  *bytes = t.total_bytes - align_up(t.sort_tmp_bytes);
  return M3G_OK;
}
extern "C" int m3g_topology_debug_last_path(int32_t* path) {
  if (!path) { set_error("m3g_topology_debug_last_path: null argument"); return M3G_ERR_VALUE; }
  *path = g_last_canonical_path;
  return M3G_OK;
}

extern "C" int m3g_topology_status(int64_t N, int64_t E, int64_t T, int64_t S, const void* topo_buf, int32_t* host_status, void* stream_) {
  if (!topo_buf || !host_status) { set_error("m3g_topology_status: null argument"); return M3G_ERR_VALUE; }
  hipStream_t s = (hipStream_t)stream_;
  Topo t = topo_carve(N, E, T, S, const_cast<void*>(topo_buf));
  int32_t v = 0;
  M3G_HIP_CHECK(hipMemcpyAsync(&v, t.flags + 8, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipStreamSynchronize(s));
  *host_status = v;
  return M3G_OK;
}

extern "C" int m3g_topology_active_edges(int64_t N, int64_t E, int64_t T, int64_t S, const void* topo_buf, int64_t* host_count, void* stream_) {
  if (!topo_buf || !host_count) { set_error("m3g_topology_active_edges: null argument"); return M3G_ERR_VALUE; }
  hipStream_t s = (hipStream_t)stream_;
  Topo t = topo_carve(N, E, T, S, const_cast<void*>(topo_buf));
  int32_t a = 0;
  M3G_HIP_CHECK(hipMemcpyAsync(&a, t.n_act, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipStreamSynchronize(s));
  *host_count = a;
  return M3G_OK;
}
