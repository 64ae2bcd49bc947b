// Node reverse (B2) as a device function: k_node_reverse (m3g_node.hip) runs it on its own, k_node_tb_reverse
// (m3g_threebody.hip) runs it as one of two workgroup roles of a launch beside the three-body reverse.
#pragma once
#include "m3g_internal.h"
#include "m3g_device.h"
#include "m3g_mfma_common.h"

namespace m3g {

// ---- B2: dx_in[i] = dx_new[i] + (sum_{row(i)} dp1) W1a + (sum_{in(i)} dp1) W1b + (dv v(1-v)) W1 ----
// kNodesRev atoms per workgroup: phase 1 streams the dp1 rows (HBM-bound gather: a wave reads a whole 1-KB row per
// instruction, 16 B per lane), phase 2 applies the transposed first-layer weights once for all atoms of the group
// (the 128 KB of W1a/W1b would otherwise be re-read from L2 for every atom).
#ifndef M3G_NODES_REV
#define M3G_NODES_REV 4   // measured: 4 -> 0.259, 8 -> 0.288, 16 -> 0.293 ms per step (one atom per wave keeps more independent gathers in flight)
#endif
constexpr int kNodesRev = M3G_NODES_REV;
#ifndef M3G_NR_BATCH
#define M3G_NR_BATCH 8   // 768-byte nontemporal rows, index pairs handed out by v_readlane: 4 -> 0.172, 8 -> 0.166, 12 -> 0.171 ms per step
#endif
constexpr int kNrBatch = M3G_NR_BATCH;   // rows in flight per wave in the dp1 gather (multiple of 4)
#ifndef M3G_NR_BATCH_SMALL
#define M3G_NR_BATCH_SMALL 16   // ... of the small-system instantiation (PRELOAD): the launch is a chain of round trips there, not bandwidth
#endif
constexpr int kNrBatchSmall = M3G_NR_BATCH_SMALL;
struct NodeRevArgs {
  int C;
  int64_t N;
  const float* W;
  BlockW bw;
  const int32_t *row_ptr, *in_ptr, *in_edge;
  const float *dp1, *dgq, *v, *dx_new;
  float* dx_out;
  const float *seg_head, *seg_first;
  int with_v_term;
  const int2* in_pair;
  int dp1_packed;
  const float* dp1_scale;
  int dp1_by_dst;   // fp32 rows stored by position in the by-neighbour list (RevArgs::in_pos): the rows of an atom are consecutive
};
// DEFER_V: the dL/dg rows (dgq) are produced by OTHER workgroups of the same launch (the three-body reverse role of
// k_node_tb_reverse, m3g_threebody.hip): the dp1 gather runs first, `wait_for_dgq()` then blocks until those rows are visible, and
// the v-gradient terms are gathered in a second pass over the same in-edge list -- in the order the one-pass form adds them
// (batches of kNrBatch, pairwise inside a batch), so dx_out is bit-identical either way.
// PRELOAD (small systems): the 128 weight values a thread needs in phase 2 (its column of the W1a^T / W1b^T quarter) are requested
// at kernel entry and arrive during the gather of phase 1 -- phase 2 is otherwise a chain of 16 dependent L2 round trips per
// thread (0.3 us each), most of what a 32-atom launch costs.  128 more registers per lane: not for the large-system gather, which
// lives on the waves it can keep resident.
template <bool DEFER_V, bool PRELOAD = false, class WAIT>
__device__ __forceinline__ void node_reverse_body(const NodeRevArgs& args, int64_t vblock, WAIT wait_for_dgq) {
  // rows in flight per wave.  The sums do not depend on it: row r of an atom's list goes to accumulator r mod 4 and the dL/dg terms are
  // added four at a time in list order whatever the batch length (padding rows add +0)
  constexpr int NB = PRELOAD ? kNrBatchSmall : kNrBatch;
  const int C = args.C;
  const int64_t N = args.N;
  const float* __restrict__ W = args.W;
  const BlockW& bw = args.bw;
  const int32_t* __restrict__ row_ptr = args.row_ptr;
  const int32_t* __restrict__ in_ptr = args.in_ptr;
  const float* __restrict__ dp1 = args.dp1;
  const float* dgq = args.dgq;
  const float* __restrict__ v = args.v;
  const float* __restrict__ dx_new = args.dx_new;
  float* __restrict__ dx_out = args.dx_out;
  const float* __restrict__ seg_head = args.seg_head;
  const float* __restrict__ seg_first = args.seg_first;
  const int with_v_term = DEFER_V ? 0 : args.with_v_term;   // (deferred: the gather below takes no dL/dg rows)
  const int2* __restrict__ in_pair = args.in_pair;
  const int dp1_packed = args.dp1_packed;
  const float* __restrict__ dp1_scale = args.dp1_scale;
  __shared__ float4 sA[kNodesRev][64], sB[kNodesRev][64];   // row / in-edge sums of dp1, 256 columns as 64 float4
  __shared__ float tv[kNodesRev][kCP];
  __shared__ float part[4][kNodesRev][kDP];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int64_t n0 = vblock * kNodesRev;
  float pre_a[PRELOAD ? kDP : 1], pre_b[PRELOAD ? kDP : 1];
  if constexpr (PRELOAD) {
    const int k = tid & 63, pq = tid >> 6;
    const MlpW& mw = pq < 2 ? bw.e : bw.n;
    const int row0 = (pq & 1) * kDP;
    const float* wa = W + mw.w1a + (size_t)row0 * kDP + k;
    const float* wb = W + mw.w1b + (size_t)row0 * kDP + k;
#pragma unroll
    for (int o = 0; o < kDP; ++o) { pre_a[o] = wa[o * kDP]; pre_b[o] = wb[o * kDP]; }
  }
  const float4* rows = reinterpret_cast<const float4*>(dp1) + ln;
  // phase 1: wave wv gathers for atoms wv, wv+4 of the group
  for (int nb = wv; nb < kNodesRev; nb += 4) {
    const int64_t i = n0 + nb;
    float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
    float dv = 0.f, dvv = 0.f;
    if (i < N) {
      const int e1 = row_ptr[i + 1];
      int e = row_ptr[i];
      // small systems: the in-edge list's bounds and its first 64 (edge, three-body row) pairs are requested BEFORE the partial-row
      // sums below are waited for -- two dependent round trips less in a launch that is a chain of them (at 10,000 atoms, where the
      // gather is bandwidth, it changes nothing: 0.2035 vs 0.2042 ms per step)
      int k = 0, k1 = 0;
      int2 mine_first = make_int2(-1, -1);
      if constexpr (PRELOAD) {
        k1 = in_ptr[i + 1];
        k = in_ptr[i];
        const int ks0 = __builtin_amdgcn_readfirstlane(k), k1s0 = __builtin_amdgcn_readfirstlane(k1);
        if (ks0 + ln < k1s0) mine_first = in_pair[ks0 + ln];
      }
      if (seg_head) {
        // the fused reverse kernel already summed the rows of each centre inside its tiles: add the partial rows
        // (run starting mid-tile + first runs of the tiles whose column 0 belongs to this centre)
        if (e1 > e) {
          if (e & 15) a0 = reinterpret_cast<const float4*>(seg_first)[i * 64 + ln];
          for (int t = (e + 15) >> 4; t <= (e1 - 1) >> 4; ++t) {
            const float4 u = reinterpret_cast<const float4*>(seg_head)[(int64_t)t * 64 + ln];
            a1.x += u.x; a1.y += u.y; a1.z += u.z; a1.w += u.w;
          }
        }
        e = e1;
      }
      for (; e + 1 < e1; e += 2) {
        const float4 u = rows[(int64_t)e * 64], w2 = rows[(int64_t)(e + 1) * 64];
        a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
        a1.x += w2.x; a1.y += w2.y; a1.z += w2.z; a1.w += w2.w;
      }
      if (e < e1) {
        const float4 u = rows[(int64_t)e * 64];
        a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
      }
      // in-edge rows: 8 whole 1-KB rows in flight per wave (random rows: latency-bound unless enough bytes are in
      // flight); lanes 0-15 also pick up the matching dL/dg row elements for the v-gradient (same edge list)
      if constexpr (!PRELOAD) {
        k1 = in_ptr[i + 1];
        k = in_ptr[i];
      }
      float4 b2 = make_float4(0.f, 0.f, 0.f, 0.f), b3 = b2;
      const int cq = ln & 15;
      // NB whole 1-KB rows in flight per wave, the remainder in one guarded batch as well (a row-at-a-time tail
      // is a dependent round trip per row)
#ifndef M3G_NR_NO_CHUNK
      // the (edge, three-body row) pairs of up to 64 in-edges arrive in ONE coalesced load, a lane each, and are handed
      // out by v_readlane: a pair load per batch would put a dependent round trip in front of every batch of row loads
      const int ks = __builtin_amdgcn_readfirstlane(k), k1s = __builtin_amdgcn_readfirstlane(k1);
      for (int kc = ks; kc < k1s; kc += 64) {
        const int cnt = k1s - kc < 64 ? k1s - kc : 64;
        const int2 mine = (PRELOAD && kc == ks) ? mine_first : (ln < cnt ? in_pair[kc + ln] : make_int2(-1, -1));
      for (int b = 0; b < cnt; b += NB) {
        int2 f[NB];   // (edge id, compact three-body row or -1)
        float4 u[NB];
        float g[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int src = b + j < 64 ? b + j : 63;   // lanes >= cnt hold (-1, -1)
          f[j].x = b + j < 64 ? __builtin_amdgcn_readlane(mine.x, src) : -1;
          f[j].y = b + j < 64 ? __builtin_amdgcn_readlane(mine.y, src) : -1;
          if (args.dp1_by_dst && f[j].x >= 0) f[j].x = kc + b + j;   // rows stored in list order: a stream, not a gather
        }
#else
      for (; k < k1; k += NB) {
        int2 f[NB];   // (edge id, compact three-body row or -1)
        float4 u[NB];
        float g[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) f[j] = k + j < k1 ? in_pair[k + j] : make_int2(-1, -1);
#endif
#ifndef M3G_DP1_F32
        if (dp1_packed == kDp1Fixed) {   // rows of the fused f16x3 reverse kernel: 24-bit fixed point + a scale per 64 columns (pack24_fixed)
          u32x3 pk[NB];
          float sc[NB];
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            pk[j] = f[j].x >= 0 ? __builtin_nontemporal_load(reinterpret_cast<const u32x3_a4*>(reinterpret_cast<const unsigned*>(dp1) +
                                                                   (int64_t)f[j].x * kDp1PackedDwords + 3 * ln))
                                : u32x3{0u, 0u, 0u};   // (any bytes decode to finite numbers; the zero scale makes them 0)
            sc[j] = f[j].x >= 0 ? dp1_scale[(int64_t)f[j].x * 4 + (ln >> 4)] : 0.f;
          }
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const f32x4 t = unpack24_fixed(pk[j], sc[j]);
            u[j] = make_float4(t[0], t[1], t[2], t[3]);
            g[j] = (with_v_term && f[j].y >= 0) ? dgq[(int64_t)f[j].y * kCP + cq] : 0.f;
          }
        } else if (dp1_packed) {   // rows written by the fused bf16x3 reverse kernel: 24-bit values, 12 B per lane (m3g_mfma_common.h: pack24)
          u32x3 pk[NB];
#pragma unroll
          for (int j = 0; j < NB; ++j)
            // nontemporal: every row is read exactly once, and keeping it out of L2 leaves the cache to the weights and
            // partial rows (node reverse 0.218 -> 0.187 ms per step)
            pk[j] = f[j].x >= 0 ? __builtin_nontemporal_load(reinterpret_cast<const u32x3_a4*>(reinterpret_cast<const unsigned*>(dp1) +
                                                                   (int64_t)f[j].x * kDp1PackedDwords + 3 * ln))
                                : u32x3{0u, 0u, 0u};
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const f32x4 t = unpack24(pk[j]);
            u[j] = make_float4(t[0], t[1], t[2], t[3]);
            g[j] = (with_v_term && f[j].y >= 0) ? dgq[(int64_t)f[j].y * kCP + cq] : 0.f;
          }
        } else
#endif
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          // fp32 rows (fp32 mode, split reverse kernels): read once -> nontemporal, like the packed rows
          if (f[j].x >= 0) {
            const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dp1) + (int64_t)f[j].x * 64 + ln);
            u[j] = make_float4(t[0], t[1], t[2], t[3]);
          } else {
            u[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          // dL/dg holds one row per ACTIVE edge; other edges contribute nothing
          g[j] = (with_v_term && f[j].y >= 0) ? dgq[(int64_t)f[j].y * kCP + cq] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NB; j += 4) {
          b0.x += u[j].x; b0.y += u[j].y; b0.z += u[j].z; b0.w += u[j].w;
          b1.x += u[j + 1].x; b1.y += u[j + 1].y; b1.z += u[j + 1].z; b1.w += u[j + 1].w;
          b2.x += u[j + 2].x; b2.y += u[j + 2].y; b2.z += u[j + 2].z; b2.w += u[j + 2].w;
          b3.x += u[j + 3].x; b3.y += u[j + 3].y; b3.z += u[j + 3].z; b3.w += u[j + 3].w;
          dv += (g[j] + g[j + 1]) + (g[j + 2] + g[j + 3]);
        }
      }
#ifndef M3G_NR_NO_CHUNK
      }
#endif
      b0.x += b2.x; b0.y += b2.y; b0.z += b2.z; b0.w += b2.w;
      b1.x += b3.x; b1.y += b3.y; b1.z += b3.z; b1.w += b3.w;
#ifndef M3G_NR_NO_CHUNK
      if constexpr (DEFER_V) {
        if (args.with_v_term) {
          const bool never_came = wait_for_dgq();   // (bounded wait ran out: the rows below are stale -- NaN instead, M3G_TOPO_ERR_SYNC)
          // second pass over the in-edge list: dv in the one-pass form's order (chunks of 64 pairs, batches of NB, pairwise sums)
          for (int kc = ks; kc < k1s; kc += 64) {
            const int cnt = k1s - kc < 64 ? k1s - kc : 64;
            const int2 mine = ln < cnt ? in_pair[kc + ln] : make_int2(-1, -1);
            for (int b = 0; b < cnt; b += NB) {
              float g[NB];
#pragma unroll
              for (int j = 0; j < NB; ++j) {
                const int src = b + j < 64 ? b + j : 63;
                const int ar = b + j < 64 ? __builtin_amdgcn_readlane(mine.y, src) : -1;
                g[j] = ar >= 0 ? dgq[(int64_t)ar * kCP + cq] : 0.f;   // (written by other workgroups of this launch: read after the acquire in wait_for_dgq)
              }
#pragma unroll
              for (int j = 0; j < NB; j += 4) dv += (g[j] + g[j + 1]) + (g[j + 2] + g[j + 3]);
            }
          }
          if (never_came) dv = __builtin_nanf("");
        }
      }
#endif
      if (ln < kCP) {
        const float vv = v[i * kCP + ln];
        dvv = ln < C ? dv * vv * (1.f - vv) : 0.f;
      }
    }
    sA[nb][ln] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    sB[nb][ln] = make_float4(b0.x + b1.x, b0.y + b1.y, b0.z + b1.z, b0.w + b1.w);
    if (ln < kCP) tv[nb][ln] = dvv;
  }
  __syncthreads();
  // phase 2: quarter pq of the threads handles table columns [pq*64, pq*64+64) for output feature k
  {
    const int k = tid & 63, pq = tid >> 6;
    const MlpW& mw = pq < 2 ? bw.e : bw.n;
    const int row0 = (pq & 1) * kDP;  // row inside the MLP's [2*kDP][kDP] matrices
    const float* wa = W + mw.w1a + (size_t)row0 * kDP + k;
    const float* wb = W + mw.w1b + (size_t)row0 * kDP + k;
    float acc[kNodesRev];
#pragma unroll
    for (int nb = 0; nb < kNodesRev; ++nb) acc[nb] = 0.f;
    const float* fa = reinterpret_cast<const float*>(&sA[0][0]) + pq * kDP;
    const float* fb = reinterpret_cast<const float*>(&sB[0][0]) + pq * kDP;
    if constexpr (PRELOAD) {
#pragma unroll
      for (int o = 0; o < kDP; o += 4) {
#pragma unroll
        for (int nb = 0; nb < kNodesRev; ++nb) {
          const float4 va = *reinterpret_cast<const float4*>(fa + nb * 256 + o), vb = *reinterpret_cast<const float4*>(fb + nb * 256 + o);
          acc[nb] += (va.x * pre_a[o] + vb.x * pre_b[o]) + (va.y * pre_a[o + 1] + vb.y * pre_b[o + 1]) + (va.z * pre_a[o + 2] + vb.z * pre_b[o + 2]) +
                     (va.w * pre_a[o + 3] + vb.w * pre_b[o + 3]);
        }
      }
    } else
#ifdef M3G_DIAG_NR_NO_PHASE2   // timing diagnostic only (wrong results): what re-reading W1a^T / W1b^T per 4-atom group costs
    for (int o = 0; o < 4; o += 4) {
#else
    for (int o = 0; o < kDP; o += 4) {
#endif
      // four weight rows per trip: the row sums come from LDS as 16-byte broadcasts (a b32 read per term made this phase
      // LDS-issue-bound: 512 reads per thread), the eight weight loads of a trip are independent
      float a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = wa[(o + j) * kDP]; b[j] = wb[(o + j) * kDP]; }
#pragma unroll
      for (int nb = 0; nb < kNodesRev; ++nb) {
        const float4 va = *reinterpret_cast<const float4*>(fa + nb * 256 + o), vb = *reinterpret_cast<const float4*>(fb + nb * 256 + o);
        acc[nb] += (va.x * a[0] + vb.x * b[0]) + (va.y * a[1] + vb.y * b[1]) + (va.z * a[2] + vb.z * b[2]) + (va.w * a[3] + vb.w * b[3]);
      }
    }
#pragma unroll
    for (int nb = 0; nb < kNodesRev; ++nb) part[pq][nb][k] = acc[nb];
  }
  __syncthreads();
  for (int idx = tid; idx < kNodesRev * kDP; idx += 256) {
    const int nb = idx >> 6, k = idx & 63;
    const int64_t i = n0 + nb;
    if (i >= N) continue;
    float acc = dx_new[i * kDP + k] + ((part[0][nb][k] + part[1][nb][k]) + (part[2][nb][k] + part[3][nb][k]));
    for (int c = 0; c < C; ++c) acc += tv[nb][c] * W[bw.tb_w1 + c * kDP + k];
    dx_out[i * kDP + k] = acc;
  }
}


inline NodeRevArgs node_rev_args(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w, const float* v, const float* dx_new,
                                 float* dx_out, bool row_sums_in_seg, int dp1_packed, bool with_v_term) {
  const bool by_dst = dp1_packed == kDp1F32ByDst;   // fp32 rows either way; only their order in the array differs
  if (by_dst) dp1_packed = kDp1F32;
  return NodeRevArgs{c.C, t.N, W, bw, t.row_ptr, t.in_ptr, t.in_edge, w.dp1, w.dg, v, dx_new, dx_out, row_sums_in_seg ? w.seg_head : nullptr,
                     row_sums_in_seg ? w.seg_first : nullptr, with_v_term ? 1 : 0, reinterpret_cast<const int2*>(t.in_pair), dp1_packed,
                     dp1_scale_of(w.dp1, t.E), by_dst ? 1 : 0};
}

}  // namespace m3g
