// Per-atom kernels: embeddings (S1/B1), node pre-pass of a block (S2) and its reverse (B2),
// readout with fused reverse (S5).
// Reference: AtomFeaturizer nn/featurizer.py:33-38, EdgeAdjustor nn/featurizer.py:128-132,
// ThreeBodyInteration.linear_sigmoid1 nn/interaction.py:204-205, the x_i / x_j columns of the conv
// GatedMLP first layers nn/conv.py:91-97 + nn/core.py:61-62, AtomWiseReadout nn/readout.py:39-58.
#include "m3g_internal.h"
#include "m3g_device.h"
#include "m3g_mfma_common.h"

namespace m3g {


// x0[a,:] = emb[type[a],:]
__global__ void __launch_bounds__(256) k_embed_nodes(int64_t N, int num_types, const int64_t* __restrict__ types,
                                                     const float* __restrict__ emb, float* __restrict__ x) {
  int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= N * kDP) return;
  int64_t a = idx / kDP;
  int o = (int)(idx % kDP);
  int64_t ty = types[a];
  ty = ty < 0 ? 0 : (ty >= num_types ? num_types - 1 : ty);
  x[idx] = emb[ty * kDP + o];
}

// e0[e,:] = SiLU(W_adj h[e,:])
__global__ void __launch_bounds__(256) k_embed_edges(int R, int64_t E, const float* __restrict__ adj_t, const float* __restrict__ h,
                                                     float* __restrict__ e0) {
  int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= E * kDP) return;
  int64_t e = idx / kDP;
  int o = (int)(idx % kDP);
  float p = 0.f;
  for (int r = 0; r < R; ++r) p += adj_t[r * kDP + o] * h[e * kRP + r];
  e0[idx] = silu_f(p);
}

// dh[e,r] += sum_o de[e,o] SiLU'(pe0[e,o]) W_adj[o,r]    one wave per edge, lanes = o
__global__ void __launch_bounds__(256) k_embed_edges_reverse(int R, int64_t E, const float* __restrict__ adj_t,
                                                             const float* __restrict__ h, const float* __restrict__ de,
                                                             float* __restrict__ dh) {
  int64_t e = blockIdx.x * (int64_t)(blockDim.x / 64) + (threadIdx.x >> 6);
  int o = threadIdx.x & 63;
  if (e >= E) return;
  float p = 0.f;
  for (int r = 0; r < R; ++r) p += adj_t[r * kDP + o] * h[e * kRP + r];
  float t = de[e * kDP + o] * dsilu_f(p);
  for (int r = 0; r < R; ++r) {
    float val = t * adj_t[r * kDP + o];
    for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off);
    if (o == 0) dh[e * kRP + r] += val;
  }
}

// x_new[i,k] = x_prev[i,k] + the per-centre message sums the forward edge kernel left as partial rows (seg_head /
// seg_first, see m3g_edge_mfma.hip seg_scan): folded into the consumers of x_new instead of a kernel of its own
struct NodeSums {
  const float* x_prev;      // null: x is already final
  const float *seg_head, *seg_first;
  const int32_t* row_ptr;
  float* x_out;             // where the summed features are kept for later stages
};
__device__ __forceinline__ float node_feature(const NodeSums& ns, const float* __restrict__ x, int64_t i, int k) {
  if (!ns.x_prev) return x[i * kDP + k];
  float acc = ns.x_prev[i * kDP + k];
  const int r0 = ns.row_ptr[i], r1 = ns.row_ptr[i + 1];
  if (r1 > r0) {
    if (r0 & 15) acc += ns.seg_first[i * (4 * kDP) + k];
    for (int t = (r0 + 15) >> 4; t <= (r1 - 1) >> 4; ++t) acc += ns.seg_head[(int64_t)t * (4 * kDP) + k];
  }
  ns.x_out[i * kDP + k] = acc;
  return acc;
}

// ---- S2: v = sigmoid(W1 x + b1);  TA = [W1a_e x + b1_e | W1a_n x + b1_n];  TB = [W1b_e x | W1b_n x] ----
constexpr int kNodesPerBlock = 16;  // weights (128 KB per block of W1a/W1b) are read once per 16 atoms
__global__ void __launch_bounds__(256) k_node_pre(int C, int64_t N, const float* __restrict__ W, BlockW bw,
                                                  const float* __restrict__ x, NodeSums ns, float* __restrict__ v,
                                                  float* __restrict__ TA, float* __restrict__ TB) {
  __shared__ float xs[kNodesPerBlock][kDP];
  int64_t n0 = (int64_t)blockIdx.x * kNodesPerBlock;
  int tid = threadIdx.x;
  for (int idx = tid; idx < kNodesPerBlock * kDP; idx += 256) {
    int nb = idx >> 6, k = idx & 63;
    int64_t a = n0 + nb;
    xs[nb][k] = a < N ? node_feature(ns, x, a, k) : 0.f;
  }
  __syncthreads();
  // table column o in [0, 4*kDP): MLP e for o < 2*kDP, MLP n otherwise
  int o = tid;
  const MlpW& mw = o < 2 * kDP ? bw.e : bw.n;
  int oo = o < 2 * kDP ? o : o - 2 * kDP;
  float accA[kNodesPerBlock], accB[kNodesPerBlock];
  float bias = W[mw.b1 + oo];
#pragma unroll
  for (int nb = 0; nb < kNodesPerBlock; ++nb) { accA[nb] = bias; accB[nb] = 0.f; }
  const float* wa = W + mw.w1a_t + oo;
  const float* wb = W + mw.w1b_t + oo;
#pragma unroll 8
  for (int k = 0; k < kDP; ++k) {
    float a = wa[k * 2 * kDP], b = wb[k * 2 * kDP];
#pragma unroll
    for (int nb = 0; nb < kNodesPerBlock; ++nb) {
      float xv = xs[nb][k];
      accA[nb] += a * xv;
      accB[nb] += b * xv;
    }
  }
#pragma unroll
  for (int nb = 0; nb < kNodesPerBlock; ++nb) {
    int64_t a = n0 + nb;
    if (a < N) { TA[a * 4 * kDP + o] = accA[nb]; TB[a * 4 * kDP + o] = accB[nb]; }
  }
  static_assert(kNodesPerBlock * kCP <= 256, "one thread per (atom, c)");
  if (tid < kNodesPerBlock * kCP) {
    int nb = tid / kCP, c = tid % kCP;
    int64_t a = n0 + nb;
    if (a < N) {
      float p = W[bw.tb_b1 + c];
      for (int k = 0; k < kDP; ++k) p += W[bw.tb_w1_t + k * kCP + c] * xs[nb][k];
      v[a * kCP + c] = c < C ? sigmoid_f(p) : 0.f;
    }
  }
}

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
__global__ void __launch_bounds__(256) k_node_reverse(int C, int64_t N, const float* __restrict__ W, BlockW bw,
                                                      const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ in_ptr,
                                                      const int32_t* __restrict__ in_edge, const float* __restrict__ dp1,
                                                      const float* __restrict__ dgq, const float* __restrict__ v,
                                                      const float* __restrict__ dx_new, float* __restrict__ dx_out,
                                                      const float* __restrict__ seg_head, const float* __restrict__ seg_first,
                                                      int with_v_term, const int2* __restrict__ in_pair, int dp1_packed,
                                                      const float* __restrict__ dp1_scale) {
  __shared__ float4 sA[kNodesRev][64], sB[kNodesRev][64];   // row / in-edge sums of dp1, 256 columns as 64 float4
  __shared__ float tv[kNodesRev][kCP];
  __shared__ float part[4][kNodesRev][kDP];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int64_t n0 = (int64_t)blockIdx.x * kNodesRev;
  const float4* rows = reinterpret_cast<const float4*>(dp1) + ln;
  // phase 1: wave wv gathers for atoms wv, wv+4 of the group
  for (int nb = wv; nb < kNodesRev; nb += 4) {
    const int64_t i = n0 + nb;
    float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
    float dv = 0.f, dvv = 0.f;
    if (i < N) {
      const int e1 = row_ptr[i + 1];
      int e = row_ptr[i];
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
      const int k1 = in_ptr[i + 1];
      int k = in_ptr[i];
      float4 b2 = make_float4(0.f, 0.f, 0.f, 0.f), b3 = b2;
      const int cq = ln & 15;
      // kNrBatch whole 1-KB rows in flight per wave, the remainder in one guarded batch as well (a row-at-a-time tail
      // is a dependent round trip per row)
#ifndef M3G_NR_NO_CHUNK
      // the (edge, three-body row) pairs of up to 64 in-edges arrive in ONE coalesced load, a lane each, and are handed
      // out by v_readlane: a pair load per batch would put a dependent round trip in front of every batch of row loads
      const int ks = __builtin_amdgcn_readfirstlane(k), k1s = __builtin_amdgcn_readfirstlane(k1);
      for (int kc = ks; kc < k1s; kc += 64) {
        const int cnt = k1s - kc < 64 ? k1s - kc : 64;
        const int2 mine = ln < cnt ? in_pair[kc + ln] : make_int2(-1, -1);
      for (int b = 0; b < cnt; b += kNrBatch) {
        int2 f[kNrBatch];   // (edge id, compact three-body row or -1)
        float4 u[kNrBatch];
        float g[kNrBatch];
#pragma unroll
        for (int j = 0; j < kNrBatch; ++j) {
          const int src = b + j < 64 ? b + j : 63;   // lanes >= cnt hold (-1, -1)
          f[j].x = b + j < 64 ? __builtin_amdgcn_readlane(mine.x, src) : -1;
          f[j].y = b + j < 64 ? __builtin_amdgcn_readlane(mine.y, src) : -1;
        }
#else
      for (; k < k1; k += kNrBatch) {
        int2 f[kNrBatch];   // (edge id, compact three-body row or -1)
        float4 u[kNrBatch];
        float g[kNrBatch];
#pragma unroll
        for (int j = 0; j < kNrBatch; ++j) f[j] = k + j < k1 ? in_pair[k + j] : make_int2(-1, -1);
#endif
#ifndef M3G_DP1_F32
        if (dp1_packed == kDp1Fixed) {   // rows of the fused f16x3 reverse kernel: 24-bit fixed point + a scale per 64 columns (pack24_fixed)
          u32x3 pk[kNrBatch];
          float sc[kNrBatch];
#pragma unroll
          for (int j = 0; j < kNrBatch; ++j) {
            pk[j] = f[j].x >= 0 ? __builtin_nontemporal_load(reinterpret_cast<const u32x3_a4*>(reinterpret_cast<const unsigned*>(dp1) +
                                                                   (int64_t)f[j].x * kDp1PackedDwords + 3 * ln))
                                : u32x3{0u, 0u, 0u};   // (any bytes decode to finite numbers; the zero scale makes them 0)
            sc[j] = f[j].x >= 0 ? dp1_scale[(int64_t)f[j].x * 4 + (ln >> 4)] : 0.f;
          }
#pragma unroll
          for (int j = 0; j < kNrBatch; ++j) {
            const f32x4 t = unpack24_fixed(pk[j], sc[j]);
            u[j] = make_float4(t[0], t[1], t[2], t[3]);
            g[j] = (with_v_term && f[j].y >= 0) ? dgq[(int64_t)f[j].y * kCP + cq] : 0.f;
          }
        } else if (dp1_packed) {   // rows written by the fused bf16x3 reverse kernel: 24-bit values, 12 B per lane (m3g_mfma_common.h: pack24)
          u32x3 pk[kNrBatch];
#pragma unroll
          for (int j = 0; j < kNrBatch; ++j)
            // nontemporal: every row is read exactly once, and keeping it out of L2 leaves the cache to the weights and
            // partial rows (node reverse 0.218 -> 0.187 ms per step)
            pk[j] = f[j].x >= 0 ? __builtin_nontemporal_load(reinterpret_cast<const u32x3_a4*>(reinterpret_cast<const unsigned*>(dp1) +
                                                                   (int64_t)f[j].x * kDp1PackedDwords + 3 * ln))
                                : u32x3{0u, 0u, 0u};
#pragma unroll
          for (int j = 0; j < kNrBatch; ++j) {
            const f32x4 t = unpack24(pk[j]);
            u[j] = make_float4(t[0], t[1], t[2], t[3]);
            g[j] = (with_v_term && f[j].y >= 0) ? dgq[(int64_t)f[j].y * kCP + cq] : 0.f;
          }
        } else
#endif
#pragma unroll
        for (int j = 0; j < kNrBatch; ++j) {
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
        for (int j = 0; j < kNrBatch; j += 4) {
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

// the v-gradient share of the node reverse on its own (dx_out += (dv v (1-v)) W1), for when k_node_reverse ran without it
// beside the three-body reverse that produces dL/dg: 16 lanes per atom gather the dL/dg rows of the incoming edges
__global__ void __launch_bounds__(256) k_node_reverse_v_term(int C, int64_t N, const float* __restrict__ W, size_t tb_w1,
                                                             const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_edge,
                                                             const float* __restrict__ dgq, const float* __restrict__ v,
                                                             float* __restrict__ dx_out, const int2* __restrict__ in_pair) {
  __shared__ float tv[16][kCP];
  const int g = threadIdx.x >> 4, cq = threadIdx.x & 15;
  const int64_t i = (int64_t)blockIdx.x * 16 + g;
  float dv = 0.f;
  if (i < N) {
    int k = in_ptr[i];
    const int k1 = in_ptr[i + 1];
    auto row = [&](int kk) { const int ar = in_pair[kk].y; return ar >= 0 ? dgq[(int64_t)ar * kCP + cq] : 0.f; };   // one row per active edge
    for (; k + 3 < k1; k += 4) dv += (row(k) + row(k + 1)) + (row(k + 2) + row(k + 3));
    for (; k < k1; ++k) dv += row(k);
    (void)in_edge;
    const float vv = v[i * kCP + cq];
    dv = cq < C ? dv * vv * (1.f - vv) : 0.f;
  }
  tv[g][cq] = dv;
  __syncthreads();
  for (int idx = threadIdx.x; idx < 16 * kDP; idx += 256) {
    const int nb = idx >> 6, k = idx & 63;
    const int64_t a = (int64_t)blockIdx.x * 16 + nb;
    if (a >= N) continue;
    float acc = 0.f;
    for (int cc = 0; cc < C; ++cc) acc += tv[nb][cc] * W[tb_w1 + cc * kDP + k];
    dx_out[a * kDP + k] += acc;
  }
}

// ---- S5 readout (+ fused reverse): one wave per kRA atoms, lanes = feature ----------------------------------------
// Every weight element a lane reads from L2 is used for kRA atoms (a wave per atom re-read all seven 16-KB matrices for
// each atom and spent its time waiting on them).
constexpr int kRA = 4;
__global__ void __launch_bounds__(256) k_readout(Consts c, int64_t N, const float* __restrict__ W, ReadoutW rw,
                                                 size_t elemental_off, const int64_t* __restrict__ types,
                                                 const float* __restrict__ x, NodeSums ns, float* __restrict__ scaled_atomic,
                                                 float* __restrict__ dx, float* __restrict__ scaled_total, int64_t S) {
  __shared__ float bufA[4][kRA][kDP], bufB[4][kRA][kDP];
  // the per-structure sums are accumulated with atomics by the next kernel: cleared here instead of a memset launch
  if (blockIdx.x == 0) for (int64_t i = threadIdx.x; i < S; i += blockDim.x) scaled_total[i] = 0.f;
  const int wv = threadIdx.x >> 6, o = threadIdx.x & 63;
  const int64_t a0 = ((int64_t)blockIdx.x * 4 + wv) * kRA;
#pragma unroll
  for (int j = 0; j < kRA; ++j) bufA[wv][j][o] = a0 + j < N ? node_feature(ns, x, a0 + j, o) : 0.f;
  __syncthreads();
  float pd1[kRA], pg1[kRA], pd2[kRA], pg2[kRA];
  {
    const float bd = W[rw.b1d + o], bg = W[rw.b1g + o];
#pragma unroll
    for (int j = 0; j < kRA; ++j) { pd1[j] = bd; pg1[j] = bg; }
    for (int k = 0; k < kDP; ++k) {
      const float wd = W[rw.w1d_t + k * kDP + o], wg = W[rw.w1g_t + k * kDP + o];
#pragma unroll
      for (int j = 0; j < kRA; ++j) { const float xv = bufA[wv][j][k]; pd1[j] += wd * xv; pg1[j] += wg * xv; }
    }
  }
#pragma unroll
  for (int j = 0; j < kRA; ++j) { bufA[wv][j][o] = silu_f(pd1[j]); bufB[wv][j][o] = silu_f(pg1[j]); }
  __syncthreads();
  {
    const float bd = W[rw.b2d + o], bg = W[rw.b2g + o];
#pragma unroll
    for (int j = 0; j < kRA; ++j) { pd2[j] = bd; pg2[j] = bg; }
    for (int k = 0; k < kDP; ++k) {
      const float wd = W[rw.w2d_t + k * kDP + o], wg = W[rw.w2g_t + k * kDP + o];
#pragma unroll
      for (int j = 0; j < kRA; ++j) { pd2[j] += wd * bufA[wv][j][k]; pg2[j] += wg * bufB[wv][j][k]; }
    }
  }
  const float w3d = W[rw.w3d + o], w3g = W[rw.w3g + o];
  float od[kRA], og[kRA], sg[kRA];
#pragma unroll
  for (int j = 0; j < kRA; ++j) {
    od[j] = w3d * silu_f(pd2[j]);
    og[j] = w3g * silu_f(pg2[j]);
    for (int off = 32; off > 0; off >>= 1) { od[j] += __shfl_xor(od[j], off); og[j] += __shfl_xor(og[j], off); }
    od[j] += W[rw.b3];
    og[j] += W[rw.b3 + 1];
    sg[j] = sigmoid_f(og[j]);
    if (a0 + j < N && o == 0) {
      int64_t ty = types[a0 + j];
      ty = ty < 0 ? 0 : (ty >= c.num_types ? c.num_types - 1 : ty);
      scaled_atomic[a0 + j] = W[elemental_off + ty] / c.energy_scale + od[j] * sg[j];
    }
  }
  if (dx == nullptr) return;  // uniform
  // reverse: dL/d eps = energy_scale
#pragma unroll
  for (int j = 0; j < kRA; ++j) {
    const float d_od = c.energy_scale * sg[j], d_og = c.energy_scale * od[j] * sg[j] * (1.f - sg[j]);
    bufA[wv][j][o] = d_od * w3d * dsilu_f(pd2[j]);
    bufB[wv][j][o] = d_og * w3g * dsilu_f(pg2[j]);
  }
  __syncthreads();
  float d_hd1[kRA], d_hg1[kRA];
#pragma unroll
  for (int j = 0; j < kRA; ++j) { d_hd1[j] = 0.f; d_hg1[j] = 0.f; }
  for (int k = 0; k < kDP; ++k) {
    const float wd = W[rw.w2d + k * kDP + o], wg = W[rw.w2g + k * kDP + o];
#pragma unroll
    for (int j = 0; j < kRA; ++j) { d_hd1[j] += wd * bufA[wv][j][k]; d_hg1[j] += wg * bufB[wv][j][k]; }
  }
#pragma unroll
  for (int j = 0; j < kRA; ++j) {
    bufA[wv][j][o] = d_hd1[j] * dsilu_f(pd1[j]);
    bufB[wv][j][o] = d_hg1[j] * dsilu_f(pg1[j]);
  }
  __syncthreads();
  float acc[kRA];
#pragma unroll
  for (int j = 0; j < kRA; ++j) acc[j] = 0.f;
  for (int k = 0; k < kDP; ++k) {
    const float wd = W[rw.w1d + k * kDP + o], wg = W[rw.w1g + k * kDP + o];
#pragma unroll
    for (int j = 0; j < kRA; ++j) acc[j] += wd * bufA[wv][j][k] + wg * bufB[wv][j][k];
  }
#pragma unroll
  for (int j = 0; j < kRA; ++j)
    if (a0 + j < N) dx[(a0 + j) * kDP + o] = acc[j];
}

// per-structure energy sum: wave-level pre-reduction when the wave's atoms share a structure
__global__ void __launch_bounds__(256) k_gather_rows(int64_t n, int width, int table_stride, int table_rows,
                                                     const float* __restrict__ table, const int64_t* __restrict__ idx,
                                                     float* __restrict__ out, int transposed) {
  int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (id >= n * width) return;
  int64_t a = id / width;
  int o = (int)(id % width);
  int64_t r = idx[a];
  r = r < 0 ? 0 : (r >= table_rows ? table_rows - 1 : r);
  out[id] = transposed ? table[(int64_t)o * table_stride + r] : table[r * table_stride + o];
}

__global__ void __launch_bounds__(256) k_copy_strided(int64_t rows, int width, const float* __restrict__ in, int in_stride,
                                                      float* __restrict__ out, int out_stride) {
  int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (id >= rows * width) return;
  int64_t r = id / width;
  int o = (int)(id % width);
  out[r * out_stride + o] = in[r * in_stride + o];
}

// out[e, :width] = rows[row_id[e], :width], zeros where row_id[e] < 0 (per-edge view of an array kept per ACTIVE edge)
__global__ void __launch_bounds__(256) k_copy_expand_rows(int64_t n, int width, const int32_t* __restrict__ row_id,
                                                          const float* __restrict__ in, int in_stride, float* __restrict__ out,
                                                          int out_stride) {
  int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (id >= n * width) return;
  int64_t e = id / width;
  int o = (int)(id % width);
  const int r = row_id[e];
  out[e * out_stride + o] = r >= 0 ? in[(int64_t)r * in_stride + o] : 0.f;
}

static inline dim3 grid_for(int64_t n, int tpb = 256) { return dim3((unsigned)((n + tpb - 1) / tpb)); }

void launch_embed(const Consts& c, const float* W, const WeightLayout& wl, const Topo& t, const int64_t* types,
                  const Work& w, hipStream_t s) {
  if (t.N > 0) hipLaunchKernelGGL(k_embed_nodes, grid_for(t.N * kDP), dim3(256), 0, s, t.N, c.num_types, types, W + wl.emb, w.x[0]);
  if (t.E > 0) hipLaunchKernelGGL(k_embed_edges, grid_for(t.E * kDP), dim3(256), 0, s, c.R, t.E, W + wl.adj_t, w.h, w.e);
}

void launch_embed_nodes_only(const Consts& c, const float* W, const WeightLayout& wl, const Topo& t, const int64_t* types,
                             const Work& w, hipStream_t s) {
  if (t.N > 0) hipLaunchKernelGGL(k_embed_nodes, grid_for(t.N * kDP), dim3(256), 0, s, t.N, c.num_types, types, W + wl.emb, w.x[0]);
}

void launch_embed_reverse(const Consts& c, const float* W, const WeightLayout& wl, const Topo& t, const Work& w,
                          hipStream_t s) {
  if (t.E > 0)
    hipLaunchKernelGGL(k_embed_edges_reverse, grid_for(t.E, 4), dim3(256), 0, s, c.R, t.E, W + wl.adj_t, w.h, w.de, w.dh);
}

// x_prev != nullptr: x (= x_prev + the per-centre message sums in w.seg_*) is formed here and stored to x
void launch_node_pre(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w, const float* x_prev, float* x,
                     float* v, float* TA, float* TB, hipStream_t s) {
  NodeSums ns{x_prev, w.seg_head, w.seg_first, t.row_ptr, x};
  if (t.N > 0)
    hipLaunchKernelGGL(k_node_pre, grid_for(t.N, kNodesPerBlock), dim3(256), 0, s, c.C, t.N, W, bw, x, ns, v, TA, TB);
}

void launch_node_reverse(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w,
                         const float* v, const float* dx_new, float* dx_out, bool row_sums_in_seg, int dp1_packed, bool with_v_term,
                         hipStream_t s) {
  if (t.N > 0)
    hipLaunchKernelGGL(k_node_reverse, grid_for(t.N, kNodesRev), dim3(256), 0, s, c.C, t.N, W, bw, t.row_ptr, t.in_ptr, t.in_edge,
                       w.dp1, w.dg, v, dx_new, dx_out, row_sums_in_seg ? w.seg_head : nullptr, row_sums_in_seg ? w.seg_first : nullptr,
                       with_v_term ? 1 : 0, reinterpret_cast<const int2*>(t.in_pair), dp1_packed, dp1_scale_of(w.dp1, t.E));
}

void launch_node_reverse_v_term(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w, const float* v,
                                float* dx_out, hipStream_t s) {
  if (t.N > 0)
    hipLaunchKernelGGL(k_node_reverse_v_term, grid_for(t.N, 16), dim3(256), 0, s, c.C, t.N, W, bw.tb_w1, t.in_ptr, t.in_edge, w.dg, v,
                       dx_out, reinterpret_cast<const int2*>(t.in_pair));
}

// per-structure sums of the scaled atomic energies and total = energy_scale * sum: one launch (k_struct_energy, fixed order)
void launch_energy_sums(const Consts& c, const Topo& t, const float* scaled_atomic, float* scaled_total, float* total, hipStream_t s) {
  launch_struct_energy(c, t, scaled_atomic, scaled_total, total, s);
}

void launch_readout(const Consts& c, const float* W, const WeightLayout& wl, const Topo& t, const int64_t* types,
                    const float* x_prev, float* x, const Work& w, float* scaled_atomic, float* scaled_total, float* total,
                    bool want_grad, hipStream_t s) {
  NodeSums ns{x_prev, w.seg_head, w.seg_first, t.row_ptr, x};
  if (t.N == 0) (void)hipMemsetAsync(scaled_total, 0, sizeof(float) * t.S, s);
  if (t.N > 0) {
    hipLaunchKernelGGL(k_readout, grid_for(t.N, 4 * kRA), dim3(256), 0, s, c, t.N, W, wl.ro, wl.elemental, types, x, ns, scaled_atomic,
                       want_grad ? w.dx : nullptr, scaled_total, t.S);
  }
  launch_energy_sums(c, t, scaled_atomic, scaled_total, total, s);
}

void launch_gather_rows(const float* table, int64_t n, int width, int table_stride, int table_rows, bool transposed,
                        const int64_t* idx, float* out, hipStream_t s) {
  // stage entry points: row-major table [rows][stride], or (transposed) torch's [width][rows] weight
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_rows, grid_for(n * width), dim3(256), 0, s, n, width, table_stride, table_rows, table, idx, out,
                     transposed ? 1 : 0);
}

void launch_copy_strided(const float* in, int in_stride, float* out, int out_stride, int width, int64_t rows,
                         hipStream_t s) {
  if (rows > 0) hipLaunchKernelGGL(k_copy_strided, grid_for(rows * width), dim3(256), 0, s, rows, width, in, in_stride, out, out_stride);
}

void launch_copy_expand_rows(const int32_t* row_id, const float* in, int in_stride, float* out, int out_stride, int width, int64_t n,
                             hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_copy_expand_rows, grid_for(n * width), dim3(256), 0, s, n, width, row_id, in, in_stride, out, out_stride);
}

}  // namespace m3g
