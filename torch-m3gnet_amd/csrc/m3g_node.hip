// Per-atom kernels: embeddings (S1/B1), node pre-pass of a block (S2) and its reverse (B2),
// readout with fused reverse (S5).
// Reference: AtomFeaturizer nn/featurizer.py:33-38, EdgeAdjustor nn/featurizer.py:128-132,
// ThreeBodyInteration.linear_sigmoid1 nn/interaction.py:204-205, the x_i / x_j columns of the conv
// GatedMLP first layers nn/conv.py:91-97 + nn/core.py:61-62, AtomWiseReadout nn/readout.py:39-58.
#include "m3g_internal.h"
#include "m3g_device.h"
#include "m3g_mfma_common.h"
#include "m3g_node_rev.h"

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

__global__ void __launch_bounds__(256) k_node_reverse(NodeRevArgs a) {
  node_reverse_body<false>(a, blockIdx.x, [] { return false; });
}
__global__ void __launch_bounds__(256) k_node_reverse_small(NodeRevArgs a) {   // weights of phase 2 requested at entry (small systems)
  node_reverse_body<false, true>(a, blockIdx.x, [] { return false; });
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
                                                 float* __restrict__ dx, float* __restrict__ scaled_total, int64_t S, const int32_t* topo_flags) {
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
      bool bad;
      const int64_t ty = species_index(types[a0 + j], c.num_types, bad);
      scaled_atomic[a0 + j] = bad ? __builtin_nanf("") : W[elemental_off + ty] / c.energy_scale + od[j] * sg[j];
      if (bad) flag_bad_species(topo_flags);
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
  // (stage entry points m3g_atom_ref / m3g_atom_featurizer: the reference raises on an index outside the table, nn/atom_ref.py:27;
  //  here the table is never indexed with one and the atom's output row is NaN)
  bool bad;
  const int64_t r = species_index(idx[a], table_rows, bad);
  out[id] = bad ? __builtin_nanf("") : (transposed ? table[(int64_t)o * table_stride + r] : table[r * table_stride + o]);
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
                         hipStream_t s, bool small) {
  if (t.N > 0) {
    const NodeRevArgs a = node_rev_args(c, W, bw, t, w, v, dx_new, dx_out, row_sums_in_seg, dp1_packed, with_v_term);
    if (small) hipLaunchKernelGGL(k_node_reverse_small, grid_for(t.N, kNodesRev), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_node_reverse, grid_for(t.N, kNodesRev), dim3(256), 0, s, a);
  }
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
                       want_grad ? w.dx : nullptr, scaled_total, t.S, t.flags);
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
