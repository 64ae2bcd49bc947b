// Per-atom kernels: embeddings (S1/B1), node pre-pass of a block (S2) and its reverse (B2),
// readout with fused reverse (S5).
// Reference: AtomFeaturizer nn/featurizer.py:33-38, EdgeAdjustor nn/featurizer.py:128-132,
// ThreeBodyInteration.linear_sigmoid1 nn/interaction.py:204-205, the x_i / x_j columns of the conv
// GatedMLP first layers nn/conv.py:91-97 + nn/core.py:61-62, AtomWiseReadout nn/readout.py:39-58.
#include "m3g_internal.h"
#include "m3g_device.h"

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

// ---- S2: v = sigmoid(W1 x + b1);  TA = [W1a_e x + b1_e | W1a_n x + b1_n];  TB = [W1b_e x | W1b_n x] ----
constexpr int kNodesPerBlock = 4;
__global__ void __launch_bounds__(256) k_node_pre(int C, int64_t N, const float* __restrict__ W, BlockW bw,
                                                  const float* __restrict__ x, float* __restrict__ v, float* __restrict__ TA,
                                                  float* __restrict__ TB) {
  __shared__ float xs[kNodesPerBlock][kDP];
  int64_t n0 = (int64_t)blockIdx.x * kNodesPerBlock;
  int tid = threadIdx.x;
  {
    int nb = tid >> 6, k = tid & 63;
    int64_t a = n0 + nb;
    xs[nb][k] = a < N ? x[a * kDP + k] : 0.f;
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
__global__ void __launch_bounds__(256) k_node_reverse(int C, int64_t N, const float* __restrict__ W, BlockW bw,
                                                      const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ in_ptr,
                                                      const int32_t* __restrict__ in_edge, const float* __restrict__ dp1,
                                                      const float* __restrict__ dgq, const float* __restrict__ v,
                                                      const float* __restrict__ dx_new, float* __restrict__ dx_out) {
  __shared__ float sA[4 * kDP], sB[4 * kDP], tv[kCP], part[4][kDP];
  int64_t i = blockIdx.x;
  int tid = threadIdx.x;
  {
    // 4 independent accumulators per list: keeps 8 row loads in flight per thread (HBM-bound gather of dp1)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    const int e0 = row_ptr[i], e1 = row_ptr[i + 1];
    int e = e0;
    for (; e + 4 <= e1; e += 4) {
      a0 += dp1[(int64_t)e * 4 * kDP + tid];
      a1 += dp1[(int64_t)(e + 1) * 4 * kDP + tid];
      a2 += dp1[(int64_t)(e + 2) * 4 * kDP + tid];
      a3 += dp1[(int64_t)(e + 3) * 4 * kDP + tid];
    }
    for (; e < e1; ++e) a0 += dp1[(int64_t)e * 4 * kDP + tid];
    const int k0 = in_ptr[i], k1 = in_ptr[i + 1];
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
      const int f0 = in_edge[k], f1 = in_edge[k + 1], f2 = in_edge[k + 2], f3 = in_edge[k + 3];
      b0 += dp1[(int64_t)f0 * 4 * kDP + tid];
      b1 += dp1[(int64_t)f1 * 4 * kDP + tid];
      b2 += dp1[(int64_t)f2 * 4 * kDP + tid];
      b3 += dp1[(int64_t)f3 * 4 * kDP + tid];
    }
    for (; k < k1; ++k) b0 += dp1[(int64_t)in_edge[k] * 4 * kDP + tid];
    sA[tid] = (a0 + a1) + (a2 + a3);
    sB[tid] = (b0 + b1) + (b2 + b3);
  }
  if (tid < kCP) {
    float dv = 0.f;
    for (int k = in_ptr[i]; k < in_ptr[i + 1]; ++k) dv += dgq[(int64_t)in_edge[k] * kCP + tid];
    float vv = v[i * kCP + tid];
    tv[tid] = tid < C ? dv * vv * (1.f - vv) : 0.f;
  }
  __syncthreads();
  int k = tid & 63, pq = tid >> 6;  // quarter pq handles table columns [pq*kDP, (pq+1)*kDP)
  {
    const MlpW& mw = pq < 2 ? bw.e : bw.n;
    int row0 = (pq & 1) * kDP;  // row inside the MLP's [2*kDP][kDP] matrices
    const float* wa = W + mw.w1a + (size_t)row0 * kDP + k;
    const float* wb = W + mw.w1b + (size_t)row0 * kDP + k;
    float acc = 0.f;
    for (int o = 0; o < kDP; ++o) acc += sA[pq * kDP + o] * wa[o * kDP] + sB[pq * kDP + o] * wb[o * kDP];
    part[pq][k] = acc;
  }
  __syncthreads();
  if (tid < kDP) {
    float acc = dx_new[i * kDP + tid] + ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
    for (int c = 0; c < C; ++c) acc += tv[c] * W[bw.tb_w1 + c * kDP + tid];
    dx_out[i * kDP + tid] = acc;
  }
}

// ---- S5 readout (+ fused reverse): one wave per atom, lanes = feature ------------------------------------------
__global__ void __launch_bounds__(256) k_readout(Consts c, int64_t N, const float* __restrict__ W, ReadoutW rw,
                                                 size_t elemental_off, const int64_t* __restrict__ types,
                                                 const float* __restrict__ x, float* __restrict__ scaled_atomic,
                                                 float* __restrict__ dx) {
  __shared__ float bufA[4][kDP], bufB[4][kDP];
  int wv = threadIdx.x >> 6, o = threadIdx.x & 63;
  int64_t a = (int64_t)blockIdx.x * 4 + wv;
  bool live = a < N;
  int64_t aa = live ? a : 0;
  bufA[wv][o] = x[aa * kDP + o];
  __syncthreads();
  float pd1 = W[rw.b1d + o], pg1 = W[rw.b1g + o];
  for (int k = 0; k < kDP; ++k) {
    float xv = bufA[wv][k];
    pd1 += W[rw.w1d_t + k * kDP + o] * xv;
    pg1 += W[rw.w1g_t + k * kDP + o] * xv;
  }
  __syncthreads();
  bufA[wv][o] = silu_f(pd1);
  bufB[wv][o] = silu_f(pg1);
  __syncthreads();
  float pd2 = W[rw.b2d + o], pg2 = W[rw.b2g + o];
  for (int k = 0; k < kDP; ++k) {
    pd2 += W[rw.w2d_t + k * kDP + o] * bufA[wv][k];
    pg2 += W[rw.w2g_t + k * kDP + o] * bufB[wv][k];
  }
  float od = W[rw.w3d + o] * silu_f(pd2), og = W[rw.w3g + o] * silu_f(pg2);
  for (int off = 32; off > 0; off >>= 1) { od += __shfl_xor(od, off); og += __shfl_xor(og, off); }
  od += W[rw.b3];
  og += W[rw.b3 + 1];
  float sg = sigmoid_f(og);
  if (live && o == 0) {
    int64_t ty = types[a];
    ty = ty < 0 ? 0 : (ty >= c.num_types ? c.num_types - 1 : ty);
    scaled_atomic[a] = W[elemental_off + ty] / c.energy_scale + od * sg;
  }
  if (dx == nullptr) return;  // uniform
  // reverse: dL/d eps = energy_scale
  float d_od = c.energy_scale * sg, d_og = c.energy_scale * od * sg * (1.f - sg);
  float d_pd2 = d_od * W[rw.w3d + o] * dsilu_f(pd2), d_pg2 = d_og * W[rw.w3g + o] * dsilu_f(pg2);
  __syncthreads();
  bufA[wv][o] = d_pd2;
  bufB[wv][o] = d_pg2;
  __syncthreads();
  float d_hd1 = 0.f, d_hg1 = 0.f;
  for (int j = 0; j < kDP; ++j) {
    d_hd1 += W[rw.w2d + j * kDP + o] * bufA[wv][j];
    d_hg1 += W[rw.w2g + j * kDP + o] * bufB[wv][j];
  }
  float d_pd1 = d_hd1 * dsilu_f(pd1), d_pg1 = d_hg1 * dsilu_f(pg1);
  __syncthreads();
  bufA[wv][o] = d_pd1;
  bufB[wv][o] = d_pg1;
  __syncthreads();
  float acc = 0.f;
  for (int j = 0; j < kDP; ++j) acc += W[rw.w1d + j * kDP + o] * bufA[wv][j] + W[rw.w1g + j * kDP + o] * bufB[wv][j];
  if (live) dx[a * kDP + o] = acc;
}

// per-structure energy sum: wave-level pre-reduction when the wave's atoms share a structure
__global__ void __launch_bounds__(256) k_energy_sum(int64_t N, const int32_t* __restrict__ batch, const float* __restrict__ ea,
                                                    float* __restrict__ scaled_total) {
  int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool live = a < N;
  int s = live ? batch[a] : -1;
  float val = live ? ea[a] : 0.f;
  int s0 = __shfl(s, 0);
  bool uniform = __all(s == s0 || !live);
  if (uniform && s0 >= 0) {
    for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&scaled_total[s0], val);
  } else if (live) {
    atomicAdd(&scaled_total[s], val);
  }
}
__global__ void k_scale(int64_t S, float scale, const float* __restrict__ in, float* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < S) out[i] = scale * in[i];
}

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

void launch_node_pre(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const float* x, float* v,
                     float* TA, float* TB, hipStream_t s) {
  if (t.N > 0)
    hipLaunchKernelGGL(k_node_pre, grid_for(t.N, kNodesPerBlock), dim3(256), 0, s, c.C, t.N, W, bw, x, v, TA, TB);
}

void launch_node_reverse(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w,
                         const float* v, const float* dx_new, float* dx_out, hipStream_t s) {
  if (t.N > 0)
    hipLaunchKernelGGL(k_node_reverse, dim3((unsigned)t.N), dim3(256), 0, s, c.C, t.N, W, bw, t.row_ptr, t.in_ptr, t.in_edge,
                       w.dp1, w.dg, v, dx_new, dx_out);
}

void launch_readout(const Consts& c, const float* W, const WeightLayout& wl, const Topo& t, const int64_t* types,
                    const float* x, const Work& w, float* scaled_atomic, float* scaled_total, float* total,
                    bool want_grad, hipStream_t s) {
  (void)hipMemsetAsync(scaled_total, 0, sizeof(float) * t.S, s);
  if (t.N > 0) {
    hipLaunchKernelGGL(k_readout, grid_for(t.N, 4), dim3(256), 0, s, c, t.N, W, wl.ro, wl.elemental, types, x, scaled_atomic,
                       want_grad ? w.dx : nullptr);
    hipLaunchKernelGGL(k_energy_sum, grid_for(t.N), dim3(256), 0, s, t.N, t.batch, scaled_atomic, scaled_total);
  }
  if (t.S > 0) hipLaunchKernelGGL(k_scale, grid_for(t.S), dim3(256), 0, s, t.S, c.energy_scale, scaled_total, total);
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

}  // namespace m3g
