// Edge block, baseline ("simple") implementation: stage S4 and its reverse B4 on the vector ALUs.
// One workgroup = 16 edges x 64 features; each wave owns 4 edges, lanes = output feature, activations
// of the tile staged in LDS, weights streamed k-major from L2 (coalesced 256 B per wave).
// This is the correctness baseline and the fallback; the MFMA version (m3g_edge_mfma.hip) replaces
// it on the default path.
// Reference: ThreeBodyInteration gated update nn/interaction.py:220-221; M3GNetConv nn/conv.py:63-97;
// GatedMLP nn/core.py:61-62.  Stage algebra: oracle/staged.py (_mlp2_forward/_mlp2_backward).
#include "m3g_internal.h"
#include "m3g_device.h"

namespace m3g {

constexpr int TE = 16;   // edges per workgroup
constexpr int EPW = 4;   // edges per wave

struct TileLds {
  float X[TE][kDP];
  float Hd[TE][kDP];
  float Hg[TE][kDP];
  float ms[TE][kCP];
  float hs[TE][kRP];
  int ss[TE];
  int ds[TE];
};

// One conv GatedMLP on the tile.  X holds its edge-feature input.  out[j] = MLP(..)[o] * (W_l h)[o].
__device__ __forceinline__ void mlp_forward(const float* __restrict__ W, const MlpW& mw, int tabofs, int actofs, TileLds& L,
                                            const float* __restrict__ TA, const float* __restrict__ TB,
                                            float* __restrict__ act, int64_t e0, int64_t E, int R, int o, int eg,
                                            float out[EPW]) {
  float accd[EPW], accg[EPW];
#pragma unroll
  for (int j = 0; j < EPW; ++j) {
    int le = eg * EPW + j;
    int64_t s = L.ss[le], d = L.ds[le];
    accd[j] = TA[s * 4 * kDP + tabofs + o] + TB[d * 4 * kDP + tabofs + o];
    accg[j] = TA[s * 4 * kDP + tabofs + kDP + o] + TB[d * 4 * kDP + tabofs + kDP + o];
  }
  const float* w1 = W + mw.w1c_t + o;
  for (int k = 0; k < kDP; ++k) {
    float wd = w1[k * 2 * kDP], wg = w1[k * 2 * kDP + kDP];
#pragma unroll
    for (int j = 0; j < EPW; ++j) {
      float xv = L.X[eg * EPW + j][k];
      accd[j] += wd * xv;
      accg[j] += wg * xv;
    }
  }
#pragma unroll
  for (int j = 0; j < EPW; ++j) {
    int le = eg * EPW + j;
    int64_t e = e0 + le;
    if (e < E) {
      act[e * 8 * kDP + actofs + o] = accd[j];
      act[e * 8 * kDP + actofs + kDP + o] = accg[j];
    }
    L.Hd[le][o] = silu_f(accd[j]);
    L.Hg[le][o] = silu_f(accg[j]);
  }
  __syncthreads();
  float p2d[EPW], p2g[EPW];
  float bd = W[mw.b2d + o], bg = W[mw.b2g + o];
#pragma unroll
  for (int j = 0; j < EPW; ++j) { p2d[j] = bd; p2g[j] = bg; }
  const float* w2d = W + mw.w2d_t + o;
  const float* w2g = W + mw.w2g_t + o;
  for (int k = 0; k < kDP; ++k) {
    float wd = w2d[k * kDP], wg = w2g[k * kDP];
#pragma unroll
    for (int j = 0; j < EPW; ++j) {
      p2d[j] += wd * L.Hd[eg * EPW + j][k];
      p2g[j] += wg * L.Hg[eg * EPW + j][k];
    }
  }
#pragma unroll
  for (int j = 0; j < EPW; ++j) {
    int le = eg * EPW + j;
    int64_t e = e0 + le;
    if (e < E) {
      act[e * 8 * kDP + actofs + 2 * kDP + o] = p2d[j];
      act[e * 8 * kDP + actofs + 3 * kDP + o] = p2g[j];
    }
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += W[mw.wl_t + r * kDP + o] * L.hs[le][r];
    out[j] = silu_f(p2d[j]) * sigmoid_f(p2g[j]) * s;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) k_edge_block(Consts c, int64_t E, const float* __restrict__ W, BlockW bw,
                                                    const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                    const float* __restrict__ h, const float* __restrict__ m,
                                                    const float* __restrict__ TA, const float* __restrict__ TB,
                                                    float* __restrict__ e_io, float* __restrict__ act,
                                                    float* __restrict__ x_new, const int32_t* __restrict__ act_id) {
  __shared__ TileLds L;
  int tid = threadIdx.x, o = tid & 63, eg = tid >> 6;
  int64_t e0 = (int64_t)blockIdx.x * TE;
  {
    int le = tid / kCP, cc = tid % kCP;  // 256 threads = 16 edges x 16
    int64_t e = e0 + le;
    const int arow = e < E ? act_id[e] : -1;   // m, dm: one row per active edge
    L.ms[le][cc] = arow >= 0 ? m[(int64_t)arow * kCP + cc] : 0.f;
    if (cc < kRP) L.hs[le][cc] = e < E ? h[e * kRP + cc] : 0.f;
    if (cc == 0) { L.ss[le] = e < E ? src[e] : 0; L.ds[le] = e < E ? dst[e] : 0; }
  }
  __syncthreads();
  float e1[EPW];
#pragma unroll
  for (int j = 0; j < EPW; ++j) {
    int le = eg * EPW + j;
    int64_t e = e0 + le;
    float pd = 0.f, pg = 0.f;
    for (int cc = 0; cc < c.C; ++cc) {
      float mv = L.ms[le][cc];
      pd += W[bw.tb_wd_t + cc * kDP + o] * mv;
      pg += W[bw.tb_wg_t + cc * kDP + o] * mv;
    }
    e1[j] = (e < E ? e_io[e * kDP + o] : 0.f) + silu_f(pd) * sigmoid_f(pg);
    L.X[le][o] = e1[j];
  }
  __syncthreads();
  float upd[EPW];
  mlp_forward(W, bw.e, 0, 0, L, TA, TB, act, e0, E, c.R, o, eg, upd);
  float e2[EPW];
#pragma unroll
  for (int j = 0; j < EPW; ++j) {
    int le = eg * EPW + j;
    int64_t e = e0 + le;
    e2[j] = e1[j] + upd[j];
    L.X[le][o] = e2[j];
    if (e < E) e_io[e * kDP + o] = e2[j];
  }
  __syncthreads();
  float msg[EPW];
  mlp_forward(W, bw.n, 2 * kDP, 4 * kDP, L, TA, TB, act, e0, E, c.R, o, eg, msg);
  // segment-sum onto the centre atom: combine the wave's consecutive edges that share a centre
  float run = 0.f;
  int cur = -1;
#pragma unroll
  for (int j = 0; j < EPW; ++j) {
    int le = eg * EPW + j;
    if (e0 + le < E) {
      int s = L.ss[le];
      if (s != cur) {
        if (cur >= 0) atomicAdd(&x_new[(int64_t)cur * kDP + o], run);
        cur = s;
        run = 0.f;
      }
      run += msg[j];
    }
  }
  if (cur >= 0) atomicAdd(&x_new[(int64_t)cur * kDP + o], run);
}

struct TileLdsRev {
  float A[TE][kDP];
  float G[TE][kDP];
  float P[TE][2 * kDP];
  float ms[TE][kCP];
  float hs[TE][kRP];
  int ss[TE];
};

// reverse of one conv GatedMLP; d_upd[j] = dL/d(MLP output * s); returns dL/d(edge-feature input)
__device__ __forceinline__ void mlp_reverse(const float* __restrict__ W, const MlpW& mw, int actofs, int dpofs, TileLdsRev& L,
                                            const float* __restrict__ act, float* __restrict__ dp1, float* __restrict__ dh,
                                            int64_t e0, int64_t E, int R, int k, int eg, const float d_upd[EPW],
                                            float contrib[EPW]) {
#pragma unroll
  for (int j = 0; j < EPW; ++j) {
    int le = eg * EPW + j;
    int64_t e = e0 + le;
    bool valid = e < E;
    float p2d = valid ? act[e * 8 * kDP + actofs + 2 * kDP + k] : 0.f;
    float p2g = valid ? act[e * 8 * kDP + actofs + 3 * kDP + k] : 0.f;
    float sg = sigmoid_f(p2g), sd = silu_f(p2d);
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += W[mw.wl_t + r * kDP + k] * L.hs[le][r];
    float d_out = d_upd[j] * s, d_s = d_upd[j] * sd * sg;
    for (int r = 0; r < R; ++r) {
      float val = d_s * W[mw.wl_t + r * kDP + k];
      for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off);
      if (k == 0 && valid) dh[e * kRP + r] += val;
    }
    L.A[le][k] = d_out * sg * dsilu_f(p2d);
    L.G[le][k] = d_out * sd * sg * (1.f - sg);
  }
  __syncthreads();
  float dhd[EPW], dhg[EPW];
#pragma unroll
  for (int j = 0; j < EPW; ++j) { dhd[j] = 0.f; dhg[j] = 0.f; }
  const float* w2d = W + mw.w2d + k;
  const float* w2g = W + mw.w2g + k;
  for (int o = 0; o < kDP; ++o) {
    float wd = w2d[o * kDP], wg = w2g[o * kDP];
#pragma unroll
    for (int j = 0; j < EPW; ++j) {
      dhd[j] += wd * L.A[eg * EPW + j][o];
      dhg[j] += wg * L.G[eg * EPW + j][o];
    }
  }
#pragma unroll
  for (int j = 0; j < EPW; ++j) {
    int le = eg * EPW + j;
    int64_t e = e0 + le;
    bool valid = e < E;
    float p1d = valid ? act[e * 8 * kDP + actofs + k] : 0.f;
    float p1g = valid ? act[e * 8 * kDP + actofs + kDP + k] : 0.f;
    float a = dhd[j] * dsilu_f(p1d), b = dhg[j] * dsilu_f(p1g);
    L.P[le][k] = a;
    L.P[le][kDP + k] = b;
    if (valid) {
      dp1[e * 4 * kDP + dpofs + k] = a;
      dp1[e * 4 * kDP + dpofs + kDP + k] = b;
    }
    contrib[j] = 0.f;
  }
  __syncthreads();
  const float* w1 = W + mw.w1c + k;
  for (int o = 0; o < 2 * kDP; ++o) {
    float wv = w1[o * kDP];
#pragma unroll
    for (int j = 0; j < EPW; ++j) contrib[j] += wv * L.P[eg * EPW + j][o];
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) k_edge_block_reverse(Consts c, int64_t E, const float* __restrict__ W, BlockW bw,
                                                            const int32_t* __restrict__ src, const float* __restrict__ h,
                                                            const float* __restrict__ m, const float* __restrict__ act,
                                                            const float* __restrict__ dx_new, float* __restrict__ de_io,
                                                            float* __restrict__ dm, float* __restrict__ dh,
                                                            float* __restrict__ dp1, const int32_t* __restrict__ act_id) {
  __shared__ TileLdsRev L;
  int tid = threadIdx.x, k = tid & 63, eg = tid >> 6;
  int64_t e0 = (int64_t)blockIdx.x * TE;
  {
    int le = tid / kCP, cc = tid % kCP;
    int64_t e = e0 + le;
    const int arow = e < E ? act_id[e] : -1;   // m, dm: one row per active edge
    L.ms[le][cc] = arow >= 0 ? m[(int64_t)arow * kCP + cc] : 0.f;
    if (cc < kRP) L.hs[le][cc] = e < E ? h[e * kRP + cc] : 0.f;
    if (cc == 0) L.ss[le] = e < E ? src[e] : 0;
  }
  __syncthreads();
  float d_in[EPW], d_msg[EPW], contrib[EPW];
#pragma unroll
  for (int j = 0; j < EPW; ++j) {
    int le = eg * EPW + j;
    int64_t e = e0 + le;
    d_in[j] = e < E ? de_io[e * kDP + k] : 0.f;
    d_msg[j] = e < E ? dx_new[(int64_t)L.ss[le] * kDP + k] : 0.f;
  }
  mlp_reverse(W, bw.n, 4 * kDP, 2 * kDP, L, act, dp1, dh, e0, E, c.R, k, eg, d_msg, contrib);
  float d_e2[EPW];
#pragma unroll
  for (int j = 0; j < EPW; ++j) d_e2[j] = d_in[j] + contrib[j];
  mlp_reverse(W, bw.e, 0, 0, L, act, dp1, dh, e0, E, c.R, k, eg, d_e2, contrib);
#pragma unroll
  for (int j = 0; j < EPW; ++j) {
    int le = eg * EPW + j;
    int64_t e = e0 + le;
    float d_e1 = d_e2[j] + contrib[j];
    if (e < E) de_io[e * kDP + k] = d_e1;
    float pd = 0.f, pg = 0.f;
    for (int cc = 0; cc < c.C; ++cc) {
      float mv = L.ms[le][cc];
      pd += W[bw.tb_wd_t + cc * kDP + k] * mv;
      pg += W[bw.tb_wg_t + cc * kDP + k] * mv;
    }
    float sg = sigmoid_f(pg);
    L.A[le][k] = d_e1 * sg * dsilu_f(pd);
    L.G[le][k] = d_e1 * silu_f(pd) * sg * (1.f - sg);
  }
  __syncthreads();
  {
    int le = tid / kCP, cc = tid % kCP;
    int64_t e = e0 + le;
    if (e < E) {
      float acc = 0.f;
      if (cc < c.C) {
        for (int o = 0; o < kDP; ++o)
          acc += L.A[le][o] * W[bw.tb_wd + o * kCP + cc] + L.G[le][o] * W[bw.tb_wg + o * kCP + cc];
      }
      const int arow = act_id[e];
      if (arow >= 0) dm[(int64_t)arow * kCP + cc] = acc;
    }
  }
}

static inline dim3 grid_for(int64_t n, int per) { return dim3((unsigned)((n + per - 1) / per)); }

void launch_edge_block(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w, int b,
                       float* x_new, hipStream_t s) {
  if (t.E == 0) return;
  hipLaunchKernelGGL(k_edge_block, grid_for(t.E, TE), dim3(256), 0, s, c, t.E, W, bw, t.src, t.dst, w.h, w.m[b], w.TA, w.TB,
                     w.e, w.act[b], x_new, t.act_id);
}

void launch_edge_block_reverse(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w, int b,
                               const float* dx_new, hipStream_t s) {
  if (t.E == 0) return;
  hipLaunchKernelGGL(k_edge_block_reverse, grid_for(t.E, TE), dim3(256), 0, s, c, t.E, W, bw, t.src, w.h, w.m[b], w.act[b],
                     dx_new, w.de, w.dm, w.dh, w.dp1, t.act_id);
}

}  // namespace m3g
