// Three-body aggregate (stage S3) and its reverse (B3).
// Reference: ThreeBodyInteration.forward, nn/interaction.py:187-217 (the scatter_sum over triplets),
// LegendreCosPolynomial nn/interaction.py:353-365, clamp of cos(theta) nn/invariant.py:40.
//
//   m[e1,c] = fc(d_e1) * sum_{t in T1(e1)} Y_l(cos_t) g[e2(t),c],   g[e,c] = q[e,c] v[dst(e),c],  c = l*R+n
//
// Triplets are CSR-grouped by e1 (forward, and the e1 half of the reverse) and by e2 (the e2 half of the
// reverse), so every sum is a private register accumulation -- no atomics, run-to-run reproducible.
// Rows are the ACTIVE edges only (edges inside the three-body cutoff that have a partner; Topo::act_list): with
// r_c = 5 / r_3 = 4 on fcc Cu 24 of an atom's 42 edges carry no triplet, and a thread per edge left 57 % of the
// lanes idle.  One thread per active row, 128 consecutive rows per workgroup.  The partners of those rows are
// active edges of the same centre atoms, i.e. one contiguous window of the compacted list: the workgroup stages that
// window's unit vectors and per-edge payload rows (g = q*v[dst] for the forward / e1 half, dS = fc*dm for the e2
// half) in LDS once -- the per-atom triplet tile -- and the triplet loops read LDS instead of gathering from
// L2.  Partners outside the staged window (only possible for degrees beyond the LDS budget) fall back to global
// memory, so correctness does not depend on the window size.
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#include "m3g_internal.h"
#include "m3g_node_rev.h"
#include "m3g_geometry_body.h"

namespace m3g {


// first C floats of a 16-float (64-byte aligned) row as 16-byte loads: a scalar load per element makes every lane of a
// wave touch its own cache line once per element
template <int C>
__device__ __forceinline__ void load_row(const float* __restrict__ row, float* out) {
#pragma unroll
  for (int k = 0; k < C; k += 4) {
    const float4 t = *(const float4*)(row + k);
    out[k] = t.x;
    if (k + 1 < C) out[k + 1] = t.y;
    if (k + 2 < C) out[k + 2] = t.z;
    if (k + 3 < C) out[k + 3] = t.w;
  }
}

template <int L>
__device__ __forceinline__ void legendre(float x, float* P, float* dP) {
  P[0] = 1.f; dP[0] = 0.f;
  if (L > 1) { P[1] = x; dP[1] = 1.f; }
#pragma unroll
  for (int n = 1; n < L - 1; ++n) {
    P[n + 1] = ((float)(2 * n + 1) * x * P[n] - (float)n * P[n - 1]) / (float)(n + 1);
    dP[n + 1] = ((float)(2 * n + 1) * (P[n] + x * dP[n]) - (float)n * dP[n - 1]) / (float)(n + 1);
  }
}

struct TbArgs {
  int64_t E;
  const int32_t *act_list, *act_dst, *tb_win, *n_act;
  const int32_t *t_ptr, *t_other;     // triplets grouped by e1, partners e2 as compacted ids
  const uint8_t* t_bytes;             // the same partners as window-relative bytes (255: use t_other)
  const float *u, *fc3, *q, *v;       // q: one row per active edge
  float* m;                           // out [A][kCP], one row per active edge
};

// Forward: m[e1,:] = fc(d_e1) sum_t Y_l(cos_t) g[e2(t),:]
// LPR lanes share one row: each walks every LPR-th partner of the row's list and the partial sums are combined inside the lane
// group (fixed order: deterministic).  The kernels are latency-bound -- a dependent LDS id read, then the partner's unit vector
// and payload row, then the Legendre recurrence, per partner -- so the serial chain per row shrinks LPR-fold.
template <int LPR>
__device__ __forceinline__ float group_sum(float x) {
  if (LPR >= 2) x += __shfl_xor(x, 1);
  if (LPR >= 4) x += __shfl_xor(x, 2);
  if (LPR >= 8) x += __shfl_xor(x, 4);
  return x;
}
// lanes per row, measured (profiles/r02_config5_sweep.txt): 10k-atom Cu cell (17 partners per row) three-body forward / reverse
// 0.054 / 0.108 ms per step at 1 lane, 0.047 / 0.102 at 2, 0.052 / 0.130 at 4, 0.066 / 0.198 at 8; dense cell (58 partners per
// row) 24 / 86 us per launch at 1, 20 / 53 at 2, 19 / 50 at 4, 23 / 70 at 8
constexpr int kTbLprShort = 2, kTbLprLong = 4;

template <int L, int R, int LIST, int CAP, int LPR>
__global__ void __launch_bounds__(kTbRows * LPR) k_threebody_fwd(Consts c, TbArgs a) {
  constexpr int C = L * R;
  constexpr int kThreads = kTbRows * LPR;
  constexpr int kTbListCap = LIST * kTbRows;   // staged partner ids, one byte each
  // CAP <= kTbCap rows of the window are staged; the topology's byte ids count from the window start up to kTbCap, ids at or
  // beyond CAP take the global-memory path like the 255 marker
  __shared__ float su[CAP * 3];
  __shared__ float sp[CAP * C];
  __shared__ unsigned char s_other[kTbListCap];
  // The number of active rows A lives on the device: the grid is a fixed number of workgroups (at most what the chip holds at
  // once) that walk the row blocks, instead of one workgroup per block of the worst case A = E of which, on the benchmark cell,
  // 57 % were dispatched only to read A and leave.
  const int A = *a.n_act;
  for (int blk = blockIdx.x; blk * kTbRows < A; blk += gridDim.x) {
  const int rb = blk * kTbRows;
  const int lo = a.tb_win[6 * blk], hi_full = a.tb_win[6 * blk + 1];
  const int t_lo = a.tb_win[6 * blk + 2], t_hi = a.tb_win[6 * blk + 3];
  // the rows' partner lists are one contiguous range: staged once, coalesced, as window-relative byte ids (precomputed
  // in the topology) -- a global load per triplet inside the loop serialises ~17 L2 round trips per thread
  const int n_list = (t_hi - t_lo) < kTbListCap ? (t_hi - t_lo) : kTbListCap;
  for (int k = threadIdx.x; k < n_list; k += kThreads) s_other[k] = a.t_bytes[t_lo + k];
  const int sub = threadIdx.x % LPR;
  const int r = rb + (int)threadIdx.x / LPR;
  const bool live = r < A;
  const int64_t e = a.act_list[live ? r : A - 1];
  const int n = (hi_full - lo) < CAP ? (hi_full - lo) : CAP;
  for (int idx = threadIdx.x; idx < n; idx += kThreads) {
    const int64_t es = a.act_list[lo + idx], ks = a.act_dst[lo + idx];
    su[idx * 3 + 0] = a.u[es * 3];
    su[idx * 3 + 1] = a.u[es * 3 + 1];
    su[idx * 3 + 2] = a.u[es * 3 + 2];
    float qr[C], vr[C];
    load_row<C>(a.q + (int64_t)(lo + idx) * kCP, qr);   // compact rows: the window is contiguous
    load_row<C>(a.v + ks * kCP, vr);
#pragma unroll
    for (int cc = 0; cc < C; ++cc) sp[idx * C + cc] = qr[cc] * vr[cc];
  }
  // everything this row needs from global memory is requested before the barrier
  const int t0 = a.t_ptr[e], t1 = a.t_ptr[e + 1];
  const float ux = a.u[e * 3], uy = a.u[e * 3 + 1], uz = a.u[e * 3 + 2];
  const float fc = a.fc3[e];
  __syncthreads();
  float acc[C];
#pragma unroll
  for (int k = 0; k < C; ++k) acc[k] = 0.f;
  for (int t = live ? t0 + sub : t1; t < t1; t += LPR) {
    const int kk = t - t_lo;
    const int bid = kk < kTbListCap ? s_other[kk] : 255;
    const int idx = bid < CAP ? bid : a.t_other[t] - lo;
    float vx, vy, vz, pr[C];
    if (bid < CAP) {
      vx = su[idx * 3]; vy = su[idx * 3 + 1]; vz = su[idx * 3 + 2];
#pragma unroll
      for (int k = 0; k < C; ++k) pr[k] = sp[idx * C + k];
    } else {
      const int64_t eo = a.act_list[lo + idx], ko = a.act_dst[lo + idx];
      vx = a.u[eo * 3]; vy = a.u[eo * 3 + 1]; vz = a.u[eo * 3 + 2];
#pragma unroll
      for (int k = 0; k < C; ++k) pr[k] = a.q[(int64_t)(lo + idx) * kCP + k] * a.v[ko * kCP + k];
    }
    const float cs = fminf(1.f, fmaxf(-1.f, ux * vx + uy * vy + uz * vz));
    float P[L], dP[L];
    legendre<L>(cs, P, dP);
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float y = c.ynorm[l] * P[l];
#pragma unroll
      for (int nn = 0; nn < R; ++nn) acc[l * R + nn] += y * pr[l * R + nn];
    }
  }
#pragma unroll
  for (int k = 0; k < C; ++k) acc[k] = group_sum<LPR>(acc[k]);
  if (live && sub == 0) {
#pragma unroll
    for (int k = 0; k < kCP; k += 4) {
      float4 o;
      o.x = k + 0 < C ? fc * acc[k + 0 < C ? k + 0 : 0] : 0.f;
      o.y = k + 1 < C ? fc * acc[k + 1 < C ? k + 1 : 0] : 0.f;
      o.z = k + 2 < C ? fc * acc[k + 2 < C ? k + 2 : 0] : 0.f;
      o.w = k + 3 < C ? fc * acc[k + 3 < C ? k + 3 : 0] : 0.f;
      *(float4*)(a.m + (int64_t)r * kCP + k) = o;
    }
  }
  __syncthreads();   // the staged window is rewritten by the next row block
  }
}

// Reverse, both halves in one launch (rows = active edges, once as e1 and once as e2 of their triplets): the two passes
// share the staged unit vectors, the row data and -- what these latency-bound kernels pay most for -- the chain of
// dependent loads in front of the barrier.  Payload rows g (for the e1 half) and dS = fc*dm (for the e2 half) are both
// staged; partner ids are kept as one byte each (window-relative; 255 = outside the staged window, read from global).
struct TbRevArgs {
  int64_t E;
  const int32_t *act_list, *act_dst, *tb_win, *n_act;
  const int32_t *t1_ptr, *t1_other, *t2_ptr, *t2_other;
  const uint8_t *t1_bytes, *t2_bytes;
  const float *u, *fc3, *fc3p, *q, *qp, *v, *dm;
  float *dd, *du, *dgq;   // dd [A], du [A,3]: geometry gradients of the three-body term, one row per active edge
  int first;              // first reverse launch of the step (last block): dd/du are written, later launches accumulate
  int ref_legendre;       // option "legendre_backward" = 1: the reference's own backward of P_l (below)
};

// LegendreCosPolynomial.backward (nn/interaction.py:373-382) multiplies grad_output in at EVERY level of its recurrence,
//   grad_n = (n P_{n-1} + x grad_{n-1}) go   =>   grad_n = go k_n,  k_1 = 1,  k_n = n P_{n-1} + x go k_{n-1},
// which is the derivative P_n' go only for n <= 1 (SURVEY finding 2).  The engine computes the true derivative; with the option
// set, the list kernels return this k_n instead of P_n' so that forces and stresses reproduce the reference's own numbers.
// `go` is the gradient arriving at legendre_cos(cos, l)'s output for ONE triplet and ONE l (the reference calls it once per l).
template <int L>
__device__ __forceinline__ float legendre_ref_k(int l, float x, const float* P, float go) {
  if (l == 0) return 0.f;
  float k = 1.f;
#pragma unroll
  for (int n = 2; n < L; ++n)
    if (n <= l) k = (float)n * P[n - 1] + x * go * k;
  return k;
}

template <int L, int R, int LIST, int CAP, int LPR>
__global__ void __launch_bounds__(kTbRows * LPR) k_threebody_rev(Consts c, TbRevArgs a) {
  constexpr int C = L * R;
  constexpr int kThreads = kTbRows * LPR;
  constexpr int kTbRevList = LIST * kTbRows;   // staged partner ids per list (bytes)
  __shared__ float su[CAP * 3];
  __shared__ float sg[CAP * C];
  __shared__ float ss[CAP * C];
  __shared__ unsigned char s1[kTbRevList], s2[kTbRevList];
  const int A = *a.n_act;
  for (int blk = blockIdx.x; blk * kTbRows < A; blk += gridDim.x) {   // fixed grid walking the row blocks (see k_threebody_fwd)
  const int rb = blk * kTbRows;
  const int lo = a.tb_win[6 * blk], hi_full = a.tb_win[6 * blk + 1];
  const int t1_lo = a.tb_win[6 * blk + 2], t1_hi = a.tb_win[6 * blk + 3];
  const int t2_lo = a.tb_win[6 * blk + 4], t2_hi = a.tb_win[6 * blk + 5];
  const int sub = threadIdx.x % LPR;
  const int r = rb + (int)threadIdx.x / LPR;
  const bool live = r < A;
  const int rr = live ? r : A - 1;           // compact row of this thread
  const int64_t e = a.act_list[rr];
  const int64_t kd = a.act_dst[rr];
  const int n = (hi_full - lo) < CAP ? (hi_full - lo) : CAP;
  const int n1 = (t1_hi - t1_lo) < kTbRevList ? (t1_hi - t1_lo) : kTbRevList;
  const int n2 = (t2_hi - t2_lo) < kTbRevList ? (t2_hi - t2_lo) : kTbRevList;
  for (int k = threadIdx.x; k < n1; k += kThreads) s1[k] = a.t1_bytes[t1_lo + k];
  for (int k = threadIdx.x; k < n2; k += kThreads) s2[k] = a.t2_bytes[t2_lo + k];
  for (int idx = threadIdx.x; idx < n; idx += kThreads) {
    const int64_t es = a.act_list[lo + idx], ks = a.act_dst[lo + idx];
    su[idx * 3 + 0] = a.u[es * 3];
    su[idx * 3 + 1] = a.u[es * 3 + 1];
    su[idx * 3 + 2] = a.u[es * 3 + 2];
    const float f = a.fc3[es];
    float qr[C], vr[C], dr[C];
    load_row<C>(a.q + (int64_t)(lo + idx) * kCP, qr);    // q, dm: one row per active edge, the window is contiguous
    load_row<C>(a.v + ks * kCP, vr);
    load_row<C>(a.dm + (int64_t)(lo + idx) * kCP, dr);
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
      sg[idx * C + cc] = qr[cc] * vr[cc];
      ss[idx * C + cc] = f * dr[cc];
    }
  }
  // this row's own data, requested before the barrier
  const int t10 = a.t1_ptr[e], t11 = a.t1_ptr[e + 1], t20 = a.t2_ptr[e], t21 = a.t2_ptr[e + 1];
  const float ux = a.u[e * 3], uy = a.u[e * 3 + 1], uz = a.u[e * 3 + 2];
  const float fc = a.fc3[e], fcp = a.fc3p[e];
  float dd0 = 0.f, du0 = 0.f, du1 = 0.f, du2 = 0.f;
  if (!a.first) { dd0 = a.dd[rr]; du0 = a.du[(int64_t)rr * 3]; du1 = a.du[(int64_t)rr * 3 + 1]; du2 = a.du[(int64_t)rr * 3 + 2]; }
  float dmv[C], gv[C], qv[C], qpv[C], vv[C];
  load_row<C>(a.dm + (int64_t)rr * kCP, dmv);
  load_row<C>(a.q + (int64_t)rr * kCP, qv);
  load_row<C>(a.qp + (int64_t)rr * kCP, qpv);
  load_row<C>(a.v + kd * kCP, vv);
#pragma unroll
  for (int k = 0; k < C; ++k) gv[k] = qv[k] * vv[k];
  __syncthreads();
  // partner (window-relative id, or the global fallback) -> unit vector and payload row
  auto fetch = [&](int id, const int32_t* list, int t, const float* sp, bool want_s, float& vx, float& vy, float& vz, float* pr) {
    if (id < CAP) {
      vx = su[id * 3]; vy = su[id * 3 + 1]; vz = su[id * 3 + 2];
#pragma unroll
      for (int k = 0; k < C; ++k) pr[k] = sp[id * C + k];
    } else {
      const int64_t ro = list[t];
      const int64_t eo = a.act_list[ro], ko = a.act_dst[ro];
      vx = a.u[eo * 3]; vy = a.u[eo * 3 + 1]; vz = a.u[eo * 3 + 2];
      const float f = a.fc3[eo];
#pragma unroll
      for (int k = 0; k < C; ++k) pr[k] = want_s ? f * a.dm[ro * kCP + k] : a.q[ro * kCP + k] * a.v[ko * kCP + k];
    }
  };
  float S[C], dg[C];
#pragma unroll
  for (int k = 0; k < C; ++k) { S[k] = 0.f; dg[k] = 0.f; }
  float a1x = 0.f, a1y = 0.f, a1z = 0.f, a2x = 0.f, a2y = 0.f, a2z = 0.f;
  for (int t = live ? t10 + sub : t11; t < t11; t += LPR) {      // this edge as e1: S += Y g[e2], d cos += dm1 dY g[e2]
    const int kk = t - t1_lo;
    float vx, vy, vz, pr[C];
    fetch(kk < kTbRevList ? s1[kk] : 255, a.t1_other, t, sg, false, vx, vy, vz, pr);
    const float raw = ux * vx + uy * vy + uz * vz;
    const float cs = fminf(1.f, fmaxf(-1.f, raw));
    float P[L], dP[L];
    legendre<L>(cs, P, dP);
    float dcos = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float y = c.ynorm[l] * P[l], dy = c.ynorm[l] * dP[l];
#pragma unroll
      for (int nn = 0; nn < R; ++nn) {
        const int k = l * R + nn;
        S[k] += y * pr[k];
        dcos += dmv[k] * dy * pr[k];
      }
    }
    if (a.ref_legendre) {   // go = fc ynorm_l sum_n dm g[e2]; this row's fc multiplies the sum at the end
      dcos = 0.f;
#pragma unroll
      for (int l = 1; l < L; ++l) {
        float G = 0.f;
#pragma unroll
        for (int nn = 0; nn < R; ++nn) G += dmv[l * R + nn] * pr[l * R + nn];
        const float g1 = c.ynorm[l] * G;
        dcos += g1 * legendre_ref_k<L>(l, cs, P, fc * g1);
      }
    }
    dcos = (raw >= -1.f && raw <= 1.f) ? dcos : 0.f;   // torch.clamp passes the gradient only inside [-1, 1]
    a1x += dcos * vx; a1y += dcos * vy; a1z += dcos * vz;
  }
  for (int t = live ? t20 + sub : t21; t < t21; t += LPR) {      // this edge as e2: dg += dS[e1] Y, d cos += dS[e1] dY g_own
    const int kk = t - t2_lo;
    float vx, vy, vz, pr[C];
    fetch(kk < kTbRevList ? s2[kk] : 255, a.t2_other, t, ss, true, vx, vy, vz, pr);
    const float raw = ux * vx + uy * vy + uz * vz;
    const float cs = fminf(1.f, fmaxf(-1.f, raw));
    float P[L], dP[L];
    legendre<L>(cs, P, dP);
    float dcos = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float y = c.ynorm[l] * P[l], dy = c.ynorm[l] * dP[l];
#pragma unroll
      for (int nn = 0; nn < R; ++nn) {
        const int k = l * R + nn;
        dg[k] += pr[k] * y;
        dcos += pr[k] * dy * gv[k];
      }
    }
    if (a.ref_legendre) {   // pr = fc[e1] dm[e1]: go is complete
      dcos = 0.f;
#pragma unroll
      for (int l = 1; l < L; ++l) {
        float G = 0.f;
#pragma unroll
        for (int nn = 0; nn < R; ++nn) G += pr[l * R + nn] * gv[l * R + nn];
        const float go = c.ynorm[l] * G;
        dcos += go * legendre_ref_k<L>(l, cs, P, go);
      }
    }
    dcos = (raw >= -1.f && raw <= 1.f) ? dcos : 0.f;
    a2x += dcos * vx; a2y += dcos * vy; a2z += dcos * vz;
  }
  // combine the lane group's partial sums (S enters only through dfc = sum_k dm_k S_k, which is linear: reduce that scalar)
  float dfc = 0.f;
#pragma unroll
  for (int k = 0; k < C; ++k) { dfc += dmv[k] * S[k]; dg[k] = group_sum<LPR>(dg[k]); }
  dfc = group_sum<LPR>(dfc);
  a1x = group_sum<LPR>(a1x); a1y = group_sum<LPR>(a1y); a1z = group_sum<LPR>(a1z);
  a2x = group_sum<LPR>(a2x); a2y = group_sum<LPR>(a2y); a2z = group_sum<LPR>(a2z);
  if (live && sub == 0) {
    float ddv = 0.f, val[kCP];
#pragma unroll
    for (int k = 0; k < kCP; ++k) {
      val[k] = 0.f;
      if (k < C) {
        const int kc = k < C ? k : 0;
        ddv += dg[kc] * vv[kc] * qpv[kc];
        val[k] = dg[kc] * qv[kc];
      }
    }
#pragma unroll
    for (int k = 0; k < kCP; k += 4) *(float4*)(a.dgq + (int64_t)r * kCP + k) = float4{val[k], val[k + 1], val[k + 2], val[k + 3]};
    a.dd[r] = (dd0 + fcp * dfc) + ddv;
    a.du[(int64_t)r * 3] = (du0 + fc * a1x) + a2x;
    a.du[(int64_t)r * 3 + 1] = (du1 + fc * a1y) + a2y;
    a.du[(int64_t)r * 3 + 2] = (du2 + fc * a1z) + a2z;
  }
  __syncthreads();   // the staged window is rewritten by the next row block
  }
}

// ---- moment path -------------------------------------------------------------------------------------------------------------
// When the partner lists of a centre atom are COMPLETE (every ordered pair of its active edges, which is what the graph builders
// emit: data/material_graph.py:196-254) the sums over partners separate: P_l(u1.u2) is a polynomial in the components of u2, so
//   sum_{e2 != e1} P_l(u1.u2) g[e2,(l,n)] = contraction of u1 with the per-atom MOMENTS  G0[n] = sum g[.,(0,n)],
//   G1[n] = sum u g[.,(1,n)],  G2[n] = sum u u^T g[.,(2,n)],  G2s[n] = sum g[.,(2,n)]          -- minus the row's own term.
// 11 sums per radial index n and atom replace ~17 (58 in dense cells) Legendre evaluations per edge: the kernels become O(edges)
// instead of O(triplets), and what is left is the latency of their three dependent loads.  The reverse pass needs the same
// moments of the incoming gradient rows dS = fc dm.  Exactness: the polynomial is evaluated on the raw dot product instead of the
// clamped one (nn/invariant.py:40 clamps the rounding overshoot |u1.u2| - 1 <= ~1e-7 of collinear partners; the clamp's zero
// gradient there only removes a component parallel to u1, which the projection onto the displacement removes anyway); l_max <= 3
// only (higher l needs rank-3 moments: those models take the list path).  Whether a graph qualifies is decided in the topology
// build (Topo::tb_fast, m3g_topology_hints): every window complete -> these kernels, anything else (filtered, one-sided,
// permuted-with-duplicates lists) -> the list kernels above.  One thread per row, 128 rows per workgroup, the window of the
// workgroup's centre atoms staged in LDS as in the list kernels.
// moment j of one (atom, n): weight W[ia] W[ib] with W = (1, ux, uy, uz), payload channel l:  j = 0: G0 | 1-3: G1 | 4-9: G2 xx yy zz
// xy xz yz | 10: G2s
__device__ __forceinline__ int mom_ia(int j) { return (int)((0x02113210000ull >> (4 * j)) & 15); }
__device__ __forceinline__ int mom_ib(int j) { return (int)((0x03323213210ull >> (4 * j)) & 15); }
__device__ __forceinline__ int mom_l(int j) { return (int)((0x22222221110ull >> (4 * j)) & 15); }
template <int L>
constexpr int mom_count() { return L == 1 ? 1 : L == 2 ? 4 : 11; }
// moments of the staged payload rows `pay` ([rows][C]) for the atoms of the window (row ranges s_arow[at] .. s_arow[at + 1]); fixed
// summation order -> reproducible
template <int L, int R, int THREADS>
__device__ __forceinline__ void window_moments(int na, const int* s_arow, const float* su4, const float* pay, float* mom) {
  constexpr int C = L * R, NM = mom_count<L>();
  for (int p = threadIdx.x; p < na * R * NM; p += THREADS) {
    const int at = p / (R * NM), rem = p % (R * NM), nn = rem / NM, j = rem % NM;
    const int ia = mom_ia(j), ib = mom_ib(j), ch = mom_l(j) * R + nn;
    // four rows in flight (the loop is a chain of LDS round trips otherwise); the sum keeps its row order
    const int i1 = s_arow[at + 1];
    float acc = 0.f;
    int i = s_arow[at];
    for (; i + 4 <= i1; i += 4) {
      float w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = su4[(i + k) * 4 + ia] * su4[(i + k) * 4 + ib] * pay[(i + k) * C + ch];
      acc = (((acc + w[0]) + w[1]) + w[2]) + w[3];
    }
    for (; i < i1; ++i) acc += su4[i * 4 + ia] * su4[i * 4 + ib] * pay[i * C + ch];
    mom[p] = acc;
  }
}
// contraction of the moments M (one n) with the row's own direction, own term subtracted: S_l = sum over the OTHER rows of
// P_l(u.u') pay';  for l >= 1 also the vector  V_l = sum over the other rows of P_l'(u.u') u' pay'  (gradient with respect to u)
struct MomEval { float S0, S1, S2; float V1x, V1y, V1z, V2x, V2y, V2z; };
template <int L>
__device__ __forceinline__ MomEval mom_eval(const float* M, float ux, float uy, float uz, float s, float p0, float p1, float p2) {
  MomEval r{};
  r.S0 = M[0] - p0;
  if constexpr (L >= 2) {
    r.S1 = (ux * M[1] + uy * M[2] + uz * M[3]) - s * p1;
    r.V1x = M[1] - ux * p1; r.V1y = M[2] - uy * p1; r.V1z = M[3] - uz * p1;
  }
  if constexpr (L >= 3) {
    const float gx = M[4] * ux + M[7] * uy + M[8] * uz, gy = M[7] * ux + M[5] * uy + M[9] * uz, gz = M[8] * ux + M[9] * uy + M[6] * uz;
    const float quad = ux * gx + uy * gy + uz * gz;
    r.S2 = 1.5f * (quad - s * s * p2) - 0.5f * (M[10] - p2);
    const float sp = s * p2;
    r.V2x = 3.f * (gx - sp * ux); r.V2y = 3.f * (gy - sp * uy); r.V2z = 3.f * (gz - sp * uz);
  }
  return r;
}

struct TbMomArgs {
  int blocks;   // E / kTbRows + 1: block indices with a window record
  const int32_t *src, *arow_ptr, *tb_fast, *act_list, *act_dst, *tb_win, *n_act;
  int32_t* flags;   // Topo::flags: [7] the hints word certified for this buffer, [8] sticky error
  int hints;        // the word the caller handed in
  const float *u, *fc3, *fc3p, *q, *qp, *v, *dm;   // reverse only: fc3p, qp, dm
  float* m;                                        // forward out
  float *dd, *du, *dgq;                            // reverse out
  int first;
  // FINAL (the step's last three-body reverse, block 0): dE/dr of every edge is formed here as well (k_geometry_reverse's work)
  GeomRev geom;
  float* dr;
};

// LDS of one workgroup for windows of at most `rows` rows over at most `atoms` centre atoms (both from the topology hints)
template <int L, int R, bool REV>
constexpr size_t mom_lds_bytes(int rows, int atoms) {
  return sizeof(float) * ((size_t)rows * 4 + (size_t)rows * L * R * (REV ? 2 : 1) + (size_t)(REV ? 2 : 1) * atoms * R * mom_count<L>()) + sizeof(int) * (atoms + 1);
}
// THREADS >= kTbRows threads work on a block of kTbRows rows: all of them stage the window and form the moments, the first kTbRows
// own a row each (the stand-alone kernel has THREADS = kTbRows; as a role of k_node_tb_reverse the workgroup has 256).
// vblock / vgrid: this workgroup's index among the `vgrid` workgroups that walk the row blocks.
template <int L, int R, bool REV, int THREADS, bool FINAL = false>
__device__ __forceinline__ void tb_moments_body(const Consts& c, const TbMomArgs& a, int cap_rows, int cap_atoms, int vblock, int vgrid, float* lds_mom) {
  constexpr int C = L * R, NM = mom_count<L>(), kThreads = THREADS;
  static_assert(THREADS >= kTbRows, "a thread per row");
  float* su = lds_mom;                                        // (1, ux, uy, uz) per window row
  float* sg = su + cap_rows * 4;                              // g = q v[dst]
  float* ss = sg + cap_rows * C;                              // dS = fc dm (reverse)
  float* s_mom = ss + (REV ? cap_rows * C : 0);               // moments of g, then (reverse) of dS
  int* s_arow = reinterpret_cast<int*>(s_mom + (REV ? 2 : 1) * cap_atoms * R * NM);
  // The launch sizes (cap_rows, cap_atoms -> the LDS carve-up above) come from the caller's hints word; they are only valid for the
  // topology buffer that word was certified for.  Anything else flags an error and touches nothing.
  if (a.flags[7] != a.hints) {
    if (vblock == 0 && threadIdx.x == 0) atomicOr(a.flags + 8, M3G_TOPO_ERR_HINTS);
    return;
  }
  // (the row count A lives on the device; a workgroup beyond it finds an empty window -- k_tb_windows writes one for every block
  // index the grid can reach -- so nothing here waits for A: one dependent load less in a kernel that is a chain of them)
  for (int blk = vblock; blk < a.blocks; blk += vgrid) {
    const int rb = blk * kTbRows;
    const int lo = a.tb_win[6 * blk], n = a.tb_win[6 * blk + 1] - lo;
    if (n <= 0) break;   // windows are in row order: the first empty one ends the list
    const int A = lo + n;   // rows of this block are < lo + n (the window covers them), rows >= A of the last block are not
    const int na = a.tb_fast[2 * blk], a0 = a.tb_fast[2 * blk + 1];   // (the launch is only made for graphs whose every window
                                                                     // qualifies, with the sizes from the certified hints)
    if (na <= 0 || na > cap_atoms || n > cap_rows) {                  // cannot happen with a certified word: never overrun the LDS
      if (threadIdx.x == 0) atomicOr(a.flags + 8, M3G_TOPO_ERR_HINTS);
      continue;
    }
    const int r = rb + (int)threadIdx.x;
    const bool live = (int)threadIdx.x < kTbRows && r < A;
    const int rr = live ? r : A - 1;
    const int64_t e = a.act_list[rr];
    const int64_t kd = a.act_dst[rr];
    for (int idx = threadIdx.x; idx < n; idx += kThreads) {
      const int64_t es = a.act_list[lo + idx], ks = a.act_dst[lo + idx];
      su[idx * 4 + 0] = 1.f;
      su[idx * 4 + 1] = a.u[es * 3];
      su[idx * 4 + 2] = a.u[es * 3 + 1];
      su[idx * 4 + 3] = a.u[es * 3 + 2];
      float qr[C], vr[C];
      load_row<C>(a.q + (int64_t)(lo + idx) * kCP, qr);
      load_row<C>(a.v + ks * kCP, vr);
#pragma unroll
      for (int cc = 0; cc < C; ++cc) sg[idx * C + cc] = qr[cc] * vr[cc];
      if constexpr (REV) {
        const float f = a.fc3[es];
        float dr[C];
        load_row<C>(a.dm + (int64_t)(lo + idx) * kCP, dr);
#pragma unroll
        for (int cc = 0; cc < C; ++cc) ss[idx * C + cc] = f * dr[cc];
      }
    }
    if ((int)threadIdx.x <= na) s_arow[threadIdx.x] = a.arow_ptr[a0 + threadIdx.x] - lo;
    // this row's own data, requested before the barrier
    const int at = a.src[e] - a0;
    const float fc = a.fc3[e];
    float fcp = 0.f, dd0 = 0.f, du0 = 0.f, du1 = 0.f, du2 = 0.f;
    float dmv[C], qv[C], qpv[C], vv[C];
    if constexpr (REV) {
      fcp = a.fc3p[e];
      if (!a.first) { dd0 = a.dd[rr]; du0 = a.du[(int64_t)rr * 3]; du1 = a.du[(int64_t)rr * 3 + 1]; du2 = a.du[(int64_t)rr * 3 + 2]; }
      load_row<C>(a.dm + (int64_t)rr * kCP, dmv);
      load_row<C>(a.q + (int64_t)rr * kCP, qv);
      load_row<C>(a.qp + (int64_t)rr * kCP, qpv);
      load_row<C>(a.v + kd * kCP, vv);
    }
    __syncthreads();
    window_moments<L, R, kThreads>(na, s_arow, su, sg, s_mom);
    if constexpr (REV) window_moments<L, R, kThreads>(na, s_arow, su, ss, s_mom + cap_atoms * R * NM);
    __syncthreads();
    if (live) {
      const int own = r - lo;
      const float ux = su[own * 4 + 1], uy = su[own * 4 + 2], uz = su[own * 4 + 3], s2 = ux * ux + uy * uy + uz * uz;
      const float* g = sg + own * C;
      if constexpr (!REV) {
        float acc[kCP];
#pragma unroll
        for (int k = 0; k < kCP; ++k) acc[k] = 0.f;
#pragma unroll
        for (int nn = 0; nn < R; ++nn) {
          const MomEval ev = mom_eval<L>(s_mom + (at * R + nn) * NM, ux, uy, uz, s2, g[nn], L >= 2 ? g[R + nn] : 0.f, L >= 3 ? g[2 * R + nn] : 0.f);
          acc[nn] = fc * c.ynorm[0] * ev.S0;
          if constexpr (L >= 2) acc[R + nn] = fc * c.ynorm[1] * ev.S1;
          if constexpr (L >= 3) acc[2 * R + nn] = fc * c.ynorm[2] * ev.S2;
        }
#pragma unroll
        for (int k = 0; k < kCP; k += 4) *(float4*)(a.m + (int64_t)r * kCP + k) = float4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
      } else {
        const float* ds = ss + own * C;   // fc dm of this row
        float dg[kCP], dfc = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int k = 0; k < kCP; ++k) dg[k] = 0.f;
#pragma unroll
        for (int nn = 0; nn < R; ++nn) {
          const float g0 = g[nn], g1 = L >= 2 ? g[R + nn] : 0.f, g2 = L >= 3 ? g[2 * R + nn] : 0.f;
          const float d0 = ds[nn], d1 = L >= 2 ? ds[R + nn] : 0.f, d2 = L >= 3 ? ds[2 * R + nn] : 0.f;
          const MomEval eg = mom_eval<L>(s_mom + (at * R + nn) * NM, ux, uy, uz, s2, g0, g1, g2);
          const MomEval eh = mom_eval<L>(s_mom + ((cap_atoms + at) * R + nn) * NM, ux, uy, uz, s2, d0, d1, d2);
          dfc += dmv[nn] * (c.ynorm[0] * eg.S0);
          dg[nn] = c.ynorm[0] * eh.S0;
          if constexpr (L >= 2) {
            dfc += dmv[R + nn] * (c.ynorm[1] * eg.S1);
            dg[R + nn] = c.ynorm[1] * eh.S1;
            ax += c.ynorm[1] * (d1 * eg.V1x + g1 * eh.V1x);
            ay += c.ynorm[1] * (d1 * eg.V1y + g1 * eh.V1y);
            az += c.ynorm[1] * (d1 * eg.V1z + g1 * eh.V1z);
          }
          if constexpr (L >= 3) {
            dfc += dmv[2 * R + nn] * (c.ynorm[2] * eg.S2);
            dg[2 * R + nn] = c.ynorm[2] * eh.S2;
            ax += c.ynorm[2] * (d2 * eg.V2x + g2 * eh.V2x);
            ay += c.ynorm[2] * (d2 * eg.V2y + g2 * eh.V2y);
            az += c.ynorm[2] * (d2 * eg.V2z + g2 * eh.V2z);
          }
        }
        float ddv = 0.f, val[kCP];
#pragma unroll
        for (int k = 0; k < kCP; ++k) {
          val[k] = 0.f;
          if (k < C) {
            const int kc = k < C ? k : 0;
            ddv += dg[kc] * vv[kc] * qpv[kc];
            val[k] = dg[kc] * qv[kc];
          }
        }
#pragma unroll
        for (int k = 0; k < kCP; k += 4) *(float4*)(a.dgq + (int64_t)r * kCP + k) = float4{val[k], val[k + 1], val[k + 2], val[k + 3]};
        a.dd[r] = (dd0 + fcp * dfc) + ddv;
        a.du[(int64_t)r * 3] = du0 + ax;
        a.du[(int64_t)r * 3 + 1] = du1 + ay;
        a.du[(int64_t)r * 3 + 2] = du2 + az;
        if constexpr (FINAL) {
          // dd / du of this edge are final (nothing adds to them after block 0's three-body reverse): its dE/dr right away -- the
          // same edge_dr k_geometry_reverse runs, reading back this thread's own stores
          float rx, ry, rz;
          edge_dr(a.geom, e, rx, ry, rz);
          a.dr[e * 3] = rx; a.dr[e * 3 + 1] = ry; a.dr[e * 3 + 2] = rz;
        }
      }
    }
    __syncthreads();   // the staged window is rewritten by the next row block
  }
  if constexpr (FINAL) {
    // edges without a three-body row (beyond the three-body cutoff, or without a partner): dE/dr from the radial-basis share alone
    for (int64_t e = (int64_t)vblock * THREADS + threadIdx.x; e < a.geom.E; e += (int64_t)vgrid * THREADS) {
      if (a.geom.act_id[e] >= 0) continue;
      float rx, ry, rz;
      edge_dr(a.geom, e, rx, ry, rz);
      a.dr[e * 3] = rx; a.dr[e * 3 + 1] = ry; a.dr[e * 3 + 2] = rz;
    }
  }
}
#ifndef M3G_TB_MOM_THREADS
#define M3G_TB_MOM_THREADS 128   // (256 measured: forward -3.6 us, reverse +2 us per step on the 10k-atom cell -- no gain) threads per workgroup of the stand-alone moment kernels (kTbRows of them own a row; all stage and sum)
#endif
constexpr int kTbMomThreads = M3G_TB_MOM_THREADS;
template <int L, int R, bool REV>
__global__ void __launch_bounds__(kTbMomThreads) k_threebody_moments(Consts c, TbMomArgs a, int cap_rows, int cap_atoms) {
  extern __shared__ __attribute__((aligned(16))) float lds_mom[];
  tb_moments_body<L, R, REV, kTbMomThreads>(c, a, cap_rows, cap_atoms, (int)blockIdx.x, (int)gridDim.x, lds_mom);
}
// the step's LAST three-body reverse (block 0) also forms dE/dr of every edge: one launch less (k_geometry_reverse)
template <int L, int R>
__global__ void __launch_bounds__(kTbMomThreads) k_threebody_moments_final(Consts c, TbMomArgs a, int cap_rows, int cap_atoms) {
  extern __shared__ __attribute__((aligned(16))) float lds_mom[];
  tb_moments_body<L, R, true, kTbMomThreads, true>(c, a, cap_rows, cap_atoms, (int)blockIdx.x, (int)gridDim.x, lds_mom);
}

// Three-body reverse (moment path) and node reverse of one block as the two workgroup ROLES of one launch.  Both consume what
// k_edge_rev_* of the block wrote (dL/dm resp. the dL/dp1 rows) and neither fills the chip for long: the three-body reverse is a
// chain of dependent loads (25 us on the 10k-atom cell at a fifth of the HBM rate), the node reverse an HBM gather (100 us).
//   * workgroups [0, n_tb) take the three-body role, then publish: device-scope fence + one atomic increment of `done`;
//   * workgroups [n_tb, ...) take the node role: the dp1 gather needs nothing from the three-body role; its last term (the
//     v-gradient, from the dL/dg rows the three-body role writes) waits until `done` == n_tb.
// No deadlock: the launcher takes this form only when EVERY workgroup of the launch is resident at once (n_tb + n_node <= what the
// device holds of this kernel at its LDS footprint, asked of the runtime -- launch_node_tb_reverse; otherwise the two launches), so
// the three-body workgroups a node workgroup waits for are running, and those never wait for anything.  The wait is bounded all the
// same: when it runs out the node role replaces the v-gradient term it waited for by NaN (forces of the call: NaN) and sets the
// sticky M3G_TOPO_ERR_SYNC bit -- never stale rows.
struct NodeTbArgs {
  int n_tb;          // workgroups of the three-body role
  int32_t* done;     // Work::sync word of this block (cleared at the start of every step)
  int max_polls;     // bound of the node role's wait (2^22 polls ~ 2 s; tests shrink it through option "debug_node_tb_polls")
  int wait_extra;    // tests: the node role waits for this many increments MORE than will ever come (its time-out path)
};
template <int L, int R>
__global__ void __launch_bounds__(256) k_node_tb_reverse(Consts c, TbMomArgs ta, int cap_rows, int cap_atoms, NodeRevArgs na, NodeTbArgs f) {
  extern __shared__ __attribute__((aligned(16))) float lds_mom[];
  if ((int)blockIdx.x < f.n_tb) {
    tb_moments_body<L, R, true, 256>(c, ta, cap_rows, cap_atoms, (int)blockIdx.x, f.n_tb, lds_mom);
    // this workgroup's dL/dg rows are visible device-wide (release at agent scope = write-back of its XCD's L2: the L2s of different
    // XCDs are not coherent with each other for ordinary device memory; agent-scope stores / loads alone do NOT make the rows
    // visible -- tried, stale rows) ...
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(f.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before it counts as done
    return;
  }
  node_reverse_body<true, true>(na, (int64_t)blockIdx.x - f.n_tb, [&] {
    // ONE lane per wave polls (64 lanes polling one word from every waiting wave starve the increments they wait for: 73 us
    // instead of 24 on the 864-atom cell), ~0.5 us apart
    int never_came = 0;
    if ((threadIdx.x & 63) == 0) {
      int spins = 0;
      while (__hip_atomic_load(f.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < f.n_tb + f.wait_extra) {
        __builtin_amdgcn_s_sleep(16);
        if (++spins > f.max_polls) { atomicOr(ta.flags + 8, M3G_TOPO_ERR_SYNC); never_came = 1; break; }   // (cannot happen: see above)
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the rows read below are the ones published before the increments
    return __builtin_amdgcn_readfirstlane(never_came) != 0;
  });
}

// at most kTbGridCap workgroups (what 256 CUs hold at once at this LDS footprint and then some): they walk the row blocks
#ifndef M3G_TB_GRID_CAP
#define M3G_TB_GRID_CAP 2048
#endif
static inline dim3 grid_rows(int64_t n) {
  const int64_t blocks = (n + kTbRows - 1) / kTbRows;
  return dim3((unsigned)(blocks < M3G_TB_GRID_CAP ? blocks : M3G_TB_GRID_CAP));
}
// Long partner lists?  Triplets per edge is a host-side lower bound of triplets per ACTIVE edge (the number of active edges
// lives on the device); either choice is correct, the wrong one only costs time (global-memory fallback or LDS footprint).
static inline bool long_lists(const Topo& t) { return t.T > 24 * t.E; }

// the moment kernels apply when the topology build found every window complete (hint bit, read back by the caller once per
// topology: m3g_topology_hints) and l_max <= 3
static inline bool use_moments(const Consts& c, int topo_hints) {
  // (L, R outside M3G_DISPATCH_LR3's cases take the list kernels)
  return (topo_hints & M3G_TOPO_TB_COMPLETE) && c.L >= 1 && c.L <= 3 && c.R >= 1 && c.R <= 4 && ((topo_hints >> 8) & 0xff) > 0 &&
         ((topo_hints >> 16) & 0xff) > 0;
}
#define M3G_DISPATCH_LR3(Lv, Rv, BODY)                          \
  switch ((Lv) * 8 + (Rv)) {                                    \
    case 1 * 8 + 1: { constexpr int L = 1, R = 1; BODY; } break; \
    case 1 * 8 + 2: { constexpr int L = 1, R = 2; BODY; } break; \
    case 1 * 8 + 3: { constexpr int L = 1, R = 3; BODY; } break; \
    case 1 * 8 + 4: { constexpr int L = 1, R = 4; BODY; } break; \
    case 2 * 8 + 1: { constexpr int L = 2, R = 1; BODY; } break; \
    case 2 * 8 + 2: { constexpr int L = 2, R = 2; BODY; } break; \
    case 2 * 8 + 3: { constexpr int L = 2, R = 3; BODY; } break; \
    case 2 * 8 + 4: { constexpr int L = 2, R = 4; BODY; } break; \
    case 3 * 8 + 1: { constexpr int L = 3, R = 1; BODY; } break; \
    case 3 * 8 + 2: { constexpr int L = 3, R = 2; BODY; } break; \
    case 3 * 8 + 3: { constexpr int L = 3, R = 3; BODY; } break; \
    case 3 * 8 + 4: { constexpr int L = 3, R = 4; BODY; } break; \
    default: break;                                             \
  }

void launch_threebody(const Consts& c, const Topo& t, const Work& w, const float* v, float* m, hipStream_t s, int topo_hints) {
  if (t.E == 0) return;
  if (t.T == 0) return;   // no active edge: every consumer reads zeros through act_id < 0
  if (use_moments(c, topo_hints)) {
    TbMomArgs a{(int)(t.E / kTbRows + 1), t.src, t.arow_ptr, t.tb_fast, t.act_list, t.act_dst, t.tb_win, t.n_act, t.flags, topo_hints, w.u, w.fc3, nullptr, w.q, nullptr, v, nullptr, m, nullptr, nullptr, nullptr, 0};
    const int rows = (topo_hints >> 8) & 0xff, atoms = (topo_hints >> 16) & 0xff;
    M3G_DISPATCH_LR3(c.L, c.R, hipLaunchKernelGGL((k_threebody_moments<L, R, false>), grid_rows(t.E), dim3(kTbMomThreads), (mom_lds_bytes<L, R, false>(rows, atoms)), s, c, a, rows, atoms));
    return;
  }
  TbArgs a{t.E, t.act_list, t.act_dst, t.tb_win, t.n_act, t.t1_ptr, t.t1_e2c, t.t1_b, w.u, w.fc3, w.q, v, m};
  if (long_lists(t)) { M3G_DISPATCH_LR(c.L, c.R, hipLaunchKernelGGL((k_threebody_fwd<L, R, kTbListLong, kTbCap, kTbLprLong>), grid_rows(t.E), dim3(kTbRows * kTbLprLong), 0, s, c, a)); }
  else { M3G_DISPATCH_LR(c.L, c.R, hipLaunchKernelGGL((k_threebody_fwd<L, R, kTbListShort, kTbCapShort, kTbLprShort>), grid_rows(t.E), dim3(kTbRows * kTbLprShort), 0, s, c, a)); }
}

// The in-launch wait of k_node_tb_reverse is deadlock-free only when all workgroups of the launch are resident together: ask the
// runtime how many this kernel fits per CU at this LDS footprint (cached per instantiation, device and footprint); false = take the
// two launches.
template <int L, int R>
static bool node_tb_launch(const Consts& c, const TbMomArgs& a, int rows, int atoms, const NodeRevArgs& na, int n_tb, int n_node, int32_t* done,
                           int debug_polls, hipStream_t s) {
  const size_t lds = mom_lds_bytes<L, R, true>(rows, atoms);
  static std::mutex mu;
  static std::map<std::pair<int, size_t>, int> resident;   // (device, dynamic LDS bytes) -> workgroups the device holds at once
  int dev = 0, cap = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = resident.find(std::make_pair(dev, lds));
    if (it == resident.end()) {
      int per_cu = 0, cus = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_node_tb_reverse<L, R>, 256, lds) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        per_cu = cus = 0;
      it = resident.emplace(std::make_pair(dev, lds), per_cu * cus).first;
    }
    cap = it->second;
  }
  if (n_tb + n_node > cap) return false;
  // (debug_polls, tests only: k > 0 bounds the wait at k polls; k < 0 makes the node role wait for an increment that never comes, |k| polls)
  const NodeTbArgs f{n_tb, done, debug_polls > 0 ? debug_polls : debug_polls < 0 ? -debug_polls : 1 << 22, debug_polls < 0 ? 1 : 0};
  hipLaunchKernelGGL((k_node_tb_reverse<L, R>), dim3((unsigned)(n_tb + n_node)), dim3(256), lds, s, c, a, rows, atoms, na, f);
  return true;
}

// three-body reverse + node reverse of a block in one launch (k_node_tb_reverse); false: not applicable (list kernels, no atoms, no
// sync words) -- the caller then launches the two kernels
bool launch_node_tb_reverse(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w, const float* v, bool first,
                            const float* dx_new, float* dx_out, int dp1_packed, int block, hipStream_t s, int topo_hints, int debug_polls) {
  // Small cells only.  Measured: 32 atoms 20.7 -> 15.8 us for the pair, 108 atoms a gain, 256 atoms a small loss, 864 atoms 24 -> 47 us,
  // 10,000 atoms 2.40 -> 2.62 ms per step: publishing costs an L2 write-back per three-body workgroup and an L2 invalidate per
  // waiting wave (the XCDs' L2s are not coherent with each other), which a launch boundary does once for everybody.
  if (t.N > kNodeTbFusedMaxAtoms) return false;
  if (t.E == 0 || t.T == 0 || t.N == 0 || !w.sync || !use_moments(c, topo_hints) || kSyncNodeRev + block >= kSyncWords) return false;
  TbMomArgs a{(int)(t.E / kTbRows + 1), t.src, t.arow_ptr, t.tb_fast, t.act_list, t.act_dst, t.tb_win, t.n_act, t.flags, topo_hints, w.u, w.fc3, w.fc3p, w.q, w.qp, v, w.dm, nullptr, w.dd, w.du, w.dg, first ? 1 : 0};
  const int rows = (topo_hints >> 8) & 0xff, atoms = (topo_hints >> 16) & 0xff;
  const NodeRevArgs na = node_rev_args(c, W, bw, t, w, v, dx_new, dx_out, /*row_sums_in_seg=*/true, dp1_packed, /*with_v_term=*/true);
  // the three-body role walks its row blocks with at most 128 workgroups: each publishes once, and a publish is a write-back of
  // its XCD's L2 (one per row block -- 284 on the 864-atom cell -- cost 59 us instead of the 24 us of the two separate kernels);
  // beside the node role's gather a longer walk costs nothing
  const int n_tb = (int)std::min<unsigned>(grid_rows(t.E).x, 128u), n_node = (int)((t.N + kNodesRev - 1) / kNodesRev);
  bool launched = false;
  M3G_DISPATCH_LR3(c.L, c.R, launched = (node_tb_launch<L, R>(c, a, rows, atoms, na, n_tb, n_node, w.sync + kSyncNodeRev + block, debug_polls, s)));
  return launched;
}

// the step's last three-body reverse + the geometry reverse (dE/dr of every edge) in one launch; false: not the moment path / no
// triplets -- the caller then launches launch_threebody_reverse and k_geometry_reverse
bool launch_threebody_reverse_final(const Consts& c, const Topo& t, const Work& w, const float* v, bool first, const float* dh, int dh_parts,
                                    hipStream_t s, int topo_hints) {
  if (t.E == 0 || t.T == 0 || c.B == 0 || !use_moments(c, topo_hints)) return false;
  TbMomArgs a{(int)(t.E / kTbRows + 1), t.src, t.arow_ptr, t.tb_fast, t.act_list, t.act_dst, t.tb_win, t.n_act, t.flags, topo_hints, w.u, w.fc3, w.fc3p, w.q, w.qp, v, w.dm, nullptr, w.dd, w.du, w.dg, first ? 1 : 0,
              GeomRev{t.E, w.u, w.d, w.hp, dh, dh_parts, w.dd, w.du, t.act_id}, w.dr};
  const int rows = (topo_hints >> 8) & 0xff, atoms = (topo_hints >> 16) & 0xff;
  M3G_DISPATCH_LR3(c.L, c.R, hipLaunchKernelGGL((k_threebody_moments_final<L, R>), grid_rows(t.E), dim3(kTbMomThreads), (mom_lds_bytes<L, R, true>(rows, atoms)), s, c, a, rows, atoms));
  return true;
}

void launch_threebody_reverse(const Consts& c, const Topo& t, const Work& w, const float* v, bool first, hipStream_t s, int topo_hints,
                              bool ref_legendre) {
  if (t.E == 0) return;
  if (t.T == 0) return;
  if (use_moments(c, topo_hints)) {
    TbMomArgs a{(int)(t.E / kTbRows + 1), t.src, t.arow_ptr, t.tb_fast, t.act_list, t.act_dst, t.tb_win, t.n_act, t.flags, topo_hints, w.u, w.fc3, w.fc3p, w.q, w.qp, v, w.dm, nullptr, w.dd, w.du, w.dg, first ? 1 : 0};
    const int rows = (topo_hints >> 8) & 0xff, atoms = (topo_hints >> 16) & 0xff;
    M3G_DISPATCH_LR3(c.L, c.R, hipLaunchKernelGGL((k_threebody_moments<L, R, true>), grid_rows(t.E), dim3(kTbMomThreads), (mom_lds_bytes<L, R, true>(rows, atoms)), s, c, a, rows, atoms));
    return;
  }
  TbRevArgs a{t.E, t.act_list, t.act_dst, t.tb_win, t.n_act, t.t1_ptr, t.t1_e2c, t.t2_ptr, t.t2_e1c, t.t1_b, t.t2_b, w.u, w.fc3, w.fc3p, w.q,
              w.qp, v,
              w.dm, w.dd, w.du, w.dg, first ? 1 : 0, ref_legendre ? 1 : 0};
  if (long_lists(t)) { M3G_DISPATCH_LR(c.L, c.R, hipLaunchKernelGGL((k_threebody_rev<L, R, kTbListLong, kTbCap, kTbLprLong>), grid_rows(t.E), dim3(kTbRows * kTbLprLong), 0, s, c, a)); }
  else { M3G_DISPATCH_LR(c.L, c.R, hipLaunchKernelGGL((k_threebody_rev<L, R, kTbListShort, kTbCapShort, kTbLprShort>), grid_rows(t.E), dim3(kTbRows * kTbLprShort), 0, s, c, a)); }
}

}  // namespace m3g
