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
// lanes idle.  One thread per active row, 256 consecutive rows per workgroup.  The partners of those rows are
// active edges of the same centre atoms, i.e. one contiguous window of the compacted list: the workgroup stages that
// window's unit vectors and per-edge payload rows (g = q*v[dst] for the forward / e1 half, dS = fc*dm for the e2
// half) in LDS once -- the per-atom triplet tile -- and the triplet loops read LDS instead of gathering from
// L2.  Partners outside the staged window (only possible for degrees beyond the LDS budget) fall back to global
// memory, so correctness does not depend on the window size.
#include "m3g_internal.h"

namespace m3g {

constexpr int kTbListCap = 3072;   // staged partner ids (128 rows x 24 partners; longer lists continue from global memory)
constexpr int kTbCap = 192;    // staged window capacity: kTbRows (128) rows + boundary rows (overflow -> global reads); 9 KB per workgroup

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
  const int32_t *t_ptr, *t_other;     // triplet CSR of this pass (by e1: partner = e2; by e2: partner = e1), partners as compacted ids
  const float *u, *fc3, *fc3p, *q, *qp, *v;
  const float* dm;                    // reverse: dL/dm [E][kCP]
  float* m;                           // forward out [E][kCP]
  float *dd, *du, *dgq;               // reverse in/out
};

// payload of edge e (neighbour atom k) for this pass: MODE 0/1 -> g[e,:] = q*v[k];  MODE 2 -> dS[e,:] = fc3*dm
template <int C, int MODE>
__device__ __forceinline__ void payload(const TbArgs& a, int64_t e, int64_t k, float* out) {
  if (MODE == 2) {
    const float f = a.fc3[e];
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = f * a.dm[e * kCP + c];
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = a.q[e * kCP + c] * a.v[k * kCP + c];
  }
}

// MODE 0: forward (rows = e1);  MODE 1: reverse e1 half (d fc(d_e1), d u_e1);  MODE 2: reverse e2 half
template <int L, int R, int MODE>
__global__ void __launch_bounds__(kTbRows) k_threebody_tile(Consts c, TbArgs a) {
  constexpr int C = L * R;
  __shared__ float su[kTbCap * 3];
  __shared__ float sp[kTbCap * C];
  // independent first-level loads: A, this workgroup's window, this thread's row
  const int A = *a.n_act;
  const int rb = blockIdx.x * kTbRows;
  if (rb >= A) return;                       // the grid is sized for the worst case A = E
  __shared__ int s_other[kTbListCap];
  const int lo = a.tb_win[6 * blockIdx.x];
  const int hi_full = a.tb_win[6 * blockIdx.x + 1];
  const int t_lo = a.tb_win[6 * blockIdx.x + (MODE == 2 ? 4 : 2)];
  const int t_hi = a.tb_win[6 * blockIdx.x + (MODE == 2 ? 5 : 3)];
  // the rows' partner lists are one contiguous range of t_other: staged once, coalesced, as window-relative ids --
  // a global load per triplet inside the loop serialises ~17 L2 round trips per thread
  const int n_list = (t_hi - t_lo) < kTbListCap ? (t_hi - t_lo) : kTbListCap;
  for (int k = threadIdx.x; k < n_list; k += kTbRows) s_other[k] = a.t_other[t_lo + k] - lo;
  const int r = rb + threadIdx.x;
  const bool live = r < A;
  const int64_t e = a.act_list[live ? r : A - 1];
  const int64_t e_next = r + 1 < A ? a.act_list[live ? r + 1 : A - 1] : a.E;
  const int64_t kd = a.act_dst[live ? r : A - 1];
  const int n = (hi_full - lo) < kTbCap ? (hi_full - lo) : kTbCap;
  for (int idx = threadIdx.x; idx < n; idx += kTbRows) {
    const int64_t es = a.act_list[lo + idx];
    su[idx * 3 + 0] = a.u[es * 3];
    su[idx * 3 + 1] = a.u[es * 3 + 1];
    su[idx * 3 + 2] = a.u[es * 3 + 2];
    float row[C];
    payload<C, MODE>(a, es, a.act_dst[lo + idx], row);
#pragma unroll
    for (int cc = 0; cc < C; ++cc) sp[idx * C + cc] = row[cc];
  }
  // edges without triplets keep m = 0 (MODE 0) / dg = 0 (MODE 2): every thread clears the gap after its own row
  // (and the first row the edges before it), so no separate memset pass over the [E,16] arrays is needed
  if (MODE != 1 && live) {
    float* z = MODE == 0 ? a.m : a.dgq;
    for (int64_t g = (r == 0 ? 0 : e + 1); g < e_next; ++g) {
      if (g == e) continue;
#pragma unroll
      for (int k = 0; k < kCP; k += 4) *(float4*)(z + g * kCP + k) = float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // everything this row needs from global memory is requested before the barrier, so the triplet loop and the
  // epilogue wait on nothing but LDS (the pointers are not restrict: loads placed after the stores would stay there)
  const int t0 = a.t_ptr[e], t1 = a.t_ptr[e + 1];
  const float ux = a.u[e * 3], uy = a.u[e * 3 + 1], uz = a.u[e * 3 + 2];
  const float fc = a.fc3[e];
  float own[C], qv[C], qpv[C], vv[C];
  float fcp = 0.f, dd0 = 0.f, du0 = 0.f, du1 = 0.f, du2 = 0.f;
  if (MODE != 0) { dd0 = a.dd[e]; du0 = a.du[e * 3]; du1 = a.du[e * 3 + 1]; du2 = a.du[e * 3 + 2]; }
  if (MODE == 1) {
    fcp = a.fc3p[e];
#pragma unroll
    for (int k = 0; k < C; ++k) own[k] = a.dm[e * kCP + k];           // dm of this e1
  } else if (MODE == 2) {
#pragma unroll
    for (int k = 0; k < C; ++k) {
      qv[k] = a.q[e * kCP + k]; qpv[k] = a.qp[e * kCP + k]; vv[k] = a.v[kd * kCP + k];
      own[k] = qv[k] * vv[k];                                         // g of this e2
    }
  }
  __syncthreads();
  if (!live) return;
  float acc[C];
#pragma unroll
  for (int k = 0; k < C; ++k) acc[k] = 0.f;
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int t = t0; t < t1; ++t) {
    const int kk = t - t_lo;
    const int idx = kk < kTbListCap ? s_other[kk] : a.t_other[t] - lo;
    float vx, vy, vz, pr[C];
    if (idx >= 0 && idx < n) {
      vx = su[idx * 3]; vy = su[idx * 3 + 1]; vz = su[idx * 3 + 2];
#pragma unroll
      for (int k = 0; k < C; ++k) pr[k] = sp[idx * C + k];
    } else {
      const int64_t eo = a.act_list[lo + idx];
      vx = a.u[eo * 3]; vy = a.u[eo * 3 + 1]; vz = a.u[eo * 3 + 2];
      payload<C, MODE>(a, eo, a.act_dst[lo + idx], pr);
    }
    const float raw = ux * vx + uy * vy + uz * vz;
    const float cs = fminf(1.f, fmaxf(-1.f, raw));
    float P[L], dP[L];
    legendre<L>(cs, P, dP);
    if (MODE == 0) {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const float y = c.ynorm[l] * P[l];
#pragma unroll
        for (int nn = 0; nn < R; ++nn) acc[l * R + nn] += y * pr[l * R + nn];
      }
    } else {
      const bool inside = raw >= -1.f && raw <= 1.f;   // torch.clamp passes the gradient only inside [-1, 1]
      float dcos = 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const float y = c.ynorm[l] * P[l], dy = c.ynorm[l] * dP[l];
#pragma unroll
        for (int nn = 0; nn < R; ++nn) {
          const int k = l * R + nn;
          if (MODE == 1) {            // pr = g[e2]: S += Y g;  dcos += dm1 dY g
            acc[k] += y * pr[k];
            dcos += own[k] * dy * pr[k];
          } else {                    // pr = dS[e1]: dg += dS Y;  dcos += dS dY g_own
            acc[k] += pr[k] * y;
            dcos += pr[k] * dy * own[k];
          }
        }
      }
      dcos = inside ? dcos : 0.f;
      ax += dcos * vx; ay += dcos * vy; az += dcos * vz;
    }
  }
  if (MODE == 0) {
#pragma unroll
    for (int k = 0; k < kCP; k += 4) {
      float4 o;
      o.x = k + 0 < C ? fc * acc[k + 0 < C ? k + 0 : 0] : 0.f;
      o.y = k + 1 < C ? fc * acc[k + 1 < C ? k + 1 : 0] : 0.f;
      o.z = k + 2 < C ? fc * acc[k + 2 < C ? k + 2 : 0] : 0.f;
      o.w = k + 3 < C ? fc * acc[k + 3 < C ? k + 3 : 0] : 0.f;
      *(float4*)(a.m + e * kCP + k) = o;
    }
  } else if (MODE == 1) {
    float dfc = 0.f;
#pragma unroll
    for (int k = 0; k < C; ++k) dfc += own[k] * acc[k];               // acc = S[e1,:]
    a.dd[e] = dd0 + fcp * dfc;
    a.du[e * 3] = du0 + fc * ax; a.du[e * 3 + 1] = du1 + fc * ay; a.du[e * 3 + 2] = du2 + fc * az;   // dS = fc * dm
  } else {
    a.du[e * 3] = du0 + ax; a.du[e * 3 + 1] = du1 + ay; a.du[e * 3 + 2] = du2 + az;
    float ddv = 0.f, val[kCP];
#pragma unroll
    for (int cc = 0; cc < kCP; ++cc) {
      val[cc] = 0.f;
      if (cc < C) {
        const float dg = acc[cc < C ? cc : 0];                         // acc = dg[e2,:]
        ddv += dg * vv[cc < C ? cc : 0] * qpv[cc < C ? cc : 0];
        val[cc] = dg * qv[cc < C ? cc : 0];
      }
    }
#pragma unroll
    for (int k = 0; k < kCP; k += 4) *(float4*)(a.dgq + e * kCP + k) = float4{val[k], val[k + 1], val[k + 2], val[k + 3]};
    a.dd[e] = dd0 + ddv;
  }
}

static inline dim3 grid_rows(int64_t n) { return dim3((unsigned)((n + kTbRows - 1) / kTbRows)); }

#define M3G_DISPATCH_LR(L_, R_, BODY)                         \
  switch ((L_) * 8 + (R_)) {                                  \
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
    case 4 * 8 + 1: { constexpr int L = 4, R = 1; BODY; } break; \
    case 4 * 8 + 2: { constexpr int L = 4, R = 2; BODY; } break; \
    case 4 * 8 + 3: { constexpr int L = 4, R = 3; BODY; } break; \
    case 4 * 8 + 4: { constexpr int L = 4, R = 4; BODY; } break; \
    default: break;                                           \
  }

void launch_threebody(const Consts& c, const Topo& t, const Work& w, const float* v, float* m, hipStream_t s) {
  if (t.E == 0) return;
  if (t.T == 0) { (void)hipMemsetAsync(m, 0, sizeof(float) * t.E * kCP, s); return; }   // no active row to clear the gaps
  TbArgs a{t.E, t.act_list, t.act_dst, t.tb_win, t.n_act, t.t1_ptr, t.t1_e2c, w.u, w.fc3, w.fc3p, w.q, w.qp, v, nullptr, m, nullptr,
           nullptr, nullptr};
  M3G_DISPATCH_LR(c.L, c.R, hipLaunchKernelGGL((k_threebody_tile<L, R, 0>), grid_rows(t.E), dim3(kTbRows), 0, s, c, a));
}

void launch_threebody_reverse(const Consts& c, const Topo& t, const Work& w, const float* v, hipStream_t s) {
  if (t.E == 0) return;
  if (t.T == 0) { (void)hipMemsetAsync(w.dg, 0, sizeof(float) * t.E * kCP, s); return; }
  TbArgs a1{t.E, t.act_list, t.act_dst, t.tb_win, t.n_act, t.t1_ptr, t.t1_e2c, w.u, w.fc3, w.fc3p, w.q, w.qp, v, w.dm, nullptr, w.dd,
            w.du, w.dg};
  TbArgs a2 = a1;
  a2.t_ptr = t.t2_ptr;
  a2.t_other = t.t2_e1c;
  M3G_DISPATCH_LR(c.L, c.R, {
    hipLaunchKernelGGL((k_threebody_tile<L, R, 1>), grid_rows(t.E), dim3(kTbRows), 0, s, c, a1);
    hipLaunchKernelGGL((k_threebody_tile<L, R, 2>), grid_rows(t.E), dim3(kTbRows), 0, s, c, a2);
  });
}

}  // namespace m3g
