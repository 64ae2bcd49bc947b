// Three-body aggregate (stage S3) and its reverse (B3).
// Reference: ThreeBodyInteration.forward, nn/interaction.py:187-217 (the scatter_sum over triplets),
// LegendreCosPolynomial nn/interaction.py:353-365, clamp of cos(theta) nn/invariant.py:40.
//
//   m[e1,c] = fc(d_e1) * sum_{t in T1(e1)} Y_l(cos_t) g[e2(t),c],   g[e,c] = q[e,c] v[dst(e),c],  c = l*R+n
//
// Triplets are CSR-grouped by e1 (forward, and the e1 half of the reverse) and by e2 (the e2 half of the
// reverse), so every sum is a private register accumulation -- no atomics, run-to-run reproducible.
// One thread per edge row, 256 consecutive rows per workgroup.  The partners of those rows are edges of the
// same centre atoms, i.e. one contiguous window of the centre-sorted edge list: the workgroup stages that
// window's unit vectors and per-edge payload rows (g = q*v[dst] for the forward / e1 half, dS = fc*dm for the e2
// half) in LDS once -- the per-atom triplet tile -- and the triplet loops read LDS instead of gathering from
// L2.  Partners outside the staged window (only possible for degrees beyond the LDS budget) fall back to global
// memory, so correctness does not depend on the window size.
#include "m3g_internal.h"

namespace m3g {

constexpr int kTbRows = 256;   // edge rows per workgroup
constexpr int kTbCap = 384;    // staged window capacity in edges: 256 rows + boundary rows (overflow -> global reads); 18 KB -> 8 WGs per CU

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
  const int32_t *src, *dst, *row_ptr;
  const int32_t *t_ptr, *t_other;     // triplet CSR of this pass (by e1: partner = e2; by e2: partner = e1)
  const float *u, *fc3, *fc3p, *q, *qp, *v;
  const float* dm;                    // reverse: dL/dm [E][kCP]
  float* m;                           // forward out [E][kCP]
  float *dd, *du, *dgq;               // reverse in/out
};

// payload of edge e for this pass: MODE 0/1 -> g[e,:] = q*v[dst];  MODE 2 -> dS[e,:] = fc3*dm
template <int C, int MODE>
__device__ __forceinline__ void payload(const TbArgs& a, int64_t e, float* out) {
  if (MODE == 2) {
    const float f = a.fc3[e];
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = f * a.dm[e * kCP + c];
  } else {
    const int64_t k = a.dst[e];
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
  const int64_t eb = (int64_t)blockIdx.x * kTbRows;
  const int64_t elast = (eb + kTbRows - 1 < a.E ? eb + kTbRows - 1 : a.E - 1);
  const int lo = a.row_ptr[a.src[eb]];
  const int hi_full = a.row_ptr[a.src[elast] + 1];
  const int n = (hi_full - lo) < kTbCap ? (hi_full - lo) : kTbCap;
  for (int idx = threadIdx.x; idx < n; idx += kTbRows) {
    const int64_t e = lo + idx;
    su[idx * 3 + 0] = a.u[e * 3];
    su[idx * 3 + 1] = a.u[e * 3 + 1];
    su[idx * 3 + 2] = a.u[e * 3 + 2];
    float row[C];
    payload<C, MODE>(a, e, row);
#pragma unroll
    for (int cc = 0; cc < C; ++cc) sp[idx * C + cc] = row[cc];
  }
  __syncthreads();
  const int64_t e = eb + threadIdx.x;
  if (e >= a.E) return;
  const int t0 = a.t_ptr[e], t1 = a.t_ptr[e + 1];
  const float ux = a.u[e * 3], uy = a.u[e * 3 + 1], uz = a.u[e * 3 + 2];
  float acc[C], own[C];
#pragma unroll
  for (int k = 0; k < C; ++k) acc[k] = 0.f;
  if (MODE == 1) {
#pragma unroll
    for (int k = 0; k < C; ++k) own[k] = a.dm[e * kCP + k];           // dm of this e1
  } else if (MODE == 2) {
    payload<C, 0>(a, e, own);                                         // g of this e2
  }
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int t = t0; t < t1; ++t) {
    const int eo = a.t_other[t];
    const int idx = eo - lo;
    float vx, vy, vz, pr[C];
    if (idx >= 0 && idx < n) {
      vx = su[idx * 3]; vy = su[idx * 3 + 1]; vz = su[idx * 3 + 2];
#pragma unroll
      for (int k = 0; k < C; ++k) pr[k] = sp[idx * C + k];
    } else {
      vx = a.u[(int64_t)eo * 3]; vy = a.u[(int64_t)eo * 3 + 1]; vz = a.u[(int64_t)eo * 3 + 2];
      payload<C, MODE>(a, eo, pr);
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
    const float f = a.fc3[e];
#pragma unroll
    for (int k = 0; k < kCP; ++k) a.m[e * kCP + k] = k < C ? f * acc[k < C ? k : 0] : 0.f;
  } else if (MODE == 1) {
    if (t0 == t1) return;
    const float f = a.fc3[e];
    float dfc = 0.f;
#pragma unroll
    for (int k = 0; k < C; ++k) dfc += own[k] * acc[k];               // acc = S[e1,:]
    a.dd[e] += a.fc3p[e] * dfc;
    a.du[e * 3] += f * ax; a.du[e * 3 + 1] += f * ay; a.du[e * 3 + 2] += f * az;   // dS = fc * dm
  } else {
    if (t0 != t1) { a.du[e * 3] += ax; a.du[e * 3 + 1] += ay; a.du[e * 3 + 2] += az; }
    const int64_t k = a.dst[e];
    float ddv = 0.f;
#pragma unroll
    for (int cc = 0; cc < kCP; ++cc) {
      float val = 0.f;
      if (cc < C) {
        const float dg = acc[cc < C ? cc : 0];                         // acc = dg[e2,:]
        ddv += dg * a.v[k * kCP + cc] * a.qp[e * kCP + cc];
        val = dg * a.q[e * kCP + cc];
      }
      a.dgq[e * kCP + cc] = val;
    }
    if (t0 != t1) a.dd[e] += ddv;
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
  TbArgs a{t.E, t.src, t.dst, t.row_ptr, t.t1_ptr, t.t1_e2, w.u, w.fc3, w.fc3p, w.q, w.qp, v, nullptr, m, nullptr, nullptr, nullptr};
  M3G_DISPATCH_LR(c.L, c.R, hipLaunchKernelGGL((k_threebody_tile<L, R, 0>), grid_rows(t.E), dim3(kTbRows), 0, s, c, a));
}

void launch_threebody_reverse(const Consts& c, const Topo& t, const Work& w, const float* v, hipStream_t s) {
  if (t.E == 0) return;
  TbArgs a1{t.E, t.src, t.dst, t.row_ptr, t.t1_ptr, t.t1_e2, w.u, w.fc3, w.fc3p, w.q, w.qp, v, w.dm, nullptr, w.dd, w.du, w.dg};
  TbArgs a2 = a1;
  a2.t_ptr = t.t2_ptr;
  a2.t_other = t.t2_e1;
  M3G_DISPATCH_LR(c.L, c.R, {
    hipLaunchKernelGGL((k_threebody_tile<L, R, 1>), grid_rows(t.E), dim3(kTbRows), 0, s, c, a1);
    hipLaunchKernelGGL((k_threebody_tile<L, R, 2>), grid_rows(t.E), dim3(kTbRows), 0, s, c, a2);
  });
}

}  // namespace m3g
