// Three-body aggregate (stage S3) and its reverse (B3).
// Reference: ThreeBodyInteration.forward, nn/interaction.py:187-217 (the scatter_sum over triplets),
// LegendreCosPolynomial nn/interaction.py:353-365, clamp of cos(theta) nn/invariant.py:40.
//
//   m[e1,c] = fc(d_e1) * sum_{t in T1(e1)} Y_l(cos_t) g[e2(t),c],   g[e,c] = q[e,c] v[dst(e),c],  c = l*R+n
//
// Triplets are CSR-grouped by e1 (forward, and the e1 half of the reverse) and by e2 (the e2 half of
// the reverse), so every sum is a private register accumulation -- no atomics, run-to-run reproducible.
// One thread per edge row: consecutive rows share a centre atom, hence the same partner window of
// g/u rows, which stays in L1/L2.  (An LDS-staged per-atom tile version is the planned upgrade.)
#include "m3g_internal.h"

namespace m3g {

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

__global__ void __launch_bounds__(256) k_make_g(int64_t E, const int32_t* __restrict__ dst, const float* __restrict__ q,
                                                const float* __restrict__ v, float* __restrict__ g) {
  int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= E * kCP) return;
  int64_t e = idx / kCP;
  int c = (int)(idx % kCP);
  g[idx] = q[idx] * v[(int64_t)dst[e] * kCP + c];
}

template <int L, int R>
__global__ void __launch_bounds__(256) k_threebody(Consts c, int64_t E, const int32_t* __restrict__ t1_ptr,
                                                   const int32_t* __restrict__ t1_e2, const float* __restrict__ u,
                                                   const float* __restrict__ fc3, const float* __restrict__ g,
                                                   float* __restrict__ m) {
  int64_t e1 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e1 >= E) return;
  constexpr int C = L * R;
  float acc[C];
#pragma unroll
  for (int k = 0; k < C; ++k) acc[k] = 0.f;
  float ux = u[e1 * 3], uy = u[e1 * 3 + 1], uz = u[e1 * 3 + 2];
  int t0 = t1_ptr[e1], t1 = t1_ptr[e1 + 1];
  for (int t = t0; t < t1; ++t) {
    int e2 = t1_e2[t];
    float cs = ux * u[e2 * 3] + uy * u[e2 * 3 + 1] + uz * u[e2 * 3 + 2];
    cs = fminf(1.f, fmaxf(-1.f, cs));
    float P[L], dP[L];
    legendre<L>(cs, P, dP);
    const float* ge = g + (int64_t)e2 * kCP;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float y = c.ynorm[l] * P[l];
#pragma unroll
      for (int n = 0; n < R; ++n) acc[l * R + n] += y * ge[l * R + n];
    }
  }
  float f = fc3[e1];
#pragma unroll
  for (int k = 0; k < kCP; ++k) m[e1 * kCP + k] = k < C ? f * acc[k < C ? k : 0] : 0.f;
}

// reverse, e1 half: d fc(d_e1), d u_e1
template <int L, int R>
__global__ void __launch_bounds__(256) k_threebody_rev1(Consts c, int64_t E, const int32_t* __restrict__ t1_ptr,
                                                        const int32_t* __restrict__ t1_e2, const float* __restrict__ u,
                                                        const float* __restrict__ fc3, const float* __restrict__ fc3p,
                                                        const float* __restrict__ g, const float* __restrict__ dm,
                                                        float* __restrict__ dd, float* __restrict__ du) {
  int64_t e1 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e1 >= E) return;
  int t0 = t1_ptr[e1], t1 = t1_ptr[e1 + 1];
  if (t0 == t1) return;
  constexpr int C = L * R;
  float S[C], dm1[C];
  float f = fc3[e1];
#pragma unroll
  for (int k = 0; k < C; ++k) { S[k] = 0.f; dm1[k] = dm[e1 * kCP + k]; }
  float ux = u[e1 * 3], uy = u[e1 * 3 + 1], uz = u[e1 * 3 + 2];
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int t = t0; t < t1; ++t) {
    int e2 = t1_e2[t];
    float vx = u[e2 * 3], vy = u[e2 * 3 + 1], vz = u[e2 * 3 + 2];
    float raw = ux * vx + uy * vy + uz * vz;
    bool inside = raw >= -1.f && raw <= 1.f;
    float cs = fminf(1.f, fmaxf(-1.f, raw));
    float P[L], dP[L];
    legendre<L>(cs, P, dP);
    const float* ge = g + (int64_t)e2 * kCP;
    float dcos = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float y = c.ynorm[l] * P[l], dy = c.ynorm[l] * dP[l];
#pragma unroll
      for (int n = 0; n < R; ++n) {
        float gv = ge[l * R + n];
        S[l * R + n] += y * gv;
        dcos += dm1[l * R + n] * dy * gv;
      }
    }
    dcos = inside ? dcos * f : 0.f;
    ax += dcos * vx; ay += dcos * vy; az += dcos * vz;
  }
  float dfc = 0.f;
#pragma unroll
  for (int k = 0; k < C; ++k) dfc += dm1[k] * S[k];
  dd[e1] += fc3p[e1] * dfc;
  du[e1 * 3] += ax; du[e1 * 3 + 1] += ay; du[e1 * 3 + 2] += az;
}

// reverse, e2 half: d g[e2,:] (-> d d_e2 through q', and dgq = dg*q for the node gather), d u_e2
template <int L, int R>
__global__ void __launch_bounds__(256) k_threebody_rev2(Consts c, int64_t E, const int32_t* __restrict__ t2_ptr,
                                                        const int32_t* __restrict__ t2_e1, const int32_t* __restrict__ dst,
                                                        const float* __restrict__ u, const float* __restrict__ fc3,
                                                        const float* __restrict__ g, const float* __restrict__ q,
                                                        const float* __restrict__ qp, const float* __restrict__ v,
                                                        const float* __restrict__ dm, float* __restrict__ dd,
                                                        float* __restrict__ du, float* __restrict__ dgq) {
  int64_t e2 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e2 >= E) return;
  constexpr int C = L * R;
  int t0 = t2_ptr[e2], t1 = t2_ptr[e2 + 1];
  float dg[C], gs[C];
#pragma unroll
  for (int k = 0; k < C; ++k) { dg[k] = 0.f; gs[k] = g[e2 * kCP + k]; }
  float ux = u[e2 * 3], uy = u[e2 * 3 + 1], uz = u[e2 * 3 + 2];
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int t = t0; t < t1; ++t) {
    int e1 = t2_e1[t];
    float f = fc3[e1];
    float vx = u[e1 * 3], vy = u[e1 * 3 + 1], vz = u[e1 * 3 + 2];
    float raw = ux * vx + uy * vy + uz * vz;
    bool inside = raw >= -1.f && raw <= 1.f;
    float cs = fminf(1.f, fmaxf(-1.f, raw));
    float P[L], dP[L];
    legendre<L>(cs, P, dP);
    const float* dme = dm + (int64_t)e1 * kCP;
    float dcos = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float y = c.ynorm[l] * P[l], dy = c.ynorm[l] * dP[l];
#pragma unroll
      for (int n = 0; n < R; ++n) {
        float ds = f * dme[l * R + n];
        dg[l * R + n] += ds * y;
        dcos += ds * dy * gs[l * R + n];
      }
    }
    dcos = inside ? dcos : 0.f;
    ax += dcos * vx; ay += dcos * vy; az += dcos * vz;
  }
  if (t0 != t1) { du[e2 * 3] += ax; du[e2 * 3 + 1] += ay; du[e2 * 3 + 2] += az; }
  int k = dst[e2];
  float ddv = 0.f;
#pragma unroll
  for (int cc = 0; cc < kCP; ++cc) {
    float val = 0.f;
    if (cc < C) {
      ddv += dg[cc < C ? cc : 0] * v[(int64_t)k * kCP + cc] * qp[e2 * kCP + cc];
      val = dg[cc < C ? cc : 0] * q[e2 * kCP + cc];
    }
    dgq[e2 * kCP + cc] = val;
  }
  if (t0 != t1) dd[e2] += ddv;
}

static inline dim3 grid_for(int64_t n, int tpb = 256) { return dim3((unsigned)((n + tpb - 1) / tpb)); }

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
  hipLaunchKernelGGL(k_make_g, grid_for(t.E * kCP), dim3(256), 0, s, t.E, t.dst, w.q, v, w.g);
  M3G_DISPATCH_LR(c.L, c.R,
                  hipLaunchKernelGGL((k_threebody<L, R>), grid_for(t.E), dim3(256), 0, s, c, t.E, t.t1_ptr, t.t1_e2, w.u,
                                     w.fc3, w.g, m));
}

void launch_threebody_reverse(const Consts& c, const Topo& t, const Work& w, const float* v, hipStream_t s) {
  if (t.E == 0) return;
  // g of this block is recomputed (cheap) so blocks do not each keep a copy
  hipLaunchKernelGGL(k_make_g, grid_for(t.E * kCP), dim3(256), 0, s, t.E, t.dst, w.q, v, w.g);
  M3G_DISPATCH_LR(c.L, c.R, {
    hipLaunchKernelGGL((k_threebody_rev1<L, R>), grid_for(t.E), dim3(256), 0, s, c, t.E, t.t1_ptr, t.t1_e2, w.u, w.fc3,
                       w.fc3p, w.g, w.dm, w.dd, w.du);
    hipLaunchKernelGGL((k_threebody_rev2<L, R>), grid_for(t.E), dim3(256), 0, s, c, t.E, t.t2_ptr, t.t2_e1, t.dst, w.u,
                       w.fc3, w.g, w.q, w.qp, v, w.dm, w.dd, w.du, w.dg);
  });
}

}  // namespace m3g
