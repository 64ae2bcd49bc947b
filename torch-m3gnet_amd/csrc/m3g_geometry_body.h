// Per-edge geometry and bases (stage S0) as a device function: k_geometry (m3g_geometry.hip) runs it on its own,
// k_geometry_node_pre (m3g_node_mfma.hip) as one of two workgroup roles beside block 0's node tables (small systems).
// Reference: nn/scale.py:24-29, nn/invariant.py:20-59, nn/featurizer.py:81-100, nn/interaction.py:268-350,389-400.
#pragma once
#include "m3g_internal.h"

namespace m3g {

// torch.sinc: sin(pi x)/(pi x), and cos(pi x) from the same argument reduction
__device__ __forceinline__ float sinc_cos_pi(float x, float& cos_px) {
  const float kPi = 3.14159265358979323846f;
  float px = kPi * x, sn;
  sincosf(px, &sn, &cos_px);
  return x == 0.f ? 1.f : sn / px;
}

// radial basis h_m(d) and dh_m/dd (nn/featurizer.py:84-96)
__device__ __forceinline__ void radial_basis(const Consts& c, float d, float* h, float* hp) {
#pragma unroll
  for (int m = 0; m < kRCap; ++m) {
    if (m < c.R) {
      float x1 = c.a1[m] * d, x2 = c.a2[m] * d;
      float c1, c2;
      float s1 = sinc_cos_pi(x1, c1), s2 = sinc_cos_pi(x2, c2);
      float f = c.coeff[m] * (s1 + s2);
      float df = c.coeff[m] * ((c1 - s1) + (c2 - s2)) / d;
      if (m == 0) {
        h[0] = f;
        hp[0] = df;
      } else {
        h[m] = (f + c.rec_mul[m] * h[m - 1]) / c.rec_div[m];
        hp[m] = (df + c.rec_mul[m] * hp[m - 1]) / c.rec_div[m];
      }
    } else {
      h[m] = 0.f;
      hp[m] = 0.f;
    }
  }
}

// j_l(x), j_l'(x) for l = 0..L-1, upward recurrence with the reference's x <= 1e-8 branch
__device__ __forceinline__ void sph_bessel(int L, float x, float* j, float* dj) {
  float seq[kLCap + 1];
  if (x > 1e-8f) {
    float sn, cx;
    sincosf(x, &sn, &cx);
    float sx = sn / x;
    seq[0] = sx;
    seq[1] = (sx - cx) / x;
#pragma unroll
    for (int n = 1; n < kLCap; ++n) seq[n + 1] = (float)(2 * n + 1) / x * seq[n] - seq[n - 1];
#pragma unroll
    for (int l = 0; l < kLCap; ++l) {
      j[l] = seq[l];
      dj[l] = l == 0 ? -seq[1] : seq[l - 1] - (float)(l + 1) / x * seq[l];
    }
  } else {
    float dfact = 1.f;
#pragma unroll
    for (int l = 0; l < kLCap; ++l) {
      if (l > 0) dfact *= (float)(2 * l + 1);
      j[l] = l == 0 ? 1.f : x / dfact;
      dj[l] = l == 1 ? 1.f / 3.f : 0.f;
    }
  }
  (void)L;
}

struct GeomArgs {
  int64_t E;
  const int32_t *src, *dst, *batch;
  const float *pos, *lattice;
  const int32_t* shift;
  float *u, *dist, *h, *hp, *q, *qp, *fc3, *fc3p;
  const int32_t* act_id;
  int32_t* sync;
};
// vblock: this workgroup's index among the workgroups of the geometry stage (k_geometry: blockIdx.x; k_geometry_node_pre: its role index)
template <bool FULL, int L, int R>
__device__ __forceinline__ void geometry_body(const Consts& c, const GeomArgs& a, int64_t vblock) {
  const int64_t E = a.E;
  const int32_t* __restrict__ src = a.src;
  const int32_t* __restrict__ dst = a.dst;
  const int32_t* __restrict__ batch = a.batch;
  const float* __restrict__ pos = a.pos;
  const float* __restrict__ lattice = a.lattice;
  const int32_t* __restrict__ shift = a.shift;
  float* __restrict__ u = a.u;
  float* __restrict__ dist = a.dist;
  float* __restrict__ h = a.h;
  float* __restrict__ hp = a.hp;
  float* __restrict__ q = a.q;
  float* __restrict__ qp = a.qp;
  float* __restrict__ fc3 = a.fc3;
  float* __restrict__ fc3p = a.fc3p;
  const int32_t* __restrict__ act_id = a.act_id;
  int32_t* __restrict__ sync = a.sync;
  int64_t e = vblock * (int64_t)blockDim.x + threadIdx.x;
  // the step's "last workgroup" counters (Work::sync): cleared by the first kernel of every step
  if (sync && vblock == 0 && threadIdx.x < kSyncWords) sync[threadIdx.x] = 0;
  if (e >= E) return;
  int i = src[e], j = dst[e], s = batch[i];
  float ls = c.length_scale;
  float r[3];
  float sh0 = (float)shift[e * 3 + 0], sh1 = (float)shift[e * 3 + 1], sh2 = (float)shift[e * 3 + 2];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l0 = lattice[s * 9 + 0 + a] / ls, l1 = lattice[s * 9 + 3 + a] / ls, l2 = lattice[s * 9 + 6 + a] / ls;
    float sv = (sh0 * l0 + sh1 * l1) + sh2 * l2;
    r[a] = (pos[(int64_t)j * 3 + a] / ls + sv) - pos[(int64_t)i * 3 + a] / ls;
  }
  float d = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  dist[e] = d;
  u[e * 3 + 0] = r[0] / d;
  u[e * 3 + 1] = r[1] / d;
  u[e * 3 + 2] = r[2] / d;
  if (!FULL) return;
  float hh[kRCap], hd[kRCap];
  radial_basis(c, d, hh, hd);
  // every per-edge row leaves as 16-byte stores (a scalar store per element touched 64 cache lines per instruction)
  *(float4*)(h + e * kRP) = float4{hh[0], hh[1], hh[2], hh[3]};
  *(float4*)(hp + e * kRP) = float4{hd[0], hd[1], hd[2], hd[3]};
  // three-body cutoff envelope (nn/interaction.py:389-400) and its derivative
  float rho = d / c.rc3;
  float f = 0.f, fp = 0.f;
  if (rho <= 1.f) {
    float r2 = rho * rho, r3 = r2 * rho;
    f = 1.f - 6.f * r3 * r2 + 15.f * r2 * r2 - 10.f * r3;
    fp = (-30.f * r2 * r2 + 60.f * r3 - 30.f * r2) / c.rc3;
  }
  fc3[e] = f;
  fc3p[e] = fp;
  // q[ar,c] = chi_ln(d) fc(d),  c = l*R + n  (nn/interaction.py:268-281), one row per ACTIVE edge (ar = act_id[e]): an edge
  // without triplets -- beyond the three-body cutoff or without a partner -- needs no row and no Bessel evaluation
  const int ar = act_id[e];
  if (ar < 0) return;
  float qr[kCP], qpr[kCP];
#pragma unroll
  for (int cc = 0; cc < kCP; ++cc) { qr[cc] = 0.f; qpr[cc] = 0.f; }
#pragma unroll
  for (int n = 0; n < R; ++n) {
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float jl[kLCap], djl[kLCap];
      // the argument differs per (l,n): z_ln * d / rc
      float x = c.zeros[l][n] * d / c.rc;
      sph_bessel(L, x, jl, djl);
      float chi = jl[l] / c.factors[l][n];
      float dchi = djl[l] * (c.zeros[l][n] / c.rc) / c.factors[l][n];
      qr[l * R + n] = chi * f;
      qpr[l * R + n] = dchi * f + chi * fp;
    }
  }
#pragma unroll
  for (int cc = 0; cc < kCP; cc += 4) {
    *(float4*)(q + (int64_t)ar * kCP + cc) = float4{qr[cc], qr[cc + 1], qr[cc + 2], qr[cc + 3]};
    *(float4*)(qp + (int64_t)ar * kCP + cc) = float4{qpr[cc], qpr[cc + 1], qpr[cc + 2], qpr[cc + 3]};
  }
}


// dE/dr of one edge from dL/dd (three-body share dd + the radial-basis share dh . h') and dL/du (projected off u):
// dd / du hold one row per ACTIVE edge (act_id == nullptr: no three-body reverse ran); dL/dh arrives in `dh_parts` slices (one
// per reverse kernel that produced a share), summed here in a fixed order
struct GeomRev {
  int64_t E;
  const float *u, *dist, *hp, *dh;
  int dh_parts;
  const float *dd, *du;
  const int32_t* act_id;
};
__device__ __forceinline__ void edge_dr(const GeomRev& a, int64_t e, float& rx, float& ry, float& rz) {
  // Nothing here is left to the compiler's choice of which multiply-add pairs to contract: every fused operation is written out
  // (the placement the kernel compiled to in round 4), so the bits do not depend on the context the function is inlined into.
#pragma clang fp contract(off)
  const int ar = a.act_id ? a.act_id[e] : -1;
  float g = ar >= 0 ? a.dd[ar] : 0.f;
  float dhs[kRP] = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < a.dh_parts; ++p) {
    const float4 t = *(const float4*)(a.dh + ((int64_t)p * a.E + e) * kRP);
    dhs[0] += t.x; dhs[1] += t.y; dhs[2] += t.z; dhs[3] += t.w;
  }
  {
    const float4 t = *(const float4*)(a.hp + e * kRP);
    float dot = dhs[1] * t.y;
    dot = __builtin_fmaf(dhs[0], t.x, dot);
    dot = __builtin_fmaf(dhs[2], t.z, dot);
    dot = __builtin_fmaf(dhs[3], t.w, dot);
    g = g + dot;
  }
  const float ux = a.u[e * 3], uy = a.u[e * 3 + 1], uz = a.u[e * 3 + 2];
  float ax = 0.f, ay = 0.f, az = 0.f;
  if (ar >= 0) { ax = a.du[(int64_t)ar * 3]; ay = a.du[(int64_t)ar * 3 + 1]; az = a.du[(int64_t)ar * 3 + 2]; }
  const float proj = __builtin_fmaf(ax, ux, ay * uy) + az * uz;
  const float inv = 1.f / a.dist[e];
  rx = __builtin_fmaf(g, ux, __builtin_fmaf(-proj, ux, ax) * inv);
  ry = __builtin_fmaf(g, uy, __builtin_fmaf(-proj, uy, ay) * inv);
  rz = __builtin_fmaf(g, uz, __builtin_fmaf(-proj, uz, az) * inv);
}

inline GeomArgs geometry_args(const Topo& t, const float* pos, const float* lattice, const int32_t* shift, const Work& w) {
  return GeomArgs{t.E, t.src, t.dst, t.batch, pos, lattice, shift, w.u, w.d, w.h, w.hp, w.q, w.qp, w.fc3, w.fc3p, t.act_id, w.sync};
}

}  // namespace m3g
