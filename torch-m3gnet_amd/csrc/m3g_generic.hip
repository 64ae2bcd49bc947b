// Any-size path: the same pipeline (energies, analytic forces, virial) for model sizes beyond the tiles of the MFMA kernels --
// embedding_dim > 64, l_max or n_max > 4 (the reference accepts any width and its Bessel-root table allows l_max <= 9,
// n_max <= 10: nn/interaction.py:250-253, model/build.py:16-28).  Plain fp32 vector-ALU kernels with run-time dimensions, one
// thread per output element, unpadded row-major buffers, every activation saved for the reverse pass.  It is the
// correctness-first path (a 128-wide block does not fit the LDS-resident weight images the fast kernels are built around); it
// is also what the stand-alone `forward` of the block modules runs on (m3g_stage_* entry points below).
// Stage by stage it follows oracle/staged.py (the executable spec of the restructured forward and the hand-derived reverse
// pass), with the reference citations given there.  All sums over edges / triplets go through the topology's CSR lists: no
// atomics on the force path.
#include <cmath>
#include <cstring>

#include "m3g_internal.h"
#include "m3g_device.h"

namespace m3g {

namespace {

constexpr int kGL = 9, kGR = 10, kGC = kGL * kGR;   // the reference's table: l_max + 1 <= 10 rows, n_max <= 10 columns

struct GenConsts {
  int L, R, C, D, B, num_types;
  float length_scale, energy_scale, rc, rc3;
  float a1[kGR], a2[kGR], coeff[kGR], rec_mul[kGR], rec_div[kGR];
  float zeros[kGL][kGR], factors[kGL][kGR], ynorm[kGL];
  int ref_legendre;   // option "legendre_backward" (m3g_threebody.hip: legendre_ref_k)
};

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }
#define GEN_IDX(n_total)                                              \
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; \
  if (gid >= (n_total)) return;

// ---- dense products ---------------------------------------------------------------------------------------------------
// C[n, j] = (beta ? C[n, j] : 0) + (bias ? bias[j] : 0) + sum_k A[n*lda + k] * B[k*sbk + j*sbj]     (any strides of B: W or W^T)
// 64 x 64 output tile per workgroup, K walked in slabs of 16 staged in LDS (an A row is read once per 64 columns, B coalesced
// whichever way it is strided); each of the four waves owns 16 rows of the tile and forms its 16 x 64 outputs with
// v_mfma_f32_16x16x4_f32 -- exact fp32 products, fp32 accumulation, the arithmetic of the reference's matmul -- four column blocks
// per 4-deep k-step.  Round 2: one thread per output element (1,064 ms per step at D = 128); round 3: this tiling on the vector
// ALU (43 ms), then on the matrix pipe.
typedef float g_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kGemmTile = 64, kGemmK = 16;
// NCB: 16-column blocks per workgroup.  4 (a 64 x 64 tile) everywhere: tiles of 128 / 256 columns, which read the activation
// operand once instead of once per 64 columns, measured SLOWER at D = 96 / 128 (30.6 / 40.1 against 28.2 / 37.3 ms per step: fewer
// workgroups in flight to hide the row-strided A loads)
template <int NCB>
__global__ void __launch_bounds__(256) g_gemm(int64_t n, int cols, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                              int sbk, int sbj, const float* __restrict__ bias, float* __restrict__ Cm, int64_t ldc, int beta) {
  constexpr int TN = 16 * NCB;
  __shared__ float sa[kGemmK][kGemmTile + 1], sb[kGemmK][TN + 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * kGemmTile;
  const int col0 = blockIdx.y * TN;
  g_f32x4 acc[NCB];   // column block cb: rows 16 wave + 4 q + r, column 16 cb + m
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int c = col0 + 16 * cb + m;
    const float b0 = (bias && c < cols) ? bias[c] : 0.f;
    acc[cb] = g_f32x4{b0, b0, b0, b0};
  }
  for (int k0 = 0; k0 < K; k0 += kGemmK) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {   // A slab: 64 rows x 16 k, k fastest (rows of A are contiguous in k)
      const int idx = tid + it * 256, r = idx >> 4, kk = idx & 15;
      const int64_t row = row0 + r;
      sa[kk][r] = (row < n && k0 + kk < K) ? A[row * lda + k0 + kk] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NCB; ++it) {   // B slab: 16 k x TN columns, the contiguous index fastest
      const int idx = tid + it * 256;
      const int kk = sbj == 1 ? idx / TN : idx & 15, cc = sbj == 1 ? idx % TN : idx >> 4;
      const int c = col0 + cc;
      sb[kk][cc] = (c < cols && k0 + kk < K) ? B[(int64_t)(k0 + kk) * sbk + (int64_t)c * sbj] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < kGemmK / 4; ++ks) {   // (rows of the slab beyond K hold zeros)
      const float a = sa[4 * ks + q][16 * wave + m];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sb[4 * ks + q][16 * cb + m], acc[cb], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + 16 * wave + 4 * q + r;
    if (row >= n) continue;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int c = col0 + 16 * cb + m;
      if (c >= cols) continue;
      float* out = Cm + row * ldc + c;
      *out = beta ? *out + acc[cb][r] : acc[cb][r];
    }
  }
}
static void gemm(hipStream_t s, int64_t n, int cols, int K, const float* A, int64_t lda, const float* B, int sbk, int sbj, const float* bias,
                 float* C, int64_t ldc, bool beta = false) {
  if (n <= 0 || cols <= 0) return;
  const unsigned gx = (unsigned)((n + kGemmTile - 1) / kGemmTile);
#define M3G_GEMM_LAUNCH(NCB_) \
  hipLaunchKernelGGL(g_gemm<NCB_>, dim3(gx, (unsigned)((cols + 16 * NCB_ - 1) / (16 * NCB_))), dim3(256), 0, s, n, cols, K, A, lda, B, sbk, sbj, bias, C, ldc, beta ? 1 : 0)
  M3G_GEMM_LAUNCH(4);
#undef M3G_GEMM_LAUNCH
}
// y[n, j] = x[idx[n], j]  (row gather), optionally y += ...
__global__ void __launch_bounds__(256) g_gather(int64_t n, int w, const int32_t* __restrict__ idx, const float* __restrict__ X, int64_t ldx,
                                                float* __restrict__ Y, int64_t ldy, int beta) {
  GEN_IDX(n * w);
  const int64_t r = gid / w;
  const int j = (int)(gid % w);
  const float v = X[(int64_t)idx[r] * ldx + j];
  float* y = Y + r * ldy + j;
  *y = beta ? *y + v : v;
}
// y[i, j] (+)= sum over p in [ptr[i], ptr[i+1]) of X[(list ? list[p] : p), j]   (segment sum through a CSR list)
__global__ void __launch_bounds__(256) g_segsum(int64_t n, int w, const int32_t* __restrict__ ptr, const int32_t* __restrict__ list,
                                                const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, int beta) {
  GEN_IDX(n * w);
  const int64_t i = gid / w;
  const int j = (int)(gid % w);
  float acc = 0.f;
  for (int p = ptr[i]; p < ptr[i + 1]; ++p) acc += X[(int64_t)(list ? list[p] : p) * ldx + j];
  float* y = Y + i * ldy + j;
  *y = beta ? *y + acc : acc;
}
// elementwise maps over n values
enum { OP_SILU = 0, OP_SIGMOID = 1, OP_MUL_DSILU = 2 /* y *= silu'(x) */, OP_MUL_DSIGMOID_OF_V = 3 /* y *= x (1 - x) */ };
__global__ void __launch_bounds__(256) g_map(int64_t n, int op, const float* __restrict__ X, float* __restrict__ Y) {
  GEN_IDX(n);
  const float x = X[gid];
  if (op == OP_SILU) Y[gid] = silu_f(x);
  else if (op == OP_SIGMOID) Y[gid] = sigmoid_f(x);
  else if (op == OP_MUL_DSILU) Y[gid] *= dsilu_f(x);
  else Y[gid] *= x * (1.f - x);
}
static void map(hipStream_t s, int64_t n, int op, const float* X, float* Y) {
  if (n > 0) hipLaunchKernelGGL(g_map, grid1(n), dim3(256), 0, s, n, op, X, Y);
}

// ---- S0 geometry and bases (nn/scale.py:24-29, nn/invariant.py:20-59, nn/featurizer.py:81-100, nn/interaction.py:268-350,389-400)
__device__ __forceinline__ float g_sinc_cos_pi(float x, float& cos_px) {
  const float px = 3.14159265358979323846f * x;
  float sn;
  sincosf(px, &sn, &cos_px);
  return x == 0.f ? 1.f : sn / px;
}
__device__ __forceinline__ void g_bessel(int L, float x, float* j, float* dj) {   // j_l, j_l' for l < L (reference's x <= 1e-8 branch)
  float seq[kGL + 1];
  if (x > 1e-8f) {
    float sn, cx;
    sincosf(x, &sn, &cx);
    const float sx = sn / x;
    seq[0] = sx;
    seq[1] = (sx - cx) / x;
    for (int n = 1; n < L; ++n) seq[n + 1] = (float)(2 * n + 1) / x * seq[n] - seq[n - 1];
    for (int l = 0; l < L; ++l) {
      j[l] = seq[l];
      dj[l] = l == 0 ? -seq[1] : seq[l - 1] - (float)(l + 1) / x * seq[l];
    }
  } else {
    float dfact = 1.f;
    for (int l = 0; l < L; ++l) {
      if (l > 0) dfact *= (float)(2 * l + 1);
      j[l] = l == 0 ? 1.f : x / dfact;
      dj[l] = l == 1 ? 1.f / 3.f : 0.f;
    }
  }
}
__global__ void __launch_bounds__(256) g_geometry(GenConsts c, int64_t E, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                  const int32_t* __restrict__ batch, const float* __restrict__ pos,
                                                  const float* __restrict__ lattice, const int32_t* __restrict__ shift, float* __restrict__ u,
                                                  float* __restrict__ dist, float* __restrict__ h, float* __restrict__ hp,
                                                  float* __restrict__ fc3, float* __restrict__ fc3p, float* __restrict__ q, float* __restrict__ qp) {
  GEN_IDX(E);
  const int64_t e = gid;
  const int i = src[e], jn = dst[e], sidx = batch[i];
  const float ls = c.length_scale;
  float r[3];
  const float sh0 = (float)shift[e * 3], sh1 = (float)shift[e * 3 + 1], sh2 = (float)shift[e * 3 + 2];
  for (int a = 0; a < 3; ++a) {
    const float l0 = lattice[sidx * 9 + a] / ls, l1 = lattice[sidx * 9 + 3 + a] / ls, l2 = lattice[sidx * 9 + 6 + a] / ls;
    const float sv = (sh0 * l0 + sh1 * l1) + sh2 * l2;
    r[a] = (pos[(int64_t)jn * 3 + a] / ls + sv) - pos[(int64_t)i * 3 + a] / ls;
  }
  const float d = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  dist[e] = d;
  u[e * 3] = r[0] / d; u[e * 3 + 1] = r[1] / d; u[e * 3 + 2] = r[2] / d;
  float hprev = 0.f, hpprev = 0.f;
  for (int m = 0; m < c.R; ++m) {
    float c1, c2;
    const float s1 = g_sinc_cos_pi(c.a1[m] * d, c1), s2 = g_sinc_cos_pi(c.a2[m] * d, c2);
    float f = c.coeff[m] * (s1 + s2);
    float df = c.coeff[m] * ((c1 - s1) + (c2 - s2)) / d;
    if (m > 0) { f = (f + c.rec_mul[m] * hprev) / c.rec_div[m]; df = (df + c.rec_mul[m] * hpprev) / c.rec_div[m]; }
    h[e * c.R + m] = f;
    hp[e * c.R + m] = df;
    hprev = f; hpprev = df;
  }
  const float rho = d / c.rc3;
  float f = 0.f, fp = 0.f;
  if (rho <= 1.f) {
    const float r2 = rho * rho, r3 = r2 * rho;
    f = 1.f - 6.f * r3 * r2 + 15.f * r2 * r2 - 10.f * r3;
    fp = (-30.f * r2 * r2 + 60.f * r3 - 30.f * r2) / c.rc3;
  }
  fc3[e] = f;
  fc3p[e] = fp;
  for (int l = 0; l < c.L; ++l)
    for (int n = 0; n < c.R; ++n) {
      float jl[kGL], djl[kGL];
      g_bessel(c.L, c.zeros[l][n] * d / c.rc, jl, djl);
      const float chi = jl[l] / c.factors[l][n];
      const float dchi = djl[l] * (c.zeros[l][n] / c.rc) / c.factors[l][n];
      q[e * c.C + l * c.R + n] = chi * f;
      qp[e * c.C + l * c.R + n] = dchi * f + chi * fp;
    }
}

__global__ void __launch_bounds__(256) g_embed_x(int64_t N, int D, int num_types, const int64_t* __restrict__ types, const float* __restrict__ W,
                                                 float* __restrict__ x) {
  GEN_IDX(N * D);
  const int64_t a = gid / D;
  const int o = (int)(gid % D);
  bool bad;   // (an out-of-range species is clamped here and reported by g_atomic_energy)
  x[gid] = W[(int64_t)o * num_types + species_index(types[a], num_types, bad)];   // one_hot(types) @ W^T, W [D, num_types] (nn/featurizer.py:33-38)
}

// what the reference's LegendreCosPolynomial.backward returns per unit of its grad_output `go` (nn/interaction.py:373-382):
// k_1 = 1, k_n = n P_{n-1} + x go k_{n-1}
__device__ __forceinline__ float g_legendre_ref_k(int l, float x, const float* P, float go) {
  if (l == 0) return 0.f;
  float k = 1.f;
  for (int n = 2; n <= l; ++n) k = (float)n * P[n - 1] + x * go * k;
  return k;
}
__device__ __forceinline__ void g_legendre(int L, float x, float* P, float* dP) {
  P[0] = 1.f; dP[0] = 0.f;
  if (L > 1) { P[1] = x; dP[1] = 1.f; }
  for (int n = 1; n < L - 1; ++n) {
    P[n + 1] = ((float)(2 * n + 1) * x * P[n] - (float)n * P[n - 1]) / (float)(n + 1);
    dP[n + 1] = ((float)(2 * n + 1) * (P[n] + x * dP[n]) - (float)n * dP[n - 1]) / (float)(n + 1);
  }
}

// S3: Ssum[e1, c] = sum_{t in T1(e1)} Y_l(cos_t) q[e2, c] v[dst(e2), c]     (m = fc3 * Ssum; nn/interaction.py:187-217)
__global__ void __launch_bounds__(256) g_threebody_fwd(GenConsts c, int64_t E, const int32_t* __restrict__ t1_ptr, const int32_t* __restrict__ t1_e2,
                                                       const int32_t* __restrict__ dst, const float* __restrict__ u, const float* __restrict__ q,
                                                       const float* __restrict__ v, float* __restrict__ Ssum) {
  GEN_IDX(E * c.L);
  const int64_t e1 = gid / c.L;
  const int l = (int)(gid % c.L);
  float acc[kGR];
  for (int n = 0; n < c.R; ++n) acc[n] = 0.f;
  const float ux = u[e1 * 3], uy = u[e1 * 3 + 1], uz = u[e1 * 3 + 2];
  for (int t = t1_ptr[e1]; t < t1_ptr[e1 + 1]; ++t) {
    const int64_t e2 = t1_e2[t];
    const float cs = fminf(1.f, fmaxf(-1.f, ux * u[e2 * 3] + uy * u[e2 * 3 + 1] + uz * u[e2 * 3 + 2]));
    float P[kGL], dP[kGL];
    g_legendre(c.L, cs, P, dP);
    const float y = c.ynorm[l] * P[l];
    const int64_t k = dst[e2];
    for (int n = 0; n < c.R; ++n) acc[n] += y * q[e2 * c.C + l * c.R + n] * v[k * c.C + l * c.R + n];
  }
  for (int n = 0; n < c.R; ++n) Ssum[e1 * c.C + l * c.R + n] = acc[n];
}
// m[e, c] = fc3[e] * Ssum[e, c]
__global__ void __launch_bounds__(256) g_scale_rows(int64_t n, int w, const float* __restrict__ scale, const float* __restrict__ X, float* __restrict__ Y) {
  GEN_IDX(n * w);
  Y[gid] = scale[gid / w] * X[gid];
}
// gated product: out[i] = silu(pd[i]) * sigmoid(pg[i]) [* lin[i]]; mode 1: y = base + that
__global__ void __launch_bounds__(256) g_gated(int64_t n, const float* __restrict__ pd, const float* __restrict__ pg, const float* __restrict__ lin,
                                               const float* __restrict__ base, float* __restrict__ Y) {
  GEN_IDX(n);
  float v = silu_f(pd[gid]) * sigmoid_f(pg[gid]);
  if (lin) v *= lin[gid];
  Y[gid] = base ? base[gid] + v : v;
}
// layer-1 concat input [x_i | x_j | e] of one edge row (nn/conv.py:91-97)
__global__ void __launch_bounds__(256) g_concat(int64_t E, int D, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                const float* __restrict__ x, const float* __restrict__ e, float* __restrict__ cat) {
  GEN_IDX(E * 3 * D);
  const int64_t r = gid / (3 * D);
  const int k = (int)(gid % (3 * D));
  cat[gid] = k < D ? x[(int64_t)src[r] * D + k] : k < 2 * D ? x[(int64_t)dst[r] * D + (k - D)] : e[r * D + (k - 2 * D)];
}

// reverse of the gated product y = silu(pd) sigmoid(pg) lin with upstream du:  d_pd, d_pg (in place of pd/pg copies), d_lin
__global__ void __launch_bounds__(256) g_gated_rev(int64_t n, const float* __restrict__ pd, const float* __restrict__ pg, const float* __restrict__ lin,
                                                   const float* __restrict__ du, float* __restrict__ d_pd, float* __restrict__ d_pg,
                                                   float* __restrict__ d_lin) {
  GEN_IDX(n);
  const float p = pd[gid], g = pg[gid], sg = sigmoid_f(g), sd = silu_f(p);
  const float d_out = du[gid] * (lin ? lin[gid] : 1.f);
  d_pd[gid] = d_out * sg * dsilu_f(p);
  d_pg[gid] = d_out * sd * sg * (1.f - sg);
  if (d_lin) d_lin[gid] = du[gid] * sd * sg;
}

// B3 three-body reverse, one thread per edge in both roles (oracle/staged.py B3)
__global__ void __launch_bounds__(256) g_threebody_rev(GenConsts c, int64_t E, const int32_t* __restrict__ t1_ptr, const int32_t* __restrict__ t1_e2,
                                                       const int32_t* __restrict__ t2_ptr, const int32_t* __restrict__ t2_e1,
                                                       const int32_t* __restrict__ dst, const float* __restrict__ u, const float* __restrict__ fc3,
                                                       const float* __restrict__ fc3p, const float* __restrict__ q, const float* __restrict__ qp,
                                                       const float* __restrict__ v, const float* __restrict__ Ssum, const float* __restrict__ dm,
                                                       float* __restrict__ dd, float* __restrict__ du, float* __restrict__ dgq) {
  GEN_IDX(E);
  const int64_t e = gid;
  const int C = c.C;
  const float ux = u[e * 3], uy = u[e * 3 + 1], uz = u[e * 3 + 2];
  float ax = 0.f, ay = 0.f, az = 0.f, dfc = 0.f;
  const float fce = fc3[e];
  // as e1: d cos_t = inside * sum_c dS[e1,c] dY_l g[e2,c];   dS = fc3 * dm
  for (int t = t1_ptr[e]; t < t1_ptr[e + 1]; ++t) {
    const int64_t e2 = t1_e2[t];
    const float vx = u[e2 * 3], vy = u[e2 * 3 + 1], vz = u[e2 * 3 + 2];
    const float raw = ux * vx + uy * vy + uz * vz;
    float P[kGL], dP[kGL];
    g_legendre(c.L, fminf(1.f, fmaxf(-1.f, raw)), P, dP);
    const int64_t k = dst[e2];
    float dcos = 0.f;
    for (int l = 0; l < c.L; ++l)
      for (int n = 0; n < c.R; ++n) {
        const int cc = l * c.R + n;
        dcos += dm[e * C + cc] * c.ynorm[l] * dP[l] * q[e2 * C + cc] * v[k * C + cc];
      }
    if (c.ref_legendre) {
      dcos = 0.f;
      for (int l = 1; l < c.L; ++l) {
        float G = 0.f;
        for (int n = 0; n < c.R; ++n) G += dm[e * C + l * c.R + n] * (q[e2 * C + l * c.R + n] * v[k * C + l * c.R + n]);
        const float g1 = c.ynorm[l] * G;
        dcos += g1 * g_legendre_ref_k(l, fminf(1.f, fmaxf(-1.f, raw)), P, fce * g1);
      }
    }
    dcos = (raw >= -1.f && raw <= 1.f) ? fce * dcos : 0.f;
    ax += dcos * vx; ay += dcos * vy; az += dcos * vz;
  }
  for (int cc = 0; cc < C; ++cc) dfc += dm[e * C + cc] * Ssum[e * C + cc];
  // as e2: dg[e,c] = sum_t dS[e1,c] Y_l;   d cos_t = inside * sum_c dS[e1,c] dY_l g[e,c]
  float dg[kGC];
  for (int cc = 0; cc < C; ++cc) dg[cc] = 0.f;
  const int64_t kd = dst[e];
  for (int t = t2_ptr[e]; t < t2_ptr[e + 1]; ++t) {
    const int64_t e1 = t2_e1[t];
    const float vx = u[e1 * 3], vy = u[e1 * 3 + 1], vz = u[e1 * 3 + 2];
    const float raw = ux * vx + uy * vy + uz * vz;
    float P[kGL], dP[kGL];
    g_legendre(c.L, fminf(1.f, fmaxf(-1.f, raw)), P, dP);
    const float f1 = fc3[e1];
    float dcos = 0.f;
    for (int l = 0; l < c.L; ++l)
      for (int n = 0; n < c.R; ++n) {
        const int cc = l * c.R + n;
        const float ds = f1 * dm[e1 * C + cc];
        dg[cc] += ds * c.ynorm[l] * P[l];
        dcos += ds * c.ynorm[l] * dP[l] * q[e * C + cc] * v[kd * C + cc];
      }
    if (c.ref_legendre) {
      dcos = 0.f;
      for (int l = 1; l < c.L; ++l) {
        float G = 0.f;
        for (int n = 0; n < c.R; ++n) G += (f1 * dm[e1 * C + l * c.R + n]) * (q[e * C + l * c.R + n] * v[kd * C + l * c.R + n]);
        const float go = c.ynorm[l] * G;
        dcos += go * g_legendre_ref_k(l, fminf(1.f, fmaxf(-1.f, raw)), P, go);
      }
    }
    dcos = (raw >= -1.f && raw <= 1.f) ? dcos : 0.f;
    ax += dcos * vx; ay += dcos * vy; az += dcos * vz;
  }
  float ddv = 0.f;
  for (int cc = 0; cc < C; ++cc) {
    ddv += dg[cc] * v[kd * C + cc] * qp[e * C + cc];
    dgq[e * C + cc] = dg[cc] * q[e * C + cc];
  }
  dd[e] += fc3p[e] * dfc + ddv;
  du[e * 3] += ax; du[e * 3 + 1] += ay; du[e * 3 + 2] += az;
}

// B0: d r = dd u + (du - (du.u) u) / d, dd includes dh . h'
__global__ void __launch_bounds__(256) g_geometry_rev(int64_t E, int R, const float* __restrict__ u, const float* __restrict__ dist,
                                                      const float* __restrict__ hp, const float* __restrict__ dh, const float* __restrict__ dd,
                                                      const float* __restrict__ du, float* __restrict__ dr) {
  GEN_IDX(E);
  const int64_t e = gid;
  float g = dd[e];
  for (int r = 0; r < R; ++r) g += dh[e * R + r] * hp[e * R + r];
  const float ux = u[e * 3], uy = u[e * 3 + 1], uz = u[e * 3 + 2];
  const float ax = du[e * 3], ay = du[e * 3 + 1], az = du[e * 3 + 2];
  const float proj = ax * ux + ay * uy + az * uz, inv = 1.f / dist[e];
  dr[e * 3] = g * ux + (ax - proj * ux) * inv;
  dr[e * 3 + 1] = g * uy + (ay - proj * uy) * inv;
  dr[e * 3 + 2] = g * uz + (az - proj * uz) * inv;
}

__global__ void __launch_bounds__(256) g_atomic_energy(int64_t N, int num_types, float energy_scale, const int64_t* __restrict__ types,
                                                       const float* __restrict__ elemental, const float* __restrict__ od, const float* __restrict__ og,
                                                       float* __restrict__ ea, const int32_t* topo_flags) {
  GEN_IDX(N);
  bool bad;   // the reference raises here (nn/atom_ref.py:27): NaN energy + sticky M3G_TOPO_ERR_SPECIES
  const int64_t ty = species_index(types[gid], num_types, bad);
  ea[gid] = bad ? __builtin_nanf("") : elemental[ty] / energy_scale + od[gid] * sigmoid_f(og[gid]);
  if (bad) flag_bad_species(topo_flags);
}
__global__ void __launch_bounds__(256) g_readout_seed(int64_t N, float energy_scale, const float* __restrict__ od, const float* __restrict__ og,
                                                      float* __restrict__ d_od, float* __restrict__ d_og) {
  GEN_IDX(N);
  const float sg = sigmoid_f(og[gid]);
  d_od[gid] = energy_scale * sg;
  d_og[gid] = energy_scale * od[gid] * sg * (1.f - sg);
}
__global__ void __launch_bounds__(256) g_add(int64_t n, const float* __restrict__ a, float* __restrict__ y) {
  GEN_IDX(n);
  y[gid] += a[gid];
}

// ---- weights ------------------------------------------------------------------------------------------------------------
struct GenBlockW {
  const float *w1s, *b1s, *wd, *wg;   // ThreeBodyInteration: linear_sigmoid1 [C,D] [C], gated_mlp dense/gate [D,C]
  struct Mlp { const float *w1d, *w1g, *b1d, *b1g, *w2d, *w2g, *b2d, *b2g, *wl; } e, n;   // [D,3D] x2, [D] x2, [D,D] x2, [D] x2, [D,R]
};
struct GenW {
  const float *emb, *adj, *elemental;
  GenBlockW blk[32];
  const float *rw[2][3], *rb[2][3];   // readout [dense|gate][layer]
};

struct GenWork {
  float *u, *d, *h, *hp, *fc3, *fc3p, *q, *qp;
  float* x[33];
  float* e[33];
  float* pe0;                       // [E,D] edge-embedding pre-activation
  // saved per block; MLP index 0 = edge update, 1 = node message; every array contiguous [E,D] unless noted
  struct Blk { float *v /*[N,C]*/, *Ssum /*[E,C]*/, *m /*[E,C]*/, *pd, *pg, *e1, *p1d[2], *p1g[2], *p2d[2], *p2g[2], *lin[2]; } b[32];
  float *cat /*[E,3D]*/, *hd, *hg, *msg, *t0, *t1;     // scratch [E,D]
  float *rp1d, *rp1g, *rp2d, *rp2g, *rod, *rog, *rh0, *rh1;   // readout [N,D] / [N]
  float *dx, *dx2, *de, *dh /*[E,R]*/, *dd, *du, *dr, *dm, *dgq, *dp1d[2], *dp1g[2], *dTA /*[N,4D]*/, *dTB, *dv /*[N,C]*/;
  size_t total;
};

}  // namespace

static GenWork gen_carve(int D, int C, int R, int B, int64_t N, int64_t E, int64_t S, void* base) {
  GenWork w{};
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t n) { float* r = p ? (float*)(p + off) : nullptr; off += (n * sizeof(float) + 255) / 256 * 256; return r; };
  const size_t e = (size_t)E, n = (size_t)N;
  w.u = take(e * 3); w.d = take(e); w.h = take(e * R); w.hp = take(e * R); w.fc3 = take(e); w.fc3p = take(e); w.q = take(e * C); w.qp = take(e * C);
  for (int b = 0; b <= B; ++b) { w.x[b] = take(n * D); w.e[b] = take(e * D); }
  w.pe0 = take(e * D);
  for (int b = 0; b < B; ++b) {
    auto& k = w.b[b];
    k.v = take(n * C); k.Ssum = take(e * C); k.m = take(e * C); k.pd = take(e * D); k.pg = take(e * D); k.e1 = take(e * D);
    for (int m = 0; m < 2; ++m) { k.p1d[m] = take(e * D); k.p1g[m] = take(e * D); k.p2d[m] = take(e * D); k.p2g[m] = take(e * D); k.lin[m] = take(e * D); }
  }
  w.cat = take(e * 3 * D); w.hd = take(e * D); w.hg = take(e * D); w.msg = take(e * D); w.t0 = take(e * D); w.t1 = take(e * D);
  w.rp1d = take(n * D); w.rp1g = take(n * D); w.rp2d = take(n * D); w.rp2g = take(n * D); w.rod = take(n); w.rog = take(n);
  w.rh0 = take(n * D); w.rh1 = take(n * D);
  w.dx = take(n * D); w.dx2 = take(n * D); w.de = take(e * D); w.dh = take(e * R); w.dd = take(e); w.du = take(e * 3); w.dr = take(e * 3);
  w.dm = take(e * C); w.dgq = take(e * C);
  for (int m = 0; m < 2; ++m) { w.dp1d[m] = take(e * D); w.dp1g[m] = take(e * D); }
  w.dTA = take(n * 4 * D); w.dTB = take(n * 4 * D); w.dv = take(n * C);
  off += ((n + (size_t)S * 2 + 64) * sizeof(float) + 255) / 256 * 256;   // tail scratch for optional outputs
  w.total = off;
  return w;
}

size_t generic_workspace_bytes(const m3g_plan* plan, int64_t N, int64_t E, int64_t T, int64_t S) {
  (void)T;
  const m3g_config& c = plan->cfg;
  return gen_carve(c.embedding_dim, c.l_max * c.n_max, c.n_max, c.num_blocks, N, E, S, nullptr).total;
}

void generic_free(m3g_plan* plan) {
  if (plan->d_generic) { (void)hipFree(plan->d_generic); plan->d_generic = nullptr; }
  plan->generic_off.clear();
}

int generic_commit(m3g_plan* plan) {
  generic_free(plan);
  std::vector<float> blob;
  auto put = [&](const std::string& key, const std::vector<float>& v) {
    plan->generic_off[key] = blob.size();
    blob.insert(blob.end(), v.begin(), v.end());
    blob.resize((blob.size() + 63) / 64 * 64);
  };
  for (auto& kv : plan->params) put(kv.first, kv.second);
  put("__elemental", plan->cvals.at("elemental_energies"));
  M3G_HIP_CHECK(hipMalloc((void**)&plan->d_generic, blob.size() * sizeof(float)));
  M3G_HIP_CHECK(hipMemcpy(plan->d_generic, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
  return M3G_OK;
}

static GenConsts gen_consts(const m3g_plan* plan) {
  const m3g_config& cfg = plan->cfg;
  GenConsts c{};
  c.L = cfg.l_max; c.R = cfg.n_max; c.C = c.L * c.R; c.D = cfg.embedding_dim; c.B = cfg.num_blocks; c.num_types = cfg.num_types;
  c.length_scale = (float)cfg.length_scale; c.energy_scale = (float)cfg.energy_scale;
  const double rc = cfg.cutoff / cfg.length_scale, rc3 = cfg.threebody_cutoff / cfg.length_scale;   // model/build.py:34-35
  c.rc = (float)rc; c.rc3 = (float)rc3;
  c.ref_legendre = plan->legendre_ref ? 1 : 0;
  const float pi_f = (float)M_PI;
  const auto& em = plan->cvals.at("em");
  const auto& dm = plan->cvals.at("dm");
  for (int m = 0; m < c.R; ++m) {
    c.a1[m] = ((float)(m + 1) * pi_f) / (float)rc;  // nn/featurizer.py:87-88
    c.a2[m] = ((float)(m + 2) * pi_f) / (float)rc;
    c.coeff[m] = plan->cvals.at("coeff")[m];
    c.rec_mul[m] = m > 0 ? sqrtf(em[m] / dm[m - 1]) : 0.f;
    c.rec_div[m] = sqrtf(dm[m]);
  }
  for (int l = 0; l < c.L; ++l) {
    c.ynorm[l] = (float)std::sqrt((2 * l + 1) / (4.0 * M_PI));
    for (int n = 0; n < c.R; ++n) {
      c.zeros[l][n] = plan->cvals.at("bessel_zeros")[l * c.R + n];
      c.factors[l][n] = plan->cvals.at("factors")[l * c.R + n];
    }
  }
  return c;
}

static GenW gen_weights(const m3g_plan* plan) {
  GenW w{};
  auto at = [&](const std::string& key) -> const float* { return plan->d_generic + plan->generic_off.at(key); };
  w.emb = at("model.3.linear.weight");
  w.adj = at("model.5.linear.weight");
  w.elemental = at("__elemental");
  const int B = plan->cfg.num_blocks;
  for (int b = 0; b < B; ++b) {
    const std::string tb = "model." + std::to_string(6 + 2 * b), cv = "model." + std::to_string(7 + 2 * b);
    GenBlockW& k = w.blk[b];
    k.w1s = at(tb + ".linear_sigmoid1.weight"); k.b1s = at(tb + ".linear_sigmoid1.bias");
    k.wd = at(tb + ".gated_mlp.dense.0.weight"); k.wg = at(tb + ".gated_mlp.gate.0.weight");
    const char* names[2] = {".concat_edge_update", ".concat_node_update"};
    const char* lins[2] = {".edge_linear.weight", ".node_linear.weight"};
    for (int m = 0; m < 2; ++m) {
      GenBlockW::Mlp& q = m == 0 ? k.e : k.n;
      const std::string pre = cv + names[m];
      q.w1d = at(pre + ".dense.0.weight"); q.w1g = at(pre + ".gate.0.weight"); q.b1d = at(pre + ".dense.0.bias"); q.b1g = at(pre + ".gate.0.bias");
      q.w2d = at(pre + ".dense.2.weight"); q.w2g = at(pre + ".gate.2.weight"); q.b2d = at(pre + ".dense.2.bias"); q.b2g = at(pre + ".gate.2.bias");
      q.wl = at(cv + lins[m]);
    }
  }
  const std::string ro = "model." + std::to_string(6 + 2 * B) + ".gated.";
  const char* br[2] = {"dense", "gate"};
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 3; ++i) {
      w.rw[g][i] = at(ro + br[g] + "." + std::to_string(2 * i) + ".weight");
      w.rb[g][i] = at(ro + br[g] + "." + std::to_string(2 * i) + ".bias");
    }
  return w;
}

// torch Linear y = W x + b with W [out, in] row-major: B(k, j) = W[j*in + k]  ->  sbk = 1, sbj = in
static void linear(hipStream_t s, int64_t n, int out, int in, const float* X, int64_t ldx, const float* W, const float* b, float* Y, int64_t ldy,
                   bool beta = false) {
  gemm(s, n, out, in, X, ldx, W, 1, in, b, Y, ldy, beta);
}
// y = x W (the transposed use of the same torch weight): B(k, j) = W[k*in + j] with k over `out`  ->  sbk = in, sbj = 1
static void linear_t(hipStream_t s, int64_t n, int out, int in, const float* DY, int64_t ldd, const float* W, float* DX, int64_t ldx, bool beta = false) {
  gemm(s, n, in, out, DY, ldd, W, in, 1, nullptr, DX, ldx, beta);
}
static void gated(hipStream_t s, int64_t n, const float* pd, const float* pg, const float* lin, const float* base, float* y) {
  if (n > 0) hipLaunchKernelGGL(g_gated, grid1(n), dim3(256), 0, s, n, pd, pg, lin, base, y);
}

// one conv GatedMLP forward on the concat rows (nn/conv.py:68-97, nn/core.py:61-62): saves p1d, p1g, p2d, p2g, lin = W_l h (all [E,D]);
// out = silu(p2d) sigmoid(p2g) lin (+ base)
static void gen_mlp_forward(hipStream_t s, int64_t E, int D, int R, const GenBlockW::Mlp& q, const GenWork& w, const GenWork::Blk& k, int m,
                            const float* base, float* out) {
  linear(s, E, D, 3 * D, w.cat, 3 * D, q.w1d, q.b1d, k.p1d[m], D);
  linear(s, E, D, 3 * D, w.cat, 3 * D, q.w1g, q.b1g, k.p1g[m], D);
  map(s, E * D, OP_SILU, k.p1d[m], w.hd);
  map(s, E * D, OP_SILU, k.p1g[m], w.hg);
  linear(s, E, D, D, w.hd, D, q.w2d, q.b2d, k.p2d[m], D);
  linear(s, E, D, D, w.hg, D, q.w2g, q.b2g, k.p2g[m], D);
  linear(s, E, D, R, w.h, R, q.wl, nullptr, k.lin[m], D);
  gated(s, E * D, k.p2d[m], k.p2g[m], k.lin[m], base, out);
}

// reverse of gen_mlp_forward (oracle/staged.py _mlp2_backward): d_upd [E,D] upstream; de_out (+)= W1c^T d_p1; dh += (d_upd out) W_l;
// leaves d_p1 in w.dp1d[m] / w.dp1g[m]
static void gen_mlp_reverse(hipStream_t s, int64_t E, int D, int R, const GenBlockW::Mlp& q, const GenWork& w, const GenWork::Blk& k, int m,
                            const float* d_upd, float* de_out) {
  // gating: d_p2d, d_p2g into t0 / t1, d_lin into msg
  hipLaunchKernelGGL(g_gated_rev, grid1(E * D), dim3(256), 0, s, E * D, k.p2d[m], k.p2g[m], k.lin[m], d_upd, w.t0, w.t1, w.msg);
  linear_t(s, E, D, R, w.msg, D, q.wl, w.dh, R, /*beta=*/true);            // dL/dh += d_lin W_l          (W_l [D,R])
  linear_t(s, E, D, D, w.t0, D, q.w2d, w.dp1d[m], D);                       // d hidden dense = d_p2d W2d
  linear_t(s, E, D, D, w.t1, D, q.w2g, w.dp1g[m], D);
  map(s, E * D, OP_MUL_DSILU, k.p1d[m], w.dp1d[m]);
  map(s, E * D, OP_MUL_DSILU, k.p1g[m], w.dp1g[m]);
  // e-part of W1 (columns 2D..3D of [D,3D]): de_out[e,k] += sum_o d_p1[e,o] W1[o, 2D + k]
  gemm(s, E, D, D, w.dp1d[m], D, q.w1d + 2 * D, 3 * D, 1, nullptr, de_out, D, /*beta=*/true);
  gemm(s, E, D, D, w.dp1g[m], D, q.w1g + 2 * D, 3 * D, 1, nullptr, de_out, D, /*beta=*/true);
}

int generic_energy_forces(const m3g_plan* plan, const m3g_io* io, void* workspace, size_t workspace_bytes, hipStream_t s) {
  const int64_t N = io->n_atoms, E = io->n_edges, T = io->n_triplets, S = io->n_structs;
  const GenConsts c = gen_consts(plan);
  const int D = c.D, C = c.C, R = c.R, B = c.B;
  GenWork w = gen_carve(D, C, R, B, N, E, S, nullptr);
  if (!workspace || workspace_bytes < w.total) { set_error("workspace too small: %zu < %zu", workspace_bytes, w.total); return M3G_ERR_SIZE; }
  w = gen_carve(D, C, R, B, N, E, S, workspace);
  const GenW W = gen_weights(plan);
  Topo t = topo_carve(N, E, T, S, const_cast<void*>(io->topo));
  float* tail = (float*)((char*)workspace + w.total - ((N + S * 2 + 64) * sizeof(float) + 255) / 256 * 256);
  float* ea = io->scaled_atomic_energies ? io->scaled_atomic_energies : tail;
  float* st = io->scaled_total_energy ? io->scaled_total_energy : tail + N;
  Consts cc{};   // the shared launchers read only these fields
  cc.energy_scale = c.energy_scale; cc.length_scale = c.length_scale; cc.B = B; cc.num_types = c.num_types;

  // ---------------- forward ----------------
  if (E > 0)
    hipLaunchKernelGGL(g_geometry, grid1(E), dim3(256), 0, s, c, E, t.src, t.dst, t.batch, io->pos, io->lattice, io->edge_cell_shift, w.u, w.d, w.h,
                       w.hp, w.fc3, w.fc3p, w.q, w.qp);
  if (N > 0) hipLaunchKernelGGL(g_embed_x, grid1(N * D), dim3(256), 0, s, N, D, c.num_types, io->atom_types, W.emb, w.x[0]);
  linear(s, E, D, R, w.h, R, W.adj, nullptr, w.pe0, D);                     // e0 = SiLU(W_adj h), nn/featurizer.py:128-132
  map(s, E * D, OP_SILU, w.pe0, w.e[0]);
  for (int b = 0; b < B; ++b) {
    const GenBlockW& kw = W.blk[b];
    const GenWork::Blk& k = w.b[b];
    linear(s, N, C, D, w.x[b], D, kw.w1s, kw.b1s, k.v, C);                  // v = sigmoid(W1 x + b1), nn/interaction.py:204-205
    map(s, N * C, OP_SIGMOID, k.v, k.v);
    if (E > 0) {
      hipLaunchKernelGGL(g_threebody_fwd, grid1(E * c.L), dim3(256), 0, s, c, E, t.t1_ptr, t.t1_e2, t.dst, w.u, w.q, k.v, k.Ssum);
      hipLaunchKernelGGL(g_scale_rows, grid1(E * C), dim3(256), 0, s, E, C, w.fc3, k.Ssum, k.m);
    }
    linear(s, E, D, C, k.m, C, kw.wd, nullptr, k.pd, D);                    // three-body gated update, nn/interaction.py:220-221
    linear(s, E, D, C, k.m, C, kw.wg, nullptr, k.pg, D);
    gated(s, E * D, k.pd, k.pg, nullptr, w.e[b], k.e1);
    if (E > 0) hipLaunchKernelGGL(g_concat, grid1(E * 3 * D), dim3(256), 0, s, E, D, t.src, t.dst, w.x[b], k.e1, w.cat);
    gen_mlp_forward(s, E, D, R, kw.e, w, k, 0, k.e1, w.e[b + 1]);           // edge update
    if (E > 0) hipLaunchKernelGGL(g_concat, grid1(E * 3 * D), dim3(256), 0, s, E, D, t.src, t.dst, w.x[b], w.e[b + 1], w.cat);
    gen_mlp_forward(s, E, D, R, kw.n, w, k, 1, nullptr, w.msg);             // node message
    if (N > 0) {
      M3G_HIP_CHECK(hipMemcpyAsync(w.x[b + 1], w.x[b], sizeof(float) * N * D, hipMemcpyDeviceToDevice, s));
      hipLaunchKernelGGL(g_segsum, grid1(N * D), dim3(256), 0, s, N, D, t.row_ptr, nullptr, w.msg, (int64_t)D, w.x[b + 1], (int64_t)D, 1);
    }
  }
  // readout (nn/readout.py:39-58)
  const float* xB = w.x[B];
  linear(s, N, D, D, xB, D, W.rw[0][0], W.rb[0][0], w.rp1d, D);
  linear(s, N, D, D, xB, D, W.rw[1][0], W.rb[1][0], w.rp1g, D);
  map(s, N * D, OP_SILU, w.rp1d, w.rh0);
  map(s, N * D, OP_SILU, w.rp1g, w.rh1);
  linear(s, N, D, D, w.rh0, D, W.rw[0][1], W.rb[0][1], w.rp2d, D);
  linear(s, N, D, D, w.rh1, D, W.rw[1][1], W.rb[1][1], w.rp2g, D);
  map(s, N * D, OP_SILU, w.rp2d, w.rh0);
  map(s, N * D, OP_SILU, w.rp2g, w.rh1);
  linear(s, N, 1, D, w.rh0, D, W.rw[0][2], W.rb[0][2], w.rod, 1);
  linear(s, N, 1, D, w.rh1, D, W.rw[1][2], W.rb[1][2], w.rog, 1);
  if (N > 0) hipLaunchKernelGGL(g_atomic_energy, grid1(N), dim3(256), 0, s, N, c.num_types, c.energy_scale, io->atom_types, W.elemental, w.rod, w.rog, ea, t.flags);
  M3G_HIP_CHECK(hipMemsetAsync(st, 0, sizeof(float) * S, s));
  launch_energy_sums(cc, t, ea, st, io->total_energy, s);

  // optional outputs
  if (io->node_features && N > 0) M3G_HIP_CHECK(hipMemcpyAsync(io->node_features, w.x[B], sizeof(float) * N * D, hipMemcpyDeviceToDevice, s));
  if (io->edge_attr && E > 0) M3G_HIP_CHECK(hipMemcpyAsync(io->edge_attr, w.e[B], sizeof(float) * E * D, hipMemcpyDeviceToDevice, s));
  if (io->edge_distances && E > 0) M3G_HIP_CHECK(hipMemcpyAsync(io->edge_distances, w.d, sizeof(float) * E, hipMemcpyDeviceToDevice, s));
  if (io->edge_weights && E > 0) M3G_HIP_CHECK(hipMemcpyAsync(io->edge_weights, w.h, sizeof(float) * E * R, hipMemcpyDeviceToDevice, s));
  if (io->triplet_angles) launch_triplet_angles(t, io->triplet_edge_index, w.u, io->triplet_angles, s);
  if (io->mid_edge_features)
    for (int b = 0; b < B; ++b)
      if (E > 0) M3G_HIP_CHECK(hipMemcpyAsync(io->mid_edge_features + (size_t)b * E * C, w.b[b].m, sizeof(float) * E * C, hipMemcpyDeviceToDevice, s));

  // ---------------- reverse ----------------
  if (!io->forces) {
    if (io->stresses) { set_error("stresses require forces"); return M3G_ERR_VALUE; }
    M3G_HIP_CHECK(hipGetLastError());
    return M3G_OK;
  }
  // readout reverse: dL/d eps = energy_scale
  if (N > 0) hipLaunchKernelGGL(g_readout_seed, grid1(N), dim3(256), 0, s, N, c.energy_scale, w.rod, w.rog, w.rod, w.rog);   // in place: d_od, d_og
  linear_t(s, N, 1, D, w.rod, 1, W.rw[0][2], w.rh0, D);                    // d hidden-2 dense = d_od w3d
  linear_t(s, N, 1, D, w.rog, 1, W.rw[1][2], w.rh1, D);
  map(s, N * D, OP_MUL_DSILU, w.rp2d, w.rh0);                               // d p2
  map(s, N * D, OP_MUL_DSILU, w.rp2g, w.rh1);
  linear_t(s, N, D, D, w.rh0, D, W.rw[0][1], w.rp2d, D);                    // d hidden-1 (re-using the p2 buffers)
  linear_t(s, N, D, D, w.rh1, D, W.rw[1][1], w.rp2g, D);
  map(s, N * D, OP_MUL_DSILU, w.rp1d, w.rp2d);                              // d p1
  map(s, N * D, OP_MUL_DSILU, w.rp1g, w.rp2g);
  linear_t(s, N, D, D, w.rp2d, D, W.rw[0][0], w.dx, D);
  linear_t(s, N, D, D, w.rp2g, D, W.rw[1][0], w.dx, D, /*beta=*/true);
  M3G_HIP_CHECK(hipMemsetAsync(w.de, 0, sizeof(float) * E * D, s));
  M3G_HIP_CHECK(hipMemsetAsync(w.dh, 0, sizeof(float) * E * R, s));
  M3G_HIP_CHECK(hipMemsetAsync(w.dd, 0, sizeof(float) * E, s));
  M3G_HIP_CHECK(hipMemsetAsync(w.du, 0, sizeof(float) * E * 3, s));
  float* dx_cur = w.dx;
  float* dx_alt = w.dx2;
  for (int b = B - 1; b >= 0; --b) {
    const GenBlockW& kw = W.blk[b];
    const GenWork::Blk& k = w.b[b];
    // node-message MLP: d msg[e] = dx[centre(e)]
    if (E > 0) hipLaunchKernelGGL(g_gather, grid1(E * D), dim3(256), 0, s, E, D, t.src, dx_cur, (int64_t)D, w.hd, (int64_t)D, 0);
    gen_mlp_reverse(s, E, D, R, kw.n, w, k, 1, w.hd, w.de);                 // de (dL/de2) += node MLP's contribution
    // edge-update MLP with upstream dL/de2 (a copy: de itself receives the contribution)
    if (E > 0) M3G_HIP_CHECK(hipMemcpyAsync(w.hg, w.de, sizeof(float) * E * D, hipMemcpyDeviceToDevice, s));
    gen_mlp_reverse(s, E, D, R, kw.e, w, k, 0, w.hg, w.de);                 // de = dL/de1
    // three-body gated update reverse: d_pd, d_pg -> dm
    if (E > 0) hipLaunchKernelGGL(g_gated_rev, grid1(E * D), dim3(256), 0, s, E * D, k.pd, k.pg, nullptr, w.de, w.t0, w.t1, nullptr);
    linear_t(s, E, D, C, w.t0, D, kw.wd, w.dm, C);
    linear_t(s, E, D, C, w.t1, D, kw.wg, w.dm, C, /*beta=*/true);
    if (E > 0)
      hipLaunchKernelGGL(g_threebody_rev, grid1(E), dim3(256), 0, s, c, E, t.t1_ptr, t.t1_e2, t.t2_ptr, t.t2_e1, t.dst, w.u, w.fc3, w.fc3p, w.q, w.qp,
                         k.v, k.Ssum, w.dm, w.dd, w.du, w.dgq);
    if (b > 0 && N > 0) {   // x^0 is the species embedding: its gradient is never needed
      // d_TA / d_TB = dp1 rows summed by centre / by neighbour, columns [edge dense | edge gate | node dense | node gate]
      for (int m = 0; m < 2; ++m) {
        hipLaunchKernelGGL(g_segsum, grid1(N * D), dim3(256), 0, s, N, D, t.row_ptr, nullptr, w.dp1d[m], (int64_t)D, w.dTA + (2 * m) * D, (int64_t)4 * D, 0);
        hipLaunchKernelGGL(g_segsum, grid1(N * D), dim3(256), 0, s, N, D, t.row_ptr, nullptr, w.dp1g[m], (int64_t)D, w.dTA + (2 * m + 1) * D, (int64_t)4 * D, 0);
        hipLaunchKernelGGL(g_segsum, grid1(N * D), dim3(256), 0, s, N, D, t.in_ptr, t.in_edge, w.dp1d[m], (int64_t)D, w.dTB + (2 * m) * D, (int64_t)4 * D, 0);
        hipLaunchKernelGGL(g_segsum, grid1(N * D), dim3(256), 0, s, N, D, t.in_ptr, t.in_edge, w.dp1g[m], (int64_t)D, w.dTB + (2 * m + 1) * D, (int64_t)4 * D, 0);
      }
      hipLaunchKernelGGL(g_segsum, grid1(N * C), dim3(256), 0, s, N, C, t.in_ptr, t.in_edge, w.dgq, (int64_t)C, w.dv, (int64_t)C, 0);
      map(s, N * C, OP_MUL_DSIGMOID_OF_V, k.v, w.dv);
      M3G_HIP_CHECK(hipMemcpyAsync(dx_alt, dx_cur, sizeof(float) * N * D, hipMemcpyDeviceToDevice, s));
      const GenBlockW::Mlp* mm[2] = {&kw.e, &kw.n};
      for (int m = 0; m < 2; ++m) {   // x_i part: columns 0..D of W1, x_j part: columns D..2D
        gemm(s, N, D, D, w.dTA + (2 * m) * D, 4 * D, mm[m]->w1d, 3 * D, 1, nullptr, dx_alt, D, true);
        gemm(s, N, D, D, w.dTA + (2 * m + 1) * D, 4 * D, mm[m]->w1g, 3 * D, 1, nullptr, dx_alt, D, true);
        gemm(s, N, D, D, w.dTB + (2 * m) * D, 4 * D, mm[m]->w1d + D, 3 * D, 1, nullptr, dx_alt, D, true);
        gemm(s, N, D, D, w.dTB + (2 * m + 1) * D, 4 * D, mm[m]->w1g + D, 3 * D, 1, nullptr, dx_alt, D, true);
      }
      linear_t(s, N, C, D, w.dv, C, kw.w1s, dx_alt, D, true);
      float* tmp = dx_cur; dx_cur = dx_alt; dx_alt = tmp;
    }
  }
  // edge embedding reverse: dh += (de * SiLU'(pe0)) W_adj
  map(s, E * D, OP_MUL_DSILU, w.pe0, w.de);
  linear_t(s, E, D, R, w.de, D, W.adj, w.dh, R, true);
  if (E > 0) hipLaunchKernelGGL(g_geometry_rev, grid1(E), dim3(256), 0, s, E, R, w.u, w.d, w.hp, w.dh, w.dd, w.du, w.dr);
  launch_force_gather(c.length_scale, t, w.dr, io->forces, io->stresses, s);
  if (io->stresses) {
    if (plan->stress_mode == 1) {
      Work ww{};
      ww.u = w.u; ww.d = w.d; ww.dr = w.dr;
      launch_stress_pair(t, ww, io->lattice, io->stresses, s);
    } else {
      launch_stress(cc, t, io->pos, io->lattice, io->forces, io->stresses, s);
    }
  }
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}

}  // namespace m3g

// ---------------------------------------------------------------------------------------------- stand-alone stages
// The reference's block modules are usable on their own (tests/test_model.py:14-38 calls the bare Sequential; nn/core.py,
// nn/featurizer.py, nn/interaction.py, nn/conv.py, nn/readout.py each define a forward).  These entry points run one module
// forward on plain row-major device tensors of any size, on the same run-time-sized kernels as the any-size path.
namespace m3g {
namespace {
__global__ void __launch_bounds__(256) g_mul(int64_t n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y) {
  GEN_IDX(n);
  y[gid] = a[gid] * b[gid];
}
// chi[l, n, t] = j_l(z_ln r_t / rc) / factors[l, n]   (NormalizedSphericalBessel.forward, nn/interaction.py:268-281)
__global__ void __launch_bounds__(256) g_bessel_basis(GenConsts c, int64_t n, const float* __restrict__ rs, float* __restrict__ out) {
  GEN_IDX(n * c.C);
  const int64_t t = gid % n;
  const int cc = (int)(gid / n), l = cc / c.R, k = cc % c.R;
  float jl[kGL], djl[kGL];
  g_bessel(c.L, c.zeros[l][k] * rs[t] / c.rc, jl, djl);
  out[gid] = jl[l] / c.factors[l][k];
}
// per triplet and order l: m[e1, l*R+n] += chi_ln(d_ik) Y_l(cos) fc(d_ij) fc(d_ik) v[k, l*R+n]   (nn/interaction.py:188-217; the
// module takes cos(theta) and the distances from the graph, so the sum runs over the caller's triplet list: float atomics)
__global__ void __launch_bounds__(256) g_threebody_standalone(GenConsts c, int64_t T, int64_t E, const int64_t* __restrict__ ei,
                                                              const int64_t* __restrict__ tei, const float* __restrict__ dist,
                                                              const float* __restrict__ angles, const float* __restrict__ v, float* __restrict__ m) {
  GEN_IDX(T * c.L);
  const int64_t t = gid / c.L;
  const int l = (int)(gid % c.L);
  const int64_t e1 = tei[t], e2 = tei[T + t];
  const float d1 = dist[e1], d2 = dist[e2];
  auto fc = [&](float d) {
    const float rho = d / c.rc3;
    if (rho > 1.f) return 0.f;
    const float r2 = rho * rho, r3 = r2 * rho;
    return 1.f - 6.f * r3 * r2 + 15.f * r2 * r2 - 10.f * r3;
  };
  const float f = fc(d1) * fc(d2);
  float P[kGL], dP[kGL];
  g_legendre(c.L, angles[t], P, dP);
  const float y = c.ynorm[l] * P[l] * f;
  const int64_t k = ei[E + e2];
  for (int n = 0; n < c.R; ++n) {
    float jl[kGL], djl[kGL];
    g_bessel(c.L, c.zeros[l][n] * d2 / c.rc, jl, djl);
    atomicAdd(&m[e1 * c.C + l * c.R + n], jl[l] / c.factors[l][n] * y * v[k * c.C + l * c.R + n]);
  }
}
__global__ void __launch_bounds__(256) g_atomic_energy_standalone(int64_t N, float energy_scale, const float* __restrict__ elemental_per_atom,
                                                                  const float* __restrict__ od, const float* __restrict__ og,
                                                                  const int64_t* __restrict__ batch, float* __restrict__ ea,
                                                                  float* __restrict__ scaled_total) {
  GEN_IDX(N);
  const float v = elemental_per_atom[gid] / energy_scale + od[gid] * sigmoid_f(og[gid]);
  ea[gid] = v;
  atomicAdd(&scaled_total[batch[gid]], v);
}
__global__ void __launch_bounds__(256) g_scale(int64_t n, float a, const float* __restrict__ x, float* __restrict__ y) {
  GEN_IDX(n);
  y[gid] = a * x[gid];
}
static GenConsts basis_consts(int l_max, int n_max, double rc, double rc3, const float* zeros, const float* factors) {
  GenConsts c{};
  c.L = l_max; c.R = n_max; c.C = l_max * n_max; c.rc = (float)rc; c.rc3 = (float)rc3;
  for (int l = 0; l < l_max; ++l) {
    c.ynorm[l] = (float)std::sqrt((2 * l + 1) / (4.0 * M_PI));
    for (int n = 0; n < n_max; ++n) { c.zeros[l][n] = zeros[l * n_max + n]; c.factors[l][n] = factors[l * n_max + n]; }
  }
  return c;
}
}  // namespace
}  // namespace m3g

using namespace m3g;

// torch.nn.Linear (+ activation): Y = act(X W^T + b); act 0 none, 1 SiLU, 2 sigmoid.  X [n,in], W [out,in], Y [n,out].
extern "C" int m3g_linear(int64_t n, int32_t in, int32_t out, const float* X, const float* W, const float* b, int32_t act, float* Y, void* stream_) {
  if (n < 0 || in < 1 || out < 1 || (n > 0 && (!X || !W || !Y)) || act < 0 || act > 2) { set_error("m3g_linear: bad argument"); return M3G_ERR_VALUE; }
  hipStream_t s = (hipStream_t)stream_;
  linear(s, n, out, in, X, in, W, b, Y, out);
  if (act == 1) map(s, n * out, OP_SILU, Y, Y);
  if (act == 2) map(s, n * out, OP_SIGMOID, Y, Y);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}
// y = a * b elementwise (the final dense(x) * gate(x) of GatedMLP.forward, nn/core.py:61-62)
extern "C" int m3g_multiply(int64_t n, const float* a, const float* b, float* y, void* stream_) {
  if (n < 0 || (n > 0 && (!a || !b || !y))) { set_error("m3g_multiply: bad argument"); return M3G_ERR_VALUE; }
  if (n > 0) hipLaunchKernelGGL(g_mul, grid1(n), dim3(256), 0, (hipStream_t)stream_, n, a, b, y);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}
// NormalizedSphericalBessel.forward (nn/interaction.py:268-281): out [l_max, n_max, n]
extern "C" int m3g_bessel_basis(int32_t l_max, int32_t n_max, double cutoff, const float* host_zeros, const float* host_factors, int64_t n,
                                const float* rs, float* out, void* stream_) {
  if (l_max < 1 || l_max > kGL || n_max < 1 || n_max > kGR || !host_zeros || !host_factors || (n > 0 && (!rs || !out))) {
    set_error("m3g_bessel_basis: bad argument");
    return M3G_ERR_VALUE;
  }
  const GenConsts c = basis_consts(l_max, n_max, cutoff, cutoff, host_zeros, host_factors);
  if (n > 0) hipLaunchKernelGGL(g_bessel_basis, grid1(n * c.C), dim3(256), 0, (hipStream_t)stream_, c, n, rs, out);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}
// ThreeBodyInteration.forward (nn/interaction.py:187-223): edge_attr += GatedMLP(m), m from the graph's distances, angles and x.
// scratch: (N*C + E*C + 2*E*D) floats.  mid (optional) receives m [E,C].
extern "C" int m3g_three_body(int32_t l_max, int32_t n_max, int32_t D, double scaled_cutoff, double scaled_threebody_cutoff,
                              const float* host_zeros, const float* host_factors, int64_t N, int64_t E, int64_t T, const int64_t* edge_index,
                              const int64_t* triplet_edge_index, const float* edge_distances, const float* triplet_angles, const float* x,
                              const float* w_sigmoid, const float* b_sigmoid, const float* w_dense, const float* w_gate, float* scratch,
                              float* edge_attr, float* mid, void* stream_) {
  if (l_max < 1 || l_max > kGL || n_max < 1 || n_max > kGR || D < 1 || !host_zeros || !host_factors || !scratch) {
    set_error("m3g_three_body: bad argument");
    return M3G_ERR_VALUE;
  }
  hipStream_t s = (hipStream_t)stream_;
  const GenConsts c = basis_consts(l_max, n_max, scaled_cutoff, scaled_threebody_cutoff, host_zeros, host_factors);
  const int C = c.C;
  float* v = scratch;
  float* m = v + (size_t)N * C;
  float* pd = m + (size_t)E * C;
  float* pg = pd + (size_t)E * D;
  linear(s, N, C, D, x, D, w_sigmoid, b_sigmoid, v, C);
  map(s, N * C, OP_SIGMOID, v, v);
  M3G_HIP_CHECK(hipMemsetAsync(m, 0, sizeof(float) * E * C, s));
  if (T > 0) hipLaunchKernelGGL(g_threebody_standalone, grid1(T * c.L), dim3(256), 0, s, c, T, E, edge_index, triplet_edge_index, edge_distances,
                                triplet_angles, v, m);
  linear(s, E, D, C, m, C, w_dense, nullptr, pd, D);
  linear(s, E, D, C, m, C, w_gate, nullptr, pg, D);
  gated(s, E * D, pd, pg, nullptr, edge_attr, edge_attr);
  if (mid && E > 0) M3G_HIP_CHECK(hipMemcpyAsync(mid, m, sizeof(float) * E * C, hipMemcpyDeviceToDevice, s));
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}
// M3GNetConv.forward (nn/conv.py:63-97): edge update, then node update with centre aggregation.  host_params: 18 DEVICE pointers in
// the order edge {dense.0.weight, gate.0.weight, dense.0.bias, gate.0.bias, dense.2.weight, gate.2.weight, dense.2.bias, gate.2.bias,
// edge_linear.weight}, node {the same with node_linear.weight}.  scratch: m3g_conv_block_scratch_bytes.
extern "C" int m3g_conv_block_scratch_bytes(int32_t D, int64_t E, size_t* bytes) {
  if (!bytes || D < 1 || E < 0) { set_error("m3g_conv_block_scratch_bytes: bad argument"); return M3G_ERR_VALUE; }
  *bytes = ((size_t)E * D * (3 + 5 + 3) + 1024) * sizeof(float);
  return M3G_OK;
}
extern "C" int m3g_conv_block(int32_t D, int32_t R, int64_t N, int64_t E, int64_t T, int64_t S, const void* topo, const float* const* host_params,
                              const float* edge_weights, float* x, float* edge_attr, float* scratch, size_t scratch_bytes, void* stream_) {
  size_t need = 0;
  if (m3g_conv_block_scratch_bytes(D, E, &need) != M3G_OK || !host_params || !topo || !scratch || scratch_bytes < need) {
    set_error("m3g_conv_block: bad argument or scratch too small");
    return M3G_ERR_VALUE;
  }
  hipStream_t s = (hipStream_t)stream_;
  Topo t = topo_carve(N, E, T, S, const_cast<void*>(topo));
  GenWork w{};
  GenWork::Blk k{};
  float* p = scratch;
  auto take = [&](size_t n) { float* r = p; p += n; return r; };
  const size_t ed = (size_t)E * D;
  w.cat = take(3 * ed); w.hd = take(ed); w.hg = take(ed); w.msg = take(ed);
  w.h = const_cast<float*>(edge_weights);
  k.p1d[0] = k.p1d[1] = take(ed); k.p1g[0] = k.p1g[1] = take(ed); k.p2d[0] = k.p2d[1] = take(ed); k.p2g[0] = k.p2g[1] = take(ed);
  k.lin[0] = k.lin[1] = take(ed);
  GenBlockW::Mlp q[2];
  for (int m = 0; m < 2; ++m) {
    const float* const* a = host_params + 9 * m;
    q[m] = GenBlockW::Mlp{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]};
  }
  if (E > 0) hipLaunchKernelGGL(g_concat, grid1(E * 3 * D), dim3(256), 0, s, E, D, t.src, t.dst, x, edge_attr, w.cat);
  gen_mlp_forward(s, E, D, R, q[0], w, k, 0, edge_attr, edge_attr);   // e += GatedMLP(concat) * (W_e h)
  if (E > 0) hipLaunchKernelGGL(g_concat, grid1(E * 3 * D), dim3(256), 0, s, E, D, t.src, t.dst, x, edge_attr, w.cat);
  gen_mlp_forward(s, E, D, R, q[1], w, k, 1, nullptr, w.msg);
  if (N > 0) hipLaunchKernelGGL(g_segsum, grid1(N * D), dim3(256), 0, s, N, D, t.row_ptr, nullptr, w.msg, (int64_t)D, x, (int64_t)D, 1);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}
// AtomWiseReadout.forward (nn/readout.py:39-58).  host_params: 12 DEVICE pointers {dense.0.w, dense.0.b, dense.2.w, dense.2.b,
// dense.4.w, dense.4.b, gate.0.w, ...}; elemental_per_atom = graph["elemental_energies"] [N].  scratch: (6*N*D + 2*N) floats.
extern "C" int m3g_readout(int32_t D, int64_t N, int64_t S, const float* const* host_params, double energy_scale, const float* x,
                           const float* elemental_per_atom, const int64_t* batch, float* scaled_atomic, float* scaled_total, float* total,
                           float* scratch, void* stream_) {
  if (D < 1 || N < 0 || S < 0 || !host_params || !scaled_total || !total || (N > 0 && (!x || !elemental_per_atom || !batch || !scaled_atomic || !scratch))) {
    set_error("m3g_readout: bad argument");
    return M3G_ERR_VALUE;
  }
  hipStream_t s = (hipStream_t)stream_;
  const size_t nd = (size_t)N * D;
  float* hd = scratch; float* hg = hd + nd; float* t0 = hg + nd; float* t1 = t0 + nd; float* od = t1 + nd; float* og = od + N;
  const float* const* dp = host_params;
  const float* const* gp = host_params + 6;
  linear(s, N, D, D, x, D, dp[0], dp[1], hd, D);
  linear(s, N, D, D, x, D, gp[0], gp[1], hg, D);
  map(s, N * D, OP_SILU, hd, hd);
  map(s, N * D, OP_SILU, hg, hg);
  linear(s, N, D, D, hd, D, dp[2], dp[3], t0, D);
  linear(s, N, D, D, hg, D, gp[2], gp[3], t1, D);
  map(s, N * D, OP_SILU, t0, t0);
  map(s, N * D, OP_SILU, t1, t1);
  linear(s, N, 1, D, t0, D, dp[4], dp[5], od, 1);
  linear(s, N, 1, D, t1, D, gp[4], gp[5], og, 1);
  M3G_HIP_CHECK(hipMemsetAsync(scaled_total, 0, sizeof(float) * S, s));
  if (N > 0) hipLaunchKernelGGL(g_atomic_energy_standalone, grid1(N), dim3(256), 0, s, N, (float)energy_scale, elemental_per_atom, od, og, batch,
                                scaled_atomic, scaled_total);
  if (S > 0) hipLaunchKernelGGL(g_scale, grid1(S), dim3(256), 0, s, S, (float)energy_scale, scaled_total, total);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}
