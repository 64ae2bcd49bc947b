// Edge block on the matrix cores: stage S4 (three-body gated update + M3GNetConv edge and node MLPs) and its
// reverse B4, fused per 32-edge tile, fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32; gfx950 has no xf32).
// Reference: nn/interaction.py:220-221, nn/conv.py:63-97, nn/core.py:61-62; algebra: oracle/staged.py.
//
// Formulation: every dense layer is computed TRANSPOSED, Y^T[feature, edge] = W[feature, k] . X^T[k, edge]:
//   * the 32 edges of a tile sit on the MFMA column index (lane & 31), output features in the 16 accumulator
//     registers (row = (r&3) + 8*(r>>2) + 4*(lane>>5)), so all per-edge elementwise work is lane-local;
//   * an accumulator tile is directly the B operand of the next layer (register r of lane-half h carries
//     k = feat_of(r,h)); the matching k permutation is folded into the weight images (m3g_pack_mfma.hip),
//     which one workgroup copies into LDS once and every wave re-reads as the A operand (ds_read_b32,
//     conflict-free: 64 consecutive floats per k-step);
//   * layer-1 accumulators start from the gathered per-node tables TA[i] + TB[j] (x_i / x_j parts, bias folded);
//     layer-2 biases enter as one extra k-step against a constant-one operand.
// Edge features travel between blocks in a tile-SoA image ([tile][slot = kb*16 + r][64 lanes]) so every register
// load/store is one contiguous 256-B wave access; saved pre-activations use the same shape.
// One persistent workgroup (8 waves, 2 per SIMD) per CU; tiles are dealt so that workgroups sharing an XCD
// (blockIdx % 8) walk a contiguous edge range, keeping their TA/TB rows in that XCD's L2.
#include <utility>

#include "m3g_device.h"
#include "m3g_internal.h"

namespace m3g {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kFwdLdsFloats = 4 * kTbSteps * 64 + 2 * (4 * 2 * 16 * 64 + 2 * (2 * 2 * 16 * 64) + 2 * 2 * 64 + 2 * 2 * 64);
constexpr int kRevLdsFloats = 4 * kTbSteps * 64 + 4 * 16 * 64 + 2 * (2 * (2 * 2 * 16 * 64) + 2 * 4 * 16 * 64 + 64 * 4);
constexpr int kWaves = 8;
constexpr int kActPerTile = 2 * 128 * 64;  // floats of saved pre-activations per tile per block (both MLPs)

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f.template operator()<I>(), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float fsigmoid(float p) { return __builtin_amdgcn_rcpf(1.f + __expf(-p)); }
__device__ __forceinline__ float fsilu(float p) { return p * fsigmoid(p); }
__device__ __forceinline__ float fdsilu(float p) {
  float s = fsigmoid(p);
  return s * (1.f + p * (1.f - s));
}

// acc[ob] += Wimg(ob, :) . x   with x[kb] accumulator-layout tiles (chain image, see m3g_internal.h)
template <int OB, int KB, int XOFF = 0, int AOFF = 0, int NX, int NA>
__device__ __forceinline__ void chain(const float* img, const f32x16 (&x)[NX], f32x16 (&acc)[NA], int lane) {
  static_assert(XOFF + KB <= NX && AOFF + OB <= NA, "chain operand out of range");
  static_for<KB>([&]<int kb>() {
    static_for<16>([&]<int s>() {
      const float b = x[XOFF + kb][s];
      static_for<OB>([&]<int ob>() {
        const float a = img[((ob * KB + kb) * 16 + s) * 64 + lane];
        acc[AOFF + ob] = mfma32(a, b, acc[AOFF + ob]);
      });
    });
  });
}

__device__ __forceinline__ void zero(f32x16& v) {
  static_for<16>([&]<int r>() { v[r] = 0.f; });
}

// persistent tile walk: workgroup g of G (G % 8 == 0) -> XCD label g % 8 owns a contiguous chunk of tiles
struct TileWalk {
  int64_t tiles, per_xcd;
  int xcd, q, wgs_per_xcd, wave;
  __device__ TileWalk(int64_t n_tiles, int wave_id) {
    tiles = n_tiles;
    per_xcd = (n_tiles + 7) / 8;
    xcd = blockIdx.x & 7;
    q = blockIdx.x >> 3;
    wgs_per_xcd = gridDim.x >> 3;
    wave = wave_id;
  }
  __device__ int64_t tile(int it) const {
    int64_t local = ((int64_t)it * wgs_per_xcd + q) * kWaves + wave;
    if (local >= per_xcd) return -1;
    int64_t t = (int64_t)xcd * per_xcd + local;
    return t < tiles ? t : -1;
  }
};

__device__ __forceinline__ void load_image(float* lds, const float* __restrict__ src, int n_floats) {
  for (int i = threadIdx.x * 4; i < n_floats; i += blockDim.x * 4) *(f32x4*)(lds + i) = *(const f32x4*)(src + i);
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------- forward
struct FwdArgs {
  int64_t E, tiles;
  const float* img;        // forward weight image of this block
  const int32_t *src, *dst;
  const float *h, *m, *TA, *TB;
  float* e_soa;            // in/out
  float* act;              // [tiles][2][128][64]
  float* msg;              // [E][64] row-major
};

template <int TBS>
__device__ __forceinline__ void tb_preact(const float* tbimg, const float (&mb)[TBS], f32x16 (&pd)[2], f32x16 (&pg)[2], int lane) {
  zero(pd[0]); zero(pd[1]); zero(pg[0]); zero(pg[1]);
  static_for<TBS>([&]<int s>() {
    const float b = mb[s];
    pd[0] = mfma32(tbimg[(0 * kTbSteps + s) * 64 + lane], b, pd[0]);
    pd[1] = mfma32(tbimg[(1 * kTbSteps + s) * 64 + lane], b, pd[1]);
    pg[0] = mfma32(tbimg[(2 * kTbSteps + s) * 64 + lane], b, pg[0]);
    pg[1] = mfma32(tbimg[(3 * kTbSteps + s) * 64 + lane], b, pg[1]);
  });
}

// one conv GatedMLP, forward.  x = edge-feature input tile; out = MLP(x) * (W_l h)
__device__ __forceinline__ void mlp_forward_mfma(const float* lds, const MfmaMlpFwd& L, int mlp, const FwdArgs& a, int64_t ci,
                                                 int64_t cj, const float (&hb)[2], const f32x16 (&x)[2], float* act_tile,
                                                 f32x16 (&out)[2], int lane) {
  const int h = lane >> 5;
  f32x16 p1[4];
  {
    const float* ta = a.TA + ci * (4 * kDP) + mlp * (2 * kDP) + 4 * h;
    const float* tb = a.TB + cj * (4 * kDP) + mlp * (2 * kDP) + 4 * h;
    static_for<4>([&]<int ob>() {
      static_for<4>([&]<int g>() {
        const f32x4 va = *(const f32x4*)(ta + ob * 32 + 8 * g);
        const f32x4 vb = *(const f32x4*)(tb + ob * 32 + 8 * g);
        static_for<4>([&]<int qq>() { p1[ob][4 * g + qq] = va[qq] + vb[qq]; });
      });
    });
  }
  chain<4, 2>(lds + L.w1c, x, p1, lane);
  float* act_m = act_tile + mlp * (128 * 64) + lane;
  static_for<4>([&]<int ob>() {
    static_for<16>([&]<int r>() {
      act_m[(ob * 16 + r) * 64] = p1[ob][r];
      p1[ob][r] = fsilu(p1[ob][r]);
    });
  });
  f32x16 p2d[2], p2g[2];
  const float one = lane < 32 ? 1.f : 0.f;
  static_for<2>([&]<int ob>() {
    zero(p2d[ob]);
    zero(p2g[ob]);
    p2d[ob] = mfma32(lds[L.b2 + (0 * 2 + ob) * 64 + lane], one, p2d[ob]);
    p2g[ob] = mfma32(lds[L.b2 + (1 * 2 + ob) * 64 + lane], one, p2g[ob]);
  });
  chain<2, 2, 0, 0>(lds + L.w2d, p1, p2d, lane);  // hidden dense = p1[0..1]
  chain<2, 2, 2, 0>(lds + L.w2g, p1, p2g, lane);  // hidden gate  = p1[2..3]
  static_for<2>([&]<int ob>() {
    zero(out[ob]);
    static_for<2>([&]<int s>() { out[ob] = mfma32(lds[L.wl + (ob * 2 + s) * 64 + lane], hb[s], out[ob]); });
    static_for<16>([&]<int r>() {
      act_m[(64 + ob * 16 + r) * 64] = p2d[ob][r];
      act_m[(96 + ob * 16 + r) * 64] = p2g[ob][r];
      out[ob][r] = fsilu(p2d[ob][r]) * fsigmoid(p2g[ob][r]) * out[ob][r];
    });
  });
}

template <int TBS>
__global__ void __launch_bounds__(512, 2) k_edge_block_mfma(FwdArgs a, MfmaFwdLayout L) {
  __shared__ __attribute__((aligned(16))) float lds[kFwdLdsFloats];
  load_image(lds, a.img, kFwdLdsFloats);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  TileWalk walk(a.tiles, wave);
  for (int it = 0;; ++it) {
    const int64_t tile = walk.tile(it);
    if (tile < 0) break;
    const int64_t edge = tile * 32 + (lane & 31);
    const int64_t ec = edge < a.E ? edge : a.E - 1;
    const int64_t ci = a.src[ec], cj = a.dst[ec];
    float* e_tile = a.e_soa + tile * 2048 + lane;
    float* act_tile = a.act + tile * kActPerTile;
    f32x16 x[2];
    static_for<2>([&]<int kb>() { static_for<16>([&]<int r>() { x[kb][r] = e_tile[(kb * 16 + r) * 64]; }); });
    float mb[TBS], hb[2];
    static_for<TBS>([&]<int s>() { mb[s] = a.m[ec * kCP + 2 * s + h]; });
    hb[0] = a.h[ec * kRP + h];
    hb[1] = a.h[ec * kRP + 2 + h];
    {  // three-body gated update (nn/interaction.py:220-221)
      f32x16 pd[2], pg[2];
      tb_preact<TBS>(lds + L.tb, mb, pd, pg, lane);
      static_for<2>([&]<int kb>() { static_for<16>([&]<int r>() { x[kb][r] += fsilu(pd[kb][r]) * fsigmoid(pg[kb][r]); }); });
    }
    f32x16 out[2];
    mlp_forward_mfma(lds, L.mlp[0], 0, a, ci, cj, hb, x, act_tile, out, lane);  // edge update (nn/conv.py:68-75)
    static_for<2>([&]<int kb>() {
      static_for<16>([&]<int r>() {
        x[kb][r] += out[kb][r];
        e_tile[(kb * 16 + r) * 64] = x[kb][r];
      });
    });
    mlp_forward_mfma(lds, L.mlp[1], 1, a, ci, cj, hb, x, act_tile, out, lane);  // node message (nn/conv.py:77-89)
    if (edge < a.E) {
      float* mrow = a.msg + edge * kDP + 4 * h;
      static_for<2>([&]<int ob>() {
        static_for<4>([&]<int g>() {
          f32x4 v;
          static_for<4>([&]<int qq>() { v[qq] = out[ob][4 * g + qq]; });
          *(f32x4*)(mrow + ob * 32 + 8 * g) = v;
        });
      });
    }
  }
}

// ---------------------------------------------------------------------------------------------- reverse
struct RevArgs {
  int64_t E, tiles;
  const float* img;
  const int32_t* src;
  const float *h, *m, *act, *dx_new;
  float* de_soa;   // in: dL/d e (after this block), out: dL/d e (before this block)
  float* dm;       // [E][16]
  float* dh;       // [E][4]  (+=)
  float* dp1;      // [E][256]
};

// reverse of one conv GatedMLP: d_upd = dL/d(output); returns contrib = W1c^T d_p1, accumulates dL/dh into dhv
__device__ __forceinline__ void mlp_reverse_mfma(const float* lds, const MfmaMlpRev& L, int mlp, const RevArgs& a, int64_t edge,
                                                 const f32x4& hv, const float* act_tile, const f32x16 (&d_upd)[2],
                                                 f32x16 (&contrib)[2], f32x4& dhv, int lane) {
  const int h = lane >> 5;
  const float* act_m = act_tile + mlp * (128 * 64) + lane;
  f32x16 d2[4];  // d_p2d[0..1], d_p2g[0..1]
  // processed four registers at a time with scheduling fences: letting the compiler hoist all 64 loads and
  // 32 LDS reads of this phase costs > 256 VGPRs (spills)
  static_for<2>([&]<int ob>() {
    static_for<4>([&]<int g>() {
      static_for<4>([&]<int qq>() {
        constexpr int r = 4 * g + qq;
        const float p2d = act_m[(64 + ob * 16 + r) * 64], p2g = act_m[(96 + ob * 16 + r) * 64];
        const f32x4 w = *(const f32x4*)(lds + L.wl + (ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 4);
        const float s_lin = w[0] * hv[0] + w[1] * hv[1] + w[2] * hv[2] + w[3] * hv[3];
        const float sg = fsigmoid(p2g), sgd = fsigmoid(p2d), sd = p2d * sgd;
        const float du = d_upd[ob][r];
        const float d_out = du * s_lin, d_s = du * sd * sg;
        dhv[0] += d_s * w[0]; dhv[1] += d_s * w[1]; dhv[2] += d_s * w[2]; dhv[3] += d_s * w[3];
        d2[ob][r] = d_out * sg * (sgd * (1.f + p2d * (1.f - sgd)));
        d2[2 + ob][r] = d_out * sd * sg * (1.f - sg);
      });
      // pin the running dL/dh sums here: otherwise LLVM sinks the whole accumulation chain to its only use at
      // the end of the kernel and keeps every w / sd / sg temporary alive (1.3 KB of scratch per lane)
      asm volatile("" : "+v"(dhv[0]), "+v"(dhv[1]), "+v"(dhv[2]), "+v"(dhv[3]));
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  __builtin_amdgcn_sched_barrier(0);
  f32x16 dp1[4];
  static_for<4>([&]<int ob>() { zero(dp1[ob]); });
  chain<2, 2, 0, 0>(lds + L.w2dT, d2, dp1, lane);  // d hidden dense -> dp1[0..1]
  chain<2, 2, 2, 2>(lds + L.w2gT, d2, dp1, lane);  // d hidden gate  -> dp1[2..3]
  __builtin_amdgcn_sched_barrier(0);
  static_for<4>([&]<int ob>() {
    static_for<2>([&]<int g>() {
      static_for<8>([&]<int qq>() { dp1[ob][8 * g + qq] *= fdsilu(act_m[(ob * 16 + 8 * g + qq) * 64]); });
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  if (edge < a.E) {
    float* row = a.dp1 + edge * (4 * kDP) + mlp * (2 * kDP) + 4 * h;
    static_for<4>([&]<int ob>() {
      static_for<4>([&]<int g>() {
        f32x4 v;
        static_for<4>([&]<int qq>() { v[qq] = dp1[ob][4 * g + qq]; });
        *(f32x4*)(row + ob * 32 + 8 * g) = v;
      });
    });
  }
  __builtin_amdgcn_sched_barrier(0);
  zero(contrib[0]);
  zero(contrib[1]);
  chain<2, 4>(lds + L.w1cT, dp1, contrib, lane);
  __builtin_amdgcn_sched_barrier(0);
}

template <int TBS>
__global__ void __launch_bounds__(512, 2) k_edge_block_reverse_mfma(RevArgs a, MfmaRevLayout L) {
  __shared__ __attribute__((aligned(16))) float lds[kRevLdsFloats];
  load_image(lds, a.img, kRevLdsFloats);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  TileWalk walk(a.tiles, wave);
  for (int it = 0;; ++it) {
    const int64_t tile = walk.tile(it);
    if (tile < 0) break;
    const int64_t edge = tile * 32 + (lane & 31);
    const int64_t ec = edge < a.E ? edge : a.E - 1;
    const int64_t ci = a.src[ec];
    float* de_tile = a.de_soa + tile * 2048 + lane;
    const float* act_tile = a.act + tile * kActPerTile;
    const f32x4 hv = *(const f32x4*)(a.h + ec * kRP);
    f32x4 dhv = {0.f, 0.f, 0.f, 0.f};
    f32x16 de[2], contrib[2];
    {
      f32x16 dmsg[2];
      const float* xrow = a.dx_new + ci * kDP + 4 * h;
      static_for<2>([&]<int ob>() {
        static_for<4>([&]<int g>() {
          const f32x4 v = *(const f32x4*)(xrow + ob * 32 + 8 * g);
          static_for<4>([&]<int qq>() { dmsg[ob][4 * g + qq] = v[qq]; });
        });
      });
      mlp_reverse_mfma(lds, L.mlp[1], 1, a, edge, hv, act_tile, dmsg, contrib, dhv, lane);
    }
    // dL/d e2 = incoming + node-MLP contribution; parked in its tile image while the edge MLP is reversed
    static_for<2>([&]<int kb>() {
      static_for<16>([&]<int r>() {
        de[kb][r] = de_tile[(kb * 16 + r) * 64] + contrib[kb][r];
        de_tile[(kb * 16 + r) * 64] = de[kb][r];
      });
    });
    mlp_reverse_mfma(lds, L.mlp[0], 0, a, edge, hv, act_tile, de, contrib, dhv, lane);
    asm volatile("" ::: "memory");
    static_for<2>([&]<int kb>() {
      static_for<16>([&]<int r>() {
        de[kb][r] = de_tile[(kb * 16 + r) * 64] + contrib[kb][r];
        de_tile[(kb * 16 + r) * 64] = de[kb][r];
      });
    });
    // three-body gated update, reverse: recompute pd, pg from m
    float mb[TBS];
    static_for<TBS>([&]<int s>() { mb[s] = a.m[ec * kCP + 2 * s + h]; });
    f32x16 d4[4];
    {
      f32x16 pd[2], pg[2];
      tb_preact<TBS>(lds + L.tb, mb, pd, pg, lane);
      static_for<2>([&]<int kb>() {
        static_for<16>([&]<int r>() {
          const float p = pd[kb][r], sgd = fsigmoid(p), sg = fsigmoid(pg[kb][r]);
          d4[kb][r] = de[kb][r] * sg * (sgd * (1.f + p * (1.f - sgd)));
          d4[2 + kb][r] = de[kb][r] * (p * sgd) * sg * (1.f - sg);
        });
      });
    }
    f32x16 dmv[1];
    zero(dmv[0]);
    chain<1, 4>(lds + L.tbT, d4, dmv, lane);
    // lane-half dhv halves hold disjoint feature sets of the same edge: combine, then lanes < 32 own the edge
    static_for<4>([&]<int rr>() { dhv[rr] += __shfl_xor(dhv[rr], 32); });
    if (edge < a.E) {
      // rows c = feat_of(r, h): registers 0-3 -> c = 4h..4h+3, registers 4-7 -> c = 8+4h..
      f32x4 lo, hi;
      static_for<4>([&]<int qq>() { lo[qq] = dmv[0][qq]; hi[qq] = dmv[0][4 + qq]; });
      *(f32x4*)(a.dm + edge * kCP + 4 * h) = lo;
      *(f32x4*)(a.dm + edge * kCP + 8 + 4 * h) = hi;
      if (h == 0) {
        f32x4 old = *(f32x4*)(a.dh + edge * kRP);
        *(f32x4*)(a.dh + edge * kRP) = old + dhv;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- helpers
__device__ __forceinline__ int feat_of_dev(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// rows [E][64] -> tile-SoA [tiles][32 slots][64]; padding edges get zeros
__global__ void __launch_bounds__(256) k_rows_to_soa(int64_t E, int64_t tiles, const float* __restrict__ rows, float* __restrict__ soa) {
  int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= tiles * 2048) return;
  int64_t tile = idx >> 11;
  int slot = (int)((idx >> 6) & 31), p = (int)(idx & 63);
  int64_t edge = tile * 32 + (p & 31);
  int o = (slot >> 4) * 32 + feat_of_dev(slot & 15, p >> 5);
  soa[idx] = edge < E ? rows[edge * kDP + o] : 0.f;
}
__global__ void __launch_bounds__(256) k_soa_to_rows(int64_t E, int width, int row_stride, const float* __restrict__ soa,
                                                     float* __restrict__ rows) {
  int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= E * kDP) return;
  int64_t edge = idx >> 6;
  int o = (int)(idx & 63);
  if (o >= width) return;
  int kb = o >> 5, w = o & 31, hh = (w >> 2) & 1, r = (w & 3) + 4 * (w >> 3);
  rows[edge * row_stride + o] = soa[(edge >> 5) * 2048 + (kb * 16 + r) * 64 + hh * 32 + (edge & 31)];
}

// x_new[i,:] = x[i,:] + sum_{e in row(i)} msg[e,:]   (nn/conv.py:82-88): one wave per atom, no atomics
__global__ void __launch_bounds__(256) k_node_sum(int64_t N, const int32_t* __restrict__ row_ptr, const float* __restrict__ x,
                                                  const float* __restrict__ msg, float* __restrict__ x_new) {
  int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  int o = threadIdx.x & 63;
  if (i >= N) return;
  float acc = x[i * kDP + o];
  for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) acc += msg[(int64_t)e * kDP + o];
  x_new[i * kDP + o] = acc;
}

static inline int grid_for_tiles(int64_t tiles) {
  int64_t wgs = (tiles + kWaves - 1) / kWaves;
  wgs = (wgs + 7) / 8 * 8;
  if (wgs < 8) wgs = 8;
  if (wgs > 256) wgs = 256;
  return (int)wgs;
}

static inline int tb_steps_for(int C) { return C <= 6 ? 3 : (C <= 10 ? 5 : 8); }

void launch_rows_to_soa(const float* rows, float* soa, int64_t E, hipStream_t s) {
  int64_t tiles = (E + 31) / 32;
  if (tiles > 0) hipLaunchKernelGGL(k_rows_to_soa, dim3((unsigned)((tiles * 2048 + 255) / 256)), dim3(256), 0, s, E, tiles, rows, soa);
}
void launch_soa_to_rows(const float* soa, float* rows, int row_stride, int width, int64_t E, hipStream_t s) {
  if (E > 0) hipLaunchKernelGGL(k_soa_to_rows, dim3((unsigned)((E * kDP + 255) / 256)), dim3(256), 0, s, E, width, row_stride, soa, rows);
}

void launch_edge_block_mfma(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* x_old,
                            float* x_new, hipStream_t s) {
  const int64_t tiles = (t.E + 31) / 32;
  const MfmaFwdLayout L = mfma_fwd_layout();
  if (tiles > 0) {
    FwdArgs a{t.E, tiles, plan->d_mfma_fwd + (size_t)b * L.total, t.src, t.dst, w.h, w.m[b], w.TA, w.TB, w.e_soa, w.act[b], w.msg};
    dim3 grid(grid_for_tiles(tiles)), block(64 * kWaves);
    switch (tb_steps_for(c.C)) {
      case 3: hipLaunchKernelGGL((k_edge_block_mfma<3>), grid, block, 0, s, a, L); break;
      case 5: hipLaunchKernelGGL((k_edge_block_mfma<5>), grid, block, 0, s, a, L); break;
      default: hipLaunchKernelGGL((k_edge_block_mfma<8>), grid, block, 0, s, a, L); break;
    }
  }
  if (t.N > 0) hipLaunchKernelGGL(k_node_sum, dim3((unsigned)((t.N + 3) / 4)), dim3(256), 0, s, t.N, t.row_ptr, x_old, w.msg, x_new);
}

void launch_edge_block_reverse_mfma(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b,
                                    const float* dx_new, hipStream_t s) {
  const int64_t tiles = (t.E + 31) / 32;
  if (tiles == 0) return;
  const MfmaRevLayout L = mfma_rev_layout();
  RevArgs a{t.E, tiles, plan->d_mfma_rev + (size_t)b * L.total, t.src, w.h, w.m[b], w.act[b], dx_new, w.de_soa, w.dm, w.dh, w.dp1};
  dim3 grid(grid_for_tiles(tiles)), block(64 * kWaves);
  switch (tb_steps_for(c.C)) {
    case 3: hipLaunchKernelGGL((k_edge_block_reverse_mfma<3>), grid, block, 0, s, a, L); break;
    case 5: hipLaunchKernelGGL((k_edge_block_reverse_mfma<5>), grid, block, 0, s, a, L); break;
    default: hipLaunchKernelGGL((k_edge_block_reverse_mfma<8>), grid, block, 0, s, a, L); break;
  }
}

}  // namespace m3g
