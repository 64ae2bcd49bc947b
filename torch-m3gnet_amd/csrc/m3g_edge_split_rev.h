// Reverse edge kernel of the exact-fp32 mode with ONE 16-edge tile worked on by FOUR waves (one per SIMD), each owning a quarter of
// every layer's output rows -- the body of k_edge_rev_split (m3g_edge_small.hip: small systems), also run by the persistent kernel
// k_edge_rev_f32 (m3g_edge_rev_f32.hip) on the tiles its waves cannot share out evenly (its "tail").
// Reference: nn/conv.py:63-97, nn/interaction.py:220-221, nn/featurizer.py:128-132; algebra: oracle/staged.py.
#pragma once
#include "m3g_edge_common.h"

namespace m3g {

constexpr int kSplitWaves = 4;
// LDS floats one group of four waves needs: dL/dp2 and dL/dp1 exchange buffers and the per-wave partial dL/dm / dL/dh rows
constexpr int kRevSplitGroupFloats = 3 * 8 * 256;
constexpr int kRevSplitTabFloats = 3 * 256;   // plain W_l of both MLPs, plain W_adj (block 0): shared by all groups

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// segmented inclusive scan of one accumulator block along the DPP row (seg_scan of m3g_edge_common.h for a single block:
// x += dpp(x) * m is the same fused multiply-add the 16-value assembly block issues)
__device__ __forceinline__ void seg_scan1(f32x4& v, const SegMasks& k) {
  static_for<4>([&]<int r>() {
    float x = v[r];
    x = fmaf(row_shr_f<1>(x), k.m1, x);
    x = fmaf(row_shr_f<2>(x), k.m2, x);
    x = fmaf(row_shr_f<4>(x), k.m4, x);
    x = fmaf(row_shr_f<8>(x), k.m8, x);
    v[r] = x;
  });
}
// run-end lanes store their run's sum for row block `blk` of the 4*kDP-float row (seg_store with a run-time block)
__device__ __forceinline__ void seg_store1(const f32x4& v, const SegMasks& k, float* seg_head, float* seg_first, int64_t tile, int64_t ci,
                                           int qd, int blk) {
  if (k.run_end) {
    float* row = (k.first_run ? seg_head + tile * (4 * kDP) : seg_first + ci * (4 * kDP)) + 4 * qd;
    *(f32x4*)(row + blk * 16) = v;
  }
}

// ------------------------------------------------------------------------------------------------------------- reverse
struct RevMlpA {
  float wld;          // direct image row block w of W_l (A operand of W_l h)
  float w2t[2][16];   // W2d^T / W2g^T: this wave's 16 input-feature rows, k over the 64 outputs of the branch
  float w1ct[32];     // W1c^T: this wave's 16 edge-feature rows, k over the 128 layer-1 outputs
  const float* wl;    // LDS copy of the plain W_l [64][4] (dL/dh accumulation: rows w*16 + 4 qd + {0..3}); in registers these 16
                      // values per MLP (+ 16 of W_adj in block 0) put the kernel at 292 registers = one workgroup per CU
};
__device__ __forceinline__ void load_rev_mlp(const float* __restrict__ img, const MfmaMlpRevF32& L, int w, int lane, RevMlpA& A, const float* wl_lds) {
  const int m = lane & 15, q = lane >> 4;
  const int base = q * 256 + (m ^ (((q & 1) << 4) | ((q >> 1) << 3)));   // chain_dual32_t's lane offset (m3g_dual_f32.h)
  A.wld = img[L.wld + w * 64 + lane];
  static_for<2>([&]<int hf>() {
    static_for<4>([&]<int blk>() {
      static_for<4>([&]<int r>() { A.w2t[hf][blk * 4 + r] = img[(hf == 0 ? L.w2d : L.w2g) + blk * 1024 + r * 64 + (base ^ ((w * 16) ^ r))]; });
    });
  });
  static_for<32>([&]<int k>() { A.w1ct[k] = img[L.w1cT + (w * 32 + k) * 64 + lane]; });   // f32 chain image [4 ob][32 k-steps]
  A.wl = wl_lds + (w * 16 + 4 * q) * 4;
}

// reverse of one conv GatedMLP from its saved activations, split over the four waves: d_upd = this wave's block of
// dL/d(output); returns this wave's block of contrib = W1c^T dL/dp1; accumulates the wave's share of dL/dh into dhv.
template <bool NEED_DP1, int MLP>
__device__ __forceinline__ f32x4 mlp_reverse_split(const RevMlpA& A, const RevArgs& a, int64_t edge, int64_t drow, int64_t tile, int64_t ci, const SegMasks& sk,
                                                   float hb_sel, const f32x4& d_upd, f32x4& dhv, float* hs1, float* hs2, int w, int lane) {
  const int qd = lane >> 4;
  const float* p2_src = a.p2 + tile * (2 * kP1TileFloats) + MLP * kP1TileFloats + lane * 4;
  const float* p1_src = a.p1 + tile * (2 * kP1TileFloats) + MLP * kP1TileFloats + lane * 4;
  f32x4 d2d = load_tile4(p2_src + w * 256), d2g = load_tile4(p2_src + (4 + w) * 256);   // saved layer-2 pre-activations
  // gating derivatives; W_l h on the matrix pipe, dL/dh on the vector ALU (as mlp_reverse_f32, for row block w)
  const f32x4 sl = mfma16(A.wld, hb_sel, zero4());
  static_for<2>([&]<int k>() {
    const f32x2 p2d = {d2d[2 * k], d2d[2 * k + 1]}, p2g = {d2g[2 * k], d2g[2 * k + 1]};
    const f32x2 du = {d_upd[2 * k], d_upd[2 * k + 1]}, s_lin = {sl[2 * k], sl[2 * k + 1]};
    f32x2 sd, dsd;
    silu_pair(p2d, sd, dsd);
    const f32x2 sg = sigmoid_pair(p2g);
    const f32x2 a_g = du * sg;            // dL/d(out) sg(p2g)
    const f32x2 d_s = a_g * sd;           // dL/d(s_lin)
    const f32x2 d_o = a_g * s_lin;
    const f32x2 dd = d_o * dsd;           // dL/d(p2d)
    const f32x2 dgt = (d_s * s_lin) * (1.f - sg);   // dL/d(p2g)
    const f32x4 w0 = *(const f32x4*)(A.wl + (2 * k) * 4), w1 = *(const f32x4*)(A.wl + (2 * k + 1) * 4);
    f32x2 h01 = {dhv[0], dhv[1]}, h23 = {dhv[2], dhv[3]};
    h01 += f32x2{w0[0], w0[1]} * d_s[0]; h23 += f32x2{w0[2], w0[3]} * d_s[0];
    h01 += f32x2{w1[0], w1[1]} * d_s[1]; h23 += f32x2{w1[2], w1[3]} * d_s[1];
    dhv[0] = h01[0]; dhv[1] = h01[1]; dhv[2] = h23[0]; dhv[3] = h23[1];
    d2d[2 * k] = dd[0]; d2d[2 * k + 1] = dd[1];
    d2g[2 * k] = dgt[0]; d2g[2 * k + 1] = dgt[1];
  });
  // dL/dp2 of all waves -> every wave (B operand of the W2^T products)
  *(f32x4*)(hs1 + w * 256 + lane * 4) = d2d;
  *(f32x4*)(hs1 + (4 + w) * 256 + lane * 4) = d2g;
  __syncthreads();
  f32x4 dp1h[2];
  static_for<2>([&]<int hf>() {
    f32x4 dp1 = zero4();
    const f32x4 ds1 = load_tile4(p1_src + (4 * hf + w) * 256);   // saved SiLU'(p1), this wave's block of this half (requested ahead of its chain)
    M3G_F32_CHAIN_PRIO(1);
    static_for<4>([&]<int blk>() {   // (B operands block by block from LDS: all eight blocks at once are 32 live registers)
      const f32x4 d2b = *(const f32x4*)(hs1 + (4 * hf + blk) * 256 + lane * 4);
      static_for<4>([&]<int r>() { dp1 = mfma16(A.w2t[hf][blk * 4 + r], d2b[r], dp1); });
    });
    M3G_F32_CHAIN_PRIO(0);
    dp1 *= ds1;
    if (NEED_DP1 && edge < a.E) *(f32x4*)(a.dp1 + drow * (4 * kDP) + MLP * (2 * kDP) + hf * kDP + 4 * qd + w * 16) = dp1;
    dp1h[hf] = dp1;
    if (NEED_DP1) {   // per-centre sums of the dp1 rows (x_i half of the node reverse)
      f32x4 t = edge < a.E ? dp1 : zero4();
      seg_scan1(t, sk);
      seg_store1(t, sk, a.seg_head, a.seg_first, tile, ci, qd, MLP * 8 + 4 * hf + w);
    }
  });
  // dL/dp1 of all waves -> every wave (B operand of the W1c^T product)
  *(f32x4*)(hs2 + w * 256 + lane * 4) = dp1h[0];
  *(f32x4*)(hs2 + (4 + w) * 256 + lane * 4) = dp1h[1];
  __syncthreads();
  f32x4 contrib = zero4();
  M3G_F32_CHAIN_PRIO(1);
  static_for<8>([&]<int blk>() {
    const f32x4 db = *(const f32x4*)(hs2 + blk * 256 + lane * 4);
    static_for<4>([&]<int r>() { contrib = mfma16(A.w1ct[blk * 4 + r], db[r], contrib); });
  });
  M3G_F32_CHAIN_PRIO(0);
  return contrib;
}

// The tiles t0, t0 + step, ... < t1 through the four waves of group `g` (wave `w` of it).  EVERY wave of the workgroup that is still
// running calls this together -- it synchronises with workgroup barriers; a group that has run out of tiles simply returns (a wave
// that has ended no longer counts at a barrier).  `scratch`: LDS, groups x kRevSplitGroupFloats + kRevSplitTabFloats floats, free
// for this function once every caller has arrived (the first barrier below) -- it may be the very LDS the image `img` sits in: every
// operand is in registers before that barrier.  The A operands a wave needs -- its quarter of the weight images -- come ONCE from the
// packed image `img` (the L2-resident global copy, or a workgroup's LDS copy) straight into registers (145 per lane).
template <int TBS, bool NEED_DP1>
__device__ __forceinline__ void rev_split_run(const RevArgs& a, const MfmaRevF32Layout& L, const float* img, float* scratch, int groups, int g, int w,
                                              int lane, int64_t t0, int64_t t1, int64_t step) {
  const int qd = lane >> 4;
  float* hs1 = scratch + (size_t)g * kRevSplitGroupFloats;   // dL/dp2 of one MLP, two blocks per wave
  float* hs2 = hs1 + 8 * 256;                                // dL/dp1 of one MLP
  float* part = hs2 + 8 * 256;                               // per-wave partial dL/dm rows, then partial dL/dh rows
  float* small_tabs = scratch + (size_t)groups * kRevSplitGroupFloats;   // plain W_l of both MLPs, plain W_adj (block 0)
  constexpr bool FIRST = !NEED_DP1;   // block 0: its input is the edge embedding e0 = SiLU(W_adj h), reversed here
  float a_tb[2][TBS], a_tbt[2][4];
  static_for<2>([&]<int hf>() {
    static_for<TBS>([&]<int s>() { a_tb[hf][s] = img[L.tb + ((hf * 4 + w) * kTbSteps + s) * 64 + lane]; });
    static_for<4>([&]<int r>() { a_tbt[hf][r] = img[L.tbT + ((hf * 4 + w) * 4 + r) * 64 + lane]; });   // f32 chain image [1][32 k-steps]
  });
  RevMlpA A0, A1;
  load_rev_mlp(img, L.mlp[0], w, lane, A0, small_tabs);
  load_rev_mlp(img, L.mlp[1], w, lane, A1, small_tabs + 256);
  float a_adj = 0.f;
  const float* adjp = small_tabs + 512 + (w * 16 + 4 * qd) * 4;
  if (FIRST) a_adj = img[L.adj + w * 64 + lane];
  float tab0 = 0.f, tab1 = 0.f, tab2 = 0.f;
  const bool tab_writer = threadIdx.x < 256;
  if (tab_writer) {
    tab0 = img[L.mlp[0].wl + threadIdx.x];
    tab1 = img[L.mlp[1].wl + threadIdx.x];
    tab2 = FIRST ? img[L.adjp + threadIdx.x] : 0.f;
  }
  __syncthreads();   // every wave of the workgroup is here (or has ended): `scratch` is free
  if (tab_writer) {
    small_tabs[threadIdx.x] = tab0;
    small_tabs[256 + threadIdx.x] = tab1;
    small_tabs[512 + threadIdx.x] = tab2;
  }
  __syncthreads();
  for (int64_t tile = t0; tile < t1; tile += step) {
    const int64_t edge = tile * kTileEdges + (lane & 15);
    const int64_t ec = edge < a.E ? edge : a.E - 1;
    const int64_t ci = a.src[ec];
    const int64_t drow = (NEED_DP1 && a.in_pos) ? (int64_t)a.in_pos[ec] : edge;   // row of this edge in the dp1 array (see RevArgs::in_pos)
    const SegMasks sk = seg_masks((int)ci, lane);
    const f32x4 hv = *(const f32x4*)(a.h + ec * kRP);
    const float hb_sel = qd == 0 ? hv[0] : qd == 1 ? hv[1] : qd == 2 ? hv[2] : hv[3];
    const int arow = a.act_id[ec];   // < 0: the edge takes part in no triplet, its aggregate is zero
    float mb[TBS];
    static_for<TBS>([&]<int s>() { mb[s] = arow >= 0 ? a.m[(int64_t)arow * kCP + 4 * s + qd] : 0.f; });
    f32x4 dhv = zero4();
    // node-message MLP (nn/conv.py:77-89): d msg[e] = dx_new[centre(e)]
    const f32x4 dmsg = *(const f32x4*)(a.dx_new + ci * kDP + 4 * qd + w * 16);
    f32x4 de;
    if (!a.de_is_zero) de = load_tile4(a.de_soa + tile * kTileFloats + w * 256 + lane * 4);
    f32x4 contrib = mlp_reverse_split<NEED_DP1, 1>(A1, a, edge, drow, tile, ci, sk, hb_sel, dmsg, dhv, hs1, hs2, w, lane);
    // dL/d e2 = what flows in from later blocks + the node MLP's contribution
    if (a.de_is_zero) de = contrib;
    else de = de + contrib;
    // edge-update MLP (nn/conv.py:68-75)
    contrib = mlp_reverse_split<NEED_DP1, 0>(A0, a, edge, drow, tile, ci, sk, hb_sel, de, dhv, hs1, hs2, w, lane);
    de += contrib;   // dL/d e1
    if (!FIRST) *(f32x4*)(a.de_soa + tile * kTileFloats + w * 256 + lane * 4) = de;
    if (FIRST) {
      // edge embedding, reverse: dL/dh += W_adj^T (dL/de0 * SiLU'(W_adj h)), rows of block w
      const f32x4 pe = mfma16(a_adj, hb_sel, zero4());
      static_for<4>([&]<int r>() {
        const f32x4 wr = *(const f32x4*)(adjp + r * 4);
        const float t = de[r] * fdsilu(pe[r]);
        dhv[0] += t * wr[0]; dhv[1] += t * wr[1]; dhv[2] += t * wr[2]; dhv[3] += t * wr[3];
      });
    }
    // three-body gated update, reverse (nn/interaction.py:220-221): rows w (dense) and 4 + w (gate)
    f32x4 pd = zero4(), pg = zero4();
    static_for<TBS>([&]<int s>() {
      pd = mfma16(a_tb[0][s], mb[s], pd);
      pg = mfma16(a_tb[1][s], mb[s], pg);
    });
    static_for<2>([&]<int k>() {
      f32x2 sd, dsd;
      silu_pair(f32x2{pd[2 * k], pd[2 * k + 1]}, sd, dsd);
      const f32x2 sg = sigmoid_pair(f32x2{pg[2 * k], pg[2 * k + 1]});
      const f32x2 a_g = f32x2{de[2 * k], de[2 * k + 1]} * sg;
      const f32x2 dd = a_g * dsd, dgt = (a_g * sd) * (1.f - sg);
      pd[2 * k] = dd[0]; pd[2 * k + 1] = dd[1];
      pg[2 * k] = dgt[0]; pg[2 * k + 1] = dgt[1];
    });
    // dL/dm = W_tb^T d8 over the 128 rows: this wave's 32 rows (blocks w, 4 + w), partial results added in wave order below
    f32x4 dmv = zero4();
    static_for<4>([&]<int r>() { dmv = mfma16(a_tbt[0][r], pd[r], dmv); });
    static_for<4>([&]<int r>() { dmv = mfma16(a_tbt[1][r], pg[r], dmv); });
    static_for<4>([&]<int rr>() { dhv[rr] = sum_lane_quarters(dhv[rr]); });   // the wave's rows of every lane quarter
    *(f32x4*)(part + w * 256 + lane * 4) = dmv;
    *(f32x4*)(part + (4 + w) * 256 + lane * 4) = dhv;
    __syncthreads();
    if (w == 0) {
      f32x4 dm = *(const f32x4*)(part + lane * 4);
      f32x4 dh = *(const f32x4*)(part + 4 * 256 + lane * 4);
      static_for<3>([&]<int k>() {
        dm += *(const f32x4*)(part + (k + 1) * 256 + lane * 4);
        dh += *(const f32x4*)(part + (5 + k) * 256 + lane * 4);
      });
      if (edge < a.E && arow >= 0) *(f32x4*)(a.dm + (int64_t)arow * kCP + 4 * qd) = dm;
      if (qd == 0 && edge < a.E) *(f32x4*)(a.dh + edge * kRP) = dh;
    }
    // (`part` of the next tile is written after four more barriers, all of which wave 0 reaches after the reads above)
  }
}


}  // namespace m3g
