// Small-system forms of the two exact-fp32 edge kernels (k_edge_block_mfma<fp32, SAVE = 2> and k_edge_rev_f32): ONE 16-edge tile
// is worked on by the FOUR waves of a workgroup, one per SIMD, each owning a quarter of every layer's output rows.
// Reference: nn/interaction.py:220-221, nn/conv.py:63-97, nn/core.py:61-62, nn/featurizer.py:128-132; algebra: oracle/staged.py.
//
// Why: a tile is 544 (forward) / 577 (reverse) v_mfma_f32_16x16x4_f32 of 32 cycles each -- 7.3 / 7.8 us for the one wave the
// persistent kernels give it, and a 32-atom cell (84 tiles; BASELINE configs[0]) or an MD cell of a few hundred atoms has fewer
// tiles than the chip has SIMDs (1,024): its step is the serial latency of its launches, not throughput.  Those kernels also
// copy a 143-154 KB weight image into LDS per workgroup before the first MFMA.  Here
//   * wave w computes output row blocks w (dense branch) and 4 + w (gate branch) of every layer: 136 / 144 MFMAs per wave and tile;
//     what the next layer needs of the other waves' rows (its B operand is the whole activation vector) crosses through 4-8 KB
//     of LDS and a workgroup barrier, four (forward) / five (reverse) times per tile;
//   * the A operands a wave needs -- its quarter of the weight images -- are loaded ONCE from the L2-resident packed images
//     straight into registers (136-145 per lane) and stay there across the tiles of the workgroup: no LDS image, no LDS operand
//     reads; both kernels stay within 256 registers (two workgroups per CU: each hides the other's tile-start loads -- the
//     reverse kernel by reading its B operands block by block from the exchange buffers and its small plain tables from LDS);
//   * tile loads, stores and table gathers split four ways as well (each wave touches only its own blocks).
// Arithmetic: every output element is the same k-ordered fp32 fmaf chain the persistent kernels form, so edge features,
// saved activations, messages, dp1 rows and dL/de are BIT-IDENTICAL to theirs.  Two sums are associated differently (stated
// in DESIGN.md): dL/dh of an edge (per-wave partial sums over the wave's rows, then the four waves in order -- the
// persistent kernel runs one chain over all rows) and dL/dm (the 128-term product with W_tb^T is split over the waves' own
// rows, partial results added in wave order).  Both are deterministic (fixed order, no atomics).
// Selected by tile count in launch_edge_block_mfma / launch_edge_rev_f32 (plan option "small_tiles").
#include "m3g_edge_common.h"
#include "m3g_edge_split_rev.h"

namespace m3g {

namespace {

// one workgroup (4 waves) per tile while tiles last; beyond that workgroups loop (two per CU: 12-24 KB of LDS each)
inline int grid_for_split(int64_t tiles) {
  int64_t wgs = tiles < 512 ? tiles : 512;
  return (int)(wgs < 1 ? 1 : wgs);
}

// acc += sum_{blk < 4, r < 4} A[blk*4 + r] * x[XOFF + blk][r]   (chain_f32 restricted to one 16-row output block: same k order)
template <int XOFF, int NX>
__device__ __forceinline__ void chain_reg16(const float (&A)[16], const f32x4 (&x)[NX], f32x4& acc) {
  static_assert(XOFF + 4 <= NX, "chain_reg16 operand out of range");
  static_for<4>([&]<int blk>() { static_for<4>([&]<int r>() { acc = mfma16(A[blk * 4 + r], x[XOFF + blk][r], acc); }); });
}

// ------------------------------------------------------------------------------------------------------------- forward
// register-resident A operands of one conv GatedMLP, forward: this wave's layer-1 rows (dense block w, gate block 4 + w),
// layer-2 rows (dense out block w from the dense hidden half, gate out block w from the gate half), its bias quads and W_l rows
struct FwdMlpA {
  float w1[2][16], w2[2][16];
  f32x4 b2[2];
  float wl;
};
__device__ __forceinline__ void load_fwd_mlp(const float* __restrict__ img, const MfmaMlpFwd& L, int w, int lane, FwdMlpA& A) {
  static_for<2>([&]<int hf>() {
    static_for<16>([&]<int k>() {
      A.w1[hf][k] = img[L.w1c + ((hf * 4 + w) * 16 + k) * 64 + lane];             // chain [8 ob][16 k-steps]
      A.w2[hf][k] = img[(hf == 0 ? L.w2d : L.w2g) + (w * 16 + k) * 64 + lane];    // chain [4 ob][16 k-steps]
    });
    A.b2[hf] = *(const f32x4*)(img + L.b2 + hf * 4 * 64 + w * 64 + 4 * (lane >> 4));   // bias_step's 16-byte read
  });
  A.wl = img[L.wl + w * 64 + lane];
}

// One conv GatedMLP, forward, split over the four waves.  x = the MLP's input tile (all four blocks, every wave holds it);
// returns this wave's block of  MLP(x) * (W_l h).  `t1` = gathered table rows (layer-1 accumulator seeds) of blocks w, 4 + w.
// SAVE == 2: SiLU'(p1) and p2 of the wave's blocks are stored for the reverse pass (as mlp_preacts<.., SAVE = 2>);
// SAVE == 0 (energy-only call): nothing is stored and SiLU is evaluated value by value, as the persistent kernel does there.
template <int SAVE>
__device__ __forceinline__ f32x4 mlp_forward_split(const FwdMlpA& A, const f32x4 (&t1)[2], const f32x4 (&x)[4], float hb, float* hs, int w,
                                                   int lane, float* p1_out, float* p2_out) {
  f32x4 p1d = t1[0], p1g = t1[1];
  M3G_F32_CHAIN_PRIO(1);
  static_for<4>([&]<int blk>() {
    static_for<4>([&]<int r>() {
      const float b = x[blk][r];
      p1d = mfma16(A.w1[0][blk * 4 + r], b, p1d);
      p1g = mfma16(A.w1[1][blk * 4 + r], b, p1g);
    });
  });
  M3G_F32_CHAIN_PRIO(0);
  if constexpr (SAVE == 2) {
    f32x4 dsd, dsg;
    static_for<2>([&]<int k>() {
      f32x2 act, der;
      silu_pair(f32x2{p1d[2 * k], p1d[2 * k + 1]}, act, der);
      p1d[2 * k] = act[0]; p1d[2 * k + 1] = act[1];
      dsd[2 * k] = der[0]; dsd[2 * k + 1] = der[1];
      silu_pair(f32x2{p1g[2 * k], p1g[2 * k + 1]}, act, der);
      p1g[2 * k] = act[0]; p1g[2 * k + 1] = act[1];
      dsg[2 * k] = der[0]; dsg[2 * k + 1] = der[1];
    });
    __builtin_nontemporal_store(dsd, (f32x4*)(p1_out + w * 256));
    __builtin_nontemporal_store(dsg, (f32x4*)(p1_out + (4 + w) * 256));
  } else {
    static_for<4>([&]<int r>() { p1d[r] = fsilu(p1d[r]); p1g[r] = fsilu(p1g[r]); });
  }
  // hidden activations of all waves -> every wave (the B operand of layer 2 is the whole hidden vector of its branch)
  *(f32x4*)(hs + w * 256 + lane * 4) = p1d;
  *(f32x4*)(hs + (4 + w) * 256 + lane * 4) = p1g;
  __syncthreads();
  f32x4 hid[8];
  static_for<8>([&]<int ob>() { hid[ob] = *(const f32x4*)(hs + ob * 256 + lane * 4); });
  f32x4 p2d = A.b2[0], p2g = A.b2[1];
  M3G_F32_CHAIN_PRIO(1);
  chain_reg16<0>(A.w2[0], hid, p2d);
  chain_reg16<4>(A.w2[1], hid, p2g);
  M3G_F32_CHAIN_PRIO(0);
  if constexpr (SAVE == 2) {
    __builtin_nontemporal_store(p2d, (f32x4*)(p2_out + w * 256));
    __builtin_nontemporal_store(p2g, (f32x4*)(p2_out + (4 + w) * 256));
  }
  f32x4 out = mfma16(A.wl, hb, zero4());
  static_for<2>([&]<int k>() {
    const f32x2 v = gated_pair(f32x2{p2d[2 * k], p2d[2 * k + 1]}, f32x2{p2g[2 * k], p2g[2 * k + 1]}) * f32x2{out[2 * k], out[2 * k + 1]};
    out[2 * k] = v[0]; out[2 * k + 1] = v[1];
  });
  return out;
}

template <int TBS, bool FIRST, int SAVE>
__global__ void __launch_bounds__(64 * kSplitWaves) k_edge_fwd_split(FwdArgs a, MfmaFwdLayout L) {
  __shared__ __attribute__((aligned(16))) float xs[4 * 256];   // the MLP input tile, a block per wave
  __shared__ __attribute__((aligned(16))) float hs[8 * 256];   // hidden activations of one MLP, two blocks per wave
  const int lane = threadIdx.x & 63, qd = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if ((int64_t)blockIdx.x >= a.tiles) return;   // (whole workgroup)
  // this wave's quarter of the block's weight image, once, into registers
  float a_tb[2][TBS];
  static_for<2>([&]<int hf>() { static_for<TBS>([&]<int s>() { a_tb[hf][s] = a.img[L.tb + ((hf * 4 + w) * kTbSteps + s) * 64 + lane]; }); });
  FwdMlpA A0, A1;
  load_fwd_mlp(a.img, L.mlp[0], w, lane, A0);
  load_fwd_mlp(a.img, L.mlp[1], w, lane, A1);
  float a_adj = 0.f;
  if (FIRST) a_adj = a.img[L.adj + w * 64 + lane];
  for (int64_t tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    const int64_t edge = tile * kTileEdges + (lane & 15);
    const int64_t ec = edge < a.E ? edge : a.E - 1;
    const int64_t ci = a.src[ec], cj = a.dst[ec];
    const int arow = a.act_id[ec];   // < 0: the edge takes part in no triplet, its aggregate is zero
    const float hb = a.h[ec * kRP + qd];
    float mb[TBS];
    static_for<TBS>([&]<int s>() { mb[s] = arow >= 0 ? a.m[(int64_t)arow * kCP + 4 * s + qd] : 0.f; });
    // table rows of both MLPs (this wave's blocks), requested at the tile start
    f32x4 t1[2][2];
    static_for<2>([&]<int mlp>() {
      static_for<2>([&]<int hf>() {
        const int off = mlp * (2 * kDP) + 4 * qd + (hf * 4 + w) * 16;
        t1[mlp][hf] = *(const f32x4*)(a.TA + ci * (4 * kDP) + off) + *(const f32x4*)(a.TB + cj * (4 * kDP) + off);
      });
    });
    f32x4 xw;
    if (FIRST) {   // block 0 forms e0 = SiLU(W_adj h) itself (nn/featurizer.py:128-132)
      xw = mfma16(a_adj, hb, zero4());
      static_for<4>([&]<int r>() { xw[r] = fsilu(xw[r]); });
    } else {
      xw = load_tile4(a.e_in + tile * kTileFloats + w * 256 + lane * 4);
    }
    {  // three-body gated update (nn/interaction.py:220-221), rows w (dense) and 4 + w (gate)
      f32x4 pd = zero4(), pg = zero4();
      static_for<TBS>([&]<int s>() {
        pd = mfma16(a_tb[0][s], mb[s], pd);
        pg = mfma16(a_tb[1][s], mb[s], pg);
      });
      static_for<2>([&]<int k>() {
        const f32x2 v = gated_pair(f32x2{pd[2 * k], pd[2 * k + 1]}, f32x2{pg[2 * k], pg[2 * k + 1]});
        xw[2 * k] += v[0]; xw[2 * k + 1] += v[1];
      });
    }
    f32x4 x[4];
    *(f32x4*)(xs + w * 256 + lane * 4) = xw;
    __syncthreads();
    static_for<4>([&]<int blk>() { x[blk] = *(const f32x4*)(xs + blk * 256 + lane * 4); });
    float* p1_tile = SAVE == 2 ? a.p1_out + tile * (2 * kP1TileFloats) + lane * 4 : nullptr;
    float* p2_tile = SAVE == 2 ? a.p2_out + tile * (2 * kP1TileFloats) + lane * 4 : nullptr;
    // edge update (nn/conv.py:68-75)
    f32x4 out = mlp_forward_split<SAVE>(A0, t1[0], x, hb, hs, w, lane, p1_tile, p2_tile);
    xw += out;
    *(f32x4*)(a.e_out + tile * kTileFloats + w * 256 + lane * 4) = xw;
    *(f32x4*)(xs + w * 256 + lane * 4) = xw;   // (every wave passed the barrier inside mlp_forward_split: the e1 reads are done)
    __syncthreads();
    static_for<4>([&]<int blk>() { x[blk] = *(const f32x4*)(xs + blk * 256 + lane * 4); });
    // node message (nn/conv.py:77-89) and its sum per centre (82-88)
    out = mlp_forward_split<SAVE>(A1, t1[1], x, hb, hs, w, lane, SAVE == 2 ? p1_tile + kP1TileFloats : nullptr,
                                  SAVE == 2 ? p2_tile + kP1TileFloats : nullptr);
    if (edge >= a.E) out = zero4();   // padding lanes of the last tile
    const SegMasks sk = seg_masks((int)ci, lane);
    seg_scan1(out, sk);
    seg_store1(out, sk, a.seg_head, a.seg_first, tile, ci, qd, w);
    // (xs / hs of the next tile: its first write of xs follows this tile's last barrier, which every wave reaches only after
    //  its reads of xs; its first write of hs follows the next tile's first barrier, which follows every wave's reads of hs)
  }
}

// ------------------------------------------------------------------------------------------------------------- reverse
// (the tile body: m3g_edge_split_rev.h, shared with the tail of the persistent reverse kernel)
template <int TBS, bool NEED_DP1>
__global__ void __launch_bounds__(64 * kSplitWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) k_edge_rev_split(RevArgs a, MfmaRevF32Layout L) {
  __shared__ __attribute__((aligned(16))) float scratch[kRevSplitGroupFloats + kRevSplitTabFloats];
  if ((int64_t)blockIdx.x >= a.tiles) return;   // (whole workgroup)
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  rev_split_run<TBS, NEED_DP1>(a, L, a.img, scratch, 1, 0, w, lane, (int64_t)blockIdx.x, a.tiles, (int64_t)gridDim.x);
}

}  // namespace

bool launch_edge_fwd_split(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, bool for_reverse, hipStream_t s) {
  const int64_t tiles = tiles_for(t.E);
  if (tiles == 0) return true;
  const int save = for_reverse ? saved_activations(plan) : 0;
  if (plan->precision != kPrecF32 || save == 1 || plan->d_stamps) return false;   // (save == 1: an A/B option of the persistent kernels)
  const MfmaFwdLayout L = mfma_fwd_layout();
  FwdArgs a{t.E, tiles, plan->d_mfma_fwd[kPrecF32] + (size_t)b * L.total, t.src, t.dst, w.h, w.m[b], w.TAb[b], w.TBb[b], t.act_id,
            w.e_blk[b], w.e_blk[b + 1], w.seg_head, w.seg_first, nullptr, save == 2 ? w.p1_blk[b] : nullptr, save == 2 ? w.p2_blk[b] : nullptr, 1.f};
  const dim3 grid(grid_for_split(tiles)), block(64 * kSplitWaves);
  const bool first = b == 0 && fused_reverse(plan);
#define M3G_FWDS(FIRST_) \
  if (save == 2) { M3G_TBS_SWITCH(c.C, hipLaunchKernelGGL((k_edge_fwd_split<TBS, FIRST_, 2>), grid, block, 0, s, a, L)); } \
  else { M3G_TBS_SWITCH(c.C, hipLaunchKernelGGL((k_edge_fwd_split<TBS, FIRST_, 0>), grid, block, 0, s, a, L)); }
  if (first) { M3G_FWDS(true); } else { M3G_FWDS(false); }
#undef M3G_FWDS
  return true;
}

bool launch_edge_rev_split(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* dx_new, bool de_is_zero,
                           hipStream_t s) {
  const int64_t tiles = tiles_for(t.E);
  if (tiles == 0) return true;
  if (plan->precision != kPrecF32 || !saves_p2(plan) || plan->d_stamps) return false;
  const MfmaRevF32Layout L = mfma_rev_f32_layout();
  RevArgs ar{t.E, tiles, plan->d_mfma_revf32 + (size_t)b * L.total, t.src, t.dst, w.h, w.m[b], dx_new, t.act_id, nullptr, nullptr, nullptr, nullptr,
             w.de_soa, nullptr, de_is_zero ? 1 : 0, w.dm, w.dh_parts + (size_t)b * t.E * kRP, w.dp1, nullptr, w.seg_head, w.seg_first, w.p1_blk[b],
             w.p2_blk[b], 1.f, nullptr, dp1_rows_by_dst(plan) ? t.in_pos : nullptr};
  const dim3 grid(grid_for_split(tiles)), block(64 * kSplitWaves);
  if (b > 0) { M3G_TBS_SWITCH(c.C, hipLaunchKernelGGL((k_edge_rev_split<TBS, true>), grid, block, 0, s, ar, L)); }
  else { M3G_TBS_SWITCH(c.C, hipLaunchKernelGGL((k_edge_rev_split<TBS, false>), grid, block, 0, s, ar, L)); }
  return true;
}

}  // namespace m3g
