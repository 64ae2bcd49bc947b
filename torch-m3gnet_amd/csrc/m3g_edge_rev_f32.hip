// Fused reverse of one block in the fp32 mode (plan option "precision" = 0): node-message MLP reverse, edge-update MLP reverse,
// three-body gated-update reverse and (block 0) the edge-embedding reverse in ONE kernel per block, every dense product on
// v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate -- the reference's arithmetic, nn/core.py:61-62).
// Reference: nn/conv.py:63-97, nn/interaction.py:220-221, nn/featurizer.py:128-132; algebra: oracle/staged.py.
//
// This mode is bound by the matrix pipe (fp32 MFMAs run at 1/16 of the bf16 rate), so the kernel is built to issue as few
// MFMAs as the math allows:
//   * the forward kernel saved the layer-1 pre-activations p1 of both MLPs (RevArgs::p1, 1 KB per edge and block): no table
//     gather, no layer-1 recompute, and the MLP inputs e1 / e2 are not needed at all -- per MLP and tile 128 MFMAs for the
//     layer-2 recompute, 128 for W2^T and 128 for W1c^T (the split kernel pair recomputing layer 1 issues 512);
//   * W2 is needed in both orientations: dual-use fp32 images (m3g_dual_f32.h: one LDS copy, bank-conflict-free by rows and by
//     columns), W1c only transposed -- both MLPs + the three-body and embedding images fit in LDS together (154 KB);
//   * the dense and the gate branch of an MLP are carried through layer 2 and the transposed layers one after the other (half
//     the live registers), the per-centre sums of the dp1 rows are formed in the kernel (DPP segmented scan, see seg_scan) and
//     the rows themselves leave as fp32 (this mode keeps every hand-over in fp32).
// One persistent workgroup per CU, waves pull 16-edge tiles from an LDS counter (TileQueue); XCD-aware chunking.
#include "m3g_edge_common.h"
#include "m3g_edge_split_rev.h"

namespace m3g {

MfmaRevF32Layout mfma_rev_f32_layout() {
  MfmaRevF32Layout L{};
  int off = 0;
  auto take = [&](int n) { int r = off; off += n; return r; };
  L.tb = take(8 * kTbSteps * 64);
  L.tbT = take(1 * 32 * 64);
  for (int m = 0; m < 2; ++m) {
    L.mlp[m].w2d = take(64 * 64);
    L.mlp[m].w2g = take(64 * 64);
    L.mlp[m].w1cT = take(4 * 32 * 64);
    L.mlp[m].b2 = take(2 * 4 * 64);
    L.mlp[m].wl = take(64 * 4);
    L.mlp[m].wld = take(4 * 64);
  }
  L.adj = take(4 * 64);
  L.adjp = take(64 * 4);
  L.total = off;
  return L;
}

constexpr int kRevF32Floats = 8 * kTbSteps * 64 + 32 * 64 + 2 * (2 * 64 * 64 + 4 * 32 * 64 + 2 * 4 * 64 + 64 * 4 + 4 * 64) + 4 * 64 + 64 * 4;
#ifndef M3G_WAVES_REV_F32
#define M3G_WAVES_REV_F32 8
#endif
constexpr int kWavesRevF32 = M3G_WAVES_REV_F32;

#ifndef M3G_NO_SCHED_FENCE
#define M3G_F32_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define M3G_F32_FENCE() ((void)0)
#endif

// reverse of one conv GatedMLP from its saved layer-1 pre-activations: d_upd = dL/d(output) is pulled back; returns
// contrib = W1c^T dL/dp1 and accumulates dL/dh into dhv; NEED_DP1: stores the dL/dp1 rows (x_j half of the node reverse) and
// their per-centre sums (x_i half)
// SAVED_P2: the forward kernel also stored the layer-2 pre-activations (RevArgs::p2) and SiLU'(p1) in place of p1: no recompute
// at all, and SiLU'(p1) is loaded half by half where it is used
template <bool NEED_DP1, int MLP, bool SAVED_P2>
__device__ __forceinline__ void mlp_reverse_f32(const float* lds, const MfmaMlpRevF32& L, const RevArgs& a, int64_t edge, int64_t drow, int64_t tile,
                                                int64_t ci, const SegMasks& sk, const f32x4& hv, const f32x4 (&d_upd)[4],
                                                f32x4 (&contrib)[4], f32x4& dhv, int lane, f32x4 (&d2)[8]) {
  const int qd = lane >> 4;
  f32x4 p1[SAVED_P2 ? 1 : 8];
  const float* p1_src = a.p1 + tile * (2 * kP1TileFloats) + MLP * kP1TileFloats + (threadIdx.x & 63) * 4;
  if constexpr (SAVED_P2) {
    // d2 arrives loaded with the saved layer-2 pre-activations (load_p2: requested by the caller ahead of their use)
  } else {
    static_for<8>([&]<int ob>() { p1[ob] = load_tile4(p1_src + ob * 256); });
    bias_step<4, 0>(lds + L.b2, d2, lane);
    bias_step<4, 4>(lds + L.b2 + 4 * 64, d2, lane);
    static_for<2>([&]<int half>() {   // 0: dense branch (p1[0..3] -> d2[0..3]), 1: gate branch
      f32x4 hid[4];
      static_for<4>([&]<int ob>() {
        static_for<4>([&]<int r>() {
          const float p = p1[4 * half + ob][r], sg = fsigmoid(p);
          hid[ob][r] = p * sg;
          p1[4 * half + ob][r] = sg * (1.f + p * (1.f - sg));   // p1 is only needed again as SiLU'(p1)
        });
      });
      chain_dual32<4, 0, 4 * half>(lds + (half == 0 ? L.w2d : L.w2g), hid, d2, lane);
      M3G_F32_FENCE();
    });
  }
  // gating derivatives; W_l h on the matrix pipe (4 small MFMAs), dL/dh on the vector ALU
  const float hb_sel = qd == 0 ? hv[0] : qd == 1 ? hv[1] : qd == 2 ? hv[2] : hv[3];
  static_for<4>([&]<int ob>() {
    const f32x4 sl = mfma16(lds[L.wld + ob * 64 + lane], hb_sel, f32x4{0.f, 0.f, 0.f, 0.f});
#ifndef M3G_F32_SCALAR_ACT   // (round 4: 1,885 -> 1,457 vector instructions per tile, 211 -> 200 VGPRs, reverse -0.9 % same-box; -DM3G_F32_SCALAR_ACT for A/B)
    static_for<2>([&]<int k>() {   // value pairs on packed fp32 instructions (as the f16x3 fused kernel evaluates them)
      const f32x2 p2d = {d2[ob][2 * k], d2[ob][2 * k + 1]}, p2g = {d2[4 + ob][2 * k], d2[4 + ob][2 * k + 1]};
      const f32x2 du = {d_upd[ob][2 * k], d_upd[ob][2 * k + 1]}, s_lin = {sl[2 * k], sl[2 * k + 1]};
      f32x2 sd, dsd;
      silu_pair(p2d, sd, dsd);
      const f32x2 sg = sigmoid_pair(p2g);
      const f32x2 a_g = du * sg;            // dL/d(out) sg(p2g)
      const f32x2 d_s = a_g * sd;           // dL/d(s_lin)
      const f32x2 d_o = a_g * s_lin;
      const f32x2 dd = d_o * dsd;           // dL/d(p2d)
      const f32x2 dgt = (d_s * s_lin) * (1.f - sg);   // dL/d(p2g)
      const f32x4 w0 = *(const f32x4*)(lds + L.wl + (ob * 16 + 4 * qd + 2 * k) * 4);
      const f32x4 w1 = *(const f32x4*)(lds + L.wl + (ob * 16 + 4 * qd + 2 * k + 1) * 4);
      f32x2 h01 = {dhv[0], dhv[1]}, h23 = {dhv[2], dhv[3]};
      h01 += f32x2{w0[0], w0[1]} * d_s[0]; h23 += f32x2{w0[2], w0[3]} * d_s[0];
      h01 += f32x2{w1[0], w1[1]} * d_s[1]; h23 += f32x2{w1[2], w1[3]} * d_s[1];
      dhv[0] = h01[0]; dhv[1] = h01[1]; dhv[2] = h23[0]; dhv[3] = h23[1];
      d2[ob][2 * k] = dd[0]; d2[ob][2 * k + 1] = dd[1];
      d2[4 + ob][2 * k] = dgt[0]; d2[4 + ob][2 * k + 1] = dgt[1];
    });
#else
    static_for<4>([&]<int r>() {
      const float p2d = d2[ob][r], p2g = d2[4 + ob][r];
      const f32x4 w = *(const f32x4*)(lds + L.wl + (ob * 16 + 4 * qd + r) * 4);
      const float s_lin = sl[r];
      const float sg = fsigmoid(p2g), sgd = fsigmoid(p2d), sd = p2d * sgd;
      const float du = d_upd[ob][r];
      const float d_out = du * s_lin, d_s = du * sd * sg;
      dhv[0] += d_s * w[0]; dhv[1] += d_s * w[1]; dhv[2] += d_s * w[2]; dhv[3] += d_s * w[3];
      d2[ob][r] = d_out * sg * (sgd * (1.f + p2d * (1.f - sgd)));
      d2[4 + ob][r] = d_out * sd * sg * (1.f - sg);
    });
#endif
    // pin the running dL/dh sums: otherwise LLVM sinks the accumulation chain to its only use at the end of the kernel and
    // keeps every w / sd / sg temporary alive
    asm volatile("" : "+v"(dhv[0]), "+v"(dhv[1]), "+v"(dhv[2]), "+v"(dhv[3]));
  });
  zero(contrib);
  M3G_F32_FENCE();
  static_for<2>([&]<int half>() {
    f32x4 dp1[4];
    zero(dp1);
    f32x4 ds1[4];   // SiLU'(p1) of this half
    if constexpr (SAVED_P2) static_for<4>([&]<int ob>() { ds1[ob] = load_tile4(p1_src + (4 * half + ob) * 256); });
    chain_dual32_t<4, 4 * half, 0>(lds + (half == 0 ? L.w2d : L.w2g), d2, dp1, lane);
    if constexpr (SAVED_P2) static_for<4>([&]<int ob>() { dp1[ob] *= ds1[ob]; });
    else static_for<4>([&]<int ob>() { dp1[ob] *= p1[4 * half + ob]; });
    if (NEED_DP1 && edge < a.E) {
      float* row = a.dp1 + drow * (4 * kDP) + MLP * (2 * kDP) + half * kDP + 4 * qd;
      static_for<4>([&]<int ob>() { *(f32x4*)(row + ob * 16) = dp1[ob]; });
    }
    chain_f32<4, 4, 0, 0, 8, 4 * half>(lds + L.w1cT, dp1, contrib, lane);   // k-blocks half*4 .. +4 of the 128 layer-1 outputs
    if (NEED_DP1) {
      if (edge >= a.E) zero(dp1);   // padding lanes of the last tile
      seg_scan(dp1, sk);
      seg_store<MLP * 8 + 4 * half>(dp1, sk, a.seg_head, a.seg_first, tile, ci, qd);
    }
    M3G_F32_FENCE();
  });
}

// saved layer-2 pre-activations of MLP `mlp` of this tile (8 x 1 KB per wave, streaming loads)
__device__ __forceinline__ void load_p2(const RevArgs& a, int64_t tile, int mlp, f32x4 (&d2)[8]) {
  const float* src = a.p2 + tile * (2 * kP1TileFloats) + mlp * kP1TileFloats + (threadIdx.x & 63) * 4;
  static_for<8>([&]<int ob>() { d2[ob] = load_tile4(src + ob * 256); });
}

template <int TBS, bool NEED_DP1, int WAVES, bool SAVED_P2>
__global__ void __launch_bounds__(64 * WAVES) k_edge_rev_f32(RevArgs a, MfmaRevF32Layout L) {
  __shared__ __attribute__((aligned(16))) float lds[kRevF32Floats + 4];  // + tile-queue head
  int* q_head = reinterpret_cast<int*>(lds + kRevF32Floats);
  load_image(lds, a.img, kRevF32Floats, q_head);
  const int lane = threadIdx.x & 63, qd = lane >> 4;
  TileQueue queue(a.tiles, q_head);
  // The tail.  A workgroup's `count` tiles go to 4 SIMDs as whole tiles, so count mod 4 = 1 ends with ONE SIMD running a whole tile
  // (12 us) while the other three idle -- at 2-10 tiles per SIMD (cells of 500 - 4,000 atoms) that is 5-15 % of the launch.  That tile
  // is taken out of the queue and worked on by four waves, one per SIMD, once the whole tiles are done (rev_split_run, the body of the
  // small-system kernel: ~8 us).  Measured (profiles/r06_split_tail_sweep.txt): 864 atoms 0.389 -> 0.377 ms per step, 2,048 atoms
  // 0.629 -> 0.614, 2,916 atoms 0.827 -> 0.820; sizes whose count mod 4 is not 1 unchanged.  split_tail = 2 also takes count mod 4 = 2
  // (two tiles, both groups of four waves): a gain of 20-27 us per step at 30 tiles per workgroup (2,916 atoms, BASELINE config 5) but a
  // loss of as much at 6 and at 42 -- the pair has to wait for the slowest whole-tile wave, and two SIMDs' worth of work gains little
  // from four -- so the default is 1.
  int tail = 0;
  if constexpr (SAVED_P2 && WAVES == 8) tail = (queue.count & 3) >= 1 && (queue.count & 3) <= a.split_tail ? (queue.count & 3) : 0;   // split_tail: 0 never, 1: one tile, 2: one or two
  const int count_a = queue.count - tail;
  const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int ticket = queue.fetch(lane);
  int ci_i = 0, cj_i;
  if (ticket < count_a) load_ends(a.src, nullptr, queue.base + ticket, a.E, lane, ci_i, cj_i);
  constexpr bool FIRST = !NEED_DP1;   // block 0: its input is the edge embedding e0 = SiLU(W_adj h), reversed here
  while (ticket < count_a) {
    const int64_t tile = queue.base + ticket;
    ticket = queue.fetch(lane);  // next tile's ticket and centre atoms one tile ahead
    const bool has_next = ticket < count_a;
    int nci = 0, ncj = 0;
    if (has_next) load_ends(a.src, nullptr, queue.base + ticket, a.E, lane, nci, ncj);
    int lv = lane;               // opaque per tile: keeps the loop-invariant LDS weight reads inside the loop
    asm volatile("" : "+v"(lv));
    const int64_t edge = tile * kTileEdges + (lane & 15);
    const int64_t ec = edge < a.E ? edge : a.E - 1;
    const int64_t ci = ci_i;
    // row of this lane's edge in the dp1 array: by position in the by-neighbour list when the node reverse streams them
    const int64_t drow = (NEED_DP1 && a.in_pos) ? (int64_t)a.in_pos[ec] : edge;
    const SegMasks sk = seg_masks((int)ci, lane);
    float* de_tile = a.de_soa + tile * kTileFloats + lane * 4;
    const f32x4 hv = *(const f32x4*)(a.h + ec * kRP);
    f32x4 dhv = {0.f, 0.f, 0.f, 0.f};
    float mb[TBS];
    const int arow = a.act_id[ec];   // < 0: the edge takes part in no triplet, its aggregate is zero
    static_for<TBS>([&]<int s>() { mb[s] = arow >= 0 ? a.m[(int64_t)arow * kCP + 4 * s + qd] : 0.f; });
    f32x4 de[4], contrib[4], d2e[8];
    {
      // node-message MLP (nn/conv.py:77-89): d msg[e] = dx_new[centre(e)]
      f32x4 dmsg[4];
      const float* xrow = a.dx_new + ci * kDP + 4 * qd;
      static_for<4>([&]<int blk>() { dmsg[blk] = *(const f32x4*)(xrow + blk * 16); });
      f32x4 d2n[8];
      if constexpr (SAVED_P2) load_p2(a, tile, 1, d2n);
#ifdef M3G_P2_PREFETCH   // measured: no effect (1.011 vs 1.003 ms per step), the SIMD's other wave already covers that latency
      // the edge MLP's saved rows are requested now, a whole MLP reverse ahead of their use (32 registers; the kernel has them)
      if constexpr (SAVED_P2) load_p2(a, tile, 0, d2e);
#endif
      mlp_reverse_f32<NEED_DP1, 1, SAVED_P2>(lds, L.mlp[1], a, edge, drow, tile, ci, sk, hv, dmsg, contrib, dhv, lv, d2n);
    }
    // dL/d e2 = what flows in from later blocks + the node MLP's contribution
    if (a.de_is_zero) {
      static_for<4>([&]<int blk>() { de[blk] = contrib[blk]; });
    } else {
      static_for<4>([&]<int blk>() { de[blk] = load_tile4(de_tile + blk * 256) + contrib[blk]; });
    }
    asm volatile("" : "+v"(lv));
    M3G_F32_FENCE();
    // edge-update MLP (nn/conv.py:68-75)
#ifndef M3G_P2_PREFETCH
    if constexpr (SAVED_P2) load_p2(a, tile, 0, d2e);
#endif
    mlp_reverse_f32<NEED_DP1, 0, SAVED_P2>(lds, L.mlp[0], a, edge, drow, tile, ci, sk, hv, de, contrib, dhv, lv, d2e);
    static_for<4>([&]<int blk>() {  // dL/d e1 = dL/d e2 + contribution
      de[blk] += contrib[blk];
      if (!FIRST) *(f32x4*)(de_tile + blk * 256) = de[blk];
    });
    if (FIRST) {
      // edge embedding, reverse (nothing upstream of e0 but the radial basis): dL/dh += W_adj^T (dL/de0 * SiLU'(W_adj h))
      const float hb = qd == 0 ? hv[0] : qd == 1 ? hv[1] : qd == 2 ? hv[2] : hv[3];
      static_for<4>([&]<int blk>() {
        const f32x4 pe = mfma16(lds[L.adj + blk * 64 + lv], hb, f32x4{0.f, 0.f, 0.f, 0.f});
        static_for<4>([&]<int r>() {
          const f32x4 w = *(const f32x4*)(lds + L.adjp + (blk * 16 + 4 * qd + r) * 4);
          const float t = de[blk][r] * fdsilu(pe[r]);
          dhv[0] += t * w[0]; dhv[1] += t * w[1]; dhv[2] += t * w[2]; dhv[3] += t * w[3];
        });
        asm volatile("" : "+v"(dhv[0]), "+v"(dhv[1]), "+v"(dhv[2]), "+v"(dhv[3]));
      });
    }
    // three-body gated update, reverse (nn/interaction.py:220-221)
    f32x4 d8[8];
    tb_preact<TBS>(lds + L.tb, mb, d8, lv);
#ifndef M3G_F32_SCALAR_ACT
    static_for<4>([&]<int blk>() {
      static_for<2>([&]<int k>() {
        f32x2 sd, dsd;
        silu_pair(f32x2{d8[blk][2 * k], d8[blk][2 * k + 1]}, sd, dsd);
        const f32x2 sg = sigmoid_pair(f32x2{d8[4 + blk][2 * k], d8[4 + blk][2 * k + 1]});
        const f32x2 a_g = f32x2{de[blk][2 * k], de[blk][2 * k + 1]} * sg;
        const f32x2 dd = a_g * dsd, dgt = (a_g * sd) * (1.f - sg);
        d8[blk][2 * k] = dd[0]; d8[blk][2 * k + 1] = dd[1];
        d8[4 + blk][2 * k] = dgt[0]; d8[4 + blk][2 * k + 1] = dgt[1];
      });
    });
#else
    static_for<4>([&]<int blk>() {
      static_for<4>([&]<int r>() {
        const float p = d8[blk][r], sgd = fsigmoid(p), sg = fsigmoid(d8[4 + blk][r]);
        d8[blk][r] = de[blk][r] * sg * (sgd * (1.f + p * (1.f - sgd)));
        d8[4 + blk][r] = de[blk][r] * (p * sgd) * sg * (1.f - sg);
      });
    });
#endif
    f32x4 dmv[1];
    zero(dmv);
    chain_f32<1, 8>(lds + L.tbT, d8, dmv, lv);
    store_dh(a.dh, edge, a.E, dhv, qd);
    if (edge < a.E && arow >= 0) *(f32x4*)(a.dm + (int64_t)arow * kCP + 4 * qd) = dmv[0];
    ci_i = nci;
  }
  if constexpr (SAVED_P2 && WAVES == 8) {
    if (tail) {   // (uniform over the workgroup.  The weight image in LDS is dead once every wave has left the loop above: rev_split_run's
                  //  first barrier; its exchange buffers then take the image's place)
      rev_split_run<TBS, NEED_DP1>(a, L, lds /* the operands come from the LDS copy of the image */, lds, 2, wave_id >> 2, wave_id & 3, lane,
                                   queue.base + count_a + (wave_id >> 2), queue.base + queue.count, 2);
    }
  }
}

void launch_edge_rev_f32(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* dx_new,
                         bool de_is_zero, hipStream_t s) {
  const int64_t tiles = tiles_for(t.E);
  if (tiles == 0) return;
  if (tiles <= plan->small_tiles && launch_edge_rev_split(plan, c, t, w, b, dx_new, de_is_zero, s)) return;
  const MfmaRevF32Layout L = mfma_rev_f32_layout();
  static_assert(kRevF32Floats * 4 + 16 <= 160 * 1024, "fused fp32 reverse image exceeds the LDS");
  const float* img = plan->d_mfma_revf32 + (size_t)b * L.total;
  RevArgs ar{t.E, tiles, img, t.src, t.dst, w.h, w.m[b], dx_new, t.act_id, nullptr, nullptr, nullptr, nullptr, w.de_soa, nullptr,
             de_is_zero ? 1 : 0, w.dm, w.dh_parts + (size_t)b * t.E * kRP, w.dp1, nullptr, w.seg_head, w.seg_first, w.p1_blk[b],
             saves_p2(plan) ? w.p2_blk[b] : nullptr, 1.f, nullptr, dp1_rows_by_dst(plan) ? t.in_pos : nullptr, plan->split_tail};
  static_assert(2 * kRevSplitGroupFloats + kRevSplitTabFloats <= kRevF32Floats, "the split tail's exchange buffers must fit in the image's LDS");
  constexpr int WV = kWavesRevF32;
  dim3 grid(grid_for_tiles(tiles, WV)), block(64 * WV);
  const bool p2 = saves_p2(plan);
  // (block 0: x^0 has no position dependence, nobody reads its dp1 rows)
#define M3G_REV_F32_LAUNCH(NEED, P2) M3G_TBS_SWITCH(c.C, hipLaunchKernelGGL((k_edge_rev_f32<TBS, NEED, WV, P2>), grid, block, 0, s, ar, L))
  if (b > 0) { if (p2) { M3G_REV_F32_LAUNCH(true, true); } else { M3G_REV_F32_LAUNCH(true, false); } }
  else { if (p2) { M3G_REV_F32_LAUNCH(false, true); } else { M3G_REV_F32_LAUNCH(false, false); } }
#undef M3G_REV_F32_LAUNCH
}

}  // namespace m3g
