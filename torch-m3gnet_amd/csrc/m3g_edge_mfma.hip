// Edge block on the matrix cores: stage S4 (three-body gated update + M3GNetConv edge and node MLPs + per-centre message
// sums) and its reverse B4, fused per 16-edge tile.  Dense chains: bf16x3 split operands on v_mfma_f32_16x16x32_bf16
// with fp32 accumulation (gfx950 has no xf32); small K <= 16 products on exact-fp32 v_mfma_f32_16x16x4_f32.
// Reference: nn/interaction.py:220-221, nn/conv.py:63-97, nn/core.py:61-62, nn/featurizer.py:128-132; algebra:
// oracle/staged.py.  Design notes and measured history: DESIGN.md section 4.
//
// Formulation: every dense layer is computed TRANSPOSED, Y^T[feature, edge] = W[feature, k] . X^T[k, edge]:
//   * the 16 edges of a tile sit on the MFMA column index (lane & 15); lane quarter qd = lane >> 4 holds, in the
//     4 accumulator registers of 16-feature block blk, features blk*16 + 4*qd + {0..3}; all per-edge elementwise
//     work (SiLU / sigmoid gating, residuals, W_l h) is therefore lane-local;
//   * an accumulator block is directly the B operand of the next layer; the matching k permutation is folded into the
//     weight images (m3g_pack_mfma.hip, m3g_dual_image.h), which a workgroup copies into LDS once and every wave
//     re-reads as the A operand;
//   * layer-1 accumulators start from the gathered per-node tables TA[i] + TB[j] (x_i / x_j parts, bias folded);
//     layer-2 biases enter as one extra k-step against a constant-one operand;
//   * a lane's 4 accumulator registers are 4 consecutive features, so every tile load/store is a 16-byte access
//     (1 KiB per wave instruction);
//   * the 16 edge lanes are a DPP row and a centre's edges are consecutive, so sums over a centre's edges (messages in
//     the forward kernel, dp1 rows in the reverse kernel) are segmented scans along the row (seg_scan).
// Tile-SoA images ([tile][blk][64 lanes][4]) carry the edge features between blocks; nothing else is saved for the
// reverse pass.  One persistent workgroup per CU (forward 12 waves in the f16x3 / bf16x3 modes, 16 in the fp32 mode; fused reverse 8); workgroups sharing an XCD
// (blockIdx % 8) walk a contiguous edge range so their TA/TB rows stay in that XCD's L2.
// Kernels: k_edge_block_mfma (forward), k_edge_rev_fused (reverse, one launch per block), and the earlier reverse pair
// k_edge_rev_node_mlp / k_edge_rev_edge_mlp kept behind option rev_kernel = 0 for A/B tests.
#include "m3g_edge_common.h"

namespace m3g {

// the fp32 forward kernel evaluates its activations on value PAIRS too (packed fp32 instructions; round 4: 1,218 -> 1,038 vector
// instructions per tile, no spills at its 128-register budget any more, forward -2 % same-box; -DM3G_F32_SCALAR_ACT for A/B)
#ifndef M3G_F32_SCALAR_ACT
constexpr bool kF32Pairs = true;
#else
constexpr bool kF32Pairs = false;
#endif

// ---------------------------------------------------------------------------------------------- forward
// both layers of one conv GatedMLP from the edge-feature tile x: p1 = layer-1 pre-activations (dense 0..3, gate 4..7),
// p2 = layer-2 pre-activations.  `w1c/w2d/w2g/b2` are offsets of the forward images inside `lds`.
// p1_out != nullptr: the layer-1 pre-activations (table rows + bias + W1c e) are stored for the reverse pass, which then
// starts from them instead of gathering the tables and recomputing the layer ([8 blk][64 lanes][4] per tile and MLP)
// saved activations are written once and read once, a whole reverse pass later: streaming stores keep them out of L2
#ifndef M3G_SAVE_PLAIN_STORE
#define M3G_SAVE_STORE(ptr, val) __builtin_nontemporal_store((val), (f32x4*)(ptr))
#else
#define M3G_SAVE_STORE(ptr, val) (*(f32x4*)(ptr) = (val))
#endif
// SAVE (compile time: a run-time choice doubles the code paths and spills): 0 nothing saved, 1 p1 -> p1_out, 2 SiLU'(p1) ->
// p1_out and p2 -> p2_out
template <bool KEEP_P1, int PREC, int SAVE = 0>
__device__ __forceinline__ void mlp_preacts(const float* lds, int w1c, int w2d, int w2g, int b2, const f32x4 (&x)[4], f32x4 (&p1)[8],
                                            f32x4 (&p2)[8], int lane, float* p1_out = nullptr, float* p2_out = nullptr, float w_inv = 1.f) {
  chain_p<PREC, 8, 2>(lds + w1c, x, p1, lane, w_inv);
  if constexpr (SAVE == 1) static_for<8>([&]<int ob>() { *(f32x4*)(p1_out + ob * 256) = p1[ob]; });
  bias_step<4, 0>(lds + b2, p2, lane);
  bias_step<4, 4>(lds + b2 + 4 * 64, p2, lane);
  if constexpr (!KEEP_P1 && SAVE == 2) {
    // fp32 mode with both layers saved (saves_p2): the reverse kernel needs layer 1 only as SiLU'(p1) and layer 2 as its
    // pre-activations -- SiLU' costs three more vector instructions here, next to the sigmoid SiLU evaluates anyway
    static_for<8>([&]<int ob>() {
      f32x4 ds;
#ifndef M3G_F32_SCALAR_ACT
      static_for<2>([&]<int k>() {   // value pairs on packed fp32 instructions (silu_pair: SiLU and SiLU' from one sigmoid)
        f32x2 act, der;
        silu_pair(f32x2{p1[ob][2 * k], p1[ob][2 * k + 1]}, act, der);
        p1[ob][2 * k] = act[0]; p1[ob][2 * k + 1] = act[1];
        ds[2 * k] = der[0]; ds[2 * k + 1] = der[1];
      });
#else
      static_for<4>([&]<int r>() {
        const float p = p1[ob][r], sg = fsigmoid(p);
        p1[ob][r] = p * sg;
        ds[r] = sg * (1.f + p * (1.f - sg));
      });
#endif
      M3G_SAVE_STORE(p1_out + ob * 256, ds);
    });
    chain_p<PREC, 4, 2, 0, 0>(lds + w2d, p1, p2, lane, w_inv);
    chain_p<PREC, 4, 2, 4, 4>(lds + w2g, p1, p2, lane, w_inv);
    static_for<8>([&]<int ob>() { M3G_SAVE_STORE(p2_out + ob * 256, p2[ob]); });
    return;
  }
  if (KEEP_P1) {
    // reverse pass: p1 is only needed again as SiLU'(p1) -- leave that in p1 (one sigmoid evaluation for both)
    f32x4 hid[8];
    static_for<8>([&]<int ob>() {
      static_for<4>([&]<int r>() {
        const float p = p1[ob][r], sg = fsigmoid(p);
        hid[ob][r] = p * sg;
        p1[ob][r] = sg * (1.f + p * (1.f - sg));
      });
    });
    chain_p<PREC, 4, 2, 0, 0>(lds + w2d, hid, p2, lane, w_inv);
    chain_p<PREC, 4, 2, 4, 4>(lds + w2g, hid, p2, lane, w_inv);
  } else {
    if constexpr (PREC == kPrecF16x3) {   // value pairs on packed fp32 instructions (the 12-wave f16x3 kernel has the registers)
      static_for<8>([&]<int ob>() {
        static_for<2>([&]<int k>() {
          const f32x2 v = silu_pair(f32x2{p1[ob][2 * k], p1[ob][2 * k + 1]});
          p1[ob][2 * k] = v[0]; p1[ob][2 * k + 1] = v[1];
        });
      });
    } else {
      static_for<8>([&]<int ob>() { static_for<4>([&]<int r>() { p1[ob][r] = fsilu(p1[ob][r]); }); });
    }
    chain_p<PREC, 4, 2, 0, 0>(lds + w2d, p1, p2, lane, w_inv);  // hidden dense = p1[0..3]
    chain_p<PREC, 4, 2, 4, 4>(lds + w2g, p1, p2, lane, w_inv);  // hidden gate  = p1[4..7]
  }
}

// one conv GatedMLP, forward.  x = edge-feature input tile; out = MLP(x) * (W_l h)
template <bool ST, int S0, int PREC, int SAVE>
__device__ __forceinline__ void mlp_forward_mfma(const float* lds, const MfmaMlpFwd& L, int mlp, const FwdArgs& a, int64_t ci,
                                                 int64_t cj, float hb, const f32x4 (&x)[4], f32x4 (&out)[4], int lane,
                                                 Stamps<ST>& st, float* p1_out, float* p2_out) {
  f32x4 p1[8], p2[8];
  gather_tables(a.TA, a.TB, mlp, ci, cj, lane >> 4, p1);
  st.template mark<S0>();      // table gather
  mlp_preacts<false, PREC, SAVE>(lds, L.w1c, L.w2d, L.w2g, L.b2, x, p1, p2, lane, p1_out, p2_out, a.w_inv);
  st.template mark<S0 + 1>();  // both layers
  static_for<4>([&]<int ob>() {
    out[ob] = mfma16(lds[L.wl + ob * 64 + lane], hb, f32x4{0.f, 0.f, 0.f, 0.f});
    if constexpr (PREC == kPrecF16x3 || (kF32Pairs && PREC == kPrecF32)) {
      static_for<2>([&]<int k>() {
        const f32x2 v = gated_pair(f32x2{p2[ob][2 * k], p2[ob][2 * k + 1]}, f32x2{p2[4 + ob][2 * k], p2[4 + ob][2 * k + 1]}) *
                        f32x2{out[ob][2 * k], out[ob][2 * k + 1]};
        out[ob][2 * k] = v[0]; out[ob][2 * k + 1] = v[1];
      });
    } else {
      static_for<4>([&]<int r>() { out[ob][r] = fgated(p2[ob][r], p2[4 + ob][r]) * out[ob][r]; });
    }
  });
  st.template mark<S0 + 3>();  // gating
}

// FIRST: block 0 forms its input e0 = SiLU(W_adj h) (nn/featurizer.py:128-132) from the radial basis instead of reading
// an embedded-edge image that a separate kernel would have to write (256 B/edge) first
template <int TBS, bool ST = false, bool FIRST = false, int PREC = kPrecBf16x3, int SAVE = 0>
__global__ void __launch_bounds__(64 * fwd_waves<PREC>()) k_edge_block_mfma(FwdArgs a, MfmaFwdLayout L) {
  __shared__ __attribute__((aligned(16))) float lds[kFwdLdsFloats + 4];  // + tile-queue head
  int* q_head = reinterpret_cast<int*>(lds + kFwdLdsFloats);
  load_image(lds, a.img, kFwdLdsFloats, q_head);
  const int lane = threadIdx.x & 63, qd = lane >> 4;
  TileQueue queue(a.tiles, q_head);
  Stamps<ST> st;
#ifdef M3G_FWD_STATIC_PRIO
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= M3G_FWD_STATIC_PRIO) __builtin_amdgcn_s_setprio(1);
#endif
  int ticket = queue.fetch(lane);
  if (ticket >= queue.count) return;
  int ci_i, cj_i;
  load_ends(a.src, a.dst, queue.base + ticket, a.E, lane, ci_i, cj_i);
  int arow_i = load_arow(a.act_id, queue.base + ticket, a.E, lane);
  for (;;) {
    const int64_t tile = queue.base + ticket;
    // next tile's ticket, end atoms and active-row id one tile ahead: the table gathers and the aggregate row depend on them,
    // and a dependent round trip per tile is what these latency-bound kernels cannot afford
    ticket = queue.fetch(lane);
    const bool has_next = ticket < queue.count;   // wave-uniform
    int nci = 0, ncj = 0, narow = -1;
    if (has_next) {
      load_ends(a.src, a.dst, queue.base + ticket, a.E, lane, nci, ncj);
      narow = load_arow(a.act_id, queue.base + ticket, a.E, lane);
    }
    // the weight-image reads are loop-invariant: without this the compiler hoists hundreds of LDS loads out of
    // the tile loop and spills them; `lv` is the lane id made opaque once per tile
    int lv = lane;
    asm volatile("" : "+v"(lv));
    st.start();
    const int64_t edge = tile * kTileEdges + (lane & 15);
    const int64_t ec = edge < a.E ? edge : a.E - 1;
    const int64_t ci = ci_i, cj = cj_i;
    // Per-lane address parts computed from `lane` are loop-invariant 64-bit VGPR pairs that the compiler hoists and, at
    // 128 registers, spills: a scratch reload plus s_waitcnt vmcnt(0) in front of each load below.  Taken from the
    // opaque `lv` they are recomputed per tile (two VALU instructions).  The edge-feature tile pointers of blocks > 0
    // stay on `lane`: recomputing those costs more spills (of live tile data) than it saves.
    const int qv = lv >> 4;
    const int tl_out = FIRST ? lv : lane;
    const float* e_tile = a.e_in + tile * kTileFloats + lane * 4;
    float* e_otile = a.e_out + tile * kTileFloats + tl_out * 4;
    f32x4 x[4];
    if (!FIRST) static_for<4>([&]<int blk>() { x[blk] = load_tile4(e_tile + blk * 256); });
    const int arow = arow_i;   // < 0: the edge takes part in no triplet, its aggregate is zero
    const TbIn<PREC, TBS> tbin = tb_load<PREC, TBS>(a.m, arow, qv, a.w_inv);
    const float hb = a.h[ec * kRP + qv];
    if (FIRST) {
      static_for<4>([&]<int blk>() {
        x[blk] = mfma16(lds[L.adj + blk * 64 + lv], hb, f32x4{0.f, 0.f, 0.f, 0.f});
        static_for<4>([&]<int r>() { x[blk][r] = fsilu(x[blk][r]); });
      });
    }
    st.template mark<0>();  // tile loads issued
    {  // three-body gated update (nn/interaction.py:220-221)
      f32x4 p[8];
      tb_preact_p<PREC, TBS>(lds + L.tb, tbin, p, lv);
      if constexpr (PREC == kPrecF16x3 || (kF32Pairs && PREC == kPrecF32)) {
        static_for<4>([&]<int blk>() {
          static_for<2>([&]<int k>() {
            const f32x2 v = gated_pair(f32x2{p[blk][2 * k], p[blk][2 * k + 1]}, f32x2{p[4 + blk][2 * k], p[4 + blk][2 * k + 1]});
            x[blk][2 * k] += v[0]; x[blk][2 * k + 1] += v[1];
          });
        });
      } else {
        static_for<4>([&]<int blk>() { static_for<4>([&]<int r>() { x[blk][r] += fgated(p[blk][r], p[4 + blk][r]); }); });
      }
    }
    st.template mark<1>();  // three-body MLP
    f32x4 out[4];
    float* p1_tile = SAVE >= 1 ? a.p1_out + tile * (2 * kP1TileFloats) + lane * 4 : nullptr;
    float* p2_tile = SAVE >= 2 ? a.p2_out + tile * (2 * kP1TileFloats) + lane * 4 : nullptr;
    mlp_forward_mfma<ST, 2, PREC, SAVE>(lds, L.mlp[0], 0, a, ci, cj, hb, x, out, lv, st, p1_tile, p2_tile);  // edge update (nn/conv.py:68-75)
    static_for<4>([&]<int blk>() {
      x[blk] += out[blk];
      *(f32x4*)(e_otile + blk * 256) = x[blk];
    });
    st.template mark<6>();  // e2 residual + store
    mlp_forward_mfma<ST, 7, PREC, SAVE>(lds, L.mlp[1], 1, a, ci, cj, hb, x, out, lv, st, SAVE >= 1 ? p1_tile + kP1TileFloats : nullptr,
                                        SAVE >= 2 ? p2_tile + kP1TileFloats : nullptr);  // node message (nn/conv.py:77-89)
    {  // sum of the messages per centre instead of a [E,64] message array + a node-side pass over it (nn/conv.py:82-88)
      if (edge >= a.E) static_for<4>([&]<int blk>() { out[blk] = f32x4{0.f, 0.f, 0.f, 0.f}; });   // padding lanes of the last tile
      const SegMasks sk = seg_masks((int)ci, lane);
      seg_scan(out, sk);
      seg_store<0>(out, sk, a.seg_head, a.seg_first, tile, ci, qd);
    }
    st.template mark<11>();  // message sums
    if (!has_next) break;
    ci_i = nci;
    cj_i = ncj;
    arow_i = narow;
  }
  if (ST && lane == 0) {
    const int wave = threadIdx.x >> 6;
    unsigned long long* dst = a.stamps + ((size_t)blockIdx.x * 16 + wave) * 12;   // [256 workgroups][16 wave slots][12]
    for (int i = 0; i < 12; ++i) dst[i] = st.sum[i];
  }
}

// ---------------------------------------------------------------------------------------------- reverse
// Two kernels per block (each with its own LDS image): node-MLP reverse, then edge-MLP + three-body reverse.
// reverse of one conv GatedMLP whose edge-feature input tile is x: both layers are recomputed, then
// d_upd = dL/d(output) is pulled back; returns contrib = W1c^T d_p1 and accumulates dL/dh into dhv.
// SAVED: the forward kernel stored the layer-1 pre-activations (RevArgs::p1): no table gather, no layer-1 recompute, and
// the MLP's input tile x is not needed at all (fp32 mode, where the matrix pipe is the bound: a quarter of the reverse
// kernel's MFMAs for 512 B per edge and MLP of extra traffic each way)
template <bool ST, int PREC, bool SAVED>
__device__ __forceinline__ void mlp_reverse_mfma(const float* lds, const MfmaMlpRev& L, int mlp, const RevArgs& a, int64_t edge, int64_t tile,
                                                 int64_t ci, int64_t cj, const f32x4& hv, const f32x4 (&x)[4],
                                                 const f32x4 (&d_upd)[4], f32x4 (&contrib)[4], f32x4& dhv, int lane,
                                                 Stamps<ST>& st) {
  const int qd = lane >> 4;
  f32x4 p1[8], d2[8];  // d2: first p2 dense 0..3 / gate 4..7, then d_p2 in place
  if constexpr (SAVED) {
    const float* src = a.p1 + tile * (2 * kP1TileFloats) + mlp * kP1TileFloats + (threadIdx.x & 63) * 4;
    static_for<8>([&]<int ob>() { p1[ob] = load_tile4(src + ob * 256); });
    st.template mark<2>();
    bias_step<4, 0>(lds + L.b2, d2, lane);
    bias_step<4, 4>(lds + L.b2 + 4 * 64, d2, lane);
    f32x4 hid[8];
    static_for<8>([&]<int ob>() {
      static_for<4>([&]<int r>() {
        const float p = p1[ob][r], sg = fsigmoid(p);
        hid[ob][r] = p * sg;
        p1[ob][r] = sg * (1.f + p * (1.f - sg));
      });
    });
    chain_p<PREC, 4, 2, 0, 0>(lds + L.w2d, hid, d2, lane, a.w_inv);
    chain_p<PREC, 4, 2, 4, 4>(lds + L.w2g, hid, d2, lane, a.w_inv);
  } else {
    gather_tables(a.TA, a.TB, mlp, ci, cj, qd, p1);
    st.template mark<2>();   // table gather (+ wait for the tile loads)
    mlp_preacts<true, PREC>(lds, L.w1c, L.w2d, L.w2g, L.b2, x, p1, d2, lane, nullptr, nullptr, a.w_inv);
  }
  st.template mark<3>();   // recompute both layers
  static_for<4>([&]<int ob>() {
    static_for<4>([&]<int r>() {
      const float p2d = d2[ob][r], p2g = d2[4 + ob][r];
      const f32x4 w = *(const f32x4*)(lds + L.wl + (ob * 16 + 4 * qd + r) * 4);
      const float s_lin = w[0] * hv[0] + w[1] * hv[1] + w[2] * hv[2] + w[3] * hv[3];
      const float sg = fsigmoid(p2g), sgd = fsigmoid(p2d), sd = p2d * sgd;
      const float du = d_upd[ob][r];
      const float d_out = du * s_lin, d_s = du * sd * sg;
      dhv[0] += d_s * w[0]; dhv[1] += d_s * w[1]; dhv[2] += d_s * w[2]; dhv[3] += d_s * w[3];
      d2[ob][r] = d_out * sg * (sgd * (1.f + p2d * (1.f - sgd)));
      d2[4 + ob][r] = d_out * sd * sg * (1.f - sg);
    });
    // pin the running dL/dh sums: otherwise LLVM sinks the accumulation chain to its only use at the end of the
    // kernel and keeps every w / sd / sg temporary alive
    asm volatile("" : "+v"(dhv[0]), "+v"(dhv[1]), "+v"(dhv[2]), "+v"(dhv[3]));
  });
  st.template mark<4>();   // gating derivatives (VALU)
  f32x4 dp1[8];
  zero(dp1);
  chain_p<PREC, 4, 2, 0, 0>(lds + L.w2dT, d2, dp1, lane, a.w_inv);  // d hidden dense -> dp1[0..3]
  chain_p<PREC, 4, 2, 4, 4>(lds + L.w2gT, d2, dp1, lane, a.w_inv);  // d hidden gate  -> dp1[4..7]
  static_for<8>([&]<int ob>() { dp1[ob] *= p1[ob]; });   // p1 holds SiLU'(p1) here (mlp_preacts<true>)
  st.template mark<5>();   // layer-2 transposed chains + SiLU'
  if (edge < a.E) {
    float* row = a.dp1 + edge * (4 * kDP) + mlp * (2 * kDP) + 4 * qd;
    static_for<8>([&]<int ob>() { *(f32x4*)(row + ob * 16) = dp1[ob]; });
  }
  zero(contrib);
  chain_p<PREC, 4, 4>(lds + L.w1cT, dp1, contrib, lane, a.w_inv);
  st.template mark<6>();   // dp1 stores + layer-1 transposed chain
}

// node-message MLP (nn/conv.py:77-89), reverse: d msg[e] = dx_new[centre(e)]
template <int PREC, bool SAVED = false>
__global__ void __launch_bounds__(64 * kWavesRev) k_edge_rev_node_mlp(RevArgs a, MfmaRevLayout L) {
  __shared__ __attribute__((aligned(16))) float lds[kRevMlpFloats + 4];  // + tile-queue head
  int* q_head = reinterpret_cast<int*>(lds + kRevMlpFloats);
  load_image(lds, a.img, kRevMlpFloats, q_head);
  const int lane = threadIdx.x & 63, qd = lane >> 4;
  TileQueue queue(a.tiles, q_head);
  int ticket = queue.fetch(lane);
  if (ticket >= queue.count) return;
  int ci_i, cj_i;
  load_ends(a.src, a.dst, queue.base + ticket, a.E, lane, ci_i, cj_i);
  for (;;) {
    const int64_t tile = queue.base + ticket;
    ticket = queue.fetch(lane);  // next tile's ticket and end atoms one tile ahead (see the forward kernel)
    const bool has_next = ticket < queue.count;
    int nci = 0, ncj = 0;
    if (has_next) load_ends(a.src, a.dst, queue.base + ticket, a.E, lane, nci, ncj);
    int lv = lane;               // opaque per tile: keeps the loop-invariant LDS weight reads inside the loop
    asm volatile("" : "+v"(lv));
    const int64_t edge = tile * kTileEdges + (lane & 15);
    const int64_t ec = edge < a.E ? edge : a.E - 1;
    const int64_t ci = ci_i, cj = cj_i;
    float* dcn_tile = a.dcn + tile * kTileFloats + lane * 4;
    const float* e_tile = a.e_tile + tile * kTileFloats + lane * 4;
    const f32x4 hv = *(const f32x4*)(a.h + ec * kRP);
    f32x4 dhv = {0.f, 0.f, 0.f, 0.f};
    f32x4 contrib[4];
    {
      f32x4 dmsg[4], x[4];
      const float* xrow = a.dx_new + ci * kDP + 4 * qd;
      static_for<4>([&]<int blk>() {
        dmsg[blk] = *(const f32x4*)(xrow + blk * 16);
        if (!SAVED) x[blk] = *(const f32x4*)(e_tile + blk * 256);     // e2: the node MLP's input
      });
      Stamps<false> st0;
      mlp_reverse_mfma<false, PREC, SAVED>(lds, L.mlp, 1, a, edge, tile, ci, cj, hv, x, dmsg, contrib, dhv, lv, st0);
    }
    static_for<4>([&]<int blk>() { *(f32x4*)(dcn_tile + blk * 256) = contrib[blk]; });
    store_dh(a.dh, edge, a.E, dhv, qd);
    if (!has_next) break;
    ci_i = nci;
    cj_i = ncj;
  }
}

// edge-update MLP (nn/conv.py:68-75) + three-body gated update (nn/interaction.py:220-221), reverse
template <int TBS, bool ST = false, int PREC = kPrecBf16x3, bool SAVED = false>
__global__ void __launch_bounds__(64 * kWavesRev) k_edge_rev_edge_mlp(RevArgs a, MfmaRevLayout L) {
  Stamps<ST> st;
  __shared__ __attribute__((aligned(16))) float lds[kRevEdgeFloats + 4];  // + tile-queue head
  int* q_head = reinterpret_cast<int*>(lds + kRevEdgeFloats);
  load_image(lds, a.img, kRevEdgeFloats, q_head);
  const int lane = threadIdx.x & 63, qd = lane >> 4;
  TileQueue queue(a.tiles, q_head);
  int ticket = queue.fetch(lane);
  if (ticket >= queue.count) return;
  int ci_i, cj_i;
  load_ends(a.src, a.dst, queue.base + ticket, a.E, lane, ci_i, cj_i);
  for (;;) {
    const int64_t tile = queue.base + ticket;
    ticket = queue.fetch(lane);  // next tile's ticket and end atoms one tile ahead (see the forward kernel)
    const bool has_next = ticket < queue.count;
    int nci = 0, ncj = 0;
    if (has_next) load_ends(a.src, a.dst, queue.base + ticket, a.E, lane, nci, ncj);
    int lv = lane;               // opaque per tile: keeps the loop-invariant LDS weight reads inside the loop
    asm volatile("" : "+v"(lv));
    st.start();
    const int64_t edge = tile * kTileEdges + (lane & 15);
    const int64_t ec = edge < a.E ? edge : a.E - 1;
    const int64_t ci = ci_i, cj = cj_i;
    float* de_tile = a.de_soa + tile * kTileFloats + lane * 4;
    const float* e_tile = a.e_tile + tile * kTileFloats + lane * 4;
    const f32x4 hv = *(const f32x4*)(a.h + ec * kRP);
    f32x4 dhv = {0.f, 0.f, 0.f, 0.f};
    float mb[TBS];
    const int arow = a.act_id[ec];   // < 0: the edge takes part in no triplet, its aggregate is zero
    static_for<TBS>([&]<int s>() { mb[s] = arow >= 0 ? a.m[(int64_t)arow * kCP + 4 * s + qd] : 0.f; });
    f32x4 de[4], contrib[4];
    {
      // recompute e1 = e_in + three-body gated update: the edge MLP's input (not needed when its layer 1 was saved)
      f32x4 x[4];
      if constexpr (!SAVED) {
        f32x4 p[8];
        static_for<4>([&]<int blk>() { x[blk] = *(const f32x4*)(e_tile + blk * 256); });
        tb_preact<TBS>(lds + L.tb, mb, p, lv);
        static_for<4>([&]<int blk>() { static_for<4>([&]<int r>() { x[blk][r] += fgated(p[blk][r], p[4 + blk][r]); }); });
      }
      st.template mark<1>();   // tile loads + three-body recompute
      // dL/d e2 = what flows in from later blocks + the node MLP's contribution (both loaded here, at the tile start)
      const float* dcn_tile = a.dcn + tile * kTileFloats + lane * 4;
      if (a.de_is_zero) {
        static_for<4>([&]<int blk>() { de[blk] = *(const f32x4*)(dcn_tile + blk * 256); });
      } else {
        static_for<4>([&]<int blk>() { de[blk] = *(const f32x4*)(de_tile + blk * 256) + *(const f32x4*)(dcn_tile + blk * 256); });
      }
      mlp_reverse_mfma<ST, PREC, SAVED>(lds, L.mlp, 0, a, edge, tile, ci, cj, hv, x, de, contrib, dhv, lv, st);
    }
    static_for<4>([&]<int blk>() {  // dL/d e1 = dL/d e2 + contribution
      de[blk] += contrib[blk];
      *(f32x4*)(de_tile + blk * 256) = de[blk];
    });
    // three-body gated update, reverse: pre-activations once more from m (24 small MFMAs; cheaper than 32 live VGPRs)
    f32x4 d8[8];
    tb_preact<TBS>(lds + L.tb, mb, d8, lv);
    static_for<4>([&]<int blk>() {
      static_for<4>([&]<int r>() {
        const float p = d8[blk][r], sgd = fsigmoid(p), sg = fsigmoid(d8[4 + blk][r]);
        d8[blk][r] = de[blk][r] * sg * (sgd * (1.f + p * (1.f - sgd)));
        d8[4 + blk][r] = de[blk][r] * (p * sgd) * sg * (1.f - sg);
      });
    });
    f32x4 dmv[1];
    zero(dmv);
    chain_p<PREC, 1, 4>(lds + L.tbT, d8, dmv, lv, a.w_inv);
    store_dh(a.dh, edge, a.E, dhv, qd);
    if (edge < a.E && arow >= 0) *(f32x4*)(a.dm + (int64_t)arow * kCP + 4 * qd) = dmv[0];  // rows c = 4*qd + reg
    st.template mark<7>();   // de store + three-body reverse + dm/dh stores
    if (!has_next) break;
    ci_i = nci;
    cj_i = ncj;
  }
  if (ST && lane == 0) {
    const int wave = threadIdx.x >> 6;
    unsigned long long* dst = a.stamps + ((size_t)blockIdx.x * 16 + wave) * 12;
    for (int i = 0; i < 12; ++i) dst[i] = st.sum[i];
  }
}

// ---------------------------------------------------------------------------------------------- fused reverse
// One kernel per block instead of the two above: with the dual-use weight images (m3g_dual_image.h: one LDS copy of
// each matrix read by rows for the recompute and through ds_read_b64_tr_b16 for the transposed products) both MLPs'
// weights fit in LDS together (150 KB), so a tile goes  e_in -> e1 (three-body update) -> e2 (edge MLP forward)
// -> node-MLP reverse -> edge-MLP reverse -> three-body reverse  without leaving registers (e2 itself is read back
// from the forward pass's output image rather than evaluated a second time).  Against the split
// kernels this drops the node->edge hand-over buffer (256 B written + read per edge), the second read of the edge
// features, the second table gather, and one dh slice; block 0 also skips the dp1 rows nobody reads.
constexpr int kRevFusedFloats = 8 * kTbSteps * 64 + 8 * 4 * 64 + 2 * (128 * 64 + 2 * 64 * 64 + 2 * 4 * 64 + 64 * 4 + 4 * 64) + 4 * 64 + 64 * 4;

// dual-image chains in the precision mode of the fused reverse kernel (bf16x3: m3g_dual_chain.h chain_dual / chain_dual_t;
// f16x3: chain_dual_h / chain_dual_t_h on fp16 images of the scaled weights, per-edge input scales, results in true units)
template <int PREC, int OB, int KS, int ROWS, int XOFF = 0, int AOFF = 0, int RB0 = 0, int NX, int NA>
__device__ __forceinline__ void cdual(const float* img, const f32x4 (&x)[NX], f32x4 (&acc)[NA], int lane, float w_inv) {
  if constexpr (PREC == kPrecF16x3) chain_dual_h<OB, KS, ROWS, XOFF, AOFF, RB0>(img, x, acc, lane, w_inv);
  else chain_dual<OB, KS, ROWS, XOFF, AOFF, RB0>(img, x, acc, lane);
}
template <int PREC, int OB, int KS, int ROWS, int DOFF = 0, int AOFF = 0, int KB0 = 0, int ND, int NA>
__device__ __forceinline__ void cdual_t(const float* img, const f32x4 (&d)[ND], f32x4 (&acc)[NA], int lane, float w_inv) {
  if constexpr (PREC == kPrecF16x3) chain_dual_t_h<OB, KS, ROWS, DOFF, AOFF, KB0>(img, d, acc, lane, w_inv);
  else chain_dual_t<OB, KS, ROWS, DOFF, AOFF, KB0>(img, d, acc, lane);
}

template <bool KEEP_P1>
__device__ __forceinline__ void mlp_preacts_dual(const float* lds, const MfmaMlpFused& L, const f32x4 (&x)[4], f32x4 (&p1)[8],
                                                 f32x4 (&p2)[8], int lane) {
  chain_dual<8, 2, 128>(lds + L.w1c, x, p1, lane);
  bias_step<4, 0>(lds + L.b2, p2, lane);
  bias_step<4, 4>(lds + L.b2 + 4 * 64, p2, lane);
  if (KEEP_P1) {
    f32x4 hid[8];
    static_for<8>([&]<int ob>() {
      static_for<4>([&]<int r>() {
        const float p = p1[ob][r], sg = fsigmoid(p);
        hid[ob][r] = p * sg;
        p1[ob][r] = sg * (1.f + p * (1.f - sg));
      });
    });
    chain_dual<4, 2, 64, 0, 0>(lds + L.w2d, hid, p2, lane);
    chain_dual<4, 2, 64, 4, 4>(lds + L.w2g, hid, p2, lane);
  } else {
    static_for<8>([&]<int ob>() { static_for<4>([&]<int r>() { p1[ob][r] = fsilu(p1[ob][r]); }); });
    chain_dual<4, 2, 64, 0, 0>(lds + L.w2d, p1, p2, lane);
    chain_dual<4, 2, 64, 4, 4>(lds + L.w2g, p1, p2, lane);
  }
}

// Scheduling fence between the phases of the fused reverse kernel: the machine scheduler otherwise interleaves
// neighbouring phases to hide latency inside ONE wave, which inflates live ranges past the 168 registers that allow a
// third wave per SIMD -- and a third wave hides more latency than the interleaving does.
#ifndef M3G_NO_SCHED_FENCE
#define M3G_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define M3G_SCHED_FENCE() ((void)0)
#endif

// reverse of one conv GatedMLP at input tile x (recomputing both layers), dual-image version of mlp_reverse_mfma.
// The dense and the gate branch are independent between layer 1 and the final product, so each is carried through
// layer 2, and later through the transposed layers, on its own: 16 instead of 32 registers for the hidden values and
// for dL/dp1, which is what lets the kernel approach three waves per SIMD.
template <bool NEED_DP1, int MLP, int PREC, bool ST = false, int S0 = 0>
__device__ __forceinline__ void mlp_reverse_dual(const float* lds, const MfmaMlpFused& L, const RevArgs& a, int64_t edge, int64_t tile,
                                                 int64_t ci, int64_t cj, const SegMasks& sk, const f32x4& hv, const f32x4 (&x)[4],
                                                 const f32x4 (&d_upd)[4], f32x4 (&contrib)[4], f32x4& dhv, int lane, Stamps<ST>& st,
                                                 f32x4& dp1_inv) {
  // dp1_inv (f16x3): receives the inverse scales of this MLP's two row quarters (elements 2 MLP, 2 MLP + 1), stored by the caller
  constexpr int mlp = MLP;
  const int qd = lane >> 4;
  f32x4 p1[8], d2[8];
  gather_tables(a.TA, a.TB, mlp, ci, cj, qd, p1);
  cdual<PREC, 8, 2, 128>(lds + L.w1c, x, p1, lane, a.w_inv);
  M3G_SCHED_FENCE();
  st.template mark<S0>();       // inputs arrived, table gather, layer 1
  bias_step<4, 0>(lds + L.b2, d2, lane);
  bias_step<4, 4>(lds + L.b2 + 4 * 64, d2, lane);
  static_for<2>([&]<int half>() {   // 0: dense branch (p1[0..3] -> d2[0..3]), 1: gate branch
    f32x4 hid[4];
    static_for<4>([&]<int ob>() {
#ifndef M3G_NO_PAIR_ACT
      static_for<2>([&]<int k>() {
        f32x4& pv = p1[4 * half + ob];
        f32x2 act, der;
        silu_pair(f32x2{pv[2 * k], pv[2 * k + 1]}, act, der);
        hid[ob][2 * k] = act[0]; hid[ob][2 * k + 1] = act[1];
        pv[2 * k] = der[0]; pv[2 * k + 1] = der[1];   // p1 is only needed again as SiLU'(p1)
      });
#else
      static_for<4>([&]<int r>() {
        const float p = p1[4 * half + ob][r], sg = fsigmoid(p);
        hid[ob][r] = p * sg;
        p1[4 * half + ob][r] = sg * (1.f + p * (1.f - sg));   // p1 is only needed again as SiLU'(p1)
      });
#endif
    });
    cdual<PREC, 4, 2, 64, 0, 4 * half>(lds + (half == 0 ? L.w2d : L.w2g), hid, d2, lane, a.w_inv);
    M3G_SCHED_FENCE();
  });
  st.template mark<S0 + 1>();   // activations, layer 2
  // W_l h on the matrix pipe (4 small MFMAs, as the forward kernel) instead of a 4-term dot per element on the vector ALU
  // (64 VALU instructions per MLP; fused reverse 0.904 -> 0.889 ms per step)
  const float hb_sel = qd == 0 ? hv[0] : qd == 1 ? hv[1] : qd == 2 ? hv[2] : hv[3];
  static_for<4>([&]<int ob>() {
    const f32x4 sl = mfma16(lds[L.wld + ob * 64 + lane], hb_sel, f32x4{0.f, 0.f, 0.f, 0.f});
#ifndef M3G_NO_PAIR_ACT
    // value pairs on packed fp32 instructions (silu_pair): out = SiLU(p2d) sg(p2g) s_lin
    static_for<2>([&]<int k>() {
      const f32x2 p2d = {d2[ob][2 * k], d2[ob][2 * k + 1]}, p2g = {d2[4 + ob][2 * k], d2[4 + ob][2 * k + 1]};
      const f32x2 du = {d_upd[ob][2 * k], d_upd[ob][2 * k + 1]}, s_lin = {sl[2 * k], sl[2 * k + 1]};
      f32x2 sd, dsd;
      silu_pair(p2d, sd, dsd);
#ifdef M3G_DIAG_CHEAP_ACT
      const f32x2 sg = p2g * 0.25f + 0.5f;
#else
      const f32x2 tg = p2g * -1.4426950408889634f;
      const f32x2 dg = f32x2{__builtin_amdgcn_exp2f(tg[0]), __builtin_amdgcn_exp2f(tg[1])} + 1.f;
      const f32x2 sg = {__builtin_amdgcn_rcpf(dg[0]), __builtin_amdgcn_rcpf(dg[1])};
#endif
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
    asm volatile("" : "+v"(dhv[0]), "+v"(dhv[1]), "+v"(dhv[2]), "+v"(dhv[3]));   // see mlp_reverse_mfma
  });
  zero(contrib);
  M3G_SCHED_FENCE();
  st.template mark<S0 + 2>();   // gating derivatives
  static_for<2>([&]<int half>() {
    f32x4 dp1[4];
    zero(dp1);
    cdual_t<PREC, 4, 2, 64, 4 * half, 0>(lds + (half == 0 ? L.w2d : L.w2g), d2, dp1, lane, a.w_inv);
    static_for<4>([&]<int ob>() { dp1[ob] *= p1[4 * half + ob]; });
    if (NEED_DP1 && edge < a.E) {
      if constexpr (PREC == kPrecBf16x3) {
        // 24-bit rows (pack24): group index = column / 4 = mlp*32 + half*16 + ob*4 + qd
        unsigned* row = reinterpret_cast<unsigned*>(a.dp1) + edge * kDp1PackedDwords + 3 * (mlp * 32 + half * 16 + qd);
        static_for<4>([&]<int ob>() { *(u32x3_a4*)(row + 12 * ob) = pack24(dp1[ob]); });
      }
    }
    if constexpr (PREC == kPrecF16x3) {
      // f16x3 mode: 24-bit fixed-point rows on the scale the W1c^T chain uses for this quarter of the row anyway (pack24_fixed: within
      // 2^-22 of the quarter's largest value; a 24-bit FLOATING row, 2^-17 per value, would be this mode's largest error by two
      // orders of magnitude), 768 B instead of 1 KB per edge to write here and to gather in the node reverse
      // (stored after the chain, which finds the scale behind its first operand request; dp1 stays live for the scan below anyway)
      EdgeScale scd;
      chain_dual_t_h<4, 2, 128, 0, 0, 4 * half>(lds + L.w1c, dp1, contrib, lane, a.w_inv, &scd);   // rows half*64 .. +63 of W1c
      if (NEED_DP1 && edge < a.E) {
        unsigned* row = reinterpret_cast<unsigned*>(a.dp1) + edge * kDp1PackedDwords + 3 * (mlp * 32 + half * 16 + qd);
        const float s9 = scd.s * 512.f;
        static_for<4>([&]<int ob>() { *(u32x3_a4*)(row + 12 * ob) = pack24_fixed(dp1[ob], s9); });
      }
      dp1_inv[mlp * 2 + half] = scd.inv * (1.f / 512.f);
    } else {
      cdual_t<PREC, 4, 2, 128, 0, 0, 4 * half>(lds + L.w1c, dp1, contrib, lane, a.w_inv);   // rows half*64 .. +63 of W1c
    }
    if (NEED_DP1) {
      // sum of the dp1 rows per centre (the x_i half of the node reverse): scanned here, so the node kernel reads a few
      // partial rows per atom instead of every row of the centre.  The rows themselves are still stored above for the
      // x_j half, which is a gather by neighbour.
      if (edge >= a.E) zero(dp1);   // padding lanes of the last tile
      seg_scan(dp1, sk);
      seg_store<MLP * 8 + 4 * half>(dp1, sk, a.seg_head, a.seg_first, tile, ci, qd);
    }
    M3G_SCHED_FENCE();
    st.template mark<S0 + 3 + half>();   // transposed chains, dp1 stores, per-centre scan of one half
  });
}

template <int TBS, bool NEED_DP1, int WAVES, int PREC = kPrecBf16x3, bool ST = false>
__global__ void __launch_bounds__(64 * WAVES) k_edge_rev_fused(RevArgs a, MfmaRevFusedLayout L) {
  Stamps<ST> st;
  __shared__ __attribute__((aligned(16))) float lds[kRevFusedFloats + 4];  // + tile-queue head
  int* q_head = reinterpret_cast<int*>(lds + kRevFusedFloats);
  load_image(lds, a.img, kRevFusedFloats, q_head);
  const int lane = threadIdx.x & 63, qd = lane >> 4;
  TileQueue queue(a.tiles, q_head);
#ifdef M3G_REV_STATIC_PRIO
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= M3G_REV_STATIC_PRIO) __builtin_amdgcn_s_setprio(1);
#endif
  int ticket = queue.fetch(lane);
  if (ticket >= queue.count) return;
  int ci_i, cj_i;
  load_ends(a.src, a.dst, queue.base + ticket, a.E, lane, ci_i, cj_i);
  int arow_i = load_arow(a.act_id, queue.base + ticket, a.E, lane);
  for (;;) {
    const int64_t tile = queue.base + ticket;
    ticket = queue.fetch(lane);  // next tile's ticket, end atoms and active-row id one tile ahead (see the forward kernel)
    const bool has_next = ticket < queue.count;
    int nci = 0, ncj = 0, narow = -1;
    if (has_next) {
      load_ends(a.src, a.dst, queue.base + ticket, a.E, lane, nci, ncj);
      narow = load_arow(a.act_id, queue.base + ticket, a.E, lane);
    }
    int lv = lane;               // opaque per tile: keeps the loop-invariant LDS weight reads inside the loop
    asm volatile("" : "+v"(lv));
    st.start();
    const int64_t edge = tile * kTileEdges + (lane & 15);
    const int64_t ec = edge < a.E ? edge : a.E - 1;
    const int64_t ci = ci_i, cj = cj_i;
    const SegMasks sk = seg_masks((int)ci, lane);
    float* de_tile = a.de_soa + tile * kTileFloats + lane * 4;
    const float* e_tile = a.e_tile + tile * kTileFloats + lane * 4;
    const f32x4 hv = *(const f32x4*)(a.h + ec * kRP);
    f32x4 dhv = {0.f, 0.f, 0.f, 0.f}, dp1_inv = {0.f, 0.f, 0.f, 0.f};
    const int arow = arow_i;   // < 0: the edge takes part in no triplet, its aggregate is zero
    const TbIn<PREC, TBS> tbin = tb_load<PREC, TBS>(a.m, arow, qd, a.w_inv);
    f32x4 x[4], de[4], contrib[4];
    constexpr bool FIRST = !NEED_DP1;   // block 0: its input is the edge embedding e0 = SiLU(W_adj h), formed here
    {
      // Node-message MLP first: its input e2 is what the forward kernel stored as the next block's edge features
      // (reading it back, 256 B/edge, is cheaper for this issue-bound kernel than a second evaluation of the edge
      // MLP), so nothing of the edge-MLP side has to stay in registers across it.
      f32x4 x2[4];
      const float* e2_tile = a.e2_tile + tile * kTileFloats + lane * 4;
      static_for<4>([&]<int blk>() { x2[blk] = load_tile4(e2_tile + blk * 256); });
      // d msg[e] = dx_new[centre(e)]
      f32x4 dmsg[4];
      const float* xrow = a.dx_new + ci * kDP + 4 * qd;
      static_for<4>([&]<int blk>() { dmsg[blk] = *(const f32x4*)(xrow + blk * 16); });
      mlp_reverse_dual<NEED_DP1, 1, PREC, ST, 0>(lds, L.mlp[1], a, edge, tile, ci, cj, sk, hv, x2, dmsg, contrib, dhv, lv, st, dp1_inv);
    }
    // dL/d e2 = what flows in from later blocks + the node MLP's contribution
    if (a.de_is_zero) {
      static_for<4>([&]<int blk>() { de[blk] = contrib[blk]; });
    } else {
      static_for<4>([&]<int blk>() { de[blk] = load_tile4(de_tile + blk * 256) + contrib[blk]; });
    }
    // the opaque copy also pins the order: without it the scheduler hoists the e1 computation above the node phase
    asm volatile("" : "+v"(lv));
    M3G_SCHED_FENCE();
    if (FIRST) {
      const float hb = a.h[ec * kRP + qd];
      static_for<4>([&]<int blk>() {
        x[blk] = mfma16(lds[L.adj + blk * 64 + lv], hb, f32x4{0.f, 0.f, 0.f, 0.f});
        static_for<4>([&]<int r>() { x[blk][r] = fsilu(x[blk][r]); });
      });
    } else {
      static_for<4>([&]<int blk>() { x[blk] = load_tile4(e_tile + blk * 256); });
    }
    {  // e1 = e_in + three-body gated update (the edge MLP's input)
      f32x4 p[8];
      tb_preact_p<PREC, TBS>(lds + L.tb, tbin, p, lv);
      static_for<4>([&]<int blk>() {
        static_for<2>([&]<int k>() {   // value pairs on packed fp32 instructions (gated_pair)
          const f32x2 v = gated_pair(f32x2{p[blk][2 * k], p[blk][2 * k + 1]}, f32x2{p[4 + blk][2 * k], p[4 + blk][2 * k + 1]});
          x[blk][2 * k] += v[0]; x[blk][2 * k + 1] += v[1];
        });
      });
    }
    st.template mark<5>();   // dL/de and e images arrived, e1 recomputed (three-body MLP)
    mlp_reverse_dual<NEED_DP1, 0, PREC, ST, 6>(lds, L.mlp[0], a, edge, tile, ci, cj, sk, hv, x, de, contrib, dhv, lv, st, dp1_inv);
    // the row's four inverse scales in one 16-byte store per edge (f16x3 mode; 256 contiguous bytes per tile)
    if (NEED_DP1 && PREC == kPrecF16x3 && qd == 0 && edge < a.E) *(f32x4*)(a.dp1_scale + edge * 4) = dp1_inv;
    static_for<4>([&]<int blk>() {  // dL/d e1 = dL/d e2 + contribution
      de[blk] += contrib[blk];
      if (!FIRST) *(f32x4*)(de_tile + blk * 256) = de[blk];
    });
    if (FIRST) {
      // edge embedding, reverse (nothing upstream of e0 but the radial basis): dL/dh += W_adj^T (dL/de0 * SiLU'(W_adj h))
      const float hb = a.h[ec * kRP + qd];
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
    // three-body gated update, reverse
    f32x4 d8[8];
    tb_preact_p<PREC, TBS>(lds + L.tb, tbin, d8, lv);
    static_for<4>([&]<int blk>() {
      static_for<2>([&]<int k>() {   // value pairs on packed fp32 instructions (silu_pair)
        const f32x2 p = {d8[blk][2 * k], d8[blk][2 * k + 1]}, g = {d8[4 + blk][2 * k], d8[4 + blk][2 * k + 1]};
        const f32x2 dv = {de[blk][2 * k], de[blk][2 * k + 1]};
        f32x2 sd, dsd, sg;
        silu_pair(p, sd, dsd);
        sg = sigmoid_pair(g);
        const f32x2 a_g = dv * sg, dd = a_g * dsd, dgt = (a_g * sd) * (1.f - sg);
        d8[blk][2 * k] = dd[0]; d8[blk][2 * k + 1] = dd[1];
        d8[4 + blk][2 * k] = dgt[0]; d8[4 + blk][2 * k + 1] = dgt[1];
      });
    });
    f32x4 dmv[1];
    zero(dmv);
    chain_p<PREC, 1, 4>(lds + L.tbT, d8, dmv, lv, a.w_inv);
    store_dh(a.dh, edge, a.E, dhv, qd);
    if (edge < a.E && arow >= 0) *(f32x4*)(a.dm + (int64_t)arow * kCP + 4 * qd) = dmv[0];
    st.template mark<11>();  // dL/de store, three-body reverse, dm / dh stores
    if (!has_next) break;
    ci_i = nci;
    cj_i = ncj;
    arow_i = narow;
  }
  if (ST && lane == 0) {
    const int wave = threadIdx.x >> 6;
    unsigned long long* dst = a.stamps + ((size_t)blockIdx.x * 16 + wave) * 12;
    for (int i = 0; i < 12; ++i) dst[i] += st.sum[i];   // summed over the launches since the option was set (a slot per wave)
  }
}

// ---------------------------------------------------------------------------------------------- helpers
// tile-SoA element index of (edge, feature)
__device__ __forceinline__ int64_t soa_index(int64_t edge, int o) {
  return (edge >> 4) * kTileFloats + (o >> 4) * 256 + (((o >> 2) & 3) * 16 + (int)(edge & 15)) * 4 + (o & 3);
}

__global__ void __launch_bounds__(256) k_soa_to_rows(int64_t E, int width, int row_stride, const float* __restrict__ soa,
                                                     float* __restrict__ rows) {
  int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= E * kDP) return;
  int64_t edge = idx >> 6;
  int o = (int)(idx & 63);
  if (o < width) rows[edge * row_stride + o] = soa[soa_index(edge, o)];
}
__global__ void __launch_bounds__(256) k_rows_to_soa(int64_t E, int64_t tiles, const float* __restrict__ rows, float* __restrict__ soa) {
  int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= tiles * kTileFloats) return;
  int64_t tile = idx / kTileFloats;
  int rem = (int)(idx % kTileFloats), blk = rem >> 8, lane = (rem >> 2) & 63, reg = rem & 3;
  int64_t edge = tile * kTileEdges + (lane & 15);
  int o = blk * 16 + 4 * (lane >> 4) + reg;
  soa[idx] = edge < E ? rows[edge * kDP + o] : 0.f;
}

// e0 = SiLU(W_adj h) written straight into the tile-SoA image (nn/featurizer.py:128-132)
__global__ void __launch_bounds__(256) k_embed_edges_soa(int R, int64_t E, int64_t tiles, const float* __restrict__ adj_t,
                                                         const float* __restrict__ h, float* __restrict__ soa) {
  int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= tiles * kTileFloats) return;
  int64_t tile = idx / kTileFloats;
  int rem = (int)(idx % kTileFloats), blk = rem >> 8, lane = (rem >> 2) & 63, reg = rem & 3;
  int64_t edge = tile * kTileEdges + (lane & 15);
  int o = blk * 16 + 4 * (lane >> 4) + reg;
  float v = 0.f;
  if (edge < E) {
    float acc = 0.f;
    for (int r = 0; r < R; ++r) acc += adj_t[r * kDP + o] * h[edge * kRP + r];
    v = silu_f(acc);
  }
  soa[idx] = v;
}

// dh[e,:] += sum_o de[e,o] SiLU'(pe0[e,o]) W_adj[o,:] reading de from its tile-SoA image: one lane per (edge, quarter)
__global__ void __launch_bounds__(256) k_embed_edges_reverse_soa(int64_t E, int64_t tiles, const float* __restrict__ adj /*[64][4]*/,
                                                                 const float* __restrict__ h, const float* __restrict__ de_soa,
                                                                 float* __restrict__ dh) {
  int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t tile = gid >> 6;
  if (tile >= tiles) return;  // whole waves exit together (64 lanes per tile)
  int lane = (int)(gid & 63), qd = lane >> 4;
  int64_t edge = tile * kTileEdges + (lane & 15);
  int64_t ec = edge < E ? edge : E - 1;
  const f32x4 hv = *(const f32x4*)(h + ec * kRP);
  const float* src = de_soa + tile * kTileFloats + lane * 4;
  float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    const f32x4 dv = *(const f32x4*)(src + blk * 256);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      int o = blk * 16 + 4 * qd + reg;
      const f32x4 w = *(const f32x4*)(adj + o * kRP);
      float pe = w[0] * hv[0] + w[1] * hv[1] + w[2] * hv[2] + w[3] * hv[3];
      float t = dv[reg] * dsilu_f(pe);
      d0 += t * w[0]; d1 += t * w[1]; d2 += t * w[2]; d3 += t * w[3];
    }
  }
  d0 += __shfl_xor(d0, 16); d1 += __shfl_xor(d1, 16); d2 += __shfl_xor(d2, 16); d3 += __shfl_xor(d3, 16);
  d0 += __shfl_xor(d0, 32); d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32); d3 += __shfl_xor(d3, 32);
  if (qd == 0 && edge < E) *(f32x4*)(dh + edge * kRP) = f32x4{d0, d1, d2, d3};   // this kernel's own slice
}

void launch_rows_to_soa(const float* rows, float* soa, int64_t E, hipStream_t s) {
  int64_t tiles = tiles_for(E);
  if (tiles > 0)
    hipLaunchKernelGGL(k_rows_to_soa, dim3((unsigned)((tiles * kTileFloats + 255) / 256)), dim3(256), 0, s, E, tiles, rows, soa);
}
void launch_soa_to_rows(const float* soa, float* rows, int row_stride, int width, int64_t E, hipStream_t s) {
  if (E > 0) hipLaunchKernelGGL(k_soa_to_rows, dim3((unsigned)((E * kDP + 255) / 256)), dim3(256), 0, s, E, width, row_stride, soa, rows);
}
void launch_embed_edges_soa(const Consts& c, const float* adj_t, const float* h, float* soa, int64_t E, hipStream_t s) {
  int64_t tiles = tiles_for(E);
  if (tiles > 0)
    hipLaunchKernelGGL(k_embed_edges_soa, dim3((unsigned)((tiles * kTileFloats + 255) / 256)), dim3(256), 0, s, c.R, E, tiles, adj_t, h, soa);
}
void launch_embed_edges_reverse_soa(const float* adj, const float* h, const float* de_soa, float* dh, int64_t E, hipStream_t s) {
  int64_t tiles = tiles_for(E);
  if (tiles > 0)
    hipLaunchKernelGGL(k_embed_edges_reverse_soa, dim3((unsigned)((tiles * 64 + 255) / 256)), dim3(256), 0, s, E, tiles, adj, h, de_soa, dh);
}

// for_reverse = false (energy-only call): nothing is saved for a reverse pass that will not run (2 KB per edge and block of stores)
void launch_edge_block_mfma(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, bool for_reverse, hipStream_t s) {
  const int64_t tiles = tiles_for(t.E);
  const MfmaFwdLayout L = mfma_fwd_layout();
  if (tiles > 0 && tiles <= plan->small_tiles_fwd && launch_edge_fwd_split(plan, c, t, w, b, for_reverse, s)) return;
  if (tiles > 0) {
    FwdArgs a{t.E, tiles, plan->d_mfma_fwd[plan->precision] + (size_t)b * L.total, t.src, t.dst, w.h, w.m[b], w.TAb[b], w.TBb[b], t.act_id,
              w.e_blk[b], w.e_blk[b + 1], w.seg_head, w.seg_first, plan->d_stamps, saves_p1(plan) ? w.p1_blk[b] : nullptr,
              saves_p2(plan) ? w.p2_blk[b] : nullptr, plan->precision == kPrecF16x3 ? plan->w_scale_inv : 1.f};
    dim3 grid(grid_for_tiles(tiles));
    const bool first = b == 0 && fused_reverse(plan);   // the fused reverse kernel recomputes e0 as well: no embedded-edge image at all
    const int save = for_reverse ? saved_activations(plan) : 0;   // fp32 mode only (saves_p1 / saves_p2)
#define M3G_FWD_LAUNCH(ST_, FIRST_, PREC_, SAVE_) hipLaunchKernelGGL((k_edge_block_mfma<TBS, ST_, FIRST_, PREC_, SAVE_>), grid, dim3(64 * fwd_waves<PREC_>()), 0, s, a, L)
#define M3G_FWD_BY_MODE(ST_, FIRST_)                                                               \
  if (plan->precision == kPrecBf16x3) { M3G_FWD_LAUNCH(ST_, FIRST_, kPrecBf16x3, 0); }             \
  else if (plan->precision == kPrecF16x3) { M3G_FWD_LAUNCH(ST_, FIRST_, kPrecF16x3, 0); }          \
  else if (save == 2) { M3G_FWD_LAUNCH(ST_, FIRST_, kPrecF32, 2); }                                \
  else if (save == 1) { M3G_FWD_LAUNCH(ST_, FIRST_, kPrecF32, 1); }                                \
  else { M3G_FWD_LAUNCH(ST_, FIRST_, kPrecF32, 0); }
    if (plan->d_stamps && plan->stamp_target == 0 && tb_steps_for(c.C) == 3) {
      // diagnostic build of the default configuration (same code path as the shipped kernel, block 0 included)
      constexpr int TBS = 3;
      if (first) { M3G_FWD_BY_MODE(true, true); } else { M3G_FWD_BY_MODE(true, false); }
    } else if (first) {
      M3G_TBS_SWITCH(c.C, M3G_FWD_BY_MODE(false, true));
    } else {
      M3G_TBS_SWITCH(c.C, M3G_FWD_BY_MODE(false, false));
    }
#undef M3G_FWD_BY_MODE
#undef M3G_FWD_LAUNCH
  }
}

void launch_edge_rev_node_mlp(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* dx_new,
                              hipStream_t s) {
  (void)c;
  const int64_t tiles = tiles_for(t.E);
  if (tiles == 0) return;
  const MfmaRevLayout L = mfma_rev_layout();
  const float* img_n = plan->d_mfma_rev[plan->precision] + (size_t)b * L.per_block + L.total_e;
  const bool saved = saves_p1(plan);
  RevArgs an{t.E, tiles, img_n, t.src, t.dst, w.h, w.m[b], dx_new, t.act_id, w.TAb[b], w.TBb[b], w.e_blk[b + 1], nullptr, w.de_soa, w.dcn, 0, w.dm,
             w.dh_parts + (size_t)(2 * b + 1) * t.E * kRP, w.dp1, nullptr, nullptr, nullptr, saved ? w.p1_blk[b] : nullptr, nullptr,
             plan->precision == kPrecF16x3 ? plan->w_scale_inv : 1.f};
  if (saved) { M3G_PREC_SWITCH(plan->precision, hipLaunchKernelGGL((k_edge_rev_node_mlp<PREC, true>), dim3(grid_for_tiles(tiles)), dim3(64 * kWavesRev), 0, s, an, L)); }
  else { M3G_PREC_SWITCH(plan->precision, hipLaunchKernelGGL((k_edge_rev_node_mlp<PREC, false>), dim3(grid_for_tiles(tiles)), dim3(64 * kWavesRev), 0, s, an, L)); }
}

void launch_edge_rev_edge_mlp(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* dx_new,
                              bool de_is_zero, hipStream_t s) {
  const int64_t tiles = tiles_for(t.E);
  if (tiles == 0) return;
  const MfmaRevLayout L = mfma_rev_layout();
  const float* img_e = plan->d_mfma_rev[plan->precision] + (size_t)b * L.per_block;
  RevArgs ae{t.E, tiles, img_e, t.src, t.dst, w.h, w.m[b], dx_new, t.act_id, w.TAb[b], w.TBb[b], w.e_blk[b], nullptr, w.de_soa, w.dcn,
             de_is_zero ? 1 : 0, w.dm, w.dh_parts + (size_t)(2 * b) * t.E * kRP, w.dp1, plan->d_stamps, nullptr, nullptr,
             saves_p1(plan) ? w.p1_blk[b] : nullptr, nullptr, plan->precision == kPrecF16x3 ? plan->w_scale_inv : 1.f};
  dim3 grid(grid_for_tiles(tiles)), block(64 * kWavesRev);
  if (plan->d_stamps && plan->stamp_target == 1 && tb_steps_for(c.C) == 3 && plan->precision == kPrecBf16x3) {  // diagnostic build
    hipLaunchKernelGGL((k_edge_rev_edge_mlp<3, true>), grid, block, 0, s, ae, L);
    return;
  }
  if (saves_p1(plan)) { M3G_PREC_SWITCH(plan->precision, M3G_TBS_SWITCH(c.C, hipLaunchKernelGGL((k_edge_rev_edge_mlp<TBS, false, PREC, true>), grid, block, 0, s, ae, L))); }
  else { M3G_PREC_SWITCH(plan->precision, M3G_TBS_SWITCH(c.C, hipLaunchKernelGGL((k_edge_rev_edge_mlp<TBS, false, PREC, false>), grid, block, 0, s, ae, L))); }
}

void launch_edge_rev_fused(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* dx_new,
                           bool de_is_zero, hipStream_t s) {
  const int64_t tiles = tiles_for(t.E);
  if (tiles == 0) return;
  const MfmaRevFusedLayout L = mfma_rev_fused_layout();
  const bool f16 = plan->precision == kPrecF16x3;
  const float* img = (f16 ? plan->d_mfma_revf_h : plan->d_mfma_revf) + (size_t)b * L.total;
  RevArgs ar{t.E, tiles, img, t.src, t.dst, w.h, w.m[b], dx_new, t.act_id, w.TAb[b], w.TBb[b], w.e_blk[b], w.e_blk[b + 1], w.de_soa, nullptr,
             de_is_zero ? 1 : 0, w.dm, w.dh_parts + (size_t)b * t.E * kRP, w.dp1, plan->d_stamps, w.seg_head, w.seg_first, nullptr, nullptr,
             f16 ? plan->w_scale_inv : 1.f, dp1_scale_of(w.dp1, t.E)};
  constexpr int WV = kWavesRevFused;
  dim3 grid(grid_for_tiles(tiles)), block(64 * WV);
  if (plan->d_stamps && plan->stamp_target == 2 && f16 && tb_steps_for(c.C) == 3) {   // diagnostic build (tools/stamp_report_fused.py)
    if (b > 0) hipLaunchKernelGGL((k_edge_rev_fused<3, true, WV, kPrecF16x3, true>), grid, block, 0, s, ar, L);
    else hipLaunchKernelGGL((k_edge_rev_fused<3, false, WV, kPrecF16x3, true>), grid, block, 0, s, ar, L);
    return;
  }
#define M3G_REVF_LAUNCH(NEED_) \
  if (f16) { M3G_TBS_SWITCH(c.C, hipLaunchKernelGGL((k_edge_rev_fused<TBS, NEED_, WV, kPrecF16x3>), grid, block, 0, s, ar, L)); } \
  else { M3G_TBS_SWITCH(c.C, hipLaunchKernelGGL((k_edge_rev_fused<TBS, NEED_, WV, kPrecBf16x3>), grid, block, 0, s, ar, L)); }
  if (b > 0) { M3G_REVF_LAUNCH(true); }
  else { M3G_REVF_LAUNCH(false); }   // x^0 has no position dependence: nobody reads block 0's dp1 rows
#undef M3G_REVF_LAUNCH
}

}  // namespace m3g
