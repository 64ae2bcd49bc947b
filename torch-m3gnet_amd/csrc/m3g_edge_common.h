// Shared device code of the MFMA edge / node kernels (m3g_edge_mfma.hip, m3g_edge_rev_f32.hip, m3g_node_mfma.hip): chain
// primitives in both precision modes, the persistent tile queue, per-centre segmented scans, image staging, stamps and the
// kernel argument blocks.  Design notes: m3g_edge_mfma.hip header and DESIGN.md section 4.
#pragma once
#include <utility>

#include "m3g_device.h"
#include "m3g_internal.h"
#include "m3g_dual_chain.h"
#include "m3g_dual_f32.h"
#include "m3g_mfma_common.h"

namespace m3g {

constexpr int kFwdLdsFloats = 8 * kTbSteps * 64 + 2 * (8 * 4 * 4 * 64 + 2 * (4 * 4 * 4 * 64) + 2 * 4 * 64 + 4 * 64) + 4 * 64;
constexpr int kRevMlpFloats = 8 * 4 * 4 * 64 + 4 * (4 * 4 * 4 * 64) + 2 * 4 * 64 + 4 * 8 * 4 * 64 + 64 * 4;   // node-MLP reverse image
constexpr int kRevEdgeFloats = kRevMlpFloats + 8 * kTbSteps * 64 + 8 * 4 * 64;                // + three-body images
#ifndef M3G_WAVES_FWD
#define M3G_WAVES_FWD 16        // forward kernel: 16 waves = 4 per SIMD (<= 128 VGPRs)
#endif
constexpr int kWaves = M3G_WAVES_FWD;
#ifndef M3G_WAVES_FWD_H
#define M3G_WAVES_FWD_H 12      // f16x3 forward: 3 per SIMD (<= 168 VGPRs), which pays for the A operands requested a row block ahead (chain_h)
#endif
// waves per workgroup of the forward kernel by precision mode (kPrecF16x3 = 2, m3g_internal.h)
template <int PREC>
#ifndef M3G_WAVES_FWD_BF
#define M3G_WAVES_FWD_BF 12     // bf16x3 forward: as the f16x3 kernel
#endif
constexpr int fwd_waves() { return PREC == kPrecF16x3 ? M3G_WAVES_FWD_H : PREC == kPrecBf16x3 ? M3G_WAVES_FWD_BF : kWaves; }
#ifndef M3G_WAVES_REV_FUSED
#define M3G_WAVES_REV_FUSED 8   // 2 waves per SIMD, 256 VGPRs, no spills (12 waves: 168 VGPRs and ~120 spilled, slower)
#endif
constexpr int kWavesRevFused = M3G_WAVES_REV_FUSED;
#ifndef M3G_WAVES_REV
#define M3G_WAVES_REV 12
#endif
constexpr int kWavesRev = M3G_WAVES_REV;   // reverse kernels hold layer-1 pre-activations across the recompute: 3 per SIMD (<= 168 VGPRs)
constexpr int kTileEdges = 16;
constexpr int kTileFloats = 4 * 64 * 4;     // one 64-feature tile image: [4 blk][64 lanes][4]
constexpr int kP1TileFloats = 8 * 64 * 4;   // layer-1 pre-activations of one MLP and tile: [8 blk][64 lanes][4]
// Nothing is saved for the reverse pass except the per-block edge-feature images and node tables: with the dense
// chains on bf16x3 the matrix work is cheap, and recomputing both layers of both MLPs in the reverse kernels costs less
// than streaming 2 KB of pre-activations per edge and block through HBM (measured history: DESIGN.md section 4).

#ifdef M3G_USE_FWD_CHAIN_PRIO
#define M3G_FWD_CHAIN_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define M3G_FWD_CHAIN_PRIO(p) ((void)0)
#endif
// acc[AOFF + ob] += W(ob-th 16-row block, :) . x[XOFF .. XOFF + 2*KS)   (chain image: m3g_pack_mfma.hip)
template <int OB, int KS, int XOFF = 0, int AOFF = 0, int NX, int NA>
__device__ __forceinline__ void chain(const float* img, const f32x4 (&x)[NX], f32x4 (&acc)[NA], int lane) {
  static_assert(XOFF + 2 * KS <= NX && AOFF + OB <= NA, "chain operand out of range");
  const bf16x8* hi_img = reinterpret_cast<const bf16x8*>(img) + lane;
  const bf16x8* lo_img = hi_img + OB * KS * 64;
  M3G_FWD_CHAIN_PRIO(1);
#ifndef M3G_NO_BF16_CHAIN_PREFETCH
  // A operands of item (s, ob + 1) requested before the MFMAs of item (s, ob), as chain_h does in the f16x3 mode (bf16x3 forward
  // kernel with 12 waves: 0.426 -> 0.416 ms per step; 12 waves without it spill and are slower, 0.451)
  bf16x8 bh[KS], bl[KS];
  static_for<KS>([&]<int s>() { split8(x[XOFF + 2 * s], x[XOFF + 2 * s + 1], bh[s], bl[s]); });
  bf16x8 ah = hi_img[0], al = lo_img[0];
  static_for<KS * OB>([&]<int it>() {
    constexpr int s = it / OB, ob = it % OB;
    bf16x8 nh = ah, nl = al;
    if constexpr (it + 1 < KS * OB) {
      constexpr int s1 = (it + 1) / OB, ob1 = (it + 1) % OB;
      nh = hi_img[(ob1 * KS + s1) * 64];
      nl = lo_img[(ob1 * KS + s1) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    acc[AOFF + ob] = mfma_bf16(ah, bh[s], acc[AOFF + ob]);
    acc[AOFF + ob] = mfma_bf16(ah, bl[s], acc[AOFF + ob]);
    acc[AOFF + ob] = mfma_bf16(al, bh[s], acc[AOFF + ob]);
    __builtin_amdgcn_sched_barrier(0);
    ah = nh; al = nl;
  });
#else
  static_for<KS>([&]<int s>() {
    bf16x8 bh, bl;
    split8(x[XOFF + 2 * s], x[XOFF + 2 * s + 1], bh, bl);
    static_for<OB>([&]<int ob>() {
      const bf16x8 ah = hi_img[(ob * KS + s) * 64], al = lo_img[(ob * KS + s) * 64];
      acc[AOFF + ob] = mfma_bf16(ah, bh, acc[AOFF + ob]);
      acc[AOFF + ob] = mfma_bf16(ah, bl, acc[AOFF + ob]);
      acc[AOFF + ob] = mfma_bf16(al, bh, acc[AOFF + ob]);
    });
  });
#endif
  M3G_FWD_CHAIN_PRIO(0);
}

// exact-fp32 chain on v_mfma_f32_16x16x4_f32: acc[AOFF + ob] += W(ob-th row block, :) . x[XOFF .. XOFF + NB) with the
// accumulator blocks of x as the B operand (k-step blk*4 + r = register r of block blk; image: f32_chain_image).
// Bitwise a k-ordered fp32 fmaf chain per output element (cdna_hip_programming.md section 3): the reference's arithmetic.
// Wave priority: vector and matrix instructions share a SIMD's issue port and the arbiter serves the oldest wave first, so a
// vector instruction of an older wave that is ready when the matrix pipe frees delays the next MFMA by its 4 issue cycles
// (tools/mfma_f32_dep_probe.hip: 36 instead of 32 cycles per MFMA with one v_fma per MFMA in the stream, at 1, 2 and 4 waves
// per SIMD).  Raised priority inside the chains lets the wave that feeds the matrix pipe win that arbitration.
#if defined(M3G_F32_STAGGER_PRIO)
// experiment: the waves that share a SIMD (wave ids w, w+4, w+8, ...) enter their chains at different priorities, so two
// co-running chains do not split the matrix pipe evenly and leave it at the same moment
__device__ __forceinline__ void f32_chain_prio_on() {
  const int cls = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  if (cls == 0) __builtin_amdgcn_s_setprio(3);
  else if (cls == 1) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(1);
}
#define M3G_F32_CHAIN_PRIO(p) do { if (p) f32_chain_prio_on(); else __builtin_amdgcn_s_setprio(0); } while (0)
#elif defined(M3G_F32_INVERSE_PRIO)
// experiment: the vector phases run at raised priority, the chains at 0
#define M3G_F32_CHAIN_PRIO(p) __builtin_amdgcn_s_setprio((p) ? 0 : M3G_F32_INVERSE_PRIO)
#elif !defined(M3G_NO_F32_CHAIN_PRIO)
#define M3G_F32_CHAIN_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define M3G_F32_CHAIN_PRIO(p) ((void)0)
#endif
// NBT / KB0: the image holds NBT k-blocks per row block, this call consumes blocks KB0 .. KB0 + NB of it (a slice of the k range)
template <int OB, int NB, int XOFF = 0, int AOFF = 0, int NBT = NB, int KB0 = 0, int NX, int NA>
__device__ __forceinline__ void chain_f32(const float* img, const f32x4 (&x)[NX], f32x4 (&acc)[NA], int lane) {
  static_assert(XOFF + NB <= NX && AOFF + OB <= NA && KB0 + NB <= NBT, "chain_f32 operand out of range");
  M3G_F32_CHAIN_PRIO(1);
  static_for<NB>([&]<int blk>() {
    static_for<4>([&]<int r>() {
      const float b = x[XOFF + blk][r];
      static_for<OB>([&]<int ob>() { acc[AOFF + ob] = mfma16(img[(ob * (4 * NBT) + (KB0 + blk) * 4 + r) * 64 + lane], b, acc[AOFF + ob]); });
    });
  });
  M3G_F32_CHAIN_PRIO(0);
}

// Precision modes of the dense chains (plan option "precision"):
//   kPrecF32     every product on v_mfma_f32_16x16x4_f32 -- exact fp32 products, fp32 accumulate (the reference's arithmetic);
//   kPrecBf16x3  operands split into two bf16 parts, 3 v_mfma_f32_16x16x32_bf16 products per fp32 product, fp32 accumulate
//                (relative product error ~2^-16).
// Both read an image of the same size and offsets (an fp32 image is as large as a bf16 hi + lo pair); KS counts 32-wide
// k-steps, i.e. 2*KS accumulator blocks of x.
template <int PREC, int OB, int KS, int XOFF = 0, int AOFF = 0, int NX, int NA>
__device__ __forceinline__ void chain_p(const float* img, const f32x4 (&x)[NX], f32x4 (&acc)[NA], int lane, float w_inv = 1.f);

// ---- f16x3 mode: scaled two-part fp16 operands (edge_scale / split_h: m3g_mfma_common.h) -----------------------------------------
// out(ob, t): t = Ws(ob-th 16-row block, :) . (s x), Ws = the image's weights (scaled by the model's power of two at pack time);
// the caller unscales with EdgeScale::inv * w_inv where it consumes t (an fma with the bias / table row it would add anyway).
// One accumulator at a time: the scaled sums never coexist with the unscaled pre-activations they are folded into.
template <int OB, int KS, class OUT>
__device__ __forceinline__ void chain_h(const float* img, const HalfB<KS>& b, int lane, OUT&& out) {
  const f16x8* hi_img = reinterpret_cast<const f16x8*>(img) + lane;
  const f16x8* lo_img = hi_img + OB * KS * 64;
#ifndef M3G_NO_FWD_CHAIN_PREFETCH
  // A operands of row block ob + 1 requested before the MFMAs of row block ob (as the dual-image chains, m3g_dual_chain.h:
  // the compiler's own order leaves two or three MFMAs between a ds_read and its use, a third of the LDS round trip; forward kernel
  // with 12 waves 0.460 -> 0.427 ms per step -- at 16 waves = 128 VGPRs the extra 16 registers spill and it is slower, 0.480)
  auto fetch = [&]<int ob>() {
    DualA<KS> a;
    static_for<KS>([&]<int s>() { a.h[s] = hi_img[(ob * KS + s) * 64]; a.l[s] = lo_img[(ob * KS + s) * 64]; });
    return a;
  };
  DualA<KS> cur = fetch.template operator()<0>();
  static_for<OB>([&]<int ob>() {
    DualA<KS> nxt = cur;
    if constexpr (ob + 1 < OB) nxt = fetch.template operator()<ob + 1>();
    __builtin_amdgcn_sched_barrier(0);
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    static_for<KS>([&]<int s>() {
      t = mfma_f16(cur.h[s], b.hi[s], t);
      t = mfma_f16(cur.h[s], b.lo[s], t);
      t = mfma_f16(cur.l[s], b.hi[s], t);
    });
    out.template operator()<ob>(t);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  });
#else
  static_for<OB>([&]<int ob>() {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    static_for<KS>([&]<int s>() {
      const f16x8 ah = hi_img[(ob * KS + s) * 64], al = lo_img[(ob * KS + s) * 64];
#ifdef M3G_DIAG_NO_AL   // timing diagnostic only (wrong results): no LDS reads of the low-part image
      const f16x8& al_ = ah;
#else
      const f16x8& al_ = al;
#endif
      t = mfma_f16(ah, b.hi[s], t);
#ifndef M3G_DIAG_H1   // timing diagnostic only (wrong results): one product per k-step instead of three
      t = mfma_f16(ah, b.lo[s], t);
      t = mfma_f16(al_, b.hi[s], t);
#elif defined(M3G_DIAG_H1_KEEP)   // ... with the low-part reads and splits kept alive: the MFMAs alone
      asm volatile("" ::"v"(al_), "v"(b.lo[s]));
#endif
    });
    out.template operator()<ob>(t);
  });
#endif
}
// acc[AOFF + ob] += (W x)[ob-th row block] in true units: the f16x3 counterpart of chain_p (scale of x taken over its 2 KS blocks)
template <int OB, int KS, int XOFF = 0, int AOFF = 0, int NX, int NA>
__device__ __forceinline__ void chain_h_acc(const float* img, const f32x4 (&x)[NX], f32x4 (&acc)[NA], int lane, float w_inv) {
  static_assert(AOFF + OB <= NA, "chain_h_acc operand out of range");
  const EdgeScale sc = edge_scale<2 * KS, XOFF>(x);
  const HalfB<KS> b = split_h<KS, XOFF>(x, sc.s);
  const float inv = sc.inv * w_inv;
  chain_h<OB, KS>(img, b, lane, [&]<int ob>(const f32x4& t) {
    acc[AOFF + ob] = t * inv + acc[AOFF + ob];   // (vector form: the compiler emits two v_pk_fma_f32)
  });
}

template <int PREC, int OB, int KS, int XOFF, int AOFF, int NX, int NA>
__device__ __forceinline__ void chain_p(const float* img, const f32x4 (&x)[NX], f32x4 (&acc)[NA], int lane, float w_inv) {
  if constexpr (PREC == kPrecBf16x3) chain<OB, KS, XOFF, AOFF>(img, x, acc, lane);
  else if constexpr (PREC == kPrecF16x3) chain_h_acc<OB, KS, XOFF, AOFF>(img, x, acc, lane, w_inv);
  else chain_f32<OB, 2 * KS, XOFF, AOFF>(img, x, acc, lane);
}

// bias image: lanes < 16 of block ob carry b[ob*16 + lane] (built as the A operand of a k-step against a constant one).
// The accumulator registers of lane (m, q) are rows 4q .. 4q+3 of the block, so the same image read as one 16-byte LDS
// broadcast per block initialises the accumulators directly -- identical values, no MFMA.
#ifdef M3G_BIAS_MFMA
template <int OB, int AOFF, int NA>
__device__ __forceinline__ void bias_step(const float* img, f32x4 (&acc)[NA], int lane) {
  const float one = lane < 16 ? 1.f : 0.f;
  static_for<OB>([&]<int ob>() { acc[AOFF + ob] = mfma16(img[ob * 64 + lane], one, f32x4{0.f, 0.f, 0.f, 0.f}); });
}
#else
template <int OB, int AOFF, int NA>
__device__ __forceinline__ void bias_step(const float* img, f32x4 (&acc)[NA], int lane) {
  const int q = lane >> 4;
  static_for<OB>([&]<int ob>() { acc[AOFF + ob] = *(const f32x4*)(img + ob * 64 + 4 * q); });
}
#endif

// Persistent tile queue.  Static over workgroups, dynamic inside one:
//   * workgroups with the same blockIdx % 8 share an XCD (speed only, never correctness); that label owns one
//     contiguous eighth of the tiles, cut into equal contiguous chunks, one per workgroup -> the TA/TB rows a
//     workgroup gathers stay in its L1/L2;
//   * the 16 waves of a workgroup pull tiles of its chunk from an LDS counter.  A SIMD arbitrates its resident
//     waves oldest-first, so with equal static shares the old waves finish early and the matrix pipe runs
//     under-occupied in the tail (measured: waves 0-3 done at 320 k cycles, waves 12-15 at 680 k).
//     A global atomic head was tried and rejected: its microsecond return sits in front of every tile load on the
//     in-order vmcnt queue.
struct TileQueue {
  int* head;       // LDS counter of this workgroup
  int64_t base;
  int count;
  __device__ TileQueue(int64_t n_tiles, int* lds_head) {
    const int64_t per_xcd = (n_tiles + 7) / 8;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, wgs = gridDim.x >> 3;
    const int64_t chunk = (per_xcd + wgs - 1) / wgs;
    const int64_t lo = (int64_t)xcd * per_xcd + (int64_t)q * chunk;
    int64_t hi = lo + chunk;
    const int64_t xcd_end = (int64_t)(xcd + 1) * per_xcd < n_tiles ? (int64_t)(xcd + 1) * per_xcd : n_tiles;
    if (hi > xcd_end) hi = xcd_end;
    base = lo;
    count = hi > lo ? (int)(hi - lo) : 0;
    head = lds_head;
  }
  __device__ __forceinline__ int fetch(int lane) const {
    int v = 0;
    if (lane == 0) v = atomicAdd(head, 1);
    return __builtin_amdgcn_readfirstlane(v);
  }
};
// (Round 6, measured and dropped: the final PART round of a workgroup handed out statically, tile j to wave j, so that its tiles land on
//  different SIMDs -- 1,372 atoms -4 us per step, but 4,000 / 5,324 / 6,912 atoms +15 .. +40 us and the 10,000-atom reverse kernel +2 %:
//  the dynamic counter gives the last tiles to the waves that finish FIRST, which matters more than which SIMD they sit on.)

// streamed-once tile loads: nontemporal, so they do not evict the node tables the gathers re-use from L2
// (forward 0.425 -> 0.418 ms, fused reverse 0.892 -> 0.884 per step)
__device__ __forceinline__ f32x4 load_tile4(const float* p) { return __builtin_nontemporal_load((const f32x4*)p); }

// centre / neighbour atom of this lane's edge in `tile` (clamped for the padding lanes of the last tile)
__device__ __forceinline__ void load_ends(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t tile, int64_t E,
                                          int lane, int& ci, int& cj) {
  const int64_t edge = tile * kTileEdges + (lane & 15);
  const int64_t ec = edge < E ? edge : E - 1;
  ci = src[ec];
  cj = dst ? dst[ec] : 0;
}
// ... and its row in the per-active-edge arrays (< 0: the edge takes part in no triplet): fetched with the end atoms, one tile
// ahead, so the aggregate row `m[arow]` is one round trip away at the tile start instead of two dependent ones
__device__ __forceinline__ int load_arow(const int32_t* __restrict__ act_id, int64_t tile, int64_t E, int lane) {
  const int64_t edge = tile * kTileEdges + (lane & 15);
  return act_id[edge < E ? edge : E - 1];
}

// ---- per-centre sums inside a tile -------------------------------------------------------------------------------
// The 16 edges of a tile sit on the 16 lanes of a DPP row and edges of one centre are consecutive, so the sum over a
// centre's edges is a segmented inclusive scan along the row: 4 row_shr steps, each a DPP-sourced FMA with a 0/1 mask
// (valid because runs are contiguous: equal centres n lanes apart imply equal centres in between).  The last lane of
// a run then holds the run's sum.
struct SegMasks {
  float m1, m2, m4, m8;
  bool run_end;    // this lane's edge is the last of its run inside the tile
  bool first_run;  // the run contains the tile's column 0
};
template <int N>
__device__ __forceinline__ int row_shr_i(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x110 + N, 0xf, 0xf, false); }
template <int N>
__device__ __forceinline__ float row_shr_f(float v) {
  // bound_ctrl: lanes without a source read 0, which lets the compiler fold the DPP move into the consuming v_fmac
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ SegMasks seg_masks(int ci, int lane) {
  SegMasks k;
  k.m1 = row_shr_i<1>(ci, -1) == ci ? 1.f : 0.f;
  k.m2 = row_shr_i<2>(ci, -1) == ci ? 1.f : 0.f;
  k.m4 = row_shr_i<4>(ci, -1) == ci ? 1.f : 0.f;
  k.m8 = row_shr_i<8>(ci, -1) == ci ? 1.f : 0.f;
  const int next = __builtin_amdgcn_update_dpp(-1, ci, 0x100 + 1, 0xf, 0xf, false);   // row_shl:1 -> lane m+1 (fill -1 at m = 15)
  k.run_end = next != ci;
  k.first_run = ci == __builtin_amdgcn_readlane(ci, 0);
  return k;
}
// One scan step for the 16 values of four accumulator blocks as v_fmac_f32 with a DPP source operand: x += dpp(x) * m in
// one instruction per value.  The compiler never forms that instruction (it SLP-packs the FMAs into v_pk_fma_f32 behind
// two v_mov_b32_dpp, 1.5 instructions per value and step).
// Hazards (the recogniser does not look inside inline assembly; at an asm boundary it pads only the dst_sel-forwarding and
// 12-dword-store cases, LLVM GCNHazardRecognizer::checkInlineAsmHazards): the consumer is a DPP instruction, so the
// software-managed producers are a VALU write of a DPP-read VGPR (2 wait states), a transcendental or dst_sel write of a
// VGPR the block reads (1) and a VALU write of EXEC (5).  The block opens with `s_nop 4` = 5 wait states, the worst case of
// that table, so it is safe whatever the compiler schedules in front of it; inside, the 16 independent values keep
// consecutive steps of one value 16 instructions apart.  (An MFMA result never feeds the block directly: every input passes
// through a vector multiply first.)  tools/asm_hazard_audit.py checks a listing against the same table: margins >= 2 wait
// states in every kept listing of round 1, the rejected vectorised-activation variants included.
#define M3G_SCAN_LINE(i, SHR) "v_fmac_f32_dpp %" #i ", %" #i ", %16 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define M3G_SCAN_STEP16(SHR, m)                                                                                            \
  asm volatile("s_nop 4\n\t" M3G_SCAN_LINE(0, SHR) M3G_SCAN_LINE(1, SHR) M3G_SCAN_LINE(2, SHR) M3G_SCAN_LINE(3, SHR)         \
                   M3G_SCAN_LINE(4, SHR) M3G_SCAN_LINE(5, SHR) M3G_SCAN_LINE(6, SHR) M3G_SCAN_LINE(7, SHR)                    \
                       M3G_SCAN_LINE(8, SHR) M3G_SCAN_LINE(9, SHR) M3G_SCAN_LINE(10, SHR) M3G_SCAN_LINE(11, SHR)              \
                           M3G_SCAN_LINE(12, SHR) M3G_SCAN_LINE(13, SHR) M3G_SCAN_LINE(14, SHR) M3G_SCAN_LINE(15, SHR)        \
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),  \
                 "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])                     \
               : "v"(m))
#ifndef M3G_NO_ASM_SCAN
__device__ __forceinline__ void seg_scan(f32x4 (&v)[4], const SegMasks& k) {
  float x[16];
  static_for<16>([&]<int i>() { x[i] = v[i >> 2][i & 3]; });
  M3G_SCAN_STEP16(1, k.m1);
  M3G_SCAN_STEP16(2, k.m2);
  M3G_SCAN_STEP16(4, k.m4);
  M3G_SCAN_STEP16(8, k.m8);
  static_for<16>([&]<int i>() { v[i >> 2][i & 3] = x[i]; });
}
#endif
template <int N>
__device__ __forceinline__ void seg_scan(f32x4 (&v)[N], const SegMasks& k) {
  static_for<N>([&]<int b>() {
    static_for<4>([&]<int r>() {
      float x = v[b][r];
      x = fmaf(row_shr_f<1>(x), k.m1, x);
      x = fmaf(row_shr_f<2>(x), k.m2, x);
      x = fmaf(row_shr_f<4>(x), k.m4, x);
      x = fmaf(row_shr_f<8>(x), k.m8, x);
      v[b][r] = x;
    });
  });
}
// run-end lanes store their run's sum: the tile's first run into seg_head[tile], a run starting mid-tile into
// seg_first[centre]; row = 4*kDP floats, this call covers blocks [B0, B0+N) of it
template <int B0, int N>
__device__ __forceinline__ void seg_store(const f32x4 (&v)[N], const SegMasks& k, float* seg_head, float* seg_first, int64_t tile,
                                          int64_t ci, int qd) {
  if (k.run_end) {
    float* row = (k.first_run ? seg_head + tile * (4 * kDP) : seg_first + ci * (4 * kDP)) + 4 * qd;
    static_for<N>([&]<int b>() { *(f32x4*)(row + (B0 + b) * 16) = v[b]; });
  }
}

// weight image -> LDS with 8 independent 16-byte loads in flight per thread: a plain load-store loop is a chain of
// dependent L2 round trips (one per blockDim*16 bytes) at the start of every launch
__device__ __forceinline__ void load_image(float* lds, const float* __restrict__ src, int n_floats, int* lds_head) {
  constexpr int kBatch = 8;
  const int n_vec = n_floats >> 2, nt = (int)blockDim.x;
  for (int base = 0; base < n_vec; base += nt * kBatch) {
    f32x4 t[kBatch];
    static_for<kBatch>([&]<int j>() {
      const int i = base + j * nt + (int)threadIdx.x;
      t[j] = i < n_vec ? *(const f32x4*)(src + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
    });
    static_for<kBatch>([&]<int j>() {
      const int i = base + j * nt + (int)threadIdx.x;
      if (i < n_vec) *(f32x4*)(lds + 4 * i) = t[j];
    });
  }
  if (threadIdx.x == 0) *lds_head = 0;
  __syncthreads();
}

// In-kernel phase stamps (diagnostic build only, never in the shipped kernel): s_memtime deltas summed per
// wave into a debug buffer that no other code reads (cdna_hip_programming.md section 7, "In-kernel stamps").
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
template <bool ON>
struct Stamps {
  unsigned long long last = 0, sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  __device__ __forceinline__ void start() { if (ON) last = stamp_now(); }
  template <int I>
  __device__ __forceinline__ void mark() {
    if (ON) { unsigned long long t = stamp_now(); sum[I] += t - last; last = t; }
  }
};

// ---- kernel argument blocks ----------------------------------------------------------------------------------------
struct FwdArgs {
  int64_t E, tiles;
  const float* img;        // forward weight image of this block
  const int32_t *src, *dst;
  const float *h, *m, *TA, *TB;   // m: three-body aggregate, one row per ACTIVE edge (Topo::act_id)
  const int32_t* act_id;
  const float* e_in;       // [tiles][4][64][4] edge features before this block
  float* e_out;            // same shape, after this block
  float *seg_head, *seg_first;   // per-centre message sums (see seg_scan)
  unsigned long long* stamps;  // diagnostic build: [gridDim.x][kWaves][12] phase cycle sums
  float* p1_out;               // fp32 mode: saved layer-1 pre-activations [tiles][2 mlp][8 blk][64 lanes][4], else nullptr
  float* p2_out;               // fp32 mode, saves_p2: layer-2 pre-activations, same shape (p1_out then holds SiLU'(p1)), else nullptr
  float w_inv;                 // f16x3 mode: 1 / the model's weight scale (plan->w_scale_inv), else 1
};

// three-body MLP pre-activations: p[0..3] dense, p[4..7] gate
template <int TBS>
__device__ __forceinline__ void tb_preact(const float* tbimg, const float (&mb)[TBS], f32x4 (&p)[8], int lane) {
  zero(p);
  static_for<TBS>([&]<int s>() {
    const float b = mb[s];
    static_for<8>([&]<int ob>() { p[ob] = mfma16(tbimg[(ob * kTbSteps + s) * 64 + lane], b, p[ob]); });
  });
}

// The three-body MLP input of one tile in the form its precision mode consumes.  fp32 / bf16x3: TBS k-step values per lane for
// exact fp32 MFMAs (24 per tile -- which occupy the SIMD's fp32 datapath for 32 cycles each, section 4a of DESIGN.md).  f16x3:
// the aggregate row as ONE K = 32 f16 k-step -- lane (edge, q) carries columns 8q .. 8q+7 (q < 2; l_max n_max <= 16), scaled per edge
// and split once, used by both evaluations of the reverse kernel -- so the 128 pre-activations cost 24 f16 MFMAs on the matrix
// pipe, which runs beside the vector work, plus ~60 vector instructions (scale, split, unscale).
template <int PREC, int TBS>
struct TbIn { float mb[TBS]; };
template <int TBS>
struct TbIn<kPrecF16x3, TBS> { f16x8 hi, lo; float inv; };

template <int PREC, int TBS>
__device__ __forceinline__ TbIn<PREC, TBS> tb_load(const float* __restrict__ m, int arow, int q, float w_inv) {
  TbIn<PREC, TBS> r;
  if constexpr (PREC == kPrecF16x3) {
    f32x4 v[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (arow >= 0 && q < 2) {   // (< 0: the edge takes part in no triplet, its aggregate is zero)
      const float* row = m + (int64_t)arow * kCP + 8 * q;
      v[0] = *(const f32x4*)row;
      v[1] = *(const f32x4*)(row + 4);
    }
    const EdgeScale sc = edge_scale<2>(v);
    split8h(v[0], v[1], sc.s, r.hi, r.lo);
    r.inv = sc.inv * w_inv;
  } else {
    static_for<TBS>([&]<int s>() { r.mb[s] = arow >= 0 ? m[(int64_t)arow * kCP + 4 * s + q] : 0.f; });
  }
  return r;
}
// p[0..3] dense, p[4..7] gate pre-activations of the three-body MLP.  f16x3 image (tb_image_h, m3g_pack_mfma.hip): [hi | lo]
// [8 row blocks][32 lanes (q < 2)][8 halves]: the k range of quarters 2, 3 is beyond l_max n_max -- tb_load gives those lanes
// zero activations, so whatever finite weights they multiply do not matter and they read the rows of lanes 0..31 again (a
// conditional read costs a branch and eight register clears per row block)
template <int PREC, int TBS>
__device__ __forceinline__ void tb_preact_p(const float* tbimg, const TbIn<PREC, TBS>& in, f32x4 (&p)[8], int lane) {
  if constexpr (PREC == kPrecF16x3) {
    const f16x8* hi = reinterpret_cast<const f16x8*>(tbimg) + (lane & 31);
    const f16x8* lo = hi + 8 * 32;
    // (quarters 2, 3 re-read rows 0..31: their activation parts are zero); operands one row block ahead, as chain_h
    f16x8 ah = hi[0], al = lo[0];
    static_for<8>([&]<int ob>() {
      f16x8 nh = ah, nl = al;
      if constexpr (ob + 1 < 8) { nh = hi[(ob + 1) * 32]; nl = lo[(ob + 1) * 32]; }
#ifndef M3G_NO_FWD_CHAIN_PREFETCH
      __builtin_amdgcn_sched_barrier(0);
#endif
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      t = mfma_f16(ah, in.hi, t);
      t = mfma_f16(ah, in.lo, t);
      t = mfma_f16(al, in.hi, t);
      p[ob] = t * in.inv;
#ifndef M3G_NO_FWD_CHAIN_PREFETCH
      __builtin_amdgcn_sched_barrier(0);
#endif
      ah = nh; al = nl;
    });
  } else {
    tb_preact<TBS>(tbimg, in.mb, p, lane);
  }
}

// layer-1 accumulators start from the gathered per-node tables TA[i] + TB[j] (x_i / x_j parts, bias folded)
__device__ __forceinline__ void gather_tables(const float* __restrict__ TA, const float* __restrict__ TB, int mlp, int64_t ci,
                                              int64_t cj, int qd, f32x4 (&p1)[8]) {
#ifdef M3G_DIAG_NO_GATHER   // timing diagnostic only (wrong results): what the table gathers cost
  static_for<8>([&]<int ob>() { p1[ob] = f32x4{0.1f, 0.2f, 0.3f, 0.4f}; });
  return;
#endif
  const float* ta = TA + ci * (4 * kDP) + mlp * (2 * kDP) + 4 * qd;
  const float* tb = TB + cj * (4 * kDP) + mlp * (2 * kDP) + 4 * qd;
  static_for<8>([&]<int ob>() { p1[ob] = *(const f32x4*)(ta + ob * 16) + *(const f32x4*)(tb + ob * 16); });
}

struct RevArgs {
  int64_t E, tiles;
  const float* img;     // reverse image of this kernel's MLP (edge image also carries the three-body images)
  const int32_t *src, *dst;
  const float *h, *m, *dx_new;   // m (and dm below): one row per ACTIVE edge (Topo::act_id)
  const int32_t* act_id;
  const float *TA, *TB;   // node tables of this block
  const float* e_tile;    // node kernel: edge features AFTER the block (input of the node MLP);
                          // edge / fused kernel: edge features BEFORE the block (three-body update + edge MLP are recomputed)
  const float* e2_tile;   // fused kernel: edge features AFTER the block
  float* de_soa;   // edge kernel: in dL/d e after this block (unless de_is_zero), out dL/d e before this block
  float* dcn;      // node kernel -> edge kernel: the node MLP's contribution to dL/d e2 (tile-SoA)
  int de_is_zero;  // last block: nothing flows in from later blocks
  float* dm;       // [E][16]   (edge kernel)
  float* dh;       // [E][4] slice of this kernel (store only)
  float* dp1;      // [E][256]  each kernel writes its MLP's 128 columns
  unsigned long long* stamps;  // diagnostic build only
  float *seg_head, *seg_first;   // fused kernel: per-centre sums of the dp1 rows (see seg_scan)
  const float* p1;      // saved layer-1 pre-activations of this block (fp32 mode; SiLU'(p1) when p2 is saved too), else nullptr
  const float* p2;      // saved layer-2 pre-activations (fp32 mode, saves_p2), else nullptr
  float w_inv;          // f16x3 mode: 1 / the model's weight scale (plan->w_scale_inv), else 1
  float* dp1_scale;     // f16x3 fused kernel: [E][4] inverse scales (x 2^-9) of the 24-bit fixed-point dp1 rows (pack24_fixed)
  const int32_t* in_pos;   // fp32 fused kernels: row of edge e in the dp1 array = in_pos[e] (Topo::in_pos); nullptr: row = e
  int split_tail;          // k_edge_rev_f32: the tiles a workgroup cannot share out evenly over its SIMDs go through the four-way split (option "split_tail")
};

// x summed over the four lane quarters (lanes l, l^16, l^32, l^48), result in every lane.  v_permlane16/32_swap
// are VALU moves (gfx950); the ds_bpermute behind __shfl_xor costs an LDS round trip per call.
__device__ __forceinline__ float sum_lane_quarters(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // rows {1,3} of vdst <-> rows {0,2} of src
  const float y = __builtin_bit_cast(float, (unsigned)r16[0]) + __builtin_bit_cast(float, (unsigned)r16[1]);
  const unsigned v = __builtin_bit_cast(unsigned, y);
  auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // lanes 32-63 of vdst <-> lanes 0-31 of src
  return __builtin_bit_cast(float, (unsigned)r32[0]) + __builtin_bit_cast(float, (unsigned)r32[1]);
}

__device__ __forceinline__ void store_dh(float* dh, int64_t edge, int64_t E, f32x4 dhv, int qd) {
  // the four lane quarters hold disjoint feature sets of the same edge: combine, then quarter 0 owns the edge.
  // Store-only into this kernel's slice: a read-modify-write here would queue its load behind the dp1 stores.
  static_for<4>([&]<int rr>() { dhv[rr] = sum_lane_quarters(dhv[rr]); });
  if (qd == 0 && edge < E) *(f32x4*)(dh + edge * kRP) = dhv;
}

// One workgroup per CU for large problems.  Small ones are spread one tile per workgroup over as many CUs as there are tiles
// instead of filling the 16 wave slots of a few CUs: a wave alone on its SIMD finishes a tile about three times sooner than
// four waves sharing the SIMD finish theirs, and a small system's step is the serial latency of its kernels.
inline int grid_for_tiles(int64_t tiles, int /*waves*/ = kWaves) {
#ifdef M3G_GRID_PACKED   // round-1 rule: fill the wave slots of ceil(tiles / waves) CUs
  int64_t wgs = (tiles + kWaves - 1) / kWaves;
#else
  int64_t wgs = tiles;
#endif
  wgs = (wgs + 7) / 8 * 8;
  if (wgs < 8) wgs = 8;
  if (wgs > 256) wgs = 256;
  return (int)wgs;
}

inline int tb_steps_for(int C) { return (C + 3) / 4; }
inline int64_t tiles_for(int64_t E) { return (E + kTileEdges - 1) / kTileEdges; }

#define M3G_TBS_SWITCH(C_, CALL)                     \
  switch (tb_steps_for(C_)) {                        \
    case 1: { constexpr int TBS = 1; CALL; } break;  \
    case 2: { constexpr int TBS = 2; CALL; } break;  \
    case 3: { constexpr int TBS = 3; CALL; } break;  \
    default: { constexpr int TBS = 4; CALL; } break; \
  }

#define M3G_PREC_SWITCH(P_, CALL)                                            \
  if ((P_) == kPrecF32) { constexpr int PREC = kPrecF32; CALL; }             \
  else if ((P_) == kPrecF16x3) { constexpr int PREC = kPrecF16x3; CALL; }    \
  else { constexpr int PREC = kPrecBf16x3; CALL; }

}  // namespace m3g
