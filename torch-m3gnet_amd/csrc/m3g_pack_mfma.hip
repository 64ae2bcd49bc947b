// Host-side packing of the per-block weight images the MFMA edge kernels copy into LDS.
// Layout rules: m3g_internal.h (MfmaFwdLayout / MfmaRevLayout).  Source tensors: the reference's state_dict
// entries of ThreeBodyInteration.gated_mlp (nn/interaction.py:180-185) and of M3GNetConv (nn/conv.py:39-61).
#include <cmath>
#include <cstring>

#include "m3g_dual_chain.h"
#include "m3g_dual_f32.h"
#include "m3g_internal.h"

namespace m3g {

MfmaFwdLayout mfma_fwd_layout() {
  MfmaFwdLayout L{};
  int off = 0;
  auto take = [&](int n) { int r = off; off += n; return r; };
  L.tb = take(8 * kTbSteps * 64);
  for (int m = 0; m < 2; ++m) {
    L.mlp[m].w1c = take(8 * 4 * 4 * 64);
    L.mlp[m].w2d = take(4 * 4 * 4 * 64);
    L.mlp[m].w2g = take(4 * 4 * 4 * 64);
    L.mlp[m].b2 = take(2 * 4 * 64);
    L.mlp[m].wl = take(4 * 1 * 64);
  }
  L.adj = take(4 * 64);
  L.total = off;
  return L;
}

MfmaRevLayout mfma_rev_layout() {
  MfmaRevLayout L{};
  int off = 0;
  auto take = [&](int n) { int r = off; off += n; return r; };
  L.mlp.w1c = take(8 * 4 * 4 * 64);
  L.mlp.w2d = take(4 * 4 * 4 * 64);
  L.mlp.w2g = take(4 * 4 * 4 * 64);
  L.mlp.b2 = take(2 * 4 * 64);
  L.mlp.w2dT = take(4 * 4 * 4 * 64);
  L.mlp.w2gT = take(4 * 4 * 4 * 64);
  L.mlp.w1cT = take(4 * 8 * 4 * 64);
  L.mlp.wl = take(64 * 4);
  L.mlp.total = off;
  L.total_n = off;
  L.tb = take(8 * kTbSteps * 64);
  L.tbT = take(1 * 8 * 4 * 64);
  L.total_e = off;
  L.per_block = L.total_e + L.total_n;
  return L;
}

MfmaRevFusedLayout mfma_rev_fused_layout() {
  MfmaRevFusedLayout L{};
  int off = 0;
  auto take = [&](int n) { int r = off; off += n; return r; };
  L.tb = take(8 * kTbSteps * 64);
  L.tbT = take(1 * 8 * 4 * 64);
  for (int m = 0; m < 2; ++m) {
    L.mlp[m].w1c = take(128 * 64);
    L.mlp[m].w2d = take(64 * 64);
    L.mlp[m].w2g = take(64 * 64);
    L.mlp[m].b2 = take(2 * 4 * 64);
    L.mlp[m].wl = take(64 * 4);
    L.mlp[m].wld = take(4 * 64);
  }
  L.adj = take(4 * 64);
  L.adjp = take(64 * 4);
  L.total = off;
  return L;
}

// bf16 helpers (round to nearest even; weights are finite)
static inline uint16_t bf16_rne(float w) {
  uint32_t u;
  memcpy(&u, &w, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_float(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// chain image for v_mfma_f32_16x16x32_bf16 with the split W = W_hi + W_lo (both bf16; residual exact in fp32):
//   part[((ob*KS + s)*64 + lane)*8 + j] = W[ob*16 + (lane&15)][feat(s, lane>>4, j)],
//   feat(s, qd, j) = (2*s + (j>>2))*16 + 4*qd + (j&3)        (k-step s consumes accumulator blocks 2s and 2s+1)
// The image holds the hi part followed by the lo part; OB*KS*256 floats each, i.e. OB*K*16 floats in total --
// the same size as an fp32 image of the layer.
template <class F>
static void chain_image(float* img, int OB, int KS, F get) {
  uint16_t* hi = reinterpret_cast<uint16_t*>(img);
  uint16_t* lo = hi + (size_t)OB * KS * 512;
  for (int ob = 0; ob < OB; ++ob)
    for (int s = 0; s < KS; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int feat = (2 * s + (j >> 2)) * 16 + 4 * (lane >> 4) + (j & 3);
          const float w = get(ob * 16 + (lane & 15), feat);
          const uint16_t h = bf16_rne(w);
          const size_t idx = (((size_t)ob * KS + s) * 64 + lane) * 8 + j;
          hi[idx] = h;
          lo[idx] = bf16_rne(w - bf16_to_float(h));
        }
}
// fp16 helpers for the f16x3 mode (round to nearest even; inputs are finite and, after scaling, inside the fp16 range)
static inline uint16_t f16_rne(float w) {
  const _Float16 h = (_Float16)w;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static inline float f16_to_float(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}
// the same image for v_mfma_f32_16x16x32_f16: W * scale = W_hi + W_lo, both fp16 (m3g_mfma_common.h, f16x3 mode)
template <class F>
static void chain_image_h(float* img, int OB, int KS, float scale, F get) {
  uint16_t* hi = reinterpret_cast<uint16_t*>(img);
  uint16_t* lo = hi + (size_t)OB * KS * 512;
  for (int ob = 0; ob < OB; ++ob)
    for (int s = 0; s < KS; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int feat = (2 * s + (j >> 2)) * 16 + 4 * (lane >> 4) + (j & 3);
          const float w = get(ob * 16 + (lane & 15), feat) * scale;
          const uint16_t h = f16_rne(w);
          const size_t idx = (((size_t)ob * KS + s) * 64 + lane) * 8 + j;
          hi[idx] = h;
          lo[idx] = f16_rne(w - f16_to_float(h));
        }
}
// three-body MLP image of the f16x3 mode (tb_preact_p, m3g_edge_common.h): one K = 32 k-step whose columns beyond 16 are zero,
// so only lane quarters 0 and 1 are stored: [hi | lo][8 row blocks][32 lanes][8 halves] = 8 KB, the size of the fp32 direct image
template <class F>
static void tb_image_h(float* img, float scale, F get) {
  uint16_t* hi = reinterpret_cast<uint16_t*>(img);
  uint16_t* lo = hi + 8 * 32 * 8;
  for (int ob = 0; ob < 8; ++ob)
    for (int lane = 0; lane < 32; ++lane)
      for (int j = 0; j < 8; ++j) {
        const float w = get(ob * 16 + (lane & 15), 8 * (lane >> 4) + j) * scale;
        const uint16_t h = f16_rne(w);
        const size_t idx = ((size_t)ob * 32 + lane) * 8 + j;
        hi[idx] = h;
        lo[idx] = f16_rne(w - f16_to_float(h));
      }
}
// direct image: img[(ob*S + s)*64 + lane] = get(row = ob*16 + (lane&15), k = 4*s + (lane>>4))
template <class F>
static void direct_image(float* img, int OB, int S, F get) {
  for (int ob = 0; ob < OB; ++ob)
    for (int s = 0; s < S; ++s)
      for (int lane = 0; lane < 64; ++lane) img[(ob * S + s) * 64 + lane] = get(ob * 16 + (lane & 15), 4 * s + (lane >> 4));
}

// exact-fp32 chain image for v_mfma_f32_16x16x4_f32 whose B operand is an accumulator block: k-step s = blk*4 + r consumes
// register r of block blk, i.e. lane quarter q supplies feature blk*16 + 4q + r:
//   img[(ob*S + s)*64 + lane] = get(ob*16 + (lane&15), (s>>2)*16 + 4*(lane>>4) + (s&3)),   S = K/4 k-steps
template <class F>
static void f32_chain_image(float* img, int OB, int S, F get) {
  for (int ob = 0; ob < OB; ++ob)
    for (int s = 0; s < S; ++s)
      for (int lane = 0; lane < 64; ++lane)
        img[(ob * S + s) * 64 + lane] = get(ob * 16 + (lane & 15), (s >> 2) * 16 + 4 * (lane >> 4) + (s & 3));
}

// chain image of a layer in the given precision mode: bf16x3 hi/lo pair or exact fp32 (KS 32-wide k-steps = 8*KS fp32 k-steps)
// (`wscale`: the f16x3 mode's power of two for the weights of this plan; unused in the other modes)
template <class F>
static void chain_image_p(int prec, float wscale, float* img, int OB, int KS, F get) {
  if (prec == kPrecBf16x3) chain_image(img, OB, KS, get);
  else if (prec == kPrecF16x3) chain_image_h(img, OB, KS, wscale, get);
  else f32_chain_image(img, OB, 8 * KS, get);
}

void free_mfma_images(m3g_plan* plan) {
  auto drop = [](float*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
  for (int prec = 0; prec < kNumPrec; ++prec) { drop(plan->d_mfma_fwd[prec]); drop(plan->d_mfma_rev[prec]); drop(plan->d_node_img[prec]); }
  drop(plan->d_mfma_revf);
  drop(plan->d_mfma_revf_h);
  drop(plan->d_mfma_revf32);
  drop(plan->d_readout_img);
  drop(plan->d_readout_img_h);
}

static int upload(float*& dst, const std::vector<float>& host) {
  if (!dst) M3G_HIP_CHECK(hipMalloc((void**)&dst, host.size() * sizeof(float)));
  M3G_HIP_CHECK(hipMemcpy(dst, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  return M3G_OK;
}

// (called by m3g_plan_commit with the device idle; buffers are allocated once per plan and device and overwritten on recommit)
int pack_mfma_images(m3g_plan* plan) {
  const m3g_config& cfg = plan->cfg;
  const int D = cfg.embedding_dim, R = cfg.n_max, C = cfg.l_max * cfg.n_max, B = cfg.num_blocks;
  const MfmaFwdLayout F = mfma_fwd_layout();
  const MfmaRevLayout Rv = mfma_rev_layout();
  const MfmaRevFusedLayout Rf = mfma_rev_fused_layout();
  const MfmaRevF32Layout R32 = mfma_rev_f32_layout();
  std::vector<float> revf((size_t)std::max(B, 1) * Rf.total, 0.f), revf32((size_t)std::max(B, 1) * R32.total, 0.f);
  std::vector<float> revfh((size_t)std::max(B, 1) * Rf.total, 0.f);   // fused reverse kernel, f16x3 mode
  float wscale = 1.f;
  {  // f16x3 mode: ONE power of two for every weight that enters a chain image, putting the largest of them into [2^12, 2^13)
     // (entries down to 2^-14 of it keep both fp16 parts normal; m3g_mfma_common.h)
    float wmax = 0.f;
    for (const auto& kv : plan->params) {
      const std::string& k = kv.first;
      const bool chain_weight = k.find(".concat_") != std::string::npos || k.find(".gated_mlp.") != std::string::npos ||
                                k.find(".linear_sigmoid1.weight") != std::string::npos;
      if (!chain_weight || k.find(".bias") != std::string::npos) continue;
      for (float v : kv.second) wmax = std::max(wmax, std::fabs(v));
    }
    int e = 0;
    if (wmax > 0.f && std::isfinite(wmax)) (void)std::frexp(wmax, &e);   // wmax in [2^(e-1), 2^e)
    wscale = std::ldexp(1.f, 13 - e);
    plan->w_scale_inv = std::ldexp(1.f, e - 13);
  }
  for (int prec = 0; prec < kNumPrec; ++prec) {
  std::vector<float> node((size_t)std::max(B, 1) * kNodeImgFloats, 0.f);
  std::vector<float> fwd((size_t)std::max(B, 1) * F.total, 0.f), rev((size_t)std::max(B, 1) * Rv.per_block, 0.f);
  for (int b = 0; b < B; ++b) {
    float* f = fwd.data() + (size_t)b * F.total;
    float* r = rev.data() + (size_t)b * Rv.per_block;  // [edge-MLP image | node-MLP image]
    const std::string tb = "model." + std::to_string(6 + 2 * b), cv = "model." + std::to_string(7 + 2 * b);
    const float* wd = plan->params.at(tb + ".gated_mlp.dense.0.weight").data();  // [D,C]
    const float* wg = plan->params.at(tb + ".gated_mlp.gate.0.weight").data();
    // rows 0-63 dense, 64-127 gate; k = c
    auto tbw = [&](int row, int k) -> float {
      const float* w = row < 64 ? wd : wg;
      int o = row & 63;
      return (o < D && k < C) ? w[(size_t)o * C + k] : 0.f;
    };
    if (prec == kPrecF16x3) tb_image_h(f + F.tb, wscale, tbw);
    else direct_image(f + F.tb, 8, kTbSteps, tbw);
    direct_image(r + Rv.tb, 8, kTbSteps, tbw);   // (the split reverse kernels keep the exact fp32 three-body products in every mode)
    float* rf = revf.data() + (size_t)b * Rf.total;
    float* rfh = revfh.data() + (size_t)b * Rf.total;
    direct_image(rf + Rf.tb, 8, kTbSteps, tbw);
    tb_image_h(rfh + Rf.tb, wscale, tbw);
    // reverse three-body: rows = c (16), k = 0..127 over (dense f | gate f)
    auto tbT = [&](int row, int k) -> float {
      const float* w = k < 64 ? wd : wg;
      int o = k & 63;
      return (row < C && o < D) ? w[(size_t)o * C + row] : 0.f;
    };
    chain_image_p(prec, wscale, r + Rv.tbT, 1, 4, tbT);
    chain_image(rf + Rf.tbT, 1, 4, tbT);
    chain_image_h(rfh + Rf.tbT, 1, 4, wscale, tbT);
    float* r32 = revf32.data() + (size_t)b * R32.total;   // fused fp32 reverse kernel
    direct_image(r32 + R32.tb, 8, kTbSteps, tbw);
    f32_chain_image(r32 + R32.tbT, 1, 32, tbT);
    {  // edge embedding (nn/featurizer.py:128-132, "model.5.linear.weight" [D,R])
      const float* wadj = plan->params.at("model.5.linear.weight").data();
      auto adj = [&](int row, int k) -> float { return (row < D && k < R) ? wadj[(size_t)row * R + k] : 0.f; };
      direct_image(f + F.adj, 4, 1, adj);
      direct_image(rf + Rf.adj, 4, 1, adj);
      direct_image(rfh + Rf.adj, 4, 1, adj);
      direct_image(r32 + R32.adj, 4, 1, adj);
      for (int o = 0; o < 64; ++o)
        for (int rr = 0; rr < 4; ++rr) rf[Rf.adjp + o * 4 + rr] = rfh[Rf.adjp + o * 4 + rr] = r32[R32.adjp + o * 4 + rr] = adj(o, rr);
    }
    {  // node tables on the matrix pipe (k_node_pre_mfma): rows = table columns, k = node feature
      float* ni = node.data() + (size_t)b * kNodeImgFloats;
      const char* names[2] = {".concat_edge_update", ".concat_node_update"};
      const float* w1[2][2];   // [mlp][dense|gate]  [D, 3D]
      const float* b1[2][2];
      for (int m = 0; m < 2; ++m) {
        w1[m][0] = plan->params.at(cv + names[m] + ".dense.0.weight").data();
        w1[m][1] = plan->params.at(cv + names[m] + ".gate.0.weight").data();
        b1[m][0] = plan->params.at(cv + names[m] + ".dense.0.bias").data();
        b1[m][1] = plan->params.at(cv + names[m] + ".gate.0.bias").data();
      }
      const float* ws = plan->params.at(tb + ".linear_sigmoid1.weight").data();   // [C, D]
      const float* bs = plan->params.at(tb + ".linear_sigmoid1.bias").data();
      auto row_w = [&](int row, int k) -> float {
        if (k >= D) return 0.f;
        if (row < 512) {
          const int part = row / 256, o = row % 256, m = o / 128, oo = o % 128, g = oo / 64, f = oo % 64;   // part 0: x_i (W1a), 1: x_j (W1b)
          return f < D ? w1[m][g][(size_t)f * 3 * D + part * D + k] : 0.f;
        }
        const int cidx = row - 512;
        return cidx < C ? ws[(size_t)cidx * D + k] : 0.f;
      };
      // three bf16x3 chain images of 11 row blocks each (the kernel keeps 11 accumulator blocks per pass)
      for (int g = 0; g < 3; ++g)
        chain_image_p(prec, wscale, ni + (size_t)g * 11 * 2 * 512, 11, 2, [&](int row, int k) -> float { return row_w(g * 176 + row, k); });
      float* bias = ni + kNodeRowBlocks * 16 * 64;
      for (int row = 0; row < kNodeRowBlocks * 16; ++row) {
        float v = 0.f;
        if (row < 256) { const int m = row / 128, oo = row % 128, g = oo / 64, f = oo % 64; v = f < D ? b1[m][g][f] : 0.f; }
        else if (row >= 512) { const int cidx = row - 512; v = cidx < C ? bs[cidx] : 0.f; }
        bias[row] = v;
      }
    }
    const char* mlps[2] = {".concat_edge_update", ".concat_node_update"};
    const char* lins[2] = {".edge_linear.weight", ".node_linear.weight"};
    for (int m = 0; m < 2; ++m) {
      const std::string pre = cv + mlps[m];
      const float* w1d = plan->params.at(pre + ".dense.0.weight").data();  // [D,3D]
      const float* w1g = plan->params.at(pre + ".gate.0.weight").data();
      const float* w2d = plan->params.at(pre + ".dense.2.weight").data();  // [D,D]
      const float* w2g = plan->params.at(pre + ".gate.2.weight").data();
      const float* b2d = plan->params.at(pre + ".dense.2.bias").data();
      const float* b2g = plan->params.at(pre + ".gate.2.bias").data();
      const float* wl = plan->params.at(cv + lins[m]).data();  // [D,R]
      // layer-1 e-part: rows 0-63 dense / 64-127 gate outputs, k = edge feature (columns 2D..3D of W1)
      auto w1c = [&](int row, int k) -> float {
        const float* w = row < 64 ? w1d : w1g;
        int o = row & 63;
        return (o < D && k < D) ? w[(size_t)o * 3 * D + 2 * D + k] : 0.f;
      };
      auto sq = [&](const float* w) { return [=](int row, int k) -> float { return (row < D && k < D) ? w[(size_t)row * D + k] : 0.f; }; };
      auto sqT = [&](const float* w) { return [=](int row, int k) -> float { return (row < D && k < D) ? w[(size_t)k * D + row] : 0.f; }; };
      auto bias_image = [&](float* img) {
        for (int g = 0; g < 2; ++g)
          for (int ob = 0; ob < 4; ++ob)
            for (int lane = 0; lane < 64; ++lane) {
              int o = ob * 16 + lane;
              img[(g * 4 + ob) * 64 + lane] = (lane < 16 && o < D) ? (g == 0 ? b2d[o] : b2g[o]) : 0.f;
            }
      };
      chain_image_p(prec, wscale, f + F.mlp[m].w1c, 8, 2, w1c);
      chain_image_p(prec, wscale, f + F.mlp[m].w2d, 4, 2, sq(w2d));
      chain_image_p(prec, wscale, f + F.mlp[m].w2g, 4, 2, sq(w2g));
      bias_image(f + F.mlp[m].b2);
      direct_image(f + F.mlp[m].wl, 4, 1, [&](int row, int k) -> float { return (row < D && k < R) ? wl[(size_t)row * R + k] : 0.f; });
      // reverse images: m == 0 (edge MLP) at offset 0, m == 1 (node MLP) after the edge image
      float* rm = r + (m == 0 ? 0 : Rv.total_e);
      chain_image_p(prec, wscale, rm + Rv.mlp.w1c, 8, 2, w1c);
      chain_image_p(prec, wscale, rm + Rv.mlp.w2d, 4, 2, sq(w2d));
      chain_image_p(prec, wscale, rm + Rv.mlp.w2g, 4, 2, sq(w2g));
      bias_image(rm + Rv.mlp.b2);
      chain_image_p(prec, wscale, rm + Rv.mlp.w2dT, 4, 2, sqT(w2d));
      chain_image_p(prec, wscale, rm + Rv.mlp.w2gT, 4, 2, sqT(w2g));
      chain_image_p(prec, wscale, rm + Rv.mlp.w1cT, 4, 4, [&](int row, int k) -> float { return w1c(k, row); });
      for (int o = 0; o < 64; ++o)
        for (int rr = 0; rr < 4; ++rr) rm[Rv.mlp.wl + o * 4 + rr] = (o < D && rr < R) ? wl[(size_t)o * R + rr] : 0.f;
      // fused reverse kernel: one dual-use image per matrix
      pack_dual_image(rf + Rf.mlp[m].w1c, 128, w1c);
      pack_dual_image(rf + Rf.mlp[m].w2d, 64, sq(w2d));
      pack_dual_image(rf + Rf.mlp[m].w2g, 64, sq(w2g));
      bias_image(rf + Rf.mlp[m].b2);
      memcpy(rf + Rf.mlp[m].wl, rm + Rv.mlp.wl, sizeof(float) * 64 * 4);
      direct_image(rf + Rf.mlp[m].wld, 4, 1, [&](int row, int k) -> float { return (row < D && k < R) ? wl[(size_t)row * R + k] : 0.f; });
      pack_dual_image_h(rfh + Rf.mlp[m].w1c, 128, wscale, w1c);
      pack_dual_image_h(rfh + Rf.mlp[m].w2d, 64, wscale, sq(w2d));
      pack_dual_image_h(rfh + Rf.mlp[m].w2g, 64, wscale, sq(w2g));
      bias_image(rfh + Rf.mlp[m].b2);
      memcpy(rfh + Rf.mlp[m].wl, rm + Rv.mlp.wl, sizeof(float) * 64 * 4);
      memcpy(rfh + Rf.mlp[m].wld, rf + Rf.mlp[m].wld, sizeof(float) * 4 * 64);
      // fused fp32 reverse kernel: W2 as dual-use fp32 images, W1c transposed only
      pack_dual32_image(r32 + R32.mlp[m].w2d, 64, sq(w2d));
      pack_dual32_image(r32 + R32.mlp[m].w2g, 64, sq(w2g));
      f32_chain_image(r32 + R32.mlp[m].w1cT, 4, 32, [&](int row, int k) -> float { return w1c(k, row); });
      bias_image(r32 + R32.mlp[m].b2);
      memcpy(r32 + R32.mlp[m].wl, rm + Rv.mlp.wl, sizeof(float) * 64 * 4);
      memcpy(r32 + R32.mlp[m].wld, rf + Rf.mlp[m].wld, sizeof(float) * 4 * 64);
    }
  }
  { int rc = upload(plan->d_mfma_fwd[prec], fwd); if (rc) return rc; }
  { int rc = upload(plan->d_mfma_rev[prec], rev); if (rc) return rc; }
  { int rc = upload(plan->d_node_img[prec], node); if (rc) return rc; }
  }   // precision modes
  {  // readout MLP ("model.<6+2B>.gated.*") as chain images
    const std::string ro = "model." + std::to_string(6 + 2 * B) + ".gated.";
    const float* w1d = plan->params.at(ro + "dense.0.weight").data();
    const float* w1g = plan->params.at(ro + "gate.0.weight").data();
    const float* w2d = plan->params.at(ro + "dense.2.weight").data();
    const float* w2g = plan->params.at(ro + "gate.2.weight").data();
    std::vector<float> img(ReadoutImg::total, 0.f);
    auto sq = [&](const float* w) { return [=](int row, int k) -> float { return (row < D && k < D) ? w[(size_t)row * D + k] : 0.f; }; };
    auto sqT = [&](const float* w) { return [=](int row, int k) -> float { return (row < D && k < D) ? w[(size_t)k * D + row] : 0.f; }; };
    auto w1 = [&](int row, int k) -> float { const float* w = row < 64 ? w1d : w1g; const int o = row & 63; return (o < D && k < D) ? w[(size_t)o * D + k] : 0.f; };
    // exact fp32 products: the readout seeds the whole reverse pass, and the reference's absolute-position virial amplifies
    // its rounding (bf16x3 here moved the Cu-32 stress from 4.5e-5 to 1.2e-4 of its fp64 value)
    f32_chain_image(img.data() + ReadoutImg::w1, 8, 16, w1);
    f32_chain_image(img.data() + ReadoutImg::w2d, 4, 16, sq(w2d));
    f32_chain_image(img.data() + ReadoutImg::w2g, 4, 16, sq(w2g));
    f32_chain_image(img.data() + ReadoutImg::w2dT, 4, 16, sqT(w2d));
    f32_chain_image(img.data() + ReadoutImg::w2gT, 4, 16, sqT(w2g));
    f32_chain_image(img.data() + ReadoutImg::w1T, 4, 32, [&](int row, int k) -> float { return w1(k, row); });
    const char* br[2] = {"dense", "gate"};
    for (int g = 0; g < 2; ++g) {
      const float* b1 = plan->params.at(ro + br[g] + ".0.bias").data();
      const float* b2 = plan->params.at(ro + br[g] + ".2.bias").data();
      const float* w3 = plan->params.at(ro + br[g] + ".4.weight").data();
      for (int o = 0; o < D; ++o) {
        img[ReadoutImg::b1 + g * 64 + o] = b1[o];
        img[ReadoutImg::b2 + g * 64 + o] = b2[o];
        img[ReadoutImg::w3 + g * 64 + o] = w3[o];
      }
      img[ReadoutImg::b3 + g] = plan->params.at(ro + br[g] + ".4.bias")[0];
    }
    { int rc = upload(plan->d_readout_img, img); if (rc) return rc; }
    // f16x3 mode: the same six matrices as scaled two-part fp16 images (fp32-grade products, a fifth of the matrix time of
    // the fp32 MFMAs; biases and the final 64 -> 1 weights stay fp32); their own power-of-two scale
    float wmax = 0.f;
    for (const float* w : {w1d, w1g, w2d, w2g})
      for (int i = 0; i < D * D; ++i) wmax = std::max(wmax, std::fabs(w[i]));
    int e = 0;
    if (wmax > 0.f) (void)std::frexp(wmax, &e);
    const float ro_scale = std::ldexp(1.f, 13 - e);
    plan->ro_w_scale_inv = std::ldexp(1.f, e - 13);
    std::vector<float> imh(img);
    chain_image_h(imh.data() + ReadoutImg::w1, 8, 2, ro_scale, w1);
    chain_image_h(imh.data() + ReadoutImg::w2d, 4, 2, ro_scale, sq(w2d));
    chain_image_h(imh.data() + ReadoutImg::w2g, 4, 2, ro_scale, sq(w2g));
    chain_image_h(imh.data() + ReadoutImg::w2dT, 4, 2, ro_scale, sqT(w2d));
    chain_image_h(imh.data() + ReadoutImg::w2gT, 4, 2, ro_scale, sqT(w2g));
    chain_image_h(imh.data() + ReadoutImg::w1T, 4, 4, ro_scale, [&](int row, int k) -> float { return w1(k, row); });
    { int rc = upload(plan->d_readout_img_h, imh); if (rc) return rc; }
  }
  { int rc = upload(plan->d_mfma_revf, revf); if (rc) return rc; }
  { int rc = upload(plan->d_mfma_revf_h, revfh); if (rc) return rc; }
  { int rc = upload(plan->d_mfma_revf32, revf32); if (rc) return rc; }
  return M3G_OK;
}

}  // namespace m3g
