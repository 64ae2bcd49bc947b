// Host-side packing of the per-block weight images the MFMA edge kernels copy into LDS.
// Layout rules: m3g_internal.h (MfmaFwdLayout / MfmaRevLayout).  Source tensors: the reference's state_dict
// entries of ThreeBodyInteration.gated_mlp (nn/interaction.py:180-185) and of M3GNetConv (nn/conv.py:39-61).
#include "m3g_internal.h"

namespace m3g {

static inline int feat_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

MfmaFwdLayout mfma_fwd_layout() {
  MfmaFwdLayout L{};
  int off = 0;
  auto take = [&](int n) { int r = off; off += n; return r; };
  L.tb = take(4 * kTbSteps * 64);
  for (int m = 0; m < 2; ++m) {
    L.mlp[m].w1c = take(4 * 2 * 16 * 64);
    L.mlp[m].w2d = take(2 * 2 * 16 * 64);
    L.mlp[m].w2g = take(2 * 2 * 16 * 64);
    L.mlp[m].b2 = take(2 * 2 * 64);
    L.mlp[m].wl = take(2 * 2 * 64);
  }
  L.total = off;
  return L;
}

MfmaRevLayout mfma_rev_layout() {
  MfmaRevLayout L{};
  int off = 0;
  auto take = [&](int n) { int r = off; off += n; return r; };
  L.tb = take(4 * kTbSteps * 64);
  L.tbT = take(1 * 4 * 16 * 64);
  for (int m = 0; m < 2; ++m) {
    L.mlp[m].w2dT = take(2 * 2 * 16 * 64);
    L.mlp[m].w2gT = take(2 * 2 * 16 * 64);
    L.mlp[m].w1cT = take(2 * 4 * 16 * 64);
    L.mlp[m].wl = take(64 * 4);
  }
  L.total = off;
  return L;
}

// chain image: img[((ob*KB + kb)*16 + s)*64 + lane] = get(row = ob*32 + (lane&31), k = kb*32 + feat_of(s, lane>>5))
template <class F>
static void chain_image(float* img, int OB, int KB, F get) {
  for (int ob = 0; ob < OB; ++ob)
    for (int kb = 0; kb < KB; ++kb)
      for (int s = 0; s < 16; ++s)
        for (int lane = 0; lane < 64; ++lane)
          img[((ob * KB + kb) * 16 + s) * 64 + lane] = get(ob * 32 + (lane & 31), kb * 32 + feat_of(s, lane >> 5));
}
// direct image: img[(ob*S + s)*64 + lane] = get(row = ob*32 + (lane&31), k = 2*s + (lane>>5))
template <class F>
static void direct_image(float* img, int OB, int S, F get) {
  for (int ob = 0; ob < OB; ++ob)
    for (int s = 0; s < S; ++s)
      for (int lane = 0; lane < 64; ++lane) img[(ob * S + s) * 64 + lane] = get(ob * 32 + (lane & 31), 2 * s + (lane >> 5));
}

int pack_mfma_images(m3g_plan* plan) {
  const m3g_config& cfg = plan->cfg;
  const int D = cfg.embedding_dim, R = cfg.n_max, C = cfg.l_max * cfg.n_max, B = cfg.num_blocks;
  const MfmaFwdLayout F = mfma_fwd_layout();
  const MfmaRevLayout Rv = mfma_rev_layout();
  std::vector<float> fwd((size_t)std::max(B, 1) * F.total, 0.f), rev((size_t)std::max(B, 1) * Rv.total, 0.f);
  for (int b = 0; b < B; ++b) {
    float* f = fwd.data() + (size_t)b * F.total;
    float* r = rev.data() + (size_t)b * Rv.total;
    const std::string tb = "model." + std::to_string(6 + 2 * b), cv = "model." + std::to_string(7 + 2 * b);
    const float* wd = plan->params.at(tb + ".gated_mlp.dense.0.weight").data();  // [D,C]
    const float* wg = plan->params.at(tb + ".gated_mlp.gate.0.weight").data();
    // rows 0-63 dense, 64-127 gate; k = c
    auto tbw = [&](int row, int k) -> float {
      const float* w = row < 64 ? wd : wg;
      int o = row & 63;
      return (o < D && k < C) ? w[(size_t)o * C + k] : 0.f;
    };
    direct_image(f + F.tb, 4, kTbSteps, tbw);
    direct_image(r + Rv.tb, 4, kTbSteps, tbw);
    // reverse three-body: rows = c (padded to 32), k = 0..127 over (dense f | gate f)
    chain_image(r + Rv.tbT, 1, 4, [&](int row, int k) -> float {
      const float* w = k < 64 ? wd : wg;
      int o = k & 63;
      return (row < C && o < D) ? w[(size_t)o * C + row] : 0.f;
    });
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
      chain_image(f + F.mlp[m].w1c, 4, 2, w1c);
      chain_image(f + F.mlp[m].w2d, 2, 2, sq(w2d));
      chain_image(f + F.mlp[m].w2g, 2, 2, sq(w2g));
      for (int g = 0; g < 2; ++g)
        for (int ob = 0; ob < 2; ++ob)
          for (int lane = 0; lane < 64; ++lane) {
            int o = ob * 32 + lane;
            f[F.mlp[m].b2 + (g * 2 + ob) * 64 + lane] = (lane < 32 && o < D) ? (g == 0 ? b2d[o] : b2g[o]) : 0.f;
          }
      direct_image(f + F.mlp[m].wl, 2, 2, [&](int row, int k) -> float { return (row < D && k < R) ? wl[(size_t)row * R + k] : 0.f; });
      // reverse images
      chain_image(r + Rv.mlp[m].w2dT, 2, 2, sqT(w2d));
      chain_image(r + Rv.mlp[m].w2gT, 2, 2, sqT(w2g));
      chain_image(r + Rv.mlp[m].w1cT, 2, 4, [&](int row, int k) -> float { return w1c(k, row); });
      for (int o = 0; o < 64; ++o)
        for (int rr = 0; rr < 4; ++rr) r[Rv.mlp[m].wl + o * 4 + rr] = (o < D && rr < R) ? wl[(size_t)o * R + rr] : 0.f;
    }
  }
  if (plan->d_mfma_fwd) { (void)hipFree(plan->d_mfma_fwd); plan->d_mfma_fwd = nullptr; }
  if (plan->d_mfma_rev) { (void)hipFree(plan->d_mfma_rev); plan->d_mfma_rev = nullptr; }
  M3G_HIP_CHECK(hipMalloc((void**)&plan->d_mfma_fwd, fwd.size() * sizeof(float)));
  M3G_HIP_CHECK(hipMalloc((void**)&plan->d_mfma_rev, rev.size() * sizeof(float)));
  M3G_HIP_CHECK(hipMemcpy(plan->d_mfma_fwd, fwd.data(), fwd.size() * sizeof(float), hipMemcpyHostToDevice));
  M3G_HIP_CHECK(hipMemcpy(plan->d_mfma_rev, rev.data(), rev.size() * sizeof(float), hipMemcpyHostToDevice));
  return M3G_OK;
}

}  // namespace m3g
