// C-ABI entry points of libm3gnet_hip.so: plan (weights + constants), workspace carving and the
// orchestration of the energy/force pipeline.  See include/m3gnet_hip.h for the contract.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "m3g_internal.h"

namespace m3g {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- weight layout ---------------------------------------------------------------------------------
static void layout_mlp(MlpW& m, size_t& off) {
  auto take = [&](size_t n) { size_t r = off; off += (n + 63) / 64 * 64; return r; };
  m.w1a_t = take(kDP * 2 * kDP); m.w1b_t = take(kDP * 2 * kDP); m.w1c_t = take(kDP * 2 * kDP);
  m.b1 = take(2 * kDP);
  m.w2d_t = take(kDP * kDP); m.w2g_t = take(kDP * kDP);
  m.b2d = take(kDP); m.b2g = take(kDP);
  m.wl_t = take(kRP * kDP);
  m.w1a = take(2 * kDP * kDP); m.w1b = take(2 * kDP * kDP); m.w1c = take(2 * kDP * kDP);
  m.w2d = take(kDP * kDP); m.w2g = take(kDP * kDP);
  m.wl = take(kDP * kRP);
}

static WeightLayout make_layout(const m3g_config& cfg) {
  WeightLayout wl{};
  size_t off = 0;
  auto take = [&](size_t n) { size_t r = off; off += (n + 63) / 64 * 64; return r; };
  wl.emb = take((size_t)cfg.num_types * kDP);
  wl.adj_t = take(kRP * kDP);
  wl.adj = take(kDP * kRP);
  wl.elemental = take(cfg.num_types);
  for (int b = 0; b < cfg.num_blocks; ++b) {
    BlockW& bw = wl.blk[b];
    bw.tb_w1_t = take(kDP * kCP); bw.tb_b1 = take(kCP); bw.tb_w1 = take(kCP * kDP);
    bw.tb_wd_t = take(kCP * kDP); bw.tb_wg_t = take(kCP * kDP);
    bw.tb_wd = take(kDP * kCP); bw.tb_wg = take(kDP * kCP);
    layout_mlp(bw.e, off);
    layout_mlp(bw.n, off);
  }
  ReadoutW& r = wl.ro;
  r.w1d_t = take(kDP * kDP); r.w1g_t = take(kDP * kDP); r.w2d_t = take(kDP * kDP); r.w2g_t = take(kDP * kDP);
  r.w1d = take(kDP * kDP); r.w1g = take(kDP * kDP); r.w2d = take(kDP * kDP); r.w2g = take(kDP * kDP);
  r.b1d = take(kDP); r.b1g = take(kDP); r.b2d = take(kDP); r.b2g = take(kDP);
  r.w3d = take(kDP); r.w3g = take(kDP); r.b3 = take(2);
  wl.total = off;
  return wl;
}

// expected numel of every state_dict key
static std::map<std::string, int64_t> expected_params(const m3g_config& c) {
  std::map<std::string, int64_t> m;
  const int64_t D = c.embedding_dim, C = (int64_t)c.l_max * c.n_max, R = c.n_max;
  m["model.3.linear.weight"] = D * c.num_types;
  m["model.5.linear.weight"] = D * R;
  char buf[128];
  for (int b = 0; b < c.num_blocks; ++b) {
    int tb = 6 + 2 * b, cv = 7 + 2 * b;
    snprintf(buf, sizeof buf, "model.%d.linear_sigmoid1.weight", tb); m[buf] = C * D;
    snprintf(buf, sizeof buf, "model.%d.linear_sigmoid1.bias", tb); m[buf] = C;
    snprintf(buf, sizeof buf, "model.%d.gated_mlp.dense.0.weight", tb); m[buf] = D * C;
    snprintf(buf, sizeof buf, "model.%d.gated_mlp.gate.0.weight", tb); m[buf] = D * C;
    for (const char* mlp : {"concat_edge_update", "concat_node_update"}) {
      for (const char* br : {"dense", "gate"}) {
        snprintf(buf, sizeof buf, "model.%d.%s.%s.0.weight", cv, mlp, br); m[buf] = D * 3 * D;
        snprintf(buf, sizeof buf, "model.%d.%s.%s.0.bias", cv, mlp, br); m[buf] = D;
        snprintf(buf, sizeof buf, "model.%d.%s.%s.2.weight", cv, mlp, br); m[buf] = D * D;
        snprintf(buf, sizeof buf, "model.%d.%s.%s.2.bias", cv, mlp, br); m[buf] = D;
      }
    }
    snprintf(buf, sizeof buf, "model.%d.edge_linear.weight", cv); m[buf] = D * R;
    snprintf(buf, sizeof buf, "model.%d.node_linear.weight", cv); m[buf] = D * R;
  }
  int ro = 6 + 2 * c.num_blocks;
  for (const char* br : {"dense", "gate"}) {
    snprintf(buf, sizeof buf, "model.%d.gated.%s.0.weight", ro, br); m[buf] = D * D;
    snprintf(buf, sizeof buf, "model.%d.gated.%s.0.bias", ro, br); m[buf] = D;
    snprintf(buf, sizeof buf, "model.%d.gated.%s.2.weight", ro, br); m[buf] = D * D;
    snprintf(buf, sizeof buf, "model.%d.gated.%s.2.bias", ro, br); m[buf] = D;
    snprintf(buf, sizeof buf, "model.%d.gated.%s.4.weight", ro, br); m[buf] = D;
    snprintf(buf, sizeof buf, "model.%d.gated.%s.4.bias", ro, br); m[buf] = 1;
  }
  return m;
}

static std::map<std::string, int64_t> expected_consts(const m3g_config& c) {
  return {{"elemental_energies", c.num_types}, {"em", c.n_max}, {"dm", c.n_max}, {"coeff", c.n_max},
          {"factors", (int64_t)c.l_max * c.n_max}, {"bessel_zeros", (int64_t)c.l_max * c.n_max}};
}

// out[k][o] (ld = ldo) = in[o][k0 + k] for o < rows, k < cols; `in` is [rows][ldi]
static void put_t(std::vector<float>& blob, size_t off, int ldo, const float* in, int rows, int ldi, int k0, int cols,
                  int col_off = 0) {
  for (int o = 0; o < rows; ++o)
    for (int k = 0; k < cols; ++k) blob[off + (size_t)k * ldo + col_off + o] = in[(size_t)o * ldi + k0 + k];
}
// out[o][k] (ld = ldo) = in[o][k0 + k]
static void put_n(std::vector<float>& blob, size_t off, int ldo, const float* in, int rows, int ldi, int k0, int cols,
                  int row_off = 0) {
  for (int o = 0; o < rows; ++o)
    for (int k = 0; k < cols; ++k) blob[off + (size_t)(row_off + o) * ldo + k] = in[(size_t)o * ldi + k0 + k];
}

static void pack_mlp(std::vector<float>& blob, const MlpW& m, const m3g_plan& p, const std::string& pre,
                     const std::string& lin, int D, int R) {
  const float* wd1 = p.params.at(pre + ".dense.0.weight").data();
  const float* wg1 = p.params.at(pre + ".gate.0.weight").data();
  const size_t parts_t[3] = {m.w1a_t, m.w1b_t, m.w1c_t};
  const size_t parts_n[3] = {m.w1a, m.w1b, m.w1c};
  for (int part = 0; part < 3; ++part) {
    put_t(blob, parts_t[part], 2 * kDP, wd1, D, 3 * D, part * D, D, 0);
    put_t(blob, parts_t[part], 2 * kDP, wg1, D, 3 * D, part * D, D, kDP);
    put_n(blob, parts_n[part], kDP, wd1, D, 3 * D, part * D, D, 0);
    put_n(blob, parts_n[part], kDP, wg1, D, 3 * D, part * D, D, kDP);
  }
  const float* bd1 = p.params.at(pre + ".dense.0.bias").data();
  const float* bg1 = p.params.at(pre + ".gate.0.bias").data();
  for (int o = 0; o < D; ++o) { blob[m.b1 + o] = bd1[o]; blob[m.b1 + kDP + o] = bg1[o]; }
  const float* wd2 = p.params.at(pre + ".dense.2.weight").data();
  const float* wg2 = p.params.at(pre + ".gate.2.weight").data();
  put_t(blob, m.w2d_t, kDP, wd2, D, D, 0, D); put_t(blob, m.w2g_t, kDP, wg2, D, D, 0, D);
  put_n(blob, m.w2d, kDP, wd2, D, D, 0, D); put_n(blob, m.w2g, kDP, wg2, D, D, 0, D);
  const float* bd2 = p.params.at(pre + ".dense.2.bias").data();
  const float* bg2 = p.params.at(pre + ".gate.2.bias").data();
  for (int o = 0; o < D; ++o) { blob[m.b2d + o] = bd2[o]; blob[m.b2g + o] = bg2[o]; }
  const float* wl = p.params.at(lin).data();  // [D,R]
  put_t(blob, m.wl_t, kDP, wl, D, R, 0, R);
  put_n(blob, m.wl, kRP, wl, D, R, 0, R);
}

Work work_carve(const Consts& c, bool mfma, int save_acts, int64_t N, int64_t E, int64_t T, int64_t S, void* base) {
  (void)T;
  Work w{};
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t n_floats) { float* r = p ? (float*)(p + off) : nullptr; off += align_up(n_floats * sizeof(float)); return r; };
  size_t e = (size_t)E, n = (size_t)N;
  w.u = take(e * 3); w.d = take(e); w.h = take(e * kRP); w.hp = take(e * kRP);
  w.q = take(e * kCP); w.qp = take(e * kCP); w.fc3 = take(e); w.fc3p = take(e);
  for (int b = 0; b <= c.B; ++b) w.x[b] = take(n * kDP);
  for (int b = 0; b < c.B; ++b) w.v[b] = take(n * kCP);
  for (int b = 0; b < c.B; ++b) w.m[b] = take(e * kCP);
  w.dx = take(n * kDP); w.dx2 = take(n * kDP);
  w.dm = take(e * kCP); w.g = take(e * kCP); w.dg = take(e * kCP);
  w.dh = take(e * kRP); w.dd = take(e); w.du = take(e * 3); w.dp1 = take(e * 4 * kDP); w.dr = take(e * 3);
  const size_t tiles16 = (e + 15) / 16;
  if (mfma) {
    // fused MFMA path: per-block edge-feature images and node tables; the reverse pass recomputes every activation
    for (int b = 0; b <= c.B; ++b) w.e_blk[b] = take(tiles16 * 1024);
    if (save_acts >= 1) for (int b = 0; b < c.B; ++b) w.p1_blk[b] = take(tiles16 * 2 * 2048);
    if (save_acts >= 2) for (int b = 0; b < c.B; ++b) w.p2_blk[b] = take(tiles16 * 2 * 2048);
    for (int b = 0; b < c.B; ++b) { w.TAb[b] = take(n * 4 * kDP); w.TBb[b] = take(n * 4 * kDP); }
    w.de_soa = take(tiles16 * 1024);
    w.dcn = take(tiles16 * 1024);
    w.dh_parts = take((size_t)(2 * c.B + 1) * e * kRP);
    w.seg_head = take((tiles16 + 1) * 4 * kDP);
    w.seg_first = take((n + 1) * 4 * kDP);
  } else {
    // vector-ALU baseline path: in-place row-major edge features and saved pre-activations [E,512] per block
    w.TA = take(n * 4 * kDP); w.TB = take(n * 4 * kDP);
    w.e = take(e * kDP);
    w.de = take(e * kDP);
    for (int b = 0; b < c.B; ++b) w.act[b] = take(e * 8 * kDP);
  }
  // tail scratch for optional outputs the caller did not ask for ([N] per-atom energies, [2 S] sums), then the step's sync words
  w.sync = p ? (int32_t*)(p + off + (n + (size_t)S * 2) * sizeof(float)) : nullptr;
  static_assert(kSyncWords <= 64, "the tail scratch reserves 64 words");
  off += align_up((n + (size_t)S * 2 + 64) * sizeof(float));
  w.total_bytes = off;
  return w;
}

}  // namespace m3g

using namespace m3g;

static void drop_graphs(const m3g_plan* plan);

// ---- stage profiler ---------------------------------------------------------------------------------
enum StageId { ST_GEOM = 0, ST_EMBED, ST_NODE_PRE, ST_THREEBODY, ST_EDGE_FWD, ST_NODE_SUM, ST_READOUT, ST_OUTPUTS, ST_EDGE_REV_NODE,
               ST_EDGE_REV, ST_THREEBODY_REV, ST_NODE_REV, ST_EMBED_REV, ST_GEOM_REV, ST_EDGE_REV_FUSED, ST_COUNT };
// edge_block_fwd / edge_rev_node_mlp / edge_rev_edge_mlp each time exactly ONE kernel launch (the MFMA kernels)
static const char* kStageNames[ST_COUNT] = {"geometry_basis", "embed", "node_pre", "threebody_fwd", "edge_block_fwd", "node_sum",
                                            "readout", "optional_outputs", "edge_rev_node_mlp", "edge_rev_edge_mlp", "threebody_rev",
                                            "node_rev", "embed_rev", "geometry_rev_forces", "edge_rev_fused"};
struct StageTimer {
  const m3g_plan* p;
  hipStream_t s;
  size_t pair = (size_t)-1;
  StageTimer(const m3g_plan* plan, int stage, hipStream_t stream) : p(plan), s(stream) {
    if (!p->profile) return;
    if (p->ev_used * 2 + 2 > p->ev_pool.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      p->ev_pool.push_back(a);
      p->ev_pool.push_back(b);
      p->ev_stage.push_back(stage);
    }
    pair = p->ev_used++;
    p->ev_stage[pair] = stage;
    (void)hipEventRecord(p->ev_pool[2 * pair], s);
  }
  ~StageTimer() {
    if (pair != (size_t)-1) (void)hipEventRecord(p->ev_pool[2 * pair + 1], s);
  }
};
#define M3G_STAGE(id) StageTimer _st_##id(plan, id, s)

extern "C" int m3g_profile_enable(m3g_plan* plan, int32_t enable) {
  if (!plan) { set_error("m3g_profile_enable: null plan"); return M3G_ERR_VALUE; }
  plan->profile = enable != 0;
  plan->ev_used = 0;
  return M3G_OK;
}

extern "C" int m3g_profile_read(m3g_plan* plan, int32_t* n_stages, const char** names, float* total_ms, int32_t* launches) {
  if (!plan || !n_stages || !names || !total_ms || !launches) { set_error("m3g_profile_read: null argument"); return M3G_ERR_VALUE; }
  static_assert(ST_COUNT <= M3G_MAX_STAGES, "stage table too large");
  *n_stages = ST_COUNT;
  for (int i = 0; i < ST_COUNT; ++i) { names[i] = kStageNames[i]; total_ms[i] = 0.f; launches[i] = 0; }
  for (size_t k = 0; k < plan->ev_used; ++k) {
    M3G_HIP_CHECK(hipEventSynchronize(plan->ev_pool[2 * k + 1]));
    float ms = 0.f;
    M3G_HIP_CHECK(hipEventElapsedTime(&ms, plan->ev_pool[2 * k], plan->ev_pool[2 * k + 1]));
    total_ms[plan->ev_stage[k]] += ms;
    launches[plan->ev_stage[k]] += 1;
  }
  plan->ev_used = 0;
  return M3G_OK;
}

extern "C" const char* m3g_last_error(void) { return g_err; }

extern "C" int m3g_get_info(m3g_info* out) {
  if (!out) return M3G_ERR_VALUE;
  memset(out, 0, sizeof(*out));
  out->abi_version = M3G_ABI_VERSION;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
  out->device_count = n;
  if (n > 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      snprintf(out->arch, sizeof(out->arch), "%s", prop.gcnArchName);
  }
  return M3G_OK;
}

extern "C" int m3g_plan_create(const m3g_config* cfg, m3g_plan** out) {
  if (!cfg || !out) { set_error("m3g_plan_create: null argument"); return M3G_ERR_VALUE; }
  // the reference's Bessel-root table is 10x10 and needs row l_max (nn/interaction.py:250-253)
  if (cfg->l_max + 1 > 10) { set_error("Too large l_max is specified."); return M3G_ERR_VALUE; }
  if (cfg->n_max > 10) { set_error("Too large n_max is specified."); return M3G_ERR_VALUE; }
  if (cfg->l_max < 1 || cfg->n_max < 1 || cfg->num_types < 1 || cfg->embedding_dim < 1 || cfg->num_blocks < 0) {
    set_error("m3g_plan_create: non-positive hyper-parameter");
    return M3G_ERR_VALUE;
  }
  if (cfg->threebody_cutoff > cfg->cutoff) {  // data/material_graph.py:149-150
    set_error("Three body cutoff raidus should be smaller than two body.");
    return M3G_ERR_VALUE;
  }
  if (cfg->num_blocks > 32 || cfg->embedding_dim > 4096) {
    set_error("unsupported size: this build handles num_blocks <= 32 and embedding_dim <= 4096");
    return M3G_ERR_UNSUPPORTED;
  }
  m3g_plan* p = new m3g_plan();
  p->cfg = *cfg;
  // sizes beyond the tiles of the MFMA kernels run on the any-size path (m3g_generic.hip)
  p->generic = cfg->l_max > kLCap || cfg->n_max > kRCap || cfg->embedding_dim > kDP || cfg->num_blocks > kMaxBlocks;
  if (const char* env = getenv("M3G_EDGE_KERNEL")) p->edge_kernel = atoi(env) != 0 ? 1 : 0;
  if (!p->generic) p->wl = make_layout(*cfg);
  *out = p;
  return M3G_OK;
}

// Streams and events are bound to the device they were created under: the internal side stream (options graph_replay / overlap),
// its fork / join events and the stage profiler's event pool.  Called with that device current; everything is recreated lazily
// (ensure_side_stream, StageTimer) under whatever device the plan lives on next.
static void release_device_handles(m3g_plan* plan) {
  for (hipEvent_t ev : plan->ev_pool) (void)hipEventDestroy(ev);
  plan->ev_pool.clear();
  plan->ev_stage.clear();
  plan->ev_used = 0;
  if (plan->ev_fork) { (void)hipEventDestroy(plan->ev_fork); plan->ev_fork = nullptr; }
  if (plan->ev_join) { (void)hipEventDestroy(plan->ev_join); plan->ev_join = nullptr; }
  if (plan->side_stream) { (void)hipStreamDestroy(plan->side_stream); plan->side_stream = nullptr; }
}

extern "C" int m3g_debug_live_handles(const m3g_plan* plan, int32_t* out) {
  if (!plan || !out) { set_error("m3g_debug_live_handles: null argument"); return M3G_ERR_VALUE; }
  *out = (int32_t)plan->ev_pool.size() + (plan->ev_fork ? 1 : 0) + (plan->ev_join ? 1 : 0) + (plan->side_stream ? 1 : 0);
  return M3G_OK;
}

extern "C" void m3g_plan_destroy(m3g_plan* plan) {
  if (!plan) return;
  if (plan->d_weights) (void)hipFree(plan->d_weights);
  free_mfma_images(plan);
  generic_free(plan);
  if (plan->d_stamps) (void)hipFree(plan->d_stamps);
  drop_graphs(plan);
  release_device_handles(plan);
  delete plan;
}

extern "C" int m3g_plan_set_param(m3g_plan* plan, const char* key, const float* host_data, int64_t numel) {
  if (!plan || !key || !host_data) { set_error("m3g_plan_set_param: null argument"); return M3G_ERR_VALUE; }
  auto exp = expected_params(plan->cfg);
  auto it = exp.find(key);
  if (it == exp.end()) { set_error("unknown parameter key '%s'", key); return M3G_ERR_VALUE; }
  if (it->second != numel) { set_error("parameter '%s': expected %lld values, got %lld", key, (long long)it->second, (long long)numel); return M3G_ERR_VALUE; }
  plan->params[key].assign(host_data, host_data + numel);
  plan->committed = false;
  return M3G_OK;
}

extern "C" int m3g_plan_set_const(m3g_plan* plan, const char* name, const float* host_data, int64_t numel) {
  if (!plan || !name || !host_data) { set_error("m3g_plan_set_const: null argument"); return M3G_ERR_VALUE; }
  auto exp = expected_consts(plan->cfg);
  auto it = exp.find(name);
  if (it == exp.end()) { set_error("unknown constant '%s'", name); return M3G_ERR_VALUE; }
  if (it->second != numel) { set_error("constant '%s': expected %lld values, got %lld", name, (long long)it->second, (long long)numel); return M3G_ERR_VALUE; }
  plan->cvals[name].assign(host_data, host_data + numel);
  plan->committed = false;
  return M3G_OK;
}

extern "C" int m3g_plan_set_option(m3g_plan* plan, const char* name, int32_t value) {
  if (!plan || !name) { set_error("m3g_plan_set_option: null argument"); return M3G_ERR_VALUE; }
  if (strcmp(name, "edge_kernel") == 0) {
    if (value < 0 || value > 2) { set_error("edge_kernel must be 0 (VALU baseline), 1 (MFMA) or 2 (any-size path)"); return M3G_ERR_VALUE; }
    if (plan->generic && value != 2) { set_error("this model size only runs on the any-size path (edge_kernel = 2)"); return M3G_ERR_UNSUPPORTED; }
    plan->edge_kernel = value;
    return M3G_OK;
  }
  if (strcmp(name, "precision") == 0) {
    if (value != kPrecF32 && value != kPrecBf16x3 && value != kPrecF16x3) {
      set_error("precision must be 0 (fp32: exact fp32 MFMA products), 1 (bf16x3 split products) or 2 (f16x3: scaled fp16 split products)");
      return M3G_ERR_VALUE;
    }
    plan->precision = value;   // both image sets are resident: no recommit needed
    return M3G_OK;
  }
  if (strcmp(name, "save_p1") == 0) {
    plan->save_p1 = value != 0;
    return M3G_OK;
  }
  if (strcmp(name, "save_p2") == 0) {
    plan->save_p2 = value != 0;
    return M3G_OK;
  }
  if (strcmp(name, "rev_kernel") == 0) {
    if (value != 0 && value != 1) { set_error("rev_kernel must be 0 (node-MLP + edge-MLP kernel pair) or 1 (fused)"); return M3G_ERR_VALUE; }
    plan->rev_kernel = value;
    return M3G_OK;
  }
  if (strcmp(name, "small_tiles") == 0) {   // graphs of at most this many 16-edge tiles take the split-tile edge kernels (m3g_edge_small.hip); 0: never
    if (value < 0) { set_error("small_tiles must be >= 0"); return M3G_ERR_VALUE; }
    plan->small_tiles = value;
    plan->small_tiles_fwd = value;   // (small_tiles_fwd, set afterwards, moves the forward kernel's threshold alone)
    drop_graphs(plan);
    return M3G_OK;
  }
  if (strcmp(name, "small_tiles_fwd") == 0) {   // the forward kernel's threshold alone (set after small_tiles)
    if (value < 0) { set_error("small_tiles_fwd must be >= 0"); return M3G_ERR_VALUE; }
    plan->small_tiles_fwd = value;
    drop_graphs(plan);
    return M3G_OK;
  }
  if (strcmp(name, "dp1_by_dst") == 0) {   // 0: dp1 rows of the exact-fp32 fused path in edge order (A/B tests; bit-identical either way)
    plan->dp1_by_dst = value != 0;
    drop_graphs(plan);
    return M3G_OK;
  }
  if (strcmp(name, "split_node_tiles") == 0) {
    if (value < 0) { set_error("split_node_tiles must be >= 0"); return M3G_ERR_VALUE; }
    plan->split_node_tiles = value;
    drop_graphs(plan);
    return M3G_OK;
  }
  if (strcmp(name, "fuse_node_tb") == 0) {   // 0: three-body reverse and node reverse as two launches (A/B tests; bit-identical either way)
    plan->fuse_node_tb = value != 0;
    drop_graphs(plan);
    return M3G_OK;
  }
  if (strcmp(name, "debug_node_tb_polls") == 0) {   // test hook: bound (k > 0) or force (k < 0) the time-out of k_node_tb_reverse's in-launch wait
    plan->debug_node_tb_polls = value;
    drop_graphs(plan);
    return M3G_OK;
  }
  if (strcmp(name, "split_tail") == 0) {   // 0: the persistent reverse kernel runs every tile whole (A/B tests; forward outputs bit-identical either way)
    if (value < 0 || value > 2) { set_error("split_tail must be 0 (never), 1 (a single left-over tile) or 2 (one or two)"); return M3G_ERR_VALUE; }
    plan->split_tail = value;
    drop_graphs(plan);
    return M3G_OK;
  }
  if (strcmp(name, "small_launches") == 0) {   // 0: never fuse the small-system launches (A/B tests; results are bit-identical either way)
    plan->small_launches = value != 0;
    drop_graphs(plan);
    return M3G_OK;
  }
  if (strcmp(name, "graph_replay") == 0) {
    plan->graph_replay = value != 0;
    if (!plan->graph_replay) drop_graphs(plan);
    return M3G_OK;
  }
  if (strcmp(name, "overlap") == 0) {
    plan->overlap = value != 0;
    return M3G_OK;
  }
  if (strcmp(name, "threebody_moments") == 0) {
    plan->tb_moments = value != 0;
    return M3G_OK;
  }
  if (strcmp(name, "legendre_backward") == 0) {
    if (value != 0 && value != 1) { set_error("legendre_backward: 0 (exact derivative) or 1 (the reference's backward)"); return M3G_ERR_VALUE; }
    plan->legendre_ref = value != 0;
    drop_graphs(plan);
    return M3G_OK;
  }
  if (strcmp(name, "readout_f16") == 0) {
    plan->readout_f16 = value != 0;
    return M3G_OK;
  }
  if (strcmp(name, "stress_mode") == 0) {
    if (value != 0 && value != 1) { set_error("stress_mode must be 0 (reference: sum pos (x) F / V) or 1 (pair virial)"); return M3G_ERR_VALUE; }
    plan->stress_mode = value;
    return M3G_OK;
  }
  if (strcmp(name, "debug_force_move") == 0) {   // test hook: the next commit takes the device-move path although the device is the same
    plan->debug_force_move = value != 0;
    plan->committed = false;
    return M3G_OK;
  }
  if (strcmp(name, "stamps") == 0) {  // diagnostic: forward edge kernel with s_memtime phase stamps
    plan->stamp_target = value == 3 ? 2 : value == 2 ? 1 : 0;   // 1: forward edge kernel, 2: reverse edge-MLP kernel, 3: fused reverse kernel (f16x3)
    if (value && !plan->d_stamps) {
      M3G_HIP_CHECK(hipMalloc((void**)&plan->d_stamps, 256 * 16 * 12 * sizeof(unsigned long long)));
      M3G_HIP_CHECK(hipMemset(plan->d_stamps, 0, 256 * 16 * 12 * sizeof(unsigned long long)));
    } else if (!value && plan->d_stamps) {
      (void)hipFree(plan->d_stamps);
      plan->d_stamps = nullptr;
    }
    return M3G_OK;
  }
  set_error("unknown option '%s'", name);
  return M3G_ERR_VALUE;
}

extern "C" int m3g_debug_read_stamps(m3g_plan* plan, uint64_t* host_out /* [256*16*12] */) {
  if (!plan || !host_out || !plan->d_stamps) { set_error("stamps not enabled"); return M3G_ERR_STATE; }
  M3G_HIP_CHECK(hipMemcpy(host_out, plan->d_stamps, 256 * 16 * 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return M3G_OK;
}

extern "C" int m3g_plan_commit(m3g_plan* plan) {
  if (!plan) { set_error("m3g_plan_commit: null plan"); return M3G_ERR_VALUE; }
  drop_graphs(plan);   // captured launch sequences point at the buffers this call replaces
  const m3g_config& cfg = plan->cfg;
  for (auto& kv : expected_params(cfg))
    if (!plan->params.count(kv.first)) { set_error("parameter '%s' was never set", kv.first.c_str()); return M3G_ERR_STATE; }
  for (auto& kv : expected_consts(cfg))
    if (!plan->cvals.count(kv.first)) { set_error("constant '%s' was never set", kv.first.c_str()); return M3G_ERR_STATE; }
  const int D = cfg.embedding_dim, R = cfg.n_max, L = cfg.l_max, C = L * R, B = cfg.num_blocks;
  {
    int dev = 0;
    M3G_HIP_CHECK(hipGetDevice(&dev));
    if (plan->device >= 0 && (plan->device != dev || plan->debug_force_move)) {   // the plan moves to the current device
      M3G_HIP_CHECK(hipSetDevice(plan->device));
      M3G_HIP_CHECK(hipDeviceSynchronize());
      if (plan->d_weights) { (void)hipFree(plan->d_weights); plan->d_weights = nullptr; }
      free_mfma_images(plan);
      generic_free(plan);
      if (plan->d_stamps) { (void)hipFree(plan->d_stamps); plan->d_stamps = nullptr; }
      release_device_handles(plan);   // the side stream, its events and the profiler's events belong to the old device too
      M3G_HIP_CHECK(hipSetDevice(dev));
      plan->debug_force_move = false;
    }
    M3G_HIP_CHECK(hipDeviceSynchronize());   // kernels of earlier calls may still read the buffers this call overwrites
    plan->device = dev;
  }
  { int rc = generic_commit(plan); if (rc) return rc; }
  if (plan->generic) {   // no padded blob, no MFMA images: the any-size path reads the raw tensors
    plan->edge_kernel = 2;
    plan->committed = true;
    return M3G_OK;
  }

  // ---- constants (fp32 arithmetic in the reference's order; see oracle make_constants/radial_basis) ----
  Consts& c = plan->consts;
  memset(&c, 0, sizeof(c));
  c.L = L; c.R = R; c.C = C; c.D = D; c.B = B; c.num_types = cfg.num_types;
  c.length_scale = (float)cfg.length_scale;
  c.energy_scale = (float)cfg.energy_scale;
  c.inv_len = 1.f / c.length_scale;
  const double rc = cfg.cutoff / cfg.length_scale, rc3 = cfg.threebody_cutoff / cfg.length_scale;  // model/build.py:34-35
  c.rc = (float)rc;
  c.rc3 = (float)rc3;
  const float pi_f = (float)M_PI;
  const auto& em = plan->cvals.at("em");
  const auto& dm = plan->cvals.at("dm");
  for (int m = 0; m < R; ++m) {
    c.a1[m] = ((float)(m + 1) * pi_f) / (float)rc;  // nn/featurizer.py:87-88
    c.a2[m] = ((float)(m + 2) * pi_f) / (float)rc;
    c.coeff[m] = plan->cvals.at("coeff")[m];
    c.rec_mul[m] = m > 0 ? sqrtf(em[m] / dm[m - 1]) : 0.f;  // nn/featurizer.py:94-96
    c.rec_div[m] = sqrtf(dm[m]);
  }
  for (int l = 0; l < L; ++l) {
    c.ynorm[l] = (float)std::sqrt((2 * l + 1) / (4.0 * M_PI));  // nn/interaction.py:198
    for (int n = 0; n < R; ++n) {
      c.zeros[l][n] = plan->cvals.at("bessel_zeros")[l * R + n];
      c.factors[l][n] = plan->cvals.at("factors")[l * R + n];
    }
  }

  // ---- weights ------------------------------------------------------------------------------------
  const WeightLayout& wl = plan->wl;
  std::vector<float> blob(wl.total, 0.f);
  put_t(blob, wl.emb, kDP, plan->params.at("model.3.linear.weight").data(), D, cfg.num_types, 0, cfg.num_types);
  put_t(blob, wl.adj_t, kDP, plan->params.at("model.5.linear.weight").data(), D, R, 0, R);
  put_n(blob, wl.adj, kRP, plan->params.at("model.5.linear.weight").data(), D, R, 0, R);
  for (int t = 0; t < cfg.num_types; ++t) blob[wl.elemental + t] = plan->cvals.at("elemental_energies")[t];
  char buf[128];
  for (int b = 0; b < B; ++b) {
    const BlockW& bw = wl.blk[b];
    std::string tb = "model." + std::to_string(6 + 2 * b), cv = "model." + std::to_string(7 + 2 * b);
    const float* w1 = plan->params.at(tb + ".linear_sigmoid1.weight").data();  // [C,D]
    put_t(blob, bw.tb_w1_t, kCP, w1, C, D, 0, D);
    put_n(blob, bw.tb_w1, kDP, w1, C, D, 0, D);
    for (int cc = 0; cc < C; ++cc) blob[bw.tb_b1 + cc] = plan->params.at(tb + ".linear_sigmoid1.bias")[cc];
    const float* wd = plan->params.at(tb + ".gated_mlp.dense.0.weight").data();  // [D,C]
    const float* wg = plan->params.at(tb + ".gated_mlp.gate.0.weight").data();
    put_t(blob, bw.tb_wd_t, kDP, wd, D, C, 0, C); put_t(blob, bw.tb_wg_t, kDP, wg, D, C, 0, C);
    put_n(blob, bw.tb_wd, kCP, wd, D, C, 0, C); put_n(blob, bw.tb_wg, kCP, wg, D, C, 0, C);
    pack_mlp(blob, bw.e, *plan, cv + ".concat_edge_update", cv + ".edge_linear.weight", D, R);
    pack_mlp(blob, bw.n, *plan, cv + ".concat_node_update", cv + ".node_linear.weight", D, R);
  }
  {
    const ReadoutW& r = wl.ro;
    snprintf(buf, sizeof buf, "model.%d.gated", 6 + 2 * B);
    std::string ro = buf;
    const float* wd0 = plan->params.at(ro + ".dense.0.weight").data();
    const float* wg0 = plan->params.at(ro + ".gate.0.weight").data();
    const float* wd2 = plan->params.at(ro + ".dense.2.weight").data();
    const float* wg2 = plan->params.at(ro + ".gate.2.weight").data();
    put_t(blob, r.w1d_t, kDP, wd0, D, D, 0, D); put_t(blob, r.w1g_t, kDP, wg0, D, D, 0, D);
    put_t(blob, r.w2d_t, kDP, wd2, D, D, 0, D); put_t(blob, r.w2g_t, kDP, wg2, D, D, 0, D);
    put_n(blob, r.w1d, kDP, wd0, D, D, 0, D); put_n(blob, r.w1g, kDP, wg0, D, D, 0, D);
    put_n(blob, r.w2d, kDP, wd2, D, D, 0, D); put_n(blob, r.w2g, kDP, wg2, D, D, 0, D);
    for (int o = 0; o < D; ++o) {
      blob[r.b1d + o] = plan->params.at(ro + ".dense.0.bias")[o];
      blob[r.b1g + o] = plan->params.at(ro + ".gate.0.bias")[o];
      blob[r.b2d + o] = plan->params.at(ro + ".dense.2.bias")[o];
      blob[r.b2g + o] = plan->params.at(ro + ".gate.2.bias")[o];
      blob[r.w3d + o] = plan->params.at(ro + ".dense.4.weight")[o];
      blob[r.w3g + o] = plan->params.at(ro + ".gate.4.weight")[o];
    }
    blob[r.b3] = plan->params.at(ro + ".dense.4.bias")[0];
    blob[r.b3 + 1] = plan->params.at(ro + ".gate.4.bias")[0];
  }
  if (!plan->d_weights) M3G_HIP_CHECK(hipMalloc((void**)&plan->d_weights, wl.total * sizeof(float)));
  M3G_HIP_CHECK(hipMemcpy(plan->d_weights, blob.data(), wl.total * sizeof(float), hipMemcpyHostToDevice));
  { int rc = pack_mfma_images(plan); if (rc) return rc; }
  plan->committed = true;
  return M3G_OK;
}

extern "C" int m3g_workspace_bytes(const m3g_plan* plan, int64_t N, int64_t E, int64_t T, int64_t S, size_t* bytes) {
  if (!plan || !bytes || N < 0 || E < 0 || T < 0 || S < 0) { set_error("m3g_workspace_bytes: bad argument"); return M3G_ERR_VALUE; }
  if (plan->edge_kernel == 2) { *bytes = generic_workspace_bytes(plan, N, E, T, S); return M3G_OK; }
  Consts c{};
  c.B = plan->cfg.num_blocks;
  *bytes = work_carve(c, plan->edge_kernel == 1, saved_activations(plan), N, E, T, S, nullptr).total_bytes;
  return M3G_OK;
}

static bool ensure_side_stream(const m3g_plan* plan) {
  if (plan->side_stream) return true;
  if (hipStreamCreateWithFlags(&plan->side_stream, hipStreamNonBlocking) != hipSuccess) { plan->side_stream = nullptr; return false; }
  if (hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&plan->ev_join, hipEventDisableTiming) != hipSuccess)
    return false;
  return true;
}

static void drop_graphs(const m3g_plan* plan) {
  for (auto& g : plan->graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  plan->graphs.clear();
}

// replay path of m3g_energy_forces: launches a cached graph, or captures one around the normal enqueue code.
// The legacy default stream cannot be captured: calls on it run the graph on an internal stream, ordered after the
// caller's stream and before its later work by a fork/join pair of events.
static int energy_forces_graph(const m3g_plan* plan, const m3g_io* io, void* workspace, size_t workspace_bytes, hipStream_t caller) {
  hipStream_t s = caller;
  const bool via_side = caller == nullptr;
  if (via_side) {
    if (!ensure_side_stream(plan)) { set_error("graph_replay: could not create the internal stream"); return M3G_ERR_HIP; }
    s = plan->side_stream;
    M3G_HIP_CHECK(hipEventRecord(plan->ev_fork, caller));
    M3G_HIP_CHECK(hipStreamWaitEvent(s, plan->ev_fork, 0));
  }
  auto join = [&]() -> int {
    if (via_side) {
      M3G_HIP_CHECK(hipEventRecord(plan->ev_join, s));
      M3G_HIP_CHECK(hipStreamWaitEvent(caller, plan->ev_join, 0));
    }
    return M3G_OK;
  };
  std::vector<unsigned char> key(sizeof(m3g_io) + sizeof(void*) * 2 + sizeof(size_t) + 4 * sizeof(int));
  unsigned char* k = key.data();
  memcpy(k, io, sizeof(m3g_io)); k += sizeof(m3g_io);
  memcpy(k, &workspace, sizeof(void*)); k += sizeof(void*);
  memcpy(k, &s, sizeof(void*)); k += sizeof(void*);
  memcpy(k, &workspace_bytes, sizeof(size_t)); k += sizeof(size_t);
  const int opts[4] = {plan->edge_kernel, plan->rev_kernel + 2 * plan->save_p1 + 4 * plan->save_p2 + 8 * plan->tb_moments + 16 * plan->precision + 64 * plan->readout_f16, plan->stress_mode, plan->overlap};
  memcpy(k, opts, sizeof(opts));
  for (auto& g : plan->graphs)
    if (g.key == key) { M3G_HIP_CHECK(hipGraphLaunch(g.exec, s)); return join(); }
  if (plan->graphs.size() >= 8) drop_graphs(plan);   // bounded cache
  m3g_plan::GraphEntry ge;
  ge.key = key;
  M3G_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  plan->capturing = true;
  const int rc = m3g_energy_forces(plan, io, workspace, workspace_bytes, (void*)s);
  plan->capturing = false;
  hipError_t e = hipStreamEndCapture(s, &ge.graph);
  if (rc != M3G_OK) { if (ge.graph) (void)hipGraphDestroy(ge.graph); return rc; }
  if (e != hipSuccess || !ge.graph) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return M3G_ERR_HIP; }
  e = hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(ge.graph); set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e)); return M3G_ERR_HIP; }
  plan->graphs.push_back(ge);
  M3G_HIP_CHECK(hipGraphLaunch(ge.exec, s));
  return join();
}

// m3g_count_launches: the call's launch sequence captured (not executed) on a stream of its own, nodes counted by type
extern "C" int m3g_count_launches(const m3g_plan* plan, const m3g_io* io, void* workspace, size_t workspace_bytes, int32_t* kernel_launches,
                                  int32_t* other_operations) {
  if (!plan || !io || !kernel_launches) { set_error("m3g_count_launches: null argument"); return M3G_ERR_VALUE; }
  if (plan->capturing) { set_error("m3g_count_launches: a capture is in progress"); return M3G_ERR_STATE; }
  hipStream_t s = nullptr;
  M3G_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { (void)hipStreamDestroy(s); set_error("hipStreamBeginCapture failed: %s", hipGetErrorString(e)); return M3G_ERR_HIP; }
  const bool profile = plan->profile;
  plan->capturing = true;
  plan->profile = false;
  const int rc = m3g_energy_forces(plan, io, workspace, workspace_bytes, (void*)s);
  plan->capturing = false;
  plan->profile = profile;
  e = hipStreamEndCapture(s, &graph);
  (void)hipStreamDestroy(s);
  if (rc != M3G_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess || !graph) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return M3G_ERR_HIP; }
  size_t n = 0;
  e = hipGraphGetNodes(graph, nullptr, &n);
  std::vector<hipGraphNode_t> nodes(n);
  if (e == hipSuccess && n) e = hipGraphGetNodes(graph, nodes.data(), &n);
  int kernels = 0, other = 0;
  for (size_t i = 0; e == hipSuccess && i < n; ++i) {
    hipGraphNodeType ty;
    e = hipGraphNodeGetType(nodes[i], &ty);
    if (e != hipSuccess) break;
    if (ty == hipGraphNodeTypeKernel) ++kernels;
    else if (ty != hipGraphNodeTypeEmpty) ++other;
  }
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { set_error("m3g_count_launches: reading the captured graph failed: %s", hipGetErrorString(e)); return M3G_ERR_HIP; }
  *kernel_launches = kernels;
  if (other_operations) *other_operations = other;
  return M3G_OK;
}

// what the reverse edge kernels of this plan hand to k_node_reverse: the fused kernels write 24-bit rows (floating in the bf16x3 mode,
// fixed point + scales in the f16x3 mode), everything else fp32 rows
static int dp1_format(const m3g_plan* plan) {
  if (!fused_reverse(plan)) return kDp1F32;
  if (dp1_rows_by_dst(plan)) return kDp1F32ByDst;
  return plan->precision == kPrecBf16x3 ? kDp1Packed : plan->precision == kPrecF16x3 ? kDp1Fixed : kDp1F32;
}

extern "C" int m3g_energy_forces(const m3g_plan* plan, const m3g_io* io, void* workspace, size_t workspace_bytes,
                                 void* stream_) {
  if (!plan || !io) { set_error("m3g_energy_forces: null argument"); return M3G_ERR_VALUE; }
  if (!plan->committed) { set_error("m3g_energy_forces: plan parameters not committed"); return M3G_ERR_STATE; }
  {
    int dev = -1;
    M3G_HIP_CHECK(hipGetDevice(&dev));
    if (dev != plan->device) {
      set_error("m3g_energy_forces: the plan was committed on device %d but the current device is %d (commit again on this device)", plan->device, dev);
      return M3G_ERR_STATE;
    }
  }
  if (plan->graph_replay && !plan->capturing && !plan->profile && !plan->overlap)
    return energy_forces_graph(plan, io, workspace, workspace_bytes, (hipStream_t)stream_);
  const int64_t N = io->n_atoms, E = io->n_edges, T = io->n_triplets, S = io->n_structs;
  if (N < 0 || E < 0 || T < 0 || S < 0) { set_error("negative size"); return M3G_ERR_VALUE; }
  if (!io->total_energy || !io->topo || (N > 0 && (!io->pos || !io->atom_types)) || (S > 0 && !io->lattice) ||
      (E > 0 && !io->edge_cell_shift)) {
    set_error("m3g_energy_forces: missing required pointer");
    return M3G_ERR_VALUE;
  }
  if (io->triplet_angles && T > 0 && !io->triplet_edge_index) { set_error("triplet_angles requires triplet_edge_index"); return M3G_ERR_VALUE; }
  hipStream_t s = (hipStream_t)stream_;
  if (plan->edge_kernel == 2) return generic_energy_forces(plan, io, workspace, workspace_bytes, s);
  const Consts& c = plan->consts;
  const WeightLayout& wl = plan->wl;
  const float* W = plan->d_weights;
  Topo t = topo_carve(N, E, T, S, const_cast<void*>(io->topo));
  // (0: the list kernels, always valid; the reference's Legendre backward is not linear in the incoming gradient of a triplet,
  // so the moment sums cannot carry it)
  const int tb_hints = plan->tb_moments && !plan->legendre_ref ? io->topo_hints : 0;
  const bool mfma = plan->edge_kernel == 1;
  const bool fused_rev = fused_reverse(plan);
  Work w = work_carve(c, mfma, saved_activations(plan), N, E, T, S, nullptr);
  if (!workspace || workspace_bytes < w.total_bytes) { set_error("workspace too small: %zu < %zu", workspace_bytes, w.total_bytes); return M3G_ERR_SIZE; }
  w = work_carve(c, mfma, saved_activations(plan), N, E, T, S, workspace);
  // tail scratch: per-atom energies + per-structure sums when the caller does not want them
  float* tail = (float*)((char*)workspace + w.total_bytes - align_up(((size_t)N + (size_t)S * 2 + 64) * sizeof(float)));
  float* ea = io->scaled_atomic_energies ? io->scaled_atomic_energies : tail;
  float* st = io->scaled_total_energy ? io->scaled_total_energy : tail + N;

  // ---------------- forward ----------------
  // small systems: the geometry stage and block 0's node tables (independent of each other) as two roles of one launch
  bool np0_done = false;
  {
    M3G_STAGE(ST_GEOM);
    np0_done = mfma && !plan->profile && launch_geometry_node_pre(plan, c, t, w, io->pos, io->lattice, io->edge_cell_shift, io->atom_types, W + wl.emb, s);
    if (!np0_done) launch_geometry(c, t, io->pos, io->lattice, io->edge_cell_shift, w, s);
  }
  {
    M3G_STAGE(ST_EMBED);
    if (mfma) {
      if (c.B == 0) launch_embed_nodes_only(c, W, wl, t, io->atom_types, w, s);   // otherwise block 0's node kernel forms x^0
      if (!fused_rev || c.B == 0) launch_embed_edges_soa(c, W + wl.adj_t, w.h, w.e_blk[0], E, s);   // fused path: block 0 forms e0 in its kernels
    } else {
      launch_embed(c, W, wl, t, io->atom_types, w, s);
    }
  }
  for (int b = 0; b < c.B; ++b) {
    {
      M3G_STAGE(ST_NODE_PRE);
      // MFMA path: block b > 0 forms x^b = x^(b-1) + per-centre message sums of block b-1 while loading it
      if (mfma && b == 0 && np0_done) { /* formed beside the geometry stage */ }
      else if (mfma) launch_node_pre_mfma(plan, c, t, w, b, b > 0 ? w.x[b - 1] : nullptr, w.x[b], w.v[b], w.TAb[b], w.TBb[b],
                                          b == 0 ? io->atom_types : nullptr, W + wl.emb, s);
      else launch_node_pre(c, W, wl.blk[b], t, w, nullptr, w.x[b], w.v[b], w.TA, w.TB, s);
    }
    { M3G_STAGE(ST_THREEBODY); launch_threebody(c, t, w, w.v[b], w.m[b], s, tb_hints); }
    if (mfma) {
      { M3G_STAGE(ST_EDGE_FWD); launch_edge_block_mfma(plan, c, t, w, b, /*for_reverse=*/io->forces != nullptr, s); }
      (void)ST_NODE_SUM;   // the per-centre sums are consumed by the next node_pre / the readout
    } else {
      M3G_STAGE(ST_EDGE_FWD);
      if (N > 0) M3G_HIP_CHECK(hipMemcpyAsync(w.x[b + 1], w.x[b], sizeof(float) * N * kDP, hipMemcpyDeviceToDevice, s));
      launch_edge_block(c, W, wl.blk[b], t, w, b, w.x[b + 1], s);
    }
  }
  const bool want_f = io->forces != nullptr;
  bool energy_deferred = false;
  {
    M3G_STAGE(ST_READOUT);
    // a step that ends with the reference virial forms the per-structure energy sums in that launch (nothing in between reads them)
    energy_deferred = mfma && plan->small_launches && want_f && io->stresses && plan->stress_mode == 0 && !plan->profile;
    if (mfma) launch_readout_mfma(plan, c, wl, t, io->atom_types, c.B > 0 ? w.x[c.B - 1] : nullptr, w.x[c.B], w, ea, st, io->total_energy, want_f, s,
                                  &energy_deferred);
    else launch_readout(c, W, wl, t, io->atom_types, nullptr, w.x[c.B], w, ea, st, io->total_energy, want_f, s);
  }
  StageTimer* st_out = new StageTimer(plan, ST_OUTPUTS, s);

  if (io->node_features) launch_copy_strided(w.x[c.B], kDP, io->node_features, c.D, c.D, N, s);
  if (io->edge_attr) {
    if (mfma) launch_soa_to_rows(w.e_blk[c.B], io->edge_attr, c.D, c.D, E, s);
    else launch_copy_strided(w.e, kDP, io->edge_attr, c.D, c.D, E, s);
  }
  if (io->edge_distances && E > 0) M3G_HIP_CHECK(hipMemcpyAsync(io->edge_distances, w.d, sizeof(float) * E, hipMemcpyDeviceToDevice, s));
  if (io->edge_weights) launch_copy_strided(w.h, kRP, io->edge_weights, c.R, c.R, E, s);
  if (io->triplet_angles) launch_triplet_angles(t, io->triplet_edge_index, w.u, io->triplet_angles, s);
  if (io->mid_edge_features)
    for (int b = 0; b < c.B; ++b)   // the aggregate is kept per active edge: expand to the reference's [E, l_max*n_max]
      launch_copy_expand_rows(t.act_id, w.m[b], kCP, io->mid_edge_features + (size_t)b * E * c.C, c.C, c.C, E, s);
  delete st_out;

  // ---------------- reverse ----------------
  if (want_f) {
    float* dx_cur = w.dx;
    float* dx_alt = w.dx2;
    bool dr_done = false;
    for (int b = c.B - 1; b >= 0; --b) {
      if (b == c.B - 1 && E > 0) {
          if (!mfma) {
            M3G_HIP_CHECK(hipMemsetAsync(w.de, 0, sizeof(float) * E * kDP, s));
            M3G_HIP_CHECK(hipMemsetAsync(w.dh, 0, sizeof(float) * E * kRP, s));
          }
          // dd / du need no clearing: the first three-body reverse of the step writes its (active) rows, and the geometry
          // reverse reads active rows only
      }
      if (mfma && fused_rev) {
        M3G_STAGE(ST_EDGE_REV_FUSED);
        if (plan->precision == kPrecF32) launch_edge_rev_f32(plan, c, t, w, b, dx_cur, /*de_is_zero=*/b == c.B - 1, s);
        else launch_edge_rev_fused(plan, c, t, w, b, dx_cur, /*de_is_zero=*/b == c.B - 1, s);
      } else if (mfma) {
        { M3G_STAGE(ST_EDGE_REV_NODE); launch_edge_rev_node_mlp(plan, c, t, w, b, dx_cur, s); }
        M3G_STAGE(ST_EDGE_REV);
        launch_edge_rev_edge_mlp(plan, c, t, w, b, dx_cur, /*de_is_zero=*/b == c.B - 1, s);
      } else {
        M3G_STAGE(ST_EDGE_REV);
        launch_edge_block_reverse(c, W, wl.blk[b], t, w, b, dx_cur, s);
      }
      if (b > 0 && fused_rev && plan->overlap && !plan->profile && ensure_side_stream(plan)) {
        // the node reverse's dp1 gather needs nothing from the three-body reverse: run that short latency-bound kernel on a
        // side stream beside it, then add the v-gradient share (which needs its dL/dg) once both are done
        M3G_HIP_CHECK(hipEventRecord(plan->ev_fork, s));
        M3G_HIP_CHECK(hipStreamWaitEvent(plan->side_stream, plan->ev_fork, 0));
        launch_threebody_reverse(c, t, w, w.v[b], /*first=*/b == c.B - 1, plan->side_stream, tb_hints, plan->legendre_ref);
        M3G_HIP_CHECK(hipEventRecord(plan->ev_join, plan->side_stream));
        launch_node_reverse(c, W, wl.blk[b], t, w, w.v[b], dx_cur, dx_alt, true, /*dp1 format=*/dp1_format(plan), /*with_v_term=*/false, s);
        M3G_HIP_CHECK(hipStreamWaitEvent(s, plan->ev_join, 0));
        launch_node_reverse_v_term(c, W, wl.blk[b], t, w, w.v[b], dx_alt, s);
        float* tmp = dx_cur; dx_cur = dx_alt; dx_alt = tmp;
      } else if (b > 0 && fused_rev && plan->fuse_node_tb && !plan->profile &&
                 launch_node_tb_reverse(c, W, wl.blk[b], t, w, w.v[b], /*first=*/b == c.B - 1, dx_cur, dx_alt, dp1_format(plan), b, s, tb_hints, plan->debug_node_tb_polls)) {
        // (moment path) three-body reverse and node reverse of the block as two workgroup roles of ONE launch
        float* tmp = dx_cur; dx_cur = dx_alt; dx_alt = tmp;
      } else if (b == 0 && mfma && fused_rev && plan->small_launches && !plan->profile &&
                 launch_threebody_reverse_final(c, t, w, w.v[b], /*first=*/b == c.B - 1, w.dh_parts, c.B, s, tb_hints)) {
        dr_done = true;   // (moment path) the step's last three-body reverse formed dE/dr of every edge as well
      } else {
        { M3G_STAGE(ST_THREEBODY_REV); launch_threebody_reverse(c, t, w, w.v[b], /*first=*/b == c.B - 1, s, tb_hints, plan->legendre_ref); }
        if (b > 0) {  // x^0 is the species embedding: no position dependence, its gradient is never needed
          M3G_STAGE(ST_NODE_REV);
          launch_node_reverse(c, W, wl.blk[b], t, w, w.v[b], dx_cur, dx_alt, fused_rev, /*dp1 format=*/dp1_format(plan), /*with_v_term=*/true, s,
                              /*small=*/plan->small_launches && N <= kFusedSumsMaxAtoms);
          float* tmp = dx_cur; dx_cur = dx_alt; dx_alt = tmp;
        }
      }
    }
    {
      M3G_STAGE(ST_EMBED_REV);
      if (fused_rev) { /* block 0's fused reverse kernel already added the embedding's dL/dh share */ }
      else if (mfma) launch_embed_edges_reverse_soa(W + wl.adj, w.h, w.de_soa, w.dh_parts + (size_t)2 * c.B * E * kRP, E, s);
      else launch_embed_reverse(c, W, wl, t, w, s);
    }
    M3G_STAGE(ST_GEOM_REV);
    // few structures: the force-gather launch ends with the reference virial (one launch less, bit-identical)
    const bool fuse = mfma && plan->small_launches && plan->stress_mode == 0;
    bool tail_fused = false;
    if (mfma) tail_fused = launch_geometry_reverse(c, t, w, w.dh_parts, fused_rev ? c.B : 2 * c.B + 1, io->forces, io->stresses, s, fuse, io->pos, io->lattice,
                                                   dr_done);
    else launch_geometry_reverse(c, t, w, w.dh, 1, io->forces, io->stresses, s);
    if (io->stresses && !tail_fused) {
      if (plan->stress_mode == 1) launch_stress_pair(t, w, io->lattice, io->stresses, s);
      else launch_stress(c, t, io->pos, io->lattice, io->forces, io->stresses, s, energy_deferred ? ea : nullptr, st, io->total_energy);
    } else if (energy_deferred) {
      launch_energy_sums(c, t, ea, st, io->total_energy, s);   // (the virial was formed by the force gather's last workgroup after all)
    }
  } else if (io->stresses) {
    set_error("stresses require forces");
    return M3G_ERR_VALUE;
  }
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}

// ---------------------------------------------------------------------------------- stage entry points
extern "C" int m3g_distance_angle(double length_scale, int64_t N, int64_t E, int64_t T, int64_t S, const float* pos,
                                  const float* lattice, const int32_t* shift, const void* topo,
                                  const int64_t* triplet_edge_index, float* scratch_u, float* edge_distances,
                                  float* triplet_angles, void* stream_) {
  if (!topo || !edge_distances || (E > 0 && !scratch_u)) { set_error("m3g_distance_angle: null argument"); return M3G_ERR_VALUE; }
  hipStream_t s = (hipStream_t)stream_;
  Topo t = topo_carve(N, E, T, S, const_cast<void*>(topo));
  launch_distance_only((float)length_scale, t, pos, lattice, shift, scratch_u, edge_distances, s);
  if (triplet_angles) {
    if (T > 0 && !triplet_edge_index) { set_error("triplet_angles requires triplet_edge_index"); return M3G_ERR_VALUE; }
    launch_triplet_angles(t, triplet_edge_index, scratch_u, triplet_angles, s);
  }
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}

extern "C" int m3g_edge_featurizer(int32_t n_max, double scaled_cutoff, const float* host_em, const float* host_dm,
                                   const float* host_coeff, int64_t E, const float* edge_distances, float* edge_weights,
                                   void* stream_) {
  if (n_max < 1 || n_max > kRCap) { set_error("m3g_edge_featurizer: n_max must be in 1..%d", kRCap); return M3G_ERR_UNSUPPORTED; }
  if (!host_em || !host_dm || !host_coeff || (E > 0 && (!edge_distances || !edge_weights))) { set_error("null argument"); return M3G_ERR_VALUE; }
  Consts c{};
  c.R = n_max;
  const float pi_f = (float)M_PI;
  for (int m = 0; m < n_max; ++m) {
    c.a1[m] = ((float)(m + 1) * pi_f) / (float)scaled_cutoff;
    c.a2[m] = ((float)(m + 2) * pi_f) / (float)scaled_cutoff;
    c.coeff[m] = host_coeff[m];
    c.rec_mul[m] = m > 0 ? sqrtf(host_em[m] / host_dm[m - 1]) : 0.f;
    c.rec_div[m] = sqrtf(host_dm[m]);
  }
  launch_edge_featurizer(c, E, edge_distances, edge_weights, n_max, (hipStream_t)stream_);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}

extern "C" int m3g_atom_featurizer(int32_t num_types, int32_t dim, const float* weight, int64_t N, const int64_t* atom_types,
                                   float* x, void* stream_) {
  if (!weight || (N > 0 && (!atom_types || !x))) { set_error("m3g_atom_featurizer: null argument"); return M3G_ERR_VALUE; }
  launch_gather_rows(weight, N, dim, num_types, num_types, true, atom_types, x, (hipStream_t)stream_);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}

extern "C" int m3g_atom_ref(int32_t num_types, const float* elemental, int64_t N, const int64_t* atom_types, float* out,
                            void* stream_) {
  if (!elemental || (N > 0 && (!atom_types || !out))) { set_error("m3g_atom_ref: null argument"); return M3G_ERR_VALUE; }
  launch_gather_rows(elemental, N, 1, 1, num_types, false, atom_types, out, (hipStream_t)stream_);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}
