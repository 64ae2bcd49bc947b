// Torch-free trajectory loop over the C ABI (include/m3gnet_hip.h): candidate search (m3g_neighbor_*, m3g_verlet_rows), buffers of the
// candidates' capacity from hipMalloc, then ONE m3g_md_step per frame -- skin test, lists and topology when a pair crossed a cutoff,
// energies and forces -- with a new search whenever the library asks for one.  What the reference does per frame on the host
// (MaterialGraph.from_structure, data/material_graph.py:132-254, then Gradient.forward, nn/gradient.py:25-64), here without Python.
// tests/test_gpu_c_abi.py writes the case and the frames and compares every frame with the Python host's results, bit for bit.
//
// File format (little endian): magic "M3GC", records 'P' / 'K' / 'C' / 'G' as in m3g_c_abi_check.cpp, then
//   'T' n_frames skin(f64), positions f64[n_frames * N * 3]
// Output: per frame  path(i32)  E f32[S]  F f32[3N]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "m3gnet_hip.h"

#define CK(call)                                                                      \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, m3g_last_error()); return 2; } \
  } while (0)
#define HK(call)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 3; } \
  } while (0)

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }
template <class T>
static T* dev_alloc(size_t n) {
  T* d = nullptr;
  return hipMalloc((void**)&d, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess ? d : nullptr;
}
template <class T>
static T* to_dev(const std::vector<T>& v) {
  T* d = dev_alloc<T>(v.size());
  if (d && !v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

// (2rx+1)(2ry+1)(2rz+1) images per structure, the bound m3g_neighbor_* sizes its scratch by (torch_m3gnet/data/graph_gpu.py: max_images)
static int64_t max_images(const std::vector<double>& lat, int64_t S, double cutoff) {
  int64_t best = 1;
  for (int64_t s = 0; s < S; ++s) {
    const double* a = &lat[9 * s];
    auto cross = [&](int p, int q, double* o) {
      o[0] = a[3 * p + 1] * a[3 * q + 2] - a[3 * p + 2] * a[3 * q + 1];
      o[1] = a[3 * p + 2] * a[3 * q + 0] - a[3 * p + 0] * a[3 * q + 2];
      o[2] = a[3 * p + 0] * a[3 * q + 1] - a[3 * p + 1] * a[3 * q + 0];
    };
    double c12[3];
    cross(1, 2, c12);
    const double vol = std::fabs(a[0] * c12[0] + a[1] * c12[1] + a[2] * c12[2]);
    int64_t n = 1;
    for (int p = 0; p < 3; ++p) {
      double c[3];
      cross((p + 1) % 3, (p + 2) % 3, c);
      const double area = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
      n *= 2 * (int64_t)std::ceil((cutoff + 1e-8) * area / vol) + 1;
    }
    best = std::max(best, n);
  }
  return best;
}

struct Lists {   // everything one candidate search hands to m3g_md_set_lists
  int64_t* cand_ei = nullptr; int32_t* cand_shift = nullptr; double* cand_dist = nullptr; int32_t* rows = nullptr; uint8_t* state = nullptr;
  void* vscratch = nullptr; double* pos_ref = nullptr;
  int64_t* ei = nullptr; int32_t* shift = nullptr; int64_t* tei = nullptr; int64_t* nti = nullptr; int32_t* ntij = nullptr; float* pos32 = nullptr;
  void* topo = nullptr; void* work = nullptr;
  void release() {
    for (void* p : {(void*)cand_ei, (void*)cand_shift, (void*)cand_dist, (void*)rows, (void*)state, vscratch, (void*)pos_ref, (void*)ei, (void*)shift,
                    (void*)tei, (void*)nti, (void*)ntij, (void*)pos32, topo, work})
      if (p) (void)hipFree(p);
    *this = Lists{};
  }
};

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s case.bin out.bin\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  char magic[4];
  if (!rd(f, magic, 4) || memcmp(magic, "M3GC", 4) != 0) { fprintf(stderr, "bad magic\n"); return 1; }
  m3g_config cfg{};
  int64_t N = 0, E = 0, T = 0, S = 0, frames = 0;
  double skin = 0.0;
  std::vector<float> pos32, lattice32;
  std::vector<int64_t> types, ei, tei, batch;
  std::vector<int32_t> shift;
  std::vector<double> traj;
  struct Rec { char kind; std::string key; std::vector<float> v; };
  std::vector<Rec> recs;
  char tag;
  while (rd(f, &tag, 1)) {
    if (tag == 'P' || tag == 'K') {
      int32_t kl; int64_t n;
      if (!rd(f, &kl, 1)) return 1;
      std::string key(kl, '\0');
      if (!rd(f, key.data(), kl) || !rd(f, &n, 1)) return 1;
      Rec r{tag, key, std::vector<float>((size_t)n)};
      if (!rd(f, r.v.data(), (size_t)n)) return 1;
      recs.push_back(std::move(r));
    } else if (tag == 'G') {
      int64_t h[4];
      if (!rd(f, h, 4)) return 1;
      N = h[0]; E = h[1]; T = h[2]; S = h[3];
      pos32.resize(N * 3); types.resize(N); ei.resize(2 * E); shift.resize(3 * E); tei.resize(2 * T); lattice32.resize(9 * S); batch.resize(N);
      if (!rd(f, pos32.data(), pos32.size()) || !rd(f, types.data(), types.size()) || !rd(f, ei.data(), ei.size()) ||
          !rd(f, shift.data(), shift.size()) || !rd(f, tei.data(), tei.size()) || !rd(f, lattice32.data(), lattice32.size()) ||
          !rd(f, batch.data(), batch.size())) return 1;
    } else if (tag == 'C') {
      double d[4]; int32_t i[5];
      if (!rd(f, d, 4) || !rd(f, i, 5)) return 1;
      cfg.cutoff = d[0]; cfg.threebody_cutoff = d[1]; cfg.energy_scale = d[2]; cfg.length_scale = d[3];
      cfg.l_max = i[0]; cfg.n_max = i[1]; cfg.num_types = i[2]; cfg.embedding_dim = i[3]; cfg.num_blocks = i[4];
    } else if (tag == 'T') {
      if (!rd(f, &frames, 1) || !rd(f, &skin, 1)) return 1;
      traj.resize((size_t)frames * N * 3);
      if (!rd(f, traj.data(), traj.size())) return 1;
    } else { fprintf(stderr, "unknown record '%c'\n", tag); return 1; }
  }
  fclose(f);
  if (frames < 1 || N < 1) { fprintf(stderr, "no frames\n"); return 1; }

  m3g_plan* plan = nullptr;
  CK(m3g_plan_create(&cfg, &plan));
  for (const Rec& r : recs) {
    if (r.kind == 'P') CK(m3g_plan_set_param(plan, r.key.c_str(), r.v.data(), (int64_t)r.v.size()));
    else CK(m3g_plan_set_const(plan, r.key.c_str(), r.v.data(), (int64_t)r.v.size()));
  }
  CK(m3g_plan_commit(plan));

  std::vector<double> lattice64(lattice32.begin(), lattice32.end());
  double* d_lat64 = to_dev(lattice64);
  float* d_lat32 = to_dev(lattice32);
  int64_t* d_types = to_dev(types);
  int64_t* d_batch = to_dev(batch);
  double* d_pos = dev_alloc<double>(3 * N);
  float *d_e = dev_alloc<float>(S), *d_f = dev_alloc<float>(3 * N), *d_s = dev_alloc<float>(6 * S);
  if (!d_lat64 || !d_lat32 || !d_types || !d_batch || !d_pos || !d_e || !d_f || !d_s) { fprintf(stderr, "device allocation failed\n"); return 3; }
  hipStream_t stream;
  HK(hipStreamCreate(&stream));
  m3g_md* md = nullptr;
  CK(m3g_md_create(&md));
  Lists L;
  const double rc = cfg.cutoff + skin, r3 = cfg.threebody_cutoff + skin;

  // candidates at the positions in d_pos: every pair within cutoff + skin; the triplets those within threebody_cutoff + skin would
  // form bound the triplets of every configuration within skin / 2 (m3g_md_lists.cap_triplets)
  auto search = [&]() -> int {
    L.release();
    const int64_t M = max_images(lattice64, S, rc);
    size_t nb = 0;
    CK(m3g_neighbor_scratch_bytes(N, S, M, &nb));
    void* scratch = nullptr;
    HK(hipMalloc(&scratch, std::max<size_t>(nb, 1)));
    int64_t Ec = 0, cap_t = 0;
    CK(m3g_neighbor_count_triplets(N, S, M, d_pos, d_lat64, d_batch, rc, (float)r3, scratch, nb, &Ec, &cap_t, stream));
    L.cand_ei = dev_alloc<int64_t>(2 * Ec); L.cand_shift = dev_alloc<int32_t>(3 * Ec); L.cand_dist = dev_alloc<double>(Ec);
    CK(m3g_neighbor_fill(N, S, M, d_batch, rc, scratch, Ec, L.cand_ei, L.cand_shift, L.cand_dist, stream));
    HK(hipStreamSynchronize(stream));
    HK(hipFree(scratch));
    L.rows = dev_alloc<int32_t>(N + 2);
    CK(m3g_verlet_rows(N, Ec, L.cand_ei, L.rows, stream));
    size_t vb = 0, tb = 0, wb = 0;
    CK(m3g_verlet_scratch_bytes(N, Ec, &vb));
    CK(m3g_topology_bytes(N, Ec, cap_t, S, &tb));
    CK(m3g_workspace_bytes(plan, N, Ec, cap_t, S, &wb));
    HK(hipMalloc(&L.vscratch, std::max<size_t>(vb, 1)));
    HK(hipMalloc(&L.topo, tb));
    HK(hipMalloc(&L.work, wb));
    L.state = dev_alloc<uint8_t>(Ec + 16);
    L.pos_ref = dev_alloc<double>(3 * N);
    HK(hipMemcpyAsync(L.pos_ref, d_pos, sizeof(double) * 3 * N, hipMemcpyDeviceToDevice, stream));
    L.ei = dev_alloc<int64_t>(2 * Ec); L.shift = dev_alloc<int32_t>(3 * Ec); L.tei = dev_alloc<int64_t>(2 * cap_t); L.nti = dev_alloc<int64_t>(N);
    L.ntij = dev_alloc<int32_t>(Ec); L.pos32 = dev_alloc<float>(3 * N);
    m3g_md_lists ml{};
    ml.n_atoms = N; ml.n_structs = S; ml.n_cand = Ec; ml.cap_edges = Ec; ml.cap_triplets = cap_t;
    ml.cutoff = cfg.cutoff; ml.threebody_cutoff = cfg.threebody_cutoff; ml.skin = skin;
    ml.pos_ref = L.pos_ref; ml.lattice = d_lat64; ml.lattice32 = d_lat32; ml.batch = d_batch; ml.atom_types = d_types;
    ml.cand_edge_index = L.cand_ei; ml.cand_shift = L.cand_shift; ml.cand_row_ptr = L.rows; ml.cand_state = L.state;
    ml.verlet_scratch = L.vscratch; ml.verlet_scratch_bytes = vb;
    ml.edge_index = L.ei; ml.edge_cell_shift = L.shift; ml.triplet_edge_index = L.tei; ml.num_triplet_i = L.nti; ml.num_triplet_ij = L.ntij;
    ml.pos32 = L.pos32; ml.topo = L.topo; ml.topo_bytes = tb; ml.workspace = L.work; ml.workspace_bytes = wb;
    CK(m3g_md_set_lists(md, &ml));
    return 0;
  };

  FILE* o = fopen(argv[2], "wb");
  if (!o) { perror(argv[2]); return 1; }
  std::vector<float> e(S), fo(3 * N);
  int counts[3] = {0, 0, 0};
  for (int64_t k = 0; k < frames; ++k) {
    HK(hipMemcpyAsync(d_pos, &traj[(size_t)k * N * 3], sizeof(double) * 3 * N, hipMemcpyHostToDevice, stream));
    if (k == 0) { int r = search(); if (r) return r; }
    m3g_md_result res{};
    CK(m3g_md_step(md, plan, d_pos, d_e, d_f, d_s, 0, &res, stream));
    bool searched = k == 0;
    if (res.path == M3G_MD_NEED_SEARCH) {   // an atom left its skin: nothing was evaluated
      int r = search();
      if (r) return r;
      searched = true;
      CK(m3g_md_step(md, plan, d_pos, d_e, d_f, d_s, 0, &res, stream));
    }
    if (res.path != M3G_MD_REUSE && res.path != M3G_MD_REFILL) { fprintf(stderr, "frame %lld: path %d\n", (long long)k, res.path); return 4; }
    HK(hipStreamSynchronize(stream));
    HK(hipMemcpy(e.data(), d_e, sizeof(float) * S, hipMemcpyDeviceToHost));
    HK(hipMemcpy(fo.data(), d_f, sizeof(float) * 3 * N, hipMemcpyDeviceToHost));
    const int32_t path = searched ? 2 : res.path;
    ++counts[path];
    fwrite(&path, sizeof(int32_t), 1, o);
    fwrite(e.data(), sizeof(float), e.size(), o);
    fwrite(fo.data(), sizeof(float), fo.size(), o);
  }
  fclose(o);
  printf("trajectory of %lld frames, N=%lld S=%lld: %d on standing lists, %d with re-derived lists, %d after a search\n", (long long)frames,
         (long long)N, (long long)S, counts[0], counts[1], counts[2]);
  L.release();
  m3g_md_destroy(md);
  m3g_plan_destroy(plan);
  return 0;
}
