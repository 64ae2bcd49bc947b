// Torch-free host of the C ABI (include/m3gnet_hip.h): reads one case from a flat binary file written by
// tests/test_gpu_c_abi.py, drives plan -> topology -> workspace -> m3g_energy_forces with plain hipMalloc'd buffers,
// and writes total energies, forces and stresses back.  Shows that the drop-in boundary needs nothing but the HIP
// runtime: the Python package is one possible host, not part of the library.
//
// File format (little endian): magic "M3GC", then records
//   'P' key_len key numel float[numel]     parameter (state_dict key)
//   'K' key_len key numel float[numel]     constant
//   'G' N E T S, pos f32[N*3], types i64[N], edge_index i64[2E], shift i32[3E], triplets i64[2T], lattice f32[9S], batch i64[N]
//   'C' cutoff threebody_cutoff energy_scale length_scale (f64 x4) l_max n_max num_types embedding_dim num_blocks (i32 x5)
// Build: hipcc -O2 tests/c_abi/m3g_c_abi_check.cpp -Iinclude -Ltorch-m3gnet_amd/lib -lm3gnet_hip -o torch-m3gnet_amd/lib/m3g_c_abi_check
#include <hip/hip_runtime.h>

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
static T* to_dev(const std::vector<T>& v) {
  T* d = nullptr;
  if (hipMalloc((void**)&d, std::max<size_t>(v.size(), 1) * sizeof(T)) != hipSuccess) return nullptr;
  if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s case.bin out.bin\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  char magic[4];
  if (!rd(f, magic, 4) || memcmp(magic, "M3GC", 4) != 0) { fprintf(stderr, "bad magic\n"); return 1; }
  m3g_config cfg{};
  m3g_plan* plan = nullptr;
  int64_t N = 0, E = 0, T = 0, S = 0;
  std::vector<float> pos, lattice;
  std::vector<int64_t> types, ei, tei, batch;
  std::vector<int32_t> shift;
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
      pos.resize(N * 3); types.resize(N); ei.resize(2 * E); shift.resize(3 * E); tei.resize(2 * T); lattice.resize(9 * S); batch.resize(N);
      if (!rd(f, pos.data(), pos.size()) || !rd(f, types.data(), types.size()) || !rd(f, ei.data(), ei.size()) ||
          !rd(f, shift.data(), shift.size()) || !rd(f, tei.data(), tei.size()) || !rd(f, lattice.data(), lattice.size()) ||
          !rd(f, batch.data(), batch.size())) return 1;
    } else if (tag == 'C') {
      double d[4]; int32_t i[5];
      if (!rd(f, d, 4) || !rd(f, i, 5)) return 1;
      cfg.cutoff = d[0]; cfg.threebody_cutoff = d[1]; cfg.energy_scale = d[2]; cfg.length_scale = d[3];
      cfg.l_max = i[0]; cfg.n_max = i[1]; cfg.num_types = i[2]; cfg.embedding_dim = i[3]; cfg.num_blocks = i[4];
    } else { fprintf(stderr, "unknown record '%c'\n", tag); return 1; }
  }
  fclose(f);

  m3g_info info{};
  CK(m3g_get_info(&info));
  printf("libm3gnet_hip abi %d, %d device(s), arch %s\n", info.abi_version, info.device_count, info.arch);
  CK(m3g_plan_create(&cfg, &plan));
  for (const Rec& r : recs) {
    if (r.kind == 'P') CK(m3g_plan_set_param(plan, r.key.c_str(), r.v.data(), (int64_t)r.v.size()));
    else CK(m3g_plan_set_const(plan, r.key.c_str(), r.v.data(), (int64_t)r.v.size()));
  }
  CK(m3g_plan_commit(plan));

  float* d_pos = to_dev(pos); float* d_lat = to_dev(lattice);
  int64_t* d_types = to_dev(types); int64_t* d_ei = to_dev(ei); int64_t* d_tei = to_dev(tei); int64_t* d_batch = to_dev(batch);
  int32_t* d_shift = to_dev(shift);
  if (!d_pos || !d_lat || !d_types || !d_ei || !d_tei || !d_batch || !d_shift) { fprintf(stderr, "device upload failed\n"); return 3; }
  hipStream_t stream;
  HK(hipStreamCreate(&stream));
  size_t topo_bytes = 0, work_bytes = 0;
  CK(m3g_topology_bytes(N, E, T, S, &topo_bytes));
  void *topo = nullptr, *work = nullptr;
  HK(hipMalloc(&topo, topo_bytes));
  HK(hipMemset(topo, 0, topo_bytes));   // (the two-phase build below is compared with this buffer byte by byte)
  int32_t flags = 0;
  CK(m3g_topology_build(N, E, T, S, d_ei, d_tei, d_batch, topo, topo_bytes, &flags, stream));
  HK(hipStreamSynchronize(stream));
  if (flags) { fprintf(stderr, "malformed graph, flags %d\n", flags); return 2; }
  CK(m3g_workspace_bytes(plan, N, E, T, S, &work_bytes));
  HK(hipMalloc(&work, work_bytes));
  float *d_e, *d_f, *d_s;
  HK(hipMalloc((void**)&d_e, sizeof(float) * std::max<int64_t>(S, 1)));
  HK(hipMalloc((void**)&d_f, sizeof(float) * 3 * std::max<int64_t>(N, 1)));
  HK(hipMalloc((void**)&d_s, sizeof(float) * 6 * std::max<int64_t>(S, 1)));
  m3g_io io{};
  io.n_atoms = N; io.n_edges = E; io.n_triplets = T; io.n_structs = S;
  io.pos = d_pos; io.atom_types = d_types; io.edge_cell_shift = d_shift; io.lattice = d_lat; io.topo = topo;
  io.triplet_edge_index = d_tei;
  CK(m3g_topology_hints(N, E, T, S, topo, &io.topo_hints, stream));   // complete partner lists -> the three-body moment kernels
  // (before the hot call, which may leave sticky status bits on `topo`) the same topology through the two-phase canonical build (the fixture's lists are in the library's canonical order): queued with
  // its verdict in pinned host memory, the host free until _end -- the list part of the buffer must equal the one built above
  {
    size_t data_bytes = 0;
    CK(m3g_topology_data_bytes(N, E, T, S, &data_bytes));
    void* topo2 = nullptr;
    int32_t* verdict = nullptr;
    HK(hipMalloc(&topo2, topo_bytes));
    HK(hipMemset(topo2, 0, topo_bytes));
    HK(hipHostMalloc((void**)&verdict, 16 * sizeof(int32_t), hipHostMallocDefault));
    int32_t flags2 = 0, hints2 = 0, path = -1;
    CK(m3g_topology_build_canonical_begin(N, E, T, S, d_ei, d_tei, d_batch, topo2, topo_bytes, verdict, stream));
    CK(m3g_topology_build_canonical_end(N, E, T, S, d_ei, d_tei, d_batch, topo2, topo_bytes, verdict, &flags2, &hints2, stream));
    HK(hipStreamSynchronize(stream));
    CK(m3g_topology_debug_last_path(&path));
    std::vector<unsigned char> a(data_bytes), b(data_bytes);
    HK(hipMemcpy(a.data(), topo, data_bytes, hipMemcpyDeviceToHost));
    HK(hipMemcpy(b.data(), topo2, data_bytes, hipMemcpyDeviceToHost));
    if (flags2 || hints2 != io.topo_hints || memcmp(a.data(), b.data(), data_bytes) != 0) {
      fprintf(stderr, "two-phase canonical build differs: flags %d hints %#x vs %#x path %d\n", flags2, hints2, io.topo_hints, path);
      return 4;
    }
    printf("two-phase canonical topology build: %zu list bytes identical, hints %#x, path %d\n", data_bytes, hints2, path);
    HK(hipFree(topo2));
    HK(hipHostFree(verdict));
  }
  io.total_energy = d_e; io.forces = d_f; io.stresses = d_s;
  CK(m3g_energy_forces(plan, &io, work, work_bytes, stream));
  HK(hipStreamSynchronize(stream));
  std::vector<float> e(S), fo(3 * N), st(6 * S);
  HK(hipMemcpy(e.data(), d_e, sizeof(float) * S, hipMemcpyDeviceToHost));
  HK(hipMemcpy(fo.data(), d_f, sizeof(float) * 3 * N, hipMemcpyDeviceToHost));
  HK(hipMemcpy(st.data(), d_s, sizeof(float) * 6 * S, hipMemcpyDeviceToHost));
  {
    // sticky error bits the hot call left on the topology buffer (M3G_TOPO_ERR_*): a caller without the Python host's checks polls them
    int32_t status = 0;
    CK(m3g_topology_status(N, E, T, S, topo, &status, stream));
    printf("topology status %d\n", status);
  }
  FILE* o = fopen(argv[2], "wb");
  if (!o) { perror(argv[2]); return 1; }
  fwrite(e.data(), sizeof(float), e.size(), o);
  fwrite(fo.data(), sizeof(float), fo.size(), o);
  fwrite(st.data(), sizeof(float), st.size(), o);
  fclose(o);
  printf("N=%lld E=%lld T=%lld S=%lld  E[0]=%.6f\n", (long long)N, (long long)E, (long long)T, (long long)S, S ? e[0] : 0.f);
  m3g_plan_destroy(plan);
  return 0;
}
