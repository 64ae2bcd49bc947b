// One trajectory step behind ONE call of the C ABI: skin-list test, (re-derived lists + topology when a pair crossed a cutoff),
// energies / forces / stresses.  The reference rebuilds its graph on the host for every frame (data/material_graph.py:132-254) and
// evaluates it (nn/gradient.py:25-64); torch_m3gnet/data/md.py sequences the same steps through m3g_verlet_* / m3g_topology_* /
// m3g_energy_forces from Python, and on small cells that sequencing IS the iteration: ~100 us of interpreter between ~28 launches,
// around 60 us of kernels that are not the step (profiles/r05_small_md_paths.txt).  Here the library sequences them itself, on
// buffers the caller allocates once per candidate search at the candidates' capacity, so that a refill allocates nothing and the
// host's share of an iteration is the two waits it cannot avoid (the verdict's sizes; the topology's certificate).
// Same kernels on the same inputs as the Python path: bit-identical results (tests/test_gpu_md.py).
#include <cmath>
#include <cstring>

#include "m3g_internal.h"

namespace m3g {
namespace {
// pos -> fp32, and (first thread) the sticky error word the EARLIER steps left on the topology buffer -> pinned host memory, where
// m3g_md_step reads it behind the wait it has anyway (status == nullptr: no topology stands yet)
__global__ void __launch_bounds__(256) k_md_pos32(int64_t n, const double* __restrict__ pos, float* __restrict__ out, const int32_t* status,
                                                  uint64_t* host_status) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)pos[i];   // (round to nearest: what pos.to(torch.float) does)
  if (i == 0) *host_status = status ? (uint64_t)(uint32_t)*status : 0;
}
// species outside the model's table?  (the reference raises: elemental_energies[atom_types], nn/atom_ref.py:27)  Once per list set and model.
__global__ void __launch_bounds__(256) k_md_species(int64_t n, const int64_t* __restrict__ types, int num_types, uint64_t* host_bad) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n && (types[i] < 0 || types[i] >= num_types)) *host_bad = 1;
}
}  // namespace
}  // namespace m3g

using namespace m3g;

struct m3g_md {
  m3g_md_lists L{};
  bool have_lists = false;     // m3g_md_set_lists has been called
  bool lists_filled = false;   // the list buffers, the topology and the certificate describe SOME positions of this candidate set
  bool state_valid = false;    // cand_state describes those lists
  int64_t n_edges = 0, n_triplets = 0;
  int32_t hints = 0;
  int species_ok_for = -1;     // num_types of the model the species of this list set were checked against (-1: not yet)
  uint64_t* verdict = nullptr;   // pinned, 8 words: [0..5] m3g_verlet_update_async, [6] sticky error word of the topology, [7] species verdict
};

extern "C" int m3g_md_create(m3g_md** out) {
  if (!out) { set_error("m3g_md_create: null argument"); return M3G_ERR_VALUE; }
  m3g_md* md = new m3g_md();
  hipError_t e = hipHostMalloc((void**)&md->verdict, 8 * sizeof(uint64_t), hipHostMallocDefault);
  if (e != hipSuccess) { delete md; set_error("hipHostMalloc failed: %s", hipGetErrorString(e)); return M3G_ERR_HIP; }
  *out = md;
  return M3G_OK;
}
extern "C" void m3g_md_destroy(m3g_md* md) {
  if (!md) return;
  if (md->verdict) (void)hipHostFree(md->verdict);
  delete md;
}
extern "C" int m3g_md_set_lists(m3g_md* md, const m3g_md_lists* lists) {
  if (!md || !lists) { set_error("m3g_md_set_lists: null argument"); return M3G_ERR_VALUE; }
  const m3g_md_lists& L = *lists;
  if (L.n_atoms < 1 || L.n_structs < 1 || L.n_cand < 0 || L.cap_edges < L.n_cand || L.cap_triplets < 0 || !L.pos_ref || !L.lattice || !L.batch ||
      !L.atom_types || !L.lattice32 || !L.cand_row_ptr || !L.verlet_scratch || !L.num_triplet_i || !L.pos32 || !L.topo || !L.workspace ||
      (L.n_cand > 0 && (!L.cand_edge_index || !L.cand_shift || !L.cand_state || !L.edge_index || !L.edge_cell_shift || !L.num_triplet_ij)) ||   // (no candidates: none of them is touched)
      (L.cap_triplets > 0 && !L.triplet_edge_index) || !(L.skin > 0.0) || L.threebody_cutoff > L.cutoff) {
    set_error("m3g_md_set_lists: missing buffer or bad size");
    return M3G_ERR_VALUE;
  }
  md->L = L;
  md->have_lists = true;
  md->lists_filled = false;
  md->state_valid = false;
  md->species_ok_for = -1;
  return M3G_OK;
}
// The list buffers no longer describe what the caller thinks (it has written them through another path): the next step re-derives them.
extern "C" int m3g_md_invalidate(m3g_md* md) {
  if (!md) { set_error("m3g_md_invalidate: null argument"); return M3G_ERR_VALUE; }
  md->lists_filled = false;
  md->state_valid = false;
  return M3G_OK;
}

extern "C" int m3g_md_step(m3g_md* md, const m3g_plan* plan, const double* pos, float* total_energy, float* forces, float* stresses,
                           int32_t force_refill, m3g_md_result* res, void* stream_) {
  if (!md || !plan || !pos || !total_energy || !res) { set_error("m3g_md_step: null argument"); return M3G_ERR_VALUE; }
  if (!md->have_lists) { set_error("m3g_md_step: m3g_md_set_lists has not been called"); return M3G_ERR_STATE; }
  const m3g_md_lists& L = md->L;
  hipStream_t s = (hipStream_t)stream_;
  const int64_t N = L.n_atoms, S = L.n_structs, Ec = L.n_cand;
  *res = m3g_md_result{};
  // ---- queued ahead of the skin test: pos -> fp32 (the step's input), the sticky error word of the EARLIER steps on this topology, and
  //      once per list set and model the species check -- all read behind the one wait below
  const bool topo_stands = md->lists_filled;
  const int32_t* status_word = topo_stands ? (const int32_t*)topo_carve(N, md->n_edges, md->n_triplets, S, L.topo).flags + 8 : nullptr;
  hipLaunchKernelGGL(k_md_pos32, dim3((unsigned)((3 * N + 255) / 256)), dim3(256), 0, s, 3 * N, pos, L.pos32, status_word, md->verdict + 6);
  const bool check_species = md->species_ok_for != plan->cfg.num_types;
  if (check_species) {
    md->verdict[7] = 0;
    hipLaunchKernelGGL(k_md_species, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, N, L.atom_types, plan->cfg.num_types, md->verdict + 7);
  }
  // ---- the skin test at `pos`, and the wait for its verdict (the sizes of the lists decide every launch below)
  int rc = m3g_verlet_update_async(N, S, Ec, pos, L.pos_ref, L.lattice, L.batch, L.cand_edge_index, L.cand_shift, L.cand_row_ptr, L.cutoff,
                                   (float)L.threebody_cutoff, md->state_valid ? L.cand_state : nullptr, L.verlet_scratch, L.verlet_scratch_bytes,
                                   md->verdict, s);
  if (rc) return rc;
  M3G_HIP_CHECK(hipStreamSynchronize(s));
  if (check_species) {
    if (md->verdict[7]) {
      set_error("m3g_md_step: atom_types must lie in [0, %d] (num_types = %d)", plan->cfg.num_types - 1, plan->cfg.num_types);
      return M3G_ERR_VALUE;
    }
    md->species_ok_for = plan->cfg.num_types;
  }
  if (topo_stands && md->verdict[6]) {
    set_error("m3g_md_step: an earlier step left error bits %#llx on the topology buffer (M3G_TOPO_ERR_*): its results were invalid",
              (unsigned long long)md->verdict[6]);
    md->lists_filled = false;   // the next step re-derives lists and topology (which clears the word)
    md->state_valid = false;
    return M3G_ERR_STATE;
  }
  double d2;
  memcpy(&d2, md->verdict, sizeof(double));
  const double disp = std::sqrt(d2);
  const bool changed = md->verdict[1] != 0;
  const int64_t n_e = (int64_t)md->verdict[2], n_t = (int64_t)md->verdict[3], max_row = (int64_t)md->verdict[5];
  res->max_displacement = disp;
  res->n_edges = n_e;
  res->n_triplets = n_t;
  if (disp >= 0.5 * L.skin && disp != 0.0) {   // an atom has left its skin: the candidates must be searched again (the caller's job)
    res->path = M3G_MD_NEED_SEARCH;
    return M3G_OK;
  }
  const bool refill = changed || !md->lists_filled || force_refill != 0;
  if (refill) {
    size_t need_topo = 0, need_work = 0;
    if (max_row > M3G_VERLET_FILL_LISTS_MAX_ROW || N > 262144 || n_e > L.cap_edges || n_t > L.cap_triplets ||
        m3g_topology_bytes(N, n_e, n_t, S, &need_topo) != M3G_OK || need_topo > L.topo_bytes ||
        m3g_workspace_bytes(plan, N, n_e, n_t, S, &need_work) != M3G_OK || need_work > L.workspace_bytes) {
      res->path = M3G_MD_UNSUPPORTED;   // very long candidate rows, or lists beyond the buffers' capacity: the general calls
      md->lists_filled = false;
      md->state_valid = false;
      return M3G_OK;
    }
    rc = m3g_verlet_fill_lists(N, Ec, n_e, n_t, max_row, L.verlet_scratch, L.cand_edge_index, L.cand_shift, L.cand_row_ptr, L.edge_index,
                               L.edge_cell_shift, L.cand_state, L.triplet_edge_index, L.num_triplet_i, L.num_triplet_ij, s);
    if (rc) return rc;
    md->state_valid = true;
    md->lists_filled = false;   // (until the topology below stands)
    int32_t flags = 0, hints = 0;
    rc = m3g_topology_build_canonical(N, n_e, n_t, S, L.edge_index, L.triplet_edge_index, L.batch, L.topo, L.topo_bytes, &flags, &hints, s);
    if (rc) return rc;
    if (flags) { set_error("m3g_md_step: the library's own lists failed the topology checks (flags %d)", flags); return M3G_ERR_STATE; }
    md->n_edges = n_e;
    md->n_triplets = n_t;
    md->hints = hints;
    md->lists_filled = true;
  }
  // ---- the step on the lists that stand
  if (!refill) {   // (engine options that enlarge the workspace may have changed since the buffers were made: the caller makes new ones)
    size_t need_work = 0;
    if (m3g_workspace_bytes(plan, N, md->n_edges, md->n_triplets, S, &need_work) != M3G_OK || need_work > L.workspace_bytes) {
      res->path = M3G_MD_UNSUPPORTED;
      md->lists_filled = false;
      md->state_valid = false;
      return M3G_OK;
    }
  }
  m3g_io io{};
  io.n_atoms = N; io.n_edges = md->n_edges; io.n_triplets = md->n_triplets; io.n_structs = S;
  io.pos = L.pos32; io.atom_types = L.atom_types; io.edge_cell_shift = L.edge_cell_shift; io.lattice = L.lattice32; io.topo = L.topo;
  io.triplet_edge_index = L.triplet_edge_index;
  io.total_energy = total_energy; io.forces = forces; io.stresses = forces ? stresses : nullptr;
  io.topo_hints = md->hints;
  rc = m3g_energy_forces(plan, &io, L.workspace, L.workspace_bytes, s);
  if (rc) return rc;
  res->path = refill ? M3G_MD_REFILL : M3G_MD_REUSE;
  res->n_edges = md->n_edges;
  res->n_triplets = md->n_triplets;
  res->topo_hints = md->hints;
  return M3G_OK;
}
