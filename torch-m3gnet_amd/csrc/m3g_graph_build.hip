// Graph construction on the GPU: periodic neighbour list and three-body index
// enumeration, producing the reference's MaterialGraph index tensors directly in HBM.
//   * m3g_neighbor_*   replaces get_all_neighbors_with_cell_shifts (data/material_graph.py:168-193, pymatgen's
//                      Structure.get_all_neighbors): full directed list, self-images included, integer cell shifts.
//   * m3g_threebody_*  replaces compute_threebody (data/material_graph.py:196-254, an O(T) Python loop):
//                      every ordered pair (e1, e2), e1 != e2, of edges with d <= threebody_cutoff sharing a centre,
//                      in exactly the reference's order.
// Canonical edge order (the reference leaves the order inside a centre unspecified): centre atom, then image shift
// (sx, sy, sz) lexicographic, then neighbour index -- the same order torch_m3gnet/data/neighbors.py produces, so
// both builders can be compared element by element.
// Geometry in fp64 like pymatgen (inclusion d <= cutoff is decided in double; the stored tensors are narrowed by
// the caller).  Linked cells: atoms are radix-sorted (hipCUB) into bins at least one cutoff wide, and one thread per
// (atom, periodic image) tests only the bins of that image within reach -- O(N) pair tests for large cells, and
// the plain all-images x all-atoms search when the cell is smaller than the cutoff (one bin).  Integer/byte work,
// L2-bound.  Count -> exclusive scan (hipCUB) -> fill, so no atomics and a deterministic result.
#include <hipcub/hipcub.hpp>

#include "m3g_internal.h"

namespace m3g {

static inline size_t align_up_g(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct StructInfo {   // per structure, device
  double lat[9];      // rows
  double inv[9];      // inverse (frac = cart . inv)
  int reps[3];        // images needed along each lattice vector
  int nb[3];          // linked-cell bins along each lattice vector (bin width >= cutoff, or 1 bin when the cell is smaller)
  int reach[3];       // bins to look at either side: 1, or reps when the axis has a single bin
  int n_img;          // (2 rx + 1)(2 ry + 1)(2 rz + 1)
  int first, count;   // atom range
};

struct NbScratch {
  StructInfo* info;       // [S]
  int64_t* bin_off;       // [S+1] bins per structure, then exclusive offsets (global bin id = bin_off[s] + local bin)
  double* pos_w;          // [N,3] wrapped into the home cell
  int32_t* wrap;          // [N,3] integer offset removed by the wrap
  int32_t* binc;          // [N,3] bin coordinates of each atom
  int32_t *bin_key, *bin_key_s, *iota, *perm;   // [N] global bin id per atom, sorted copy, atom ids, atoms ordered by bin
  int32_t* bin_start;     // [Bmax+1] first slot of each bin in perm
  double* pos_s;          // [N,3] wrapped positions in bin order
  int64_t* counts;        // [N*M + 1] matches per (atom, image), then exclusive offsets
  void* tmp;              // scan / sort temporary
  size_t tmp_bytes;
  int64_t max_bins;
  size_t total;
};

static NbScratch nb_carve(int64_t N, int64_t S, int64_t M, void* base) {
  NbScratch w{};
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = p ? (void*)(p + off) : nullptr; off += align_up_g(bytes); return r; };
  w.max_bins = 2 * N + 8 * S;   // k_struct_info keeps every structure within 2 count + 8 bins
  w.info = (StructInfo*)take(sizeof(StructInfo) * (size_t)(S + 1));
  w.bin_off = (int64_t*)take(sizeof(int64_t) * (size_t)(S + 2));
  w.pos_w = (double*)take(sizeof(double) * 3 * (size_t)(N + 1));
  w.wrap = (int32_t*)take(sizeof(int32_t) * 3 * (size_t)(N + 1));
  w.binc = (int32_t*)take(sizeof(int32_t) * 3 * (size_t)(N + 1));
  w.bin_key = (int32_t*)take(sizeof(int32_t) * (size_t)(N + 1));
  w.bin_key_s = (int32_t*)take(sizeof(int32_t) * (size_t)(N + 1));
  w.iota = (int32_t*)take(sizeof(int32_t) * (size_t)(N + 1));
  w.perm = (int32_t*)take(sizeof(int32_t) * (size_t)(N + 1));
  w.bin_start = (int32_t*)take(sizeof(int32_t) * (size_t)(w.max_bins + 2));
  w.pos_s = (double*)take(sizeof(double) * 3 * (size_t)(N + 1));
  w.counts = (int64_t*)take(sizeof(int64_t) * (size_t)(N * M + 2));
  size_t t1 = 0, t2 = 0, t3 = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t1, (const int64_t*)nullptr, (int64_t*)nullptr, (int)(N * M + 1));
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t2, (const int64_t*)nullptr, (int64_t*)nullptr, (int)(S + 1));
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t3, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr,
                                           (int32_t*)nullptr, (int)N);
  w.tmp_bytes = std::max(t1, std::max(t2, t3));
  w.tmp = take(w.tmp_bytes);
  w.total = off;
  return w;
}

// one thread per structure: inverse lattice, image ranges, bin grid, atom range (batch must be sorted/contiguous)
__global__ void k_struct_info(int64_t N, int64_t S, const double* __restrict__ lattice, const int64_t* __restrict__ batch,
                              double cutoff, StructInfo* info, int64_t* nbins, int* flags) {
  int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (s == 0) nbins[S] = 0;
  if (s >= S) return;
  StructInfo si;
  const double* L = lattice + s * 9;
  for (int k = 0; k < 9; ++k) si.lat[k] = L[k];
  const double c0x = L[4] * L[8] - L[5] * L[7], c0y = L[5] * L[6] - L[3] * L[8], c0z = L[3] * L[7] - L[4] * L[6];  // a1 x a2
  const double c1x = L[7] * L[2] - L[8] * L[1], c1y = L[8] * L[0] - L[6] * L[2], c1z = L[6] * L[1] - L[7] * L[0];  // a2 x a0
  const double c2x = L[1] * L[5] - L[2] * L[4], c2y = L[2] * L[3] - L[0] * L[5], c2z = L[0] * L[4] - L[1] * L[3];  // a0 x a1
  const double det = L[0] * c0x + L[1] * c0y + L[2] * c0z;
  const double vol = fabs(det);
  // inverse: columns are the cross products / det  (frac_p = cart . inv[:, p])
  si.inv[0] = c0x / det; si.inv[3] = c0y / det; si.inv[6] = c0z / det;
  si.inv[1] = c1x / det; si.inv[4] = c1y / det; si.inv[7] = c1z / det;
  si.inv[2] = c2x / det; si.inv[5] = c2y / det; si.inv[8] = c2z / det;
  // atom range by binary search on the sorted batch vector
  int64_t lo = 0, hi = N;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (batch[mid] < s) lo = mid + 1; else hi = mid; }
  si.first = (int)lo;
  hi = N;
  int64_t lo2 = lo;
  while (lo2 < hi) { int64_t mid = (lo2 + hi) >> 1; if (batch[mid] < s + 1) lo2 = mid + 1; else hi = mid; }
  si.count = (int)(lo2 - lo);
  // images needed along lattice vector p: ceil(cutoff / height_p), height_p = V / |a_q x a_r|;
  // bins along p: as many as fit with width >= cutoff (so a neighbour is at most one bin away)
  const double area[3] = {sqrt(c0x * c0x + c0y * c0y + c0z * c0z), sqrt(c1x * c1x + c1y * c1y + c1z * c1z),
                          sqrt(c2x * c2x + c2y * c2y + c2z * c2z)};
  si.n_img = 1;
  for (int p = 0; p < 3; ++p) {
    si.reps[p] = (int)ceil((cutoff + 1e-8) * area[p] / vol);
    si.n_img *= 2 * si.reps[p] + 1;
    const double nbf = floor(vol / (area[p] * (cutoff + 2e-8)));
    si.nb[p] = nbf >= 1.0 ? (nbf > 1024.0 ? 1024 : (int)nbf) : 1;
  }
  while ((int64_t)si.nb[0] * si.nb[1] * si.nb[2] > 2 * (int64_t)si.count + 8) {   // empty space: coarser bins, never finer
    int p = si.nb[0] >= si.nb[1] ? (si.nb[0] >= si.nb[2] ? 0 : 2) : (si.nb[1] >= si.nb[2] ? 1 : 2);
    si.nb[p] = (si.nb[p] + 1) / 2;
  }
  for (int p = 0; p < 3; ++p) si.reach[p] = si.nb[p] == 1 ? si.reps[p] : 1;
  info[s] = si;
  nbins[s] = (int64_t)si.nb[0] * si.nb[1] * si.nb[2];
  if (!(vol > 0.0)) atomicOr(flags, 1);
}

__global__ void k_wrap_positions(int64_t N, int64_t S, const double* __restrict__ pos, const int64_t* __restrict__ batch,
                                 const StructInfo* __restrict__ info, const int64_t* __restrict__ bin_off, double* pos_w,
                                 int32_t* wrap, int32_t* binc, int32_t* bin_key, int32_t* iota, int* flags) {
  int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (a >= N) return;
  int64_t s = batch[a];
  if (s < 0 || s >= S || (a > 0 && batch[a - 1] > s)) { atomicOr(flags, 2); s = 0; }
  const StructInfo& si = info[s];
  const double x = pos[a * 3], y = pos[a * 3 + 1], z = pos[a * 3 + 2];
  double f[3], w[3];
  int b[3];
  for (int p = 0; p < 3; ++p) {
    f[p] = x * si.inv[0 + p] + y * si.inv[3 + p] + z * si.inv[6 + p];
    w[p] = floor(f[p]);
    wrap[a * 3 + p] = (int32_t)w[p];
    f[p] -= w[p];
    b[p] = min(si.nb[p] - 1, max(0, (int)(f[p] * si.nb[p])));
    binc[a * 3 + p] = b[p];
  }
  for (int c = 0; c < 3; ++c) pos_w[a * 3 + c] = f[0] * si.lat[0 + c] + f[1] * si.lat[3 + c] + f[2] * si.lat[6 + c];
  bin_key[a] = (int32_t)(bin_off[s] + ((int64_t)b[0] * si.nb[1] + b[1]) * si.nb[2] + b[2]);
  iota[a] = (int32_t)a;
}

// bin_start[g] = first slot of bin g in the bin-sorted atom list; also gathers the wrapped positions in that order
__global__ void k_bin_ranges(int64_t N, int64_t S, const int64_t* __restrict__ bin_off, const int32_t* __restrict__ keys_sorted,
                             const int32_t* __restrict__ perm, const double* __restrict__ pos_w, int32_t* bin_start, double* pos_s) {
  int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (g < N) {
    const int64_t a = perm[g];
    pos_s[g * 3] = pos_w[a * 3]; pos_s[g * 3 + 1] = pos_w[a * 3 + 1]; pos_s[g * 3 + 2] = pos_w[a * 3 + 2];
  }
  if (g > bin_off[S]) return;
  int64_t lo = 0, hi = N;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (keys_sorted[mid] < g) lo = mid + 1; else hi = mid; }
  bin_start[g] = (int32_t)lo;
}

// One thread per (atom i, image).  The thread walks the bins of that image that can hold a neighbour (at most
// 3x3x3, or the single bin of a small cell for every image), counting (FILL = false) or writing (FILL = true)
// its matches into the slot range the exclusive scan assigned; the fill pass then orders its own short segment
// by neighbour index so the result is canonical and independent of the bin traversal.
template <bool FILL>
__global__ void __launch_bounds__(256) k_neighbors(int64_t N, int64_t M, const int64_t* __restrict__ batch,
                                                   const StructInfo* __restrict__ info, const int64_t* __restrict__ bin_off,
                                                   const int32_t* __restrict__ bin_start, const int32_t* __restrict__ perm,
                                                   const double* __restrict__ pos_s, const double* __restrict__ pos_w,
                                                   const int32_t* __restrict__ binc, const int32_t* __restrict__ wrap, double cutoff,
                                                   int64_t* __restrict__ counts, int64_t E, int64_t* __restrict__ edge_index,
                                                   int32_t* __restrict__ shift, double* __restrict__ dist) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= N * M) return;
  const int64_t i = t / M;
  const int img = (int)(t % M);
  const int s = (int)batch[i];
  const StructInfo& si = info[s];
  if (img >= si.n_img) { if (!FILL) counts[t] = 0; return; }
  const int ny = 2 * si.reps[1] + 1, nz = 2 * si.reps[2] + 1;
  const int sh[3] = {img / (ny * nz) - si.reps[0], (img / nz) % ny - si.reps[1], img % nz - si.reps[2]};
  int lo[3], hi[3];
  bool any = true;
  for (int p = 0; p < 3; ++p) {   // bins b' of the image whose unwrapped coordinate b' + s nb is within reach of this atom's bin
    const int b = binc[i * 3 + p];
    lo[p] = max(0, b - si.reach[p] - sh[p] * si.nb[p]);
    hi[p] = min(si.nb[p] - 1, b + si.reach[p] - sh[p] * si.nb[p]);
    any = any && lo[p] <= hi[p];
  }
  // image displacement minus this atom's wrapped position: |pos_w[j] + o| is the pair distance
  const double ox = sh[0] * si.lat[0] + sh[1] * si.lat[3] + sh[2] * si.lat[6] - pos_w[i * 3];
  const double oy = sh[0] * si.lat[1] + sh[1] * si.lat[4] + sh[2] * si.lat[7] - pos_w[i * 3 + 1];
  const double oz = sh[0] * si.lat[2] + sh[1] * si.lat[5] + sh[2] * si.lat[8] - pos_w[i * 3 + 2];
  const double c2 = (cutoff + 1e-8) * (cutoff + 1e-8);
  const int64_t begin = FILL ? counts[t] : 0;
  int64_t out = begin;
  if (any) {
    const int64_t b0 = bin_off[s];
    for (int bx = lo[0]; bx <= hi[0]; ++bx)
      for (int by = lo[1]; by <= hi[1]; ++by) {
        const int64_t g0 = b0 + ((int64_t)bx * si.nb[1] + by) * si.nb[2];
        const int k0 = bin_start[g0 + lo[2]], k1 = bin_start[g0 + hi[2] + 1];   // bins along z are consecutive slots
        for (int k = k0; k < k1; ++k) {
          const double dx = pos_s[(int64_t)k * 3] + ox, dy = pos_s[(int64_t)k * 3 + 1] + oy, dz = pos_s[(int64_t)k * 3 + 2] + oz;
          const double d2 = dx * dx + dy * dy + dz * dz;
          if (d2 <= c2 && d2 > 1e-16) {
            if (FILL && out < E) { edge_index[E + out] = perm[k]; dist[out] = sqrt(d2); }
            ++out;
          }
        }
      }
  }
  if (!FILL) { counts[t] = out; return; }
  const int64_t end = out < E ? out : E;
  for (int64_t a = begin + 1; a < end; ++a) {   // insertion sort of this (atom, image) segment by neighbour index
    const int64_t j = edge_index[E + a];
    const double d = dist[a];
    int64_t b = a;
    while (b > begin && edge_index[E + b - 1] > j) { edge_index[E + b] = edge_index[E + b - 1]; dist[b] = dist[b - 1]; --b; }
    edge_index[E + b] = j; dist[b] = d;
  }
  for (int64_t a = begin; a < end; ++a) {
    const int64_t j = edge_index[E + a];
    edge_index[a] = i;
    for (int p = 0; p < 3; ++p) shift[a * 3 + p] = sh[p] - wrap[j * 3 + p] + wrap[i * 3 + p];
  }
}

// ---- three-body ------------------------------------------------------------------------------------------------
struct TbScratch {
  int32_t* rank;      // [E] rank of a valid edge inside its centre's valid list, -1 for invalid edges
  int32_t* deg;       // [N] valid edges per centre
  int32_t* row_ptr;   // [N+1]
  int64_t* counts;    // [E+1] triplets per edge, then exclusive offsets
  void* scan_tmp;
  size_t scan_tmp_bytes;
  size_t total;
};
static TbScratch tb_carve(int64_t N, int64_t E, void* base) {
  TbScratch w{};
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = p ? (void*)(p + off) : nullptr; off += align_up_g(bytes); return r; };
  w.rank = (int32_t*)take(sizeof(int32_t) * (size_t)(E + 1));
  w.deg = (int32_t*)take(sizeof(int32_t) * (size_t)(N + 1));
  w.row_ptr = (int32_t*)take(sizeof(int32_t) * (size_t)(N + 2));
  w.counts = (int64_t*)take(sizeof(int64_t) * (size_t)(E + 2));
  size_t tmp = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, (const int64_t*)nullptr, (int64_t*)nullptr, (int)(E + 1));
  w.scan_tmp_bytes = tmp;
  w.scan_tmp = take(tmp);
  w.total = off;
  return w;
}

__global__ void k_rows_from_sorted(int64_t N, int64_t E, const int64_t* __restrict__ src, int32_t* row_ptr, int* flags) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r > N) return;
  int64_t lo = 0, hi = E;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (src[mid] < r) lo = mid + 1; else hi = mid; }
  row_ptr[r] = (int32_t)lo;
  if (r < N && lo < E && lo > 0 && src[lo - 1] > src[lo]) atomicOr(flags, 1);
}

// one thread per centre: rank its valid edges (d <= threebody_cutoff, decided on the fp32 distances like the reference)
__global__ void k_rank_valid(int64_t N, const int32_t* __restrict__ row_ptr, const float* __restrict__ dist, float tb_cutoff,
                             int32_t* rank, int32_t* deg, int64_t* counts) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  int d = 0;
  for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) rank[e] = dist[e] <= tb_cutoff ? d++ : -1;
  deg[i] = d;
  for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) counts[e] = rank[e] >= 0 ? d - 1 : 0;
}

// one thread per edge: write its deg-1 triplets in the reference's order (partners in edge order, itself skipped)
__global__ void k_fill_triplets(int64_t N, int64_t E, int64_t T, const int64_t* __restrict__ src, const int32_t* __restrict__ row_ptr,
                                const int32_t* __restrict__ rank, const int32_t* __restrict__ deg,
                                const int64_t* __restrict__ offsets, int64_t* __restrict__ tei, int64_t* __restrict__ num_triplet_i,
                                int32_t* __restrict__ num_triplet_ij) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e < N && num_triplet_i) { int64_t d = deg[e]; num_triplet_i[e] = d * (d - 1); }
  if (e >= E) return;
  const int64_t i = src[e];
  const bool valid = rank[e] >= 0;
  if (num_triplet_ij) num_triplet_ij[e] = valid ? deg[i] - 1 : 0;
  if (!valid) return;
  int64_t out = offsets[e];
  for (int f = row_ptr[i]; f < row_ptr[i + 1]; ++f) {
    if (f == e || rank[f] < 0) continue;
    if (out < T) { tei[out] = e; tei[T + out] = f; }
    ++out;
  }
}

}  // namespace m3g

using namespace m3g;

static inline dim3 g_for(int64_t n, int tpb = 256) { return dim3((unsigned)((n + tpb - 1) / tpb)); }

extern "C" int m3g_neighbor_scratch_bytes(int64_t N, int64_t S, int64_t max_images, size_t* bytes) {
  if (!bytes || N < 0 || S < 0 || max_images < 1) { set_error("m3g_neighbor_scratch_bytes: bad argument"); return M3G_ERR_VALUE; }
  if (N * max_images >= (int64_t(1) << 31)) { set_error("neighbour search too large: atoms x images >= 2^31"); return M3G_ERR_UNSUPPORTED; }
  *bytes = nb_carve(N, S, max_images, nullptr).total + 256;
  return M3G_OK;
}

// Phase 1: counts.  host_n_edges receives E after an internal stream synchronisation (graph construction is not a
// hot call).  max_images = upper bound of (2rx+1)(2ry+1)(2rz+1) over the structures (the Python host computes it).
extern "C" int m3g_neighbor_count(int64_t N, int64_t S, int64_t max_images, const double* pos, const double* lattice,
                                  const int64_t* batch, double cutoff, void* scratch, size_t scratch_bytes,
                                  int64_t* host_n_edges, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  size_t need = 0;
  int rc = m3g_neighbor_scratch_bytes(N, S, max_images, &need);
  if (rc) return rc;
  if (!scratch || scratch_bytes < need || !host_n_edges) { set_error("m3g_neighbor_count: scratch too small or null argument"); return M3G_ERR_SIZE; }
  NbScratch w = nb_carve(N, S, max_images, scratch);
  int* flags = (int*)((char*)scratch + w.total);
  M3G_HIP_CHECK(hipMemsetAsync(flags, 0, sizeof(int), s));
  *host_n_edges = 0;
  if (N == 0 || S == 0) return M3G_OK;
  hipLaunchKernelGGL(k_struct_info, g_for(S), dim3(256), 0, s, N, S, lattice, batch, cutoff, w.info, w.bin_off, flags);
  M3G_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(w.tmp, w.tmp_bytes, w.bin_off, w.bin_off, (int)(S + 1), s));
  hipLaunchKernelGGL(k_wrap_positions, g_for(N), dim3(256), 0, s, N, S, pos, batch, w.info, w.bin_off, w.pos_w, w.wrap, w.binc, w.bin_key,
                     w.iota, flags);
  M3G_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(w.tmp, w.tmp_bytes, w.bin_key, w.bin_key_s, w.iota, w.perm, (int)N, 0, 32, s));
  hipLaunchKernelGGL(k_bin_ranges, g_for(w.max_bins + 1), dim3(256), 0, s, N, S, w.bin_off, w.bin_key_s, w.perm, w.pos_w, w.bin_start, w.pos_s);
  const int64_t NM = N * max_images;
  hipLaunchKernelGGL((k_neighbors<false>), g_for(NM), dim3(256), 0, s, N, max_images, batch, w.info, w.bin_off, w.bin_start, w.perm, w.pos_s,
                     w.pos_w, w.binc, w.wrap, cutoff, w.counts, (int64_t)0, nullptr, nullptr, nullptr);
  M3G_HIP_CHECK(hipMemsetAsync(w.counts + NM, 0, sizeof(int64_t), s));
  M3G_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(w.tmp, w.tmp_bytes, w.counts, w.counts, (int)(NM + 1), s));
  int h_flags = 0;
  M3G_HIP_CHECK(hipMemcpyAsync(host_n_edges, w.counts + NM, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipMemcpyAsync(&h_flags, flags, sizeof(int), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipStreamSynchronize(s));
  if (h_flags & 1) { set_error("singular lattice"); return M3G_ERR_VALUE; }
  if (h_flags & 2) { set_error("batch vector must be sorted, contiguous per structure and within [0, n_structs)"); return M3G_ERR_VALUE; }
  return M3G_OK;
}

// Phase 2: fill (same scratch, untouched since the count call).
extern "C" int m3g_neighbor_fill(int64_t N, int64_t S, int64_t max_images, const int64_t* batch, double cutoff, void* scratch,
                                 int64_t n_edges, int64_t* edge_index, int32_t* edge_cell_shift, double* distances, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (n_edges == 0 || N == 0) return M3G_OK;
  if (!scratch || !edge_index || !edge_cell_shift || !distances) { set_error("m3g_neighbor_fill: null argument"); return M3G_ERR_VALUE; }
  NbScratch w = nb_carve(N, S, max_images, scratch);
  hipLaunchKernelGGL((k_neighbors<true>), g_for(N * max_images), dim3(256), 0, s, N, max_images, batch, w.info, w.bin_off, w.bin_start, w.perm,
                     w.pos_s, w.pos_w, w.binc, w.wrap, cutoff, w.counts, n_edges, edge_index, edge_cell_shift, distances);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}

extern "C" int m3g_threebody_scratch_bytes(int64_t N, int64_t E, size_t* bytes) {
  if (!bytes || N < 0 || E < 0) { set_error("m3g_threebody_scratch_bytes: bad argument"); return M3G_ERR_VALUE; }
  if (E >= (int64_t(1) << 31) - 2) { set_error("too many edges for int32 row pointers"); return M3G_ERR_UNSUPPORTED; }
  *bytes = tb_carve(N, E, nullptr).total + 256;
  return M3G_OK;
}

extern "C" int m3g_threebody_count(int64_t N, int64_t E, const int64_t* edge_index, const float* distances, float threebody_cutoff,
                                   void* scratch, size_t scratch_bytes, int64_t* host_n_triplets, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  size_t need = 0;
  int rc = m3g_threebody_scratch_bytes(N, E, &need);
  if (rc) return rc;
  if (!scratch || scratch_bytes < need || !host_n_triplets) { set_error("m3g_threebody_count: scratch too small or null argument"); return M3G_ERR_SIZE; }
  *host_n_triplets = 0;
  if (N == 0) return M3G_OK;
  TbScratch w = tb_carve(N, E, scratch);
  int* flags = (int*)((char*)scratch + w.total);
  M3G_HIP_CHECK(hipMemsetAsync(flags, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_rows_from_sorted, g_for(N + 1), dim3(256), 0, s, N, E, edge_index, w.row_ptr, flags);
  hipLaunchKernelGGL(k_rank_valid, g_for(N), dim3(256), 0, s, N, w.row_ptr, distances, threebody_cutoff, w.rank, w.deg, w.counts);
  M3G_HIP_CHECK(hipMemsetAsync(w.counts + E, 0, sizeof(int64_t), s));
  M3G_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(w.scan_tmp, w.scan_tmp_bytes, w.counts, w.counts, (int)(E + 1), s));
  int h_flags = 0;
  M3G_HIP_CHECK(hipMemcpyAsync(host_n_triplets, w.counts + E, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipMemcpyAsync(&h_flags, flags, sizeof(int), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipStreamSynchronize(s));
  if (h_flags & 1) { set_error("edge_index must be sorted by centre atom (row 0)"); return M3G_ERR_VALUE; }
  return M3G_OK;
}

extern "C" int m3g_threebody_fill(int64_t N, int64_t E, const int64_t* edge_index, void* scratch, int64_t n_triplets,
                                  int64_t* triplet_edge_index, int64_t* num_triplet_i, int32_t* num_triplet_ij, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (N == 0) return M3G_OK;
  if (!scratch || (n_triplets > 0 && !triplet_edge_index)) { set_error("m3g_threebody_fill: null argument"); return M3G_ERR_VALUE; }
  TbScratch w = tb_carve(N, E, scratch);
  hipLaunchKernelGGL(k_fill_triplets, g_for(std::max(E, N)), dim3(256), 0, s, N, E, n_triplets, edge_index, w.row_ptr, w.rank, w.deg,
                     w.counts, triplet_edge_index, num_triplet_i, num_triplet_ij);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}
