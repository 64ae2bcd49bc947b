// Graph construction on the GPU: periodic neighbour list and three-body index
// enumeration, producing the reference's MaterialGraph index tensors directly in HBM.
//   * m3g_neighbor_*   replaces get_all_neighbors_with_cell_shifts (data/material_graph.py:168-193, pymatgen's
//                      Structure.get_all_neighbors): full directed list, self-images included, integer cell shifts.
//   * m3g_threebody_*  replaces compute_threebody (data/material_graph.py:196-254, an O(T) Python loop):
//                      every ordered pair (e1, e2), e1 != e2, of edges with d <= threebody_cutoff sharing a centre,
//                      in exactly the reference's order.
// Canonical edge order (the reference leaves the order inside a centre unspecified): centre atom, then the edge's cell shift
// (sx, sy, sz) lexicographic -- the shift that refers to the GIVEN coordinates, so the order does not depend on which atoms sit
// outside the home cell: an unwrapped MD trajectory keeps its edge order while atoms cross cell faces, which is what lets a
// skin list (m3g_verlet_*) reproduce a fresh build bit for bit --, then neighbour index: the same order
// torch_m3gnet/data/neighbors.py produces, so both builders can be compared element by element.
// Geometry in fp64 like pymatgen (inclusion d <= cutoff is decided in double; the stored tensors are narrowed by
// the caller).  Linked cells: atoms are put into bins at least one cutoff wide by a counting sort (per-bin integer counters, one
// scan, one scatter), and one wave per atom walks its periodic images, testing only the bins of an image within reach -- O(N)
// pair tests for large cells, and the plain all-images x all-atoms search when the cell is smaller than the cutoff (one bin).
// Integer/byte work, L2-bound.  Count -> exclusive scan (m3g_prims.h) -> fill: no float atomics and a deterministic result.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "m3g_internal.h"
#include "m3g_prims.h"

namespace m3g {

static inline size_t align_up_g(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Geometry shared by the search and by the skin-list update (m3g_verlet_*): the update must classify a pair exactly as a fresh
// search would, so both go through these functions, compiled WITHOUT floating-point contraction (an fma formed in one caller and
// not in the other would let the two disagree on a pair that sits on the cutoff sphere to within rounding).
struct LatticeFrame { double lat[9], inv[9], cross[9], det; };
__device__ __forceinline__ LatticeFrame lattice_frame(const double* L) {
#pragma clang fp contract(off)
  LatticeFrame f;
  for (int k = 0; k < 9; ++k) f.lat[k] = L[k];
  f.cross[0] = L[4] * L[8] - L[5] * L[7]; f.cross[1] = L[5] * L[6] - L[3] * L[8]; f.cross[2] = L[3] * L[7] - L[4] * L[6];  // a1 x a2
  f.cross[3] = L[7] * L[2] - L[8] * L[1]; f.cross[4] = L[8] * L[0] - L[6] * L[2]; f.cross[5] = L[6] * L[1] - L[7] * L[0];  // a2 x a0
  f.cross[6] = L[1] * L[5] - L[2] * L[4]; f.cross[7] = L[2] * L[3] - L[0] * L[5]; f.cross[8] = L[0] * L[4] - L[1] * L[3];  // a0 x a1
  f.det = L[0] * f.cross[0] + L[1] * f.cross[1] + L[2] * f.cross[2];
  // inverse: columns are the cross products / det  (frac_p = cart . inv[:, p])
  f.inv[0] = f.cross[0] / f.det; f.inv[3] = f.cross[1] / f.det; f.inv[6] = f.cross[2] / f.det;
  f.inv[1] = f.cross[3] / f.det; f.inv[4] = f.cross[4] / f.det; f.inv[7] = f.cross[5] / f.det;
  f.inv[2] = f.cross[6] / f.det; f.inv[5] = f.cross[7] / f.det; f.inv[8] = f.cross[8] / f.det;
  return f;
}
// fractional coordinates of (x, y, z): integer part `w` (the wrap), remainder `f`, and the wrapped cartesian position `pw`
__device__ __forceinline__ void wrap_point(const double* lat, const double* inv, double x, double y, double z, double (&f)[3], double (&w)[3], double (&pw)[3]) {
#pragma clang fp contract(off)
  for (int p = 0; p < 3; ++p) {
    f[p] = x * inv[0 + p] + y * inv[3 + p] + z * inv[6 + p];
    w[p] = floor(f[p]);
    f[p] -= w[p];
  }
  for (int c = 0; c < 3; ++c) pw[c] = f[0] * lat[0 + c] + f[1] * lat[3 + c] + f[2] * lat[6 + c];
}
// displacement of image `sh` of the cell minus the wrapped position of the centre: |pos_w[j] + o| is the pair distance
__device__ __forceinline__ void image_offset(const double* lat, const int (&sh)[3], const double* pos_w_i, double& ox, double& oy, double& oz) {
#pragma clang fp contract(off)
  ox = sh[0] * lat[0] + sh[1] * lat[3] + sh[2] * lat[6] - pos_w_i[0];
  oy = sh[0] * lat[1] + sh[1] * lat[4] + sh[2] * lat[7] - pos_w_i[1];
  oz = sh[0] * lat[2] + sh[1] * lat[5] + sh[2] * lat[8] - pos_w_i[2];
}
__device__ __forceinline__ double pair_d2(const double* pos_w_j, double ox, double oy, double oz) {
#pragma clang fp contract(off)
  const double dx = pos_w_j[0] + ox, dy = pos_w_j[1] + oy, dz = pos_w_j[2] + oz;
  return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ bool pair_hit(double d2, double c2) { return d2 <= c2 && d2 > 1e-16; }
__device__ __forceinline__ double cutoff_sq(double cutoff) {
#pragma clang fp contract(off)
  return (cutoff + 1e-8) * (cutoff + 1e-8);
}

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
  int32_t *bin_key, *bin_rank, *perm;   // [N] global bin id per atom, its arrival rank inside that bin, atoms ordered by bin
  int32_t* bin_start;     // [Bmax+2] atoms per bin (counted by k_wrap_positions), then the first slot of each bin in perm
  double* pos_s;          // [N,3] wrapped positions in bin order
  int64_t* counts;        // [N*M + 1] matches per (atom, image), then exclusive offsets
  int64_t* tri;           // [N + 1] triplets per centre d (d - 1) (m3g_neighbor_count_triplets), [N] = their sum
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
  w.bin_rank = (int32_t*)take(sizeof(int32_t) * (size_t)(N + 1));
  w.perm = (int32_t*)take(sizeof(int32_t) * (size_t)(N + 1));
  w.bin_start = (int32_t*)take(sizeof(int32_t) * (size_t)(w.max_bins + 2));
  w.pos_s = (double*)take(sizeof(double) * 3 * (size_t)(N + 1));
  w.counts = (int64_t*)take(sizeof(int64_t) * (size_t)(N * M + 2));
  w.tri = (int64_t*)take(sizeof(int64_t) * (size_t)(N + 2));
  // scratch of the three scans (counts per (atom, image); bins per structure; atoms per bin), m3g_prims.h
  w.tmp_bytes = std::max(std::max(prims::scan_tmp_bytes<int64_t>(N * M + 1), prims::scan_tmp_bytes<int64_t>(S + 1)),
                         prims::scan_tmp_bytes<int32_t>(w.max_bins + 2));
  w.tmp = take(w.tmp_bytes);
  w.total = off;
  return w;
}

// one thread per structure: inverse lattice, image ranges, bin grid, atom range (batch must be sorted/contiguous)
__global__ void k_struct_info(int64_t N, int64_t S, const double* __restrict__ lattice, const int64_t* __restrict__ batch,
                              double cutoff, StructInfo* info, int64_t* nbins, int* flags, int64_t* tri_total) {
  int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (s == 0) { nbins[S] = 0; *tri_total = 0; }
  if (s >= S) return;
  StructInfo si;
  const double* L = lattice + s * 9;
  const LatticeFrame fr = lattice_frame(L);
  for (int k = 0; k < 9; ++k) { si.lat[k] = fr.lat[k]; si.inv[k] = fr.inv[k]; }
  const double c0x = fr.cross[0], c0y = fr.cross[1], c0z = fr.cross[2], c1x = fr.cross[3], c1y = fr.cross[4], c1z = fr.cross[5],
               c2x = fr.cross[6], c2y = fr.cross[7], c2z = fr.cross[8];
  const double det = fr.det;
  const double vol = fabs(det);
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
                                 int32_t* wrap, int32_t* binc, int32_t* bin_key, int32_t* bin_cnt, int32_t* bin_rank, int* flags) {
  int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (a >= N) return;
  int64_t s = batch[a];
  if (s < 0 || s >= S || (a > 0 && batch[a - 1] > s)) { atomicOr(flags, 2); s = 0; }
  const StructInfo& si = info[s];
  const double x = pos[a * 3], y = pos[a * 3 + 1], z = pos[a * 3 + 2];
  double f[3], w[3], pw[3];
  int b[3];
  wrap_point(si.lat, si.inv, x, y, z, f, w, pw);
  for (int p = 0; p < 3; ++p) {
    wrap[a * 3 + p] = (int32_t)w[p];
    b[p] = min(si.nb[p] - 1, max(0, (int)(f[p] * si.nb[p])));
    binc[a * 3 + p] = b[p];
  }
  for (int c = 0; c < 3; ++c) pos_w[a * 3 + c] = pw[c];
  const int32_t key = (int32_t)(bin_off[s] + ((int64_t)b[0] * si.nb[1] + b[1]) * si.nb[2] + b[2]);
  bin_key[a] = key;
  // counting sort of the atoms by bin: the bin's counter hands out the slots.  The order of arrival inside a bin is not fixed, and
  // nothing depends on it: every consumer of `perm` (k_neighbors) ranks its matches by neighbour index before it writes them.
  bin_rank[a] = atomicAdd(&bin_cnt[key], 1);
}

// atoms into their bins' slot ranges (bin_start: exclusive scan of the per-bin counts) + the wrapped positions in that order
__global__ void k_bin_scatter(int64_t N, const int32_t* __restrict__ bin_key, const int32_t* __restrict__ bin_rank, const int32_t* __restrict__ bin_start,
                              const double* __restrict__ pos_w, int32_t* __restrict__ perm, double* __restrict__ pos_s) {
  const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (a >= N) return;
  const int64_t g = bin_start[bin_key[a]] + bin_rank[a];
  perm[g] = (int32_t)a;
  pos_s[g * 3] = pos_w[a * 3]; pos_s[g * 3 + 1] = pos_w[a * 3 + 1]; pos_s[g * 3 + 2] = pos_w[a * 3 + 2];
}

// One wave per atom i.  The wave walks the atom's periodic images in order; for an image it visits the bins that can hold a
// neighbour (at most 3x3x3, or the single bin of a small cell for every image) with its 64 lanes striding over the bin-sorted
// candidates, and counts (FILL = false) or writes (FILL = true) the matches into the slot range the exclusive scan assigned to
// (atom, image).  A segment must come out ordered by neighbour index (canonical order, independent of the bin traversal): the
// matches are staged in LDS and every lane places its match by counting the smaller indices (a neighbour occurs once per image);
// a segment longer than the staging buffer is written and insertion-sorted by one lane in global memory.
// (Round 1 ran one THREAD per (atom, image): in a large cell only the home image of an atom has work, ~290 fp64 pair tests on
// 2-3 lanes of every wave: count + fill 0.46 ms on the 10,000-atom cell.)
constexpr int kNbStage = 512;   // staged matches per wave (one (atom, image) segment)

struct NbImage {
  int sh[3], lo[3], hi[3];
  bool any;
  double ox, oy, oz;
};
__device__ __forceinline__ NbImage nb_image(const StructInfo& si, int img, const int32_t* binc_i, const double* pos_w_i) {
  NbImage im;
  const int ny = 2 * si.reps[1] + 1, nz = 2 * si.reps[2] + 1;
  im.sh[0] = img / (ny * nz) - si.reps[0];
  im.sh[1] = (img / nz) % ny - si.reps[1];
  im.sh[2] = img % nz - si.reps[2];
  im.any = true;
  for (int p = 0; p < 3; ++p) {   // bins b' of the image whose unwrapped coordinate b' + s nb is within reach of this atom's bin
    const int b = binc_i[p];
    im.lo[p] = max(0, b - si.reach[p] - im.sh[p] * si.nb[p]);
    im.hi[p] = min(si.nb[p] - 1, b + si.reach[p] - im.sh[p] * si.nb[p]);
    im.any = im.any && im.lo[p] <= im.hi[p];
  }
  image_offset(si.lat, im.sh, pos_w_i, im.ox, im.oy, im.oz);
  return im;
}

template <bool FILL>
__global__ void __launch_bounds__(256) k_neighbors(int64_t N, int64_t M, const int64_t* __restrict__ batch,
                                                   const StructInfo* __restrict__ info, const int64_t* __restrict__ bin_off,
                                                   const int32_t* __restrict__ bin_start, const int32_t* __restrict__ perm,
                                                   const double* __restrict__ pos_s, const double* __restrict__ pos_w,
                                                   const int32_t* __restrict__ binc, const int32_t* __restrict__ wrap, double cutoff,
                                                   int64_t* __restrict__ counts, int64_t E, int64_t* __restrict__ edge_index,
                                                   int32_t* __restrict__ shift, double* __restrict__ dist, float tb_cutoff,
                                                   int64_t* __restrict__ tri) {
  __shared__ int32_t stage_j[FILL ? 4 * kNbStage : 1];
  __shared__ double stage_d[FILL ? 4 * kNbStage : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + wave;   // wave-uniform
  if (i >= N) return;
  const int s = (int)batch[i];
  const StructInfo& si = info[s];
  const double c2 = cutoff_sq(cutoff);
  const int64_t b0 = bin_off[s];
  // A cell smaller than the cutoff is one bin and has many images with at most `count` candidates each: the wave then works
  // on G = 64 / W images at a time, W lanes per image.
  const bool one_bin = si.nb[0] == 1 && si.nb[1] == 1 && si.nb[2] == 1;
  const int logw = !one_bin || si.count > 32 ? 6 : si.count > 16 ? 5 : si.count > 8 ? 4 : 3;
  const int W = 1 << logw, G = 64 >> logw, g = lane >> logw, gl = lane & (W - 1);
  const unsigned long long group_bits = W == 64 ? ~0ull : ((1ull << W) - 1ull);
  const int cap = kNbStage >> (6 - logw);   // staged matches per group
  int32_t* sj = stage_j + (FILL ? wave * kNbStage + g * cap : 0);
  double* sd = stage_d + (FILL ? wave * kNbStage + g * cap : 0);
  if (!FILL) for (int64_t img = si.n_img + lane; img < M; img += 64) counts[i * M + img] = 0;   // images this structure does not have
  int my3 = 0;   // count pass with `tri`: this lane's matches whose fp32 length is within the three-body cutoff
  for (int img0 = 0; img0 < si.n_img; img0 += G) {
    const int img = img0 + g;
    const bool valid = img < si.n_img;
    NbImage im = nb_image(si, valid ? img : 0, binc + i * 3, pos_w + i * 3);
    if (!valid) im.any = false;
    const int64_t begin = FILL && valid ? counts[i * M + img] : 0;
    int n = 0;   // matches of this image so far (uniform over the group)
    if (im.any) {
      for (int bx = im.lo[0]; bx <= im.hi[0]; ++bx)
        for (int by = im.lo[1]; by <= im.hi[1]; ++by) {
          const int64_t g0 = b0 + ((int64_t)bx * si.nb[1] + by) * si.nb[2];
          const int k0 = bin_start[g0 + im.lo[2]], k1 = bin_start[g0 + im.hi[2] + 1];   // bins along z are consecutive slots
          for (int kb = k0; kb < k1; kb += W) {
            const int k = kb + gl;
            bool hit = false;
            double d2 = 0.0;
            if (k < k1) {
              d2 = pair_d2(pos_s + (int64_t)k * 3, im.ox, im.oy, im.oz);
              hit = pair_hit(d2, c2);
              if (!FILL && tri && hit && (float)sqrt(d2) <= tb_cutoff) ++my3;   // as the reference thresholds the narrowed lengths
            }
            const unsigned long long m = (__ballot(hit) >> (g * W)) & group_bits;   // this group's lanes (all in this iteration together)
            if (FILL && hit) {
              const int at = n + __popcll(m & ((1ull << gl) - 1ull));
              if (at < cap) { sj[at] = perm[k]; sd[at] = sqrt(d2); }
            }
            n += __popcll(m);
          }
        }
    }
    if (!FILL) { if (valid && gl == 0) counts[i * M + img] = n; continue; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the staged matches of the other lanes (same wave: LDS keeps program order)
    if (n > 0 && n <= cap) {
      for (int e = gl; e < n; e += W) {
        const int32_t j = sj[e];
        int rank = 0;
        for (int f = 0; f < n; ++f) rank += sj[f] < j ? 1 : 0;
        const int64_t a = begin + rank;
        if (a < E) {
          edge_index[a] = i;
          edge_index[E + a] = j;
          dist[a] = sd[e];
          for (int p = 0; p < 3; ++p) shift[a * 3 + p] = im.sh[p] - wrap[(int64_t)j * 3 + p] + wrap[i * 3 + p];
        }
      }
    } else if (n > cap && gl == 0) {   // oversized segment: one lane, in place
      int64_t out = begin;
      for (int bx = im.lo[0]; bx <= im.hi[0]; ++bx)
        for (int by = im.lo[1]; by <= im.hi[1]; ++by) {
          const int64_t g0 = b0 + ((int64_t)bx * si.nb[1] + by) * si.nb[2];
          const int k0 = bin_start[g0 + im.lo[2]], k1 = bin_start[g0 + im.hi[2] + 1];
          for (int k = k0; k < k1; ++k) {
            const double d2 = pair_d2(pos_s + (int64_t)k * 3, im.ox, im.oy, im.oz);
            if (pair_hit(d2, c2)) {
              if (out < E) { edge_index[E + out] = perm[k]; dist[out] = sqrt(d2); }
              ++out;
            }
          }
        }
      const int64_t end = out < E ? out : E;
      for (int64_t a = begin + 1; a < end; ++a) {   // insertion sort by neighbour index
        const int64_t j = edge_index[E + a];
        const double d = dist[a];
        int64_t b = a;
        while (b > begin && edge_index[E + b - 1] > j) { edge_index[E + b] = edge_index[E + b - 1]; dist[b] = dist[b - 1]; --b; }
        edge_index[E + b] = j; dist[b] = d;
      }
      for (int64_t a = begin; a < end; ++a) {
        const int64_t j = edge_index[E + a];
        edge_index[a] = i;
        for (int p = 0; p < 3; ++p) shift[a * 3 + p] = im.sh[p] - wrap[j * 3 + p] + wrap[i * 3 + p];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // before the next image overwrites the stage
  }
  if (!FILL && tri) {
    const int64_t d = __reduce_add_sync(~0ull, my3);
    if (lane == 0) {
      tri[i] = d * (d - 1);
      atomicAdd(reinterpret_cast<unsigned long long*>(tri + N), (unsigned long long)(d * (d - 1)));   // (integers: the total does not depend on the order)
    }
  }
}

// The search above emits a centre's edges by image of the WRAPPED cell; the canonical order is by the shift relative to the given
// coordinates, shift = image - wrap[j] + wrap[i].  The two differ only in rows that hold a neighbour outside the home cell (wrap[j]
// != 0; wrap[i] moves all shifts of a row alike).  One wave per centre: rows without such a neighbour return at once, the others
// are re-ranked by (shift, j) -- unique per row -- from an LDS copy (rows longer than the stage: insertion sort by one lane).
constexpr int kCanonStage = 512;   // edges per wave (4 waves x 24 B: 48 KB of LDS)
__device__ __forceinline__ bool canon_less(int ax, int ay, int az, int aj, int bx, int by, int bz, int bj) {
  if (ax != bx) return ax < bx;
  if (ay != by) return ay < by;
  if (az != bz) return az < bz;
  return aj < bj;
}
__global__ void __launch_bounds__(256) k_rows_canonical(int64_t N, int64_t M, const int64_t* __restrict__ offsets, const int32_t* __restrict__ wrap,
                                                        int64_t E, int64_t* __restrict__ edge_index, int32_t* __restrict__ shift,
                                                        double* __restrict__ dist) {
  __shared__ int32_t s_j[4 * kCanonStage], s_x[4 * kCanonStage], s_y[4 * kCanonStage], s_z[4 * kCanonStage];
  __shared__ double s_d[4 * kCanonStage];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + wave;   // wave-uniform
  if (i >= N) return;
  const int64_t b = offsets[i * M], e = offsets[(i + 1) * M];
  const int n = (int)(e - b);
  if (n < 2 || e > E) return;
  bool moved = false;
  for (int k = lane; k < n; k += 64) {
    const int64_t j = edge_index[E + b + k];
    moved = moved || wrap[j * 3] != 0 || wrap[j * 3 + 1] != 0 || wrap[j * 3 + 2] != 0;
  }
  if (!__any(moved)) return;
  if (n <= kCanonStage) {
    int32_t *sj = s_j + wave * kCanonStage, *sx = s_x + wave * kCanonStage, *sy = s_y + wave * kCanonStage, *sz = s_z + wave * kCanonStage;
    double* sd = s_d + wave * kCanonStage;
    for (int k = lane; k < n; k += 64) {
      sj[k] = (int32_t)edge_index[E + b + k];
      sx[k] = shift[(b + k) * 3]; sy[k] = shift[(b + k) * 3 + 1]; sz[k] = shift[(b + k) * 3 + 2];
      sd[k] = dist[b + k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int k = lane; k < n; k += 64) {
      const int x = sx[k], y = sy[k], z = sz[k], j = sj[k];
      int rank = 0;
      for (int f = 0; f < n; ++f) rank += canon_less(sx[f], sy[f], sz[f], sj[f], x, y, z, j) ? 1 : 0;
      const int64_t a = b + rank;
      edge_index[E + a] = j;
      shift[a * 3] = x; shift[a * 3 + 1] = y; shift[a * 3 + 2] = z;
      dist[a] = sd[k];
    }
  } else {
    // very long rows (a tiny cell under a large cutoff): odd-even transposition over BLOCKS of kCanonStage / 2 edges -- in pass p the
    // wave stages the block pairs (p & 1, p & 1 + 1), (p & 1 + 2, ...) in its LDS share, ranks the pair's <= 512 edges as above
    // and writes them back in order (a merge-split step); after as many passes as there are blocks the row is sorted.  The keys
    // (shift, neighbour) are distinct, so the result is the one canonical order whatever the method.  (Until round 5: an
    // insertion sort by one lane on global memory, O(n^2) dependent round trips.)
    constexpr int H = kCanonStage / 2;
    int32_t *sj = s_j + wave * kCanonStage, *sx = s_x + wave * kCanonStage, *sy = s_y + wave * kCanonStage, *sz = s_z + wave * kCanonStage;
    double* sd = s_d + wave * kCanonStage;
    const int nc = (n + H - 1) / H;
    for (int pass = 0; pass < nc; ++pass) {
      for (int blk = pass & 1; blk + 1 < nc; blk += 2) {
        const int64_t b0 = b + (int64_t)blk * H;
        const int m = (int)((e - b0) < 2 * H ? (e - b0) : 2 * H);   // edges of the two blocks (the last block may be short)
        for (int k = lane; k < m; k += 64) {
          sj[k] = (int32_t)edge_index[E + b0 + k];
          sx[k] = shift[(b0 + k) * 3]; sy[k] = shift[(b0 + k) * 3 + 1]; sz[k] = shift[(b0 + k) * 3 + 2];
          sd[k] = dist[b0 + k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int k = lane; k < m; k += 64) {
          const int x = sx[k], y = sy[k], z = sz[k], j = sj[k];
          int rank = 0;
          for (int f = 0; f < m; ++f) rank += canon_less(sx[f], sy[f], sz[f], sj[f], x, y, z, j) ? 1 : 0;
          const int64_t a = b0 + rank;
          edge_index[E + a] = j;
          shift[a * 3] = x; shift[a * 3 + 1] = y; shift[a * 3 + 2] = z;
          dist[a] = sd[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the stage is rewritten by the next pair; its global writes precede the next pass's reads
      }
    }
  }
}

// ---- skin ("Verlet") list: the lists of an MD trajectory without a search per step ------------------------------------
// CANDIDATES are the neighbour list built with cutoff + skin at reference positions.  While no atom has moved by more than skin / 2
// since, every pair within `cutoff` is among them; the candidates are in canonical order and that order does not depend on the
// positions (centre, shift relative to the given coordinates, neighbour), so the candidates that pass d <= cutoff, in candidate
// order, ARE the list a fresh search would return -- same edges, same order, same shifts, and with them the same triplets.  Every
// distance goes through the search's own functions (wrap_point / image_offset / pair_d2, no contraction): the update classifies a
// pair exactly as the search does.
struct VerletScratch {
  double* pos_w;       // [N,3] wrapped positions (as the search forms them)
  int32_t* wrap;       // [N,3]
  uint8_t* state;      // [Ec] membership at the current positions: bit 0 in the list, bit 1 within the three-body cutoff
  double* dist;        // [Ec] pair distance at the current positions
  int32_t* row_keep;   // [N+1] kept candidates per centre, then exclusive offsets
  int64_t* row_tri;    // [N] triplets per centre
  void* scan_tmp;
  size_t scan_tmp_bytes;
  unsigned long long* acc;   // [4] max displacement^2 (bits of a non-negative double), changed flag, E, T
  int32_t* off_e;      // [N+1] first edge slot of each centre (m3g_verlet_fill_lists)
  int64_t* off_t;      // [N+1] first triplet slot of each centre
  size_t total;
};
static VerletScratch verlet_carve(int64_t N, int64_t Ec, void* base) {
  VerletScratch w{};
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = p ? (void*)(p + off) : nullptr; off += align_up_g(bytes); return r; };
  w.pos_w = (double*)take(sizeof(double) * 3 * (size_t)(N + 1));
  w.wrap = (int32_t*)take(sizeof(int32_t) * 3 * (size_t)(N + 1));
  w.state = (uint8_t*)take((size_t)Ec + 16);
  w.dist = (double*)take(sizeof(double) * (size_t)(Ec + 1));
  w.row_keep = (int32_t*)take(sizeof(int32_t) * (size_t)(N + 2));
  w.row_tri = (int64_t*)take(sizeof(int64_t) * (size_t)(N + 1));
  const size_t tmp = prims::scan_tmp_bytes<int32_t>(N + 1);
  w.scan_tmp_bytes = tmp;
  w.scan_tmp = take(tmp);
  w.acc = (unsigned long long*)take(sizeof(unsigned long long) * 8);   // [4]: workgroup counter of the one-launch update (small cells)
  w.off_e = (int32_t*)take(sizeof(int32_t) * (size_t)(N + 2));
  w.off_t = (int64_t*)take(sizeof(int64_t) * (size_t)(N + 2));
  w.total = off;
  return w;
}

// one thread per atom: wrapped position and wrap exactly as the search forms them; largest displacement since the reference
__global__ void k_verlet_prep(int64_t N, int64_t S, const double* __restrict__ pos, const double* __restrict__ pos_ref,
                              const double* __restrict__ lattice, const int64_t* __restrict__ batch, double* pos_w, int32_t* wrap,
                              unsigned long long* acc) {
  const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  double m = 0.0;   // lanes beyond N stay in the wave reduction below with a neutral value (a shuffle from an exited lane is undefined)
  if (a < N) {
    int64_t s = batch[a];
    if (s < 0 || s >= S) s = 0;
    const LatticeFrame fr = lattice_frame(lattice + s * 9);
    const double x = pos[a * 3], y = pos[a * 3 + 1], z = pos[a * 3 + 2];
    double f[3], w[3], pw[3];
    wrap_point(fr.lat, fr.inv, x, y, z, f, w, pw);
    for (int c = 0; c < 3; ++c) { pos_w[a * 3 + c] = pw[c]; wrap[a * 3 + c] = (int32_t)w[c]; }
    const double dx = x - pos_ref[a * 3], dy = y - pos_ref[a * 3 + 1], dz = z - pos_ref[a * 3 + 2];
    m = dx * dx + dy * dy + dz * dz;
    if (!(m >= 0.0)) m = 1e300;   // NaN positions: force the rebuild path (which reports them)
  }
  // wave maximum first: one atomic per wave (non-negative doubles order like their bit patterns)
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(acc, (unsigned long long)__double_as_longlong(m));
}

// one wave per centre over its candidate row: membership and distance of every candidate at the current positions, compared with
// the membership the caller's lists were built with; kept edges and triplets of the row
__global__ void __launch_bounds__(256) k_verlet_rows(int64_t N, int64_t S, int64_t Ec, const int64_t* __restrict__ batch,
                                                     const double* __restrict__ lattice, const int64_t* __restrict__ cand_ei,
                                                     const int32_t* __restrict__ cand_shift, const int32_t* __restrict__ row_ptr,
                                                     const double* __restrict__ pos_w, const int32_t* __restrict__ wrap, double cutoff,
                                                     float tb_cutoff, const uint8_t* __restrict__ old_state, uint8_t* state,
                                                     double* dist, int32_t* row_keep, int64_t* row_tri, unsigned long long* acc) {
  const int lane = threadIdx.x & 63;
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);   // wave-uniform
  if (i >= N) return;
  int64_t s = batch[i];
  if (s < 0 || s >= S) s = 0;
  const double* lat = lattice + s * 9;
  const double c2 = cutoff_sq(cutoff);
  const int r0 = row_ptr[i], r1 = row_ptr[i + 1];
  int n2 = 0, n3 = 0;
  bool changed = false;
  for (int base = r0; base < r1; base += 64) {
    const int c = base + lane;
    bool in2 = false, in3 = false;
    if (c < r1) {
      const int64_t j = cand_ei[Ec + c];
      // the image of the wrapped cell this edge belongs to: shift = image - wrap[j] + wrap[i]
      int sh[3];
      for (int p = 0; p < 3; ++p) sh[p] = cand_shift[(int64_t)c * 3 + p] + wrap[j * 3 + p] - wrap[i * 3 + p];
      double ox, oy, oz;
      image_offset(lat, sh, pos_w + i * 3, ox, oy, oz);
      const double d2 = pair_d2(pos_w + j * 3, ox, oy, oz);
      const double d = sqrt(d2);
      in2 = pair_hit(d2, c2);
      in3 = in2 && (float)d <= tb_cutoff;   // as the reference thresholds the narrowed lengths
      const uint8_t st = (uint8_t)((in2 ? 1 : 0) | (in3 ? 2 : 0));
      state[c] = st;
      dist[c] = d;
      changed = changed || (old_state && st != old_state[c]);
    }
    n2 += __popcll(__ballot(in2));
    n3 += __popcll(__ballot(in3));
  }
  const bool any_changed = __any(changed);
  if (lane == 0) {   // (no per-row atomics on shared totals: 10,000 waves on one address serialise -- k_verlet_totals adds the rows up)
    row_keep[i] = n2;
    row_tri[i] = (int64_t)n3 * (n3 > 0 ? n3 - 1 : 0);
    // rare along a trajectory; right after a search EVERY row differs from the zeroed membership bytes, and 10,000 atomics on one
    // address serialise (measured: 118 us) -- so look first, and only the first few waves write
    if (any_changed && __atomic_load_n(acc + 1, __ATOMIC_RELAXED) == 0ull) atomicOr(acc + 1, 1ull);
  }
}
// one workgroup: E = sum of the rows' kept candidates, T = sum of their triplets (fixed order: integers anyway)
// (acc[5] = the longest candidate row: the caller chooses the refill entry point by it without a read-back of its own)
__global__ void __launch_bounds__(1024) k_verlet_totals(int64_t N, const int32_t* __restrict__ row_keep, const int64_t* __restrict__ row_tri,
                                                        const int32_t* __restrict__ row_ptr, unsigned long long* acc) {
  __shared__ unsigned long long se[16], st[16], sm[16];
  unsigned long long e = 0, t = 0, m = 0;
  for (int64_t i = threadIdx.x; i < N; i += blockDim.x) {
    e += (unsigned long long)row_keep[i]; t += (unsigned long long)row_tri[i];
    const unsigned long long len = (unsigned long long)(row_ptr[i + 1] - row_ptr[i]);
    m = len > m ? len : m;
  }
  for (int o = 32; o > 0; o >>= 1) {
    e += __shfl_xor(e, o); t += __shfl_xor(t, o);
    const unsigned long long mo = __shfl_xor(m, o);
    m = mo > m ? mo : m;
  }
  if ((threadIdx.x & 63) == 0) { se[threadIdx.x >> 6] = e; st[threadIdx.x >> 6] = t; sm[threadIdx.x >> 6] = m; }
  __syncthreads();
  if (threadIdx.x == 0) {
    e = 0; t = 0; m = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { e += se[k]; t += st[k]; m = sm[k] > m ? sm[k] : m; }
    acc[2] = e; acc[3] = t; acc[5] = m;
  }
}

// one wave per centre: its kept candidates, in candidate order, to the slots the scan assigned to the row
__global__ void __launch_bounds__(256) k_verlet_fill(int64_t N, int64_t Ec, int64_t E, const int64_t* __restrict__ cand_ei,
                                                     const int32_t* __restrict__ cand_shift, const int32_t* __restrict__ row_ptr,
                                                     const int32_t* __restrict__ row_off, const uint8_t* __restrict__ state,
                                                     const double* __restrict__ dist_c, int64_t* __restrict__ edge_index,
                                                     int32_t* __restrict__ shift, double* __restrict__ dist, uint8_t* cand_state) {
  const int lane = threadIdx.x & 63;
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= N) return;
  const int r0 = row_ptr[i], r1 = row_ptr[i + 1];
  int64_t out = row_off[i];
  for (int base = r0; base < r1; base += 64) {
    const int c = base + lane;
    const uint8_t st = c < r1 ? state[c] : (uint8_t)0;
    const bool keep = (st & 1) != 0;
    const unsigned long long m = __ballot(keep);
    if (c < r1) cand_state[c] = st;
    if (keep) {
      const int64_t a = out + __popcll(m & ((1ull << lane) - 1ull));
      if (a < E) {
        edge_index[a] = i;
        edge_index[E + a] = cand_ei[Ec + c];
        for (int p = 0; p < 3; ++p) shift[a * 3 + p] = cand_shift[(int64_t)c * 3 + p];
        dist[a] = dist_c[c];
      }
    }
    out += __popcll(m);
  }
}

// Small cells: k_verlet_prep + k_verlet_rows + k_verlet_totals in ONE launch.  A wave per centre wraps its own position AND each
// candidate's neighbour position on the fly (wrap_point on the raw coordinates: the very arithmetic k_verlet_prep applies per atom, so
// every pair is classified exactly as before) instead of reading a wrapped copy another launch would have to write first; the
// totals are formed by the launch's last workgroup (device-scope counter, nobody waits).  Three launch boundaries less per MD step
// of a small cell (~10 us of the 0.19 ms a 32-atom iteration takes).
constexpr int64_t kVerletFusedMaxAtoms = 512;
__global__ void __launch_bounds__(256) k_verlet_update_small(int64_t N, int64_t S, int64_t Ec, const double* __restrict__ pos,
                                                             const double* __restrict__ pos_ref, const int64_t* __restrict__ batch,
                                                             const double* __restrict__ lattice, const int64_t* __restrict__ cand_ei,
                                                             const int32_t* __restrict__ cand_shift, const int32_t* __restrict__ row_ptr, double cutoff,
                                                             float tb_cutoff, const uint8_t* __restrict__ old_state, uint8_t* state, double* dist,
                                                             int32_t* row_keep, int64_t* row_tri, unsigned long long* acc) {
  const int lane = threadIdx.x & 63;
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);   // wave-uniform
  if (i < N) {
    int64_t s = batch[i];
    if (s < 0 || s >= S) s = 0;
    const LatticeFrame fr = lattice_frame(lattice + s * 9);
    const double* lat = lattice + s * 9;
    const double xi = pos[i * 3], yi = pos[i * 3 + 1], zi = pos[i * 3 + 2];
    double fi[3], wi[3], pwi[3];
    wrap_point(fr.lat, fr.inv, xi, yi, zi, fi, wi, pwi);
    const int wi0 = (int32_t)wi[0], wi1 = (int32_t)wi[1], wi2 = (int32_t)wi[2];
    if (lane == 0) {   // largest displacement since the reference (k_verlet_prep)
      const double dx = xi - pos_ref[i * 3], dy = yi - pos_ref[i * 3 + 1], dz = zi - pos_ref[i * 3 + 2];
      double m = dx * dx + dy * dy + dz * dz;
      if (!(m >= 0.0)) m = 1e300;
      atomicMax(acc, (unsigned long long)__double_as_longlong(m));
    }
    const double c2 = cutoff_sq(cutoff);
    const int r0 = row_ptr[i], r1 = row_ptr[i + 1];
    int n2 = 0, n3 = 0;
    bool changed = false;
    for (int base = r0; base < r1; base += 64) {
      const int c = base + lane;
      bool in2 = false, in3 = false;
      if (c < r1) {
        const int64_t j = cand_ei[Ec + c];
        double fj[3], wj[3], pwj[3];
        wrap_point(fr.lat, fr.inv, pos[j * 3], pos[j * 3 + 1], pos[j * 3 + 2], fj, wj, pwj);
        int sh[3];
        sh[0] = cand_shift[(int64_t)c * 3] + (int32_t)wj[0] - wi0;
        sh[1] = cand_shift[(int64_t)c * 3 + 1] + (int32_t)wj[1] - wi1;
        sh[2] = cand_shift[(int64_t)c * 3 + 2] + (int32_t)wj[2] - wi2;
        double ox, oy, oz;
        image_offset(lat, sh, pwi, ox, oy, oz);
        const double d2 = pair_d2(pwj, ox, oy, oz);
        const double d = sqrt(d2);
        in2 = pair_hit(d2, c2);
        in3 = in2 && (float)d <= tb_cutoff;
        const uint8_t st = (uint8_t)((in2 ? 1 : 0) | (in3 ? 2 : 0));
        state[c] = st;
        dist[c] = d;
        changed = changed || (old_state && st != old_state[c]);
      }
      n2 += __popcll(__ballot(in2));
      n3 += __popcll(__ballot(in3));
    }
    const bool any_changed = __any(changed);
    if (lane == 0) {
      row_keep[i] = n2;
      row_tri[i] = (int64_t)n3 * (n3 > 0 ? n3 - 1 : 0);
      if (any_changed && __atomic_load_n(acc + 1, __ATOMIC_RELAXED) == 0ull) atomicOr(acc + 1, 1ull);
    }
  }
  // totals by the last workgroup (k_verlet_totals' sums: integers, any order)
  __shared__ int s_last;
  __shared__ unsigned long long se[4], st4[4];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(acc + 4, 1ull) == (unsigned long long)gridDim.x - 1ull;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __shared__ unsigned long long sm4[4];
  unsigned long long e = 0, t = 0, m = 0;
  for (int64_t k = threadIdx.x; k < N; k += blockDim.x) {
    e += (unsigned long long)row_keep[k]; t += (unsigned long long)row_tri[k];
    const unsigned long long len = (unsigned long long)(row_ptr[k + 1] - row_ptr[k]);
    m = len > m ? len : m;
  }
  for (int o = 32; o > 0; o >>= 1) {
    e += __shfl_xor(e, o); t += __shfl_xor(t, o);
    const unsigned long long mo = __shfl_xor(m, o);
    m = mo > m ? mo : m;
  }
  if (lane == 0) { se[threadIdx.x >> 6] = e; st4[threadIdx.x >> 6] = t; sm4[threadIdx.x >> 6] = m; }
  __syncthreads();
  if (threadIdx.x == 0) {
    acc[2] = (se[0] + se[1]) + (se[2] + se[3]); acc[3] = (st4[0] + st4[1]) + (st4[2] + st4[3]);
    m = sm4[0];
    for (int k = 1; k < 4; ++k) m = sm4[k] > m ? sm4[k] : m;
    acc[5] = m;
  }
}

// ---- refill in two launches (m3g_verlet_fill_lists) ----------------------------------------------------------------------
// The update pass left, per centre, the number of kept candidates and of triplets.  One workgroup turns both into exclusive
// offsets (N + 1 = 10,001 items: a library scan is two launches per array plus a clear); one wave per centre then writes the
// centre's edges AND its triplets -- the d (d - 1) ordered pairs of its edges inside the three-body cutoff, in the reference's order
// (data/material_graph.py:239-248) -- and the per-centre / per-edge triplet counts: what m3g_verlet_fill, a dtype cast,
// m3g_threebody_build and their scans did in twelve launches.
constexpr int kRefillScanThreads = 1024;
constexpr int64_t kRefillMaxAtoms = (int64_t)kRefillScanThreads * 256;   // at most 256 chunks for the one-workgroup scan
__global__ void __launch_bounds__(kRefillScanThreads) k_verlet_offsets(int64_t N, const int32_t* __restrict__ row_keep, const int64_t* __restrict__ row_tri,
                                                                       int32_t* __restrict__ off_e, int64_t* __restrict__ off_t) {
  // Chunks of 1,024 consecutive items, a thread per item (coalesced): 32-bit inclusive scan inside the wave by lane shifts (a chunk's
  // triplets are < 2^30: rows of at most 1,024 candidates), the 16 wave totals scanned by wave 0, 64-bit carries from chunk to chunk.
  // The loads of the first kPre chunks are all in flight before the first scan.
  __shared__ unsigned w_e[16], w_t[16], x_e[17], x_t[17];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int kPre = 12;
  const int64_t chunks = (N + kRefillScanThreads - 1) / kRefillScanThreads;
  unsigned pe[kPre], pt[kPre];
#pragma unroll
  for (int c = 0; c < kPre; ++c) {
    const int64_t i = (int64_t)c * kRefillScanThreads + t;
    pe[c] = (c < chunks && i < N) ? (unsigned)row_keep[i] : 0u;
    pt[c] = (c < chunks && i < N) ? (unsigned)row_tri[i] : 0u;
  }
  long long carry_e = 0, carry_t = 0;
  auto chunk = [&](int64_t c, unsigned e, unsigned tr) {
    const int64_t i = c * kRefillScanThreads + t;
    unsigned ie = e, it = tr;   // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned ae = __shfl_up(ie, off), at = __shfl_up(it, off);
      if (lane >= off) { ie += ae; it += at; }
    }
    if (lane == 63) { w_e[wave] = ie; w_t[wave] = it; }
    __syncthreads();
    if (wave == 0) {   // exclusive scan of the 16 wave totals, chunk totals in slot 16
      unsigned ve = lane < 16 ? w_e[lane] : 0u, vt = lane < 16 ? w_t[lane] : 0u;
      const unsigned oe = ve, ot = vt;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        const unsigned ae = __shfl_up(ve, off), at = __shfl_up(vt, off);
        if (lane >= off) { ve += ae; vt += at; }
      }
      if (lane < 16) { x_e[lane] = ve - oe; x_t[lane] = vt - ot; }
      if (lane == 15) { x_e[16] = ve; x_t[16] = vt; }
    }
    __syncthreads();
    if (i < N) { off_e[i] = (int32_t)(carry_e + (long long)(x_e[wave] + ie - e)); off_t[i] = carry_t + (long long)(x_t[wave] + it - tr); }
    carry_e += x_e[16]; carry_t += x_t[16];
    // (w_* are rewritten before the next chunk's first barrier, x_* between its barriers: every read above precedes both)
  };
#pragma unroll
  for (int c = 0; c < kPre; ++c)
    if (c < chunks) chunk(c, pe[c], pt[c]);
  for (int64_t c = kPre; c < chunks; ++c) {
    const int64_t i = c * kRefillScanThreads + t;
    chunk(c, i < N ? (unsigned)row_keep[i] : 0u, i < N ? (unsigned)row_tri[i] : 0u);
  }
  if (t == 0) { off_e[N] = (int32_t)carry_e; off_t[N] = carry_t; }
}

constexpr int kRefillList = 1024;   // valid edges of a centre listed in LDS (the host checks the longest candidate row against it)
__global__ void __launch_bounds__(256) k_verlet_fill_lists(int64_t N, int64_t Ec, int64_t E, int64_t T, const int64_t* __restrict__ cand_ei,
                                                           const int32_t* __restrict__ cand_shift, const int32_t* __restrict__ row_ptr,
                                                           const int32_t* __restrict__ off_e, const int64_t* __restrict__ off_t,
                                                           const uint8_t* __restrict__ state, int64_t* __restrict__ edge_index,
                                                           int32_t* __restrict__ shift, uint8_t* cand_state, int64_t* __restrict__ tei,
                                                           int64_t* __restrict__ num_triplet_i, int32_t* __restrict__ num_triplet_ij) {
  __shared__ int32_t vlist_all[4 * kRefillList];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + wave;
  if (i >= N) return;
  int32_t* vlist = vlist_all + wave * kRefillList;
  const int r0 = row_ptr[i], r1 = row_ptr[i + 1];
  int64_t out = off_e[i];
  int d = 0;   // edges of this centre inside the three-body cutoff so far
  for (int base = r0; base < r1; base += 64) {
    const int c = base + lane;
    const uint8_t st = c < r1 ? state[c] : (uint8_t)0;
    const bool keep = (st & 1) != 0, valid = (st & 2) != 0;
    const unsigned long long m = __ballot(keep), mv = __ballot(valid);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (c < r1) cand_state[c] = st;
    if (keep) {
      const int64_t a = out + __popcll(m & below);
      if (a < E) {
        edge_index[a] = i;
        edge_index[E + a] = cand_ei[Ec + c];
        for (int p = 0; p < 3; ++p) shift[a * 3 + p] = cand_shift[(int64_t)c * 3 + p];
        if (valid) {
          const int rk = d + __popcll(mv & below);
          if (rk < kRefillList) vlist[rk] = (int32_t)a;
        }
      }
    }
    out += __popcll(m);
    d += __popcll(mv);
  }
  if (lane == 0 && num_triplet_i) num_triplet_i[i] = (int64_t)d * (d > 0 ? d - 1 : 0);
  if (num_triplet_ij) {   // per kept edge: d - 1 partners when it lies inside the three-body cutoff
    int64_t o2 = off_e[i];
    for (int base = r0; base < r1; base += 64) {
      const int c = base + lane;
      const uint8_t st = c < r1 ? state[c] : (uint8_t)0;
      const unsigned long long m = __ballot((st & 1) != 0);
      if (st & 1) {
        const int64_t a = o2 + __popcll(m & ((1ull << lane) - 1ull));
        if (a < E) num_triplet_ij[a] = (st & 2) ? d - 1 : 0;
      }
      o2 += __popcll(m);
    }
  }
  if (d < 2) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // vlist entries written by other lanes of this wave
  const int64_t out0 = off_t[i];
  const int n = d * (d - 1);
  for (int q = lane; q < n; q += 64) {   // slot q: first edge of rank q / (d - 1), partner of rank k or k + 1, k = q % (d - 1)
    const int a = q / (d - 1), k = q - a * (d - 1);
    const int64_t o = out0 + q;
    if (o < T) { tei[o] = vlist[a]; tei[T + o] = vlist[k < a ? k : k + 1]; }
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
  const size_t tmp = prims::scan_tmp_bytes<int64_t>(E + 1);
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

// One wave per centre: rank its valid edges (d <= threebody_cutoff, decided on the fp32 distances like the reference) with a
// ballot prefix count, 64 edges of the row per pass.
__global__ void __launch_bounds__(256) k_rank_valid(int64_t N, const int32_t* __restrict__ row_ptr, const float* __restrict__ dist,
                                                    float tb_cutoff, int32_t* rank, int32_t* deg, int64_t* counts) {
  const int lane = threadIdx.x & 63;
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= N) return;
  const int r0 = row_ptr[i], r1 = row_ptr[i + 1];
  int d = 0;
  for (int base = r0; base < r1; base += 64) {
    const int e = base + lane;
    const bool v = e < r1 && dist[e] <= tb_cutoff;
    const unsigned long long m = __ballot(v);
    if (e < r1) rank[e] = v ? d + __popcll(m & ((1ull << lane) - 1ull)) : -1;
    d += __popcll(m);
  }
  if (lane == 0) deg[i] = d;
  for (int base = r0; base < r1; base += 64) {
    const int e = base + lane;
    if (e < r1) counts[e] = dist[e] <= tb_cutoff ? d - 1 : 0;
  }
}

// One wave per centre: its d (d - 1) triplets in the reference's order (first edge in edge order; partners in edge order, itself
// skipped) occupy one contiguous range of the output, which the lanes write side by side: slot q -> first edge of rank
// q / (d - 1), partner of rank k or k + 1 with k = q % (d - 1).  The centre's valid edges are listed by rank in LDS; a centre
// with more than kTripletList of them falls back to one lane per first edge.
constexpr int kTripletList = 256;
__global__ void __launch_bounds__(256) k_fill_triplets(int64_t N, int64_t E, int64_t T, const int32_t* __restrict__ row_ptr,
                                                       const int32_t* __restrict__ rank, const int32_t* __restrict__ deg,
                                                       const int64_t* __restrict__ offsets, int64_t* __restrict__ tei,
                                                       int64_t* __restrict__ num_triplet_i, int32_t* __restrict__ num_triplet_ij) {
  __shared__ int32_t vlist_all[4 * kTripletList];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + wave;
  if (i >= N) return;
  int32_t* vlist = vlist_all + wave * kTripletList;
  const int r0 = row_ptr[i], r1 = row_ptr[i + 1];
  const int d = deg[i];
  if (lane == 0 && num_triplet_i) num_triplet_i[i] = (int64_t)d * (d - 1);
  int first = -1;   // the centre's first valid edge: its offset is the centre's output base
  for (int base = r0; base < r1; base += 64) {
    const int e = base + lane;
    const int rk = e < r1 ? rank[e] : -1;
    if (e < r1 && num_triplet_ij) num_triplet_ij[e] = rk >= 0 ? d - 1 : 0;
    if (rk >= 0 && rk < kTripletList) vlist[rk] = e;
    if (rk == 0) first = e;
  }
  if (d < 2) return;
  first = __builtin_amdgcn_readfirstlane(__reduce_max_sync(~0ull, first));
  const int64_t out0 = offsets[first];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // vlist entries written by other lanes of this wave
  if (d <= kTripletList) {
    const int n = d * (d - 1);
    for (int q = lane; q < n; q += 64) {
      const int a = q / (d - 1), k = q - a * (d - 1);
      const int64_t o = out0 + q;
      if (o < T) { tei[o] = vlist[a]; tei[T + o] = vlist[k < a ? k : k + 1]; }
    }
  } else {   // very long rows: one lane per first edge, partners found by walking the row
    for (int base = r0; base < r1; base += 64) {
      const int e = base + lane;
      if (e >= r1 || rank[e] < 0) continue;
      int64_t o = offsets[e];
      for (int f = r0; f < r1; ++f) {
        if (f == e || rank[f] < 0) continue;
        if (o < T) { tei[o] = e; tei[T + o] = f; }
        ++o;
      }
    }
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
// host_n_triplets != nullptr: also the number of triplets the list has under `threebody_cutoff`, from the same pass and the same
// wait -- a caller that builds both lists can then size every tensor at once (m3g_threebody_build needs no wait of its own).
extern "C" int m3g_neighbor_count_triplets(int64_t N, int64_t S, int64_t max_images, const double* pos, const double* lattice,
                                           const int64_t* batch, double cutoff, float threebody_cutoff, void* scratch, size_t scratch_bytes,
                                           int64_t* host_n_edges, int64_t* host_n_triplets, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (host_n_triplets) *host_n_triplets = 0;
  size_t need = 0;
  int rc = m3g_neighbor_scratch_bytes(N, S, max_images, &need);
  if (rc) return rc;
  if (!scratch || scratch_bytes < need || !host_n_edges) { set_error("m3g_neighbor_count: scratch too small or null argument"); return M3G_ERR_SIZE; }
  NbScratch w = nb_carve(N, S, max_images, scratch);
  int* flags = (int*)((char*)scratch + w.total);
  M3G_HIP_CHECK(hipMemsetAsync(flags, 0, sizeof(int), s));
  *host_n_edges = 0;
  if (N == 0 || S == 0) return M3G_OK;
  hipLaunchKernelGGL(k_struct_info, g_for(S), dim3(256), 0, s, N, S, lattice, batch, cutoff, w.info, w.bin_off, flags, w.tri + N);
  M3G_HIP_CHECK(prims::exclusive_scan<int64_t>(w.bin_off, w.bin_off, S + 1, w.tmp, s));
  // atoms sorted by bin: a counting sort (bin ids are < max_bins) -- per-bin counters filled while the positions are wrapped, one
  // scan, one scatter
  M3G_HIP_CHECK(hipMemsetAsync(w.bin_start, 0, sizeof(int32_t) * (size_t)(w.max_bins + 2), s));
  hipLaunchKernelGGL(k_wrap_positions, g_for(N), dim3(256), 0, s, N, S, pos, batch, w.info, w.bin_off, w.pos_w, w.wrap, w.binc, w.bin_key,
                     w.bin_start, w.bin_rank, flags);
  M3G_HIP_CHECK(prims::exclusive_scan<int32_t>(w.bin_start, w.bin_start, w.max_bins + 2, w.tmp, s));
  hipLaunchKernelGGL(k_bin_scatter, g_for(N), dim3(256), 0, s, N, w.bin_key, w.bin_rank, w.bin_start, w.pos_w, w.perm, w.pos_s);
  const int64_t NM = N * max_images;
  hipLaunchKernelGGL((k_neighbors<false>), g_for(N * 64), dim3(256), 0, s, N, max_images, batch, w.info, w.bin_off, w.bin_start, w.perm, w.pos_s,
                     w.pos_w, w.binc, w.wrap, cutoff, w.counts, (int64_t)0, nullptr, nullptr, nullptr, threebody_cutoff, host_n_triplets ? w.tri : nullptr);
  M3G_HIP_CHECK(hipMemsetAsync(w.counts + NM, 0, sizeof(int64_t), s));
  M3G_HIP_CHECK(prims::exclusive_scan<int64_t>(w.counts, w.counts, NM + 1, w.tmp, s));
  int h_flags = 0;
  if (host_n_triplets)   // (summed by k_neighbors itself: an integer atomic per centre)
    M3G_HIP_CHECK(hipMemcpyAsync(host_n_triplets, w.tri + N, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipMemcpyAsync(host_n_edges, w.counts + NM, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipMemcpyAsync(&h_flags, flags, sizeof(int), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipStreamSynchronize(s));
  if (h_flags & 1) { set_error("singular lattice"); return M3G_ERR_VALUE; }
  if (h_flags & 2) { set_error("batch vector must be sorted, contiguous per structure and within [0, n_structs)"); return M3G_ERR_VALUE; }
  return M3G_OK;
}

extern "C" int m3g_neighbor_count(int64_t N, int64_t S, int64_t max_images, const double* pos, const double* lattice,
                                  const int64_t* batch, double cutoff, void* scratch, size_t scratch_bytes,
                                  int64_t* host_n_edges, void* stream_) {
  return m3g_neighbor_count_triplets(N, S, max_images, pos, lattice, batch, cutoff, 0.f, scratch, scratch_bytes, host_n_edges, nullptr, stream_);
}

// Phase 2: fill (same scratch, untouched since the count call).
extern "C" int m3g_neighbor_fill(int64_t N, int64_t S, int64_t max_images, const int64_t* batch, double cutoff, void* scratch,
                                 int64_t n_edges, int64_t* edge_index, int32_t* edge_cell_shift, double* distances, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (n_edges == 0 || N == 0) return M3G_OK;
  if (!scratch || !edge_index || !edge_cell_shift || !distances) { set_error("m3g_neighbor_fill: null argument"); return M3G_ERR_VALUE; }
  NbScratch w = nb_carve(N, S, max_images, scratch);
  hipLaunchKernelGGL((k_neighbors<true>), g_for(N * 64), dim3(256), 0, s, N, max_images, batch, w.info, w.bin_off, w.bin_start, w.perm,
                     w.pos_s, w.pos_w, w.binc, w.wrap, cutoff, w.counts, n_edges, edge_index, edge_cell_shift, distances, 0.f, nullptr);
  // canonical order inside a centre: by the shift relative to the given coordinates (rows with a neighbour outside the home cell)
  hipLaunchKernelGGL(k_rows_canonical, g_for(N * 64), dim3(256), 0, s, N, max_images, w.counts, w.wrap, n_edges, edge_index, edge_cell_shift, distances);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}

// ---- skin list, host entries (include/m3gnet_hip.h) -----------------------------------------------------------------
extern "C" int m3g_verlet_scratch_bytes(int64_t N, int64_t n_candidates, size_t* bytes) {
  if (!bytes || N < 0 || n_candidates < 0) { set_error("m3g_verlet_scratch_bytes: bad argument"); return M3G_ERR_VALUE; }
  if (n_candidates >= (int64_t(1) << 31) - 2) { set_error("too many candidate pairs for int32 row pointers"); return M3G_ERR_UNSUPPORTED; }
  *bytes = verlet_carve(N, n_candidates, nullptr).total + 256;
  return M3G_OK;
}

extern "C" int m3g_verlet_rows(int64_t N, int64_t n_candidates, const int64_t* cand_edge_index, int32_t* cand_row_ptr, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!cand_row_ptr || (n_candidates > 0 && !cand_edge_index)) { set_error("m3g_verlet_rows: null argument"); return M3G_ERR_VALUE; }
  int* flags = (int*)(cand_row_ptr + N + 1);   // one spare word behind the row pointers (the caller allocates N + 2)
  M3G_HIP_CHECK(hipMemsetAsync(flags, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_rows_from_sorted, g_for(N + 1), dim3(256), 0, s, N, n_candidates, cand_edge_index, cand_row_ptr, flags);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}

// The pass itself, queued on the stream with its 48-byte result copy ([4] internal, [5] = the longest candidate row): host_out[0] = bits of the largest squared displacement
// (a non-negative double), [1] = changed flag, [2] = E, [3] = T.  No wait: a caller with PINNED host memory can queue the
// evaluation behind it and look at the verdict afterwards (torch_m3gnet.data.md.VerletGraph.begin / confirm).
extern "C" int m3g_verlet_update_async(int64_t N, int64_t S, int64_t Ec, const double* pos, const double* pos_ref, const double* lattice,
                                       const int64_t* batch, const int64_t* cand_edge_index, const int32_t* cand_shift,
                                       const int32_t* cand_row_ptr, double cutoff, float threebody_cutoff, const uint8_t* cand_state,
                                       void* scratch, size_t scratch_bytes, uint64_t* host_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  size_t need = 0;
  int rc = m3g_verlet_scratch_bytes(N, Ec, &need);
  if (rc) return rc;
  if (!scratch || scratch_bytes < need || !host_out) { set_error("m3g_verlet_update: scratch too small or null argument"); return M3G_ERR_SIZE; }
  if (N == 0) { for (int k = 0; k < 6; ++k) host_out[k] = 0; return M3G_OK; }
  if (!pos || !pos_ref || !lattice || !batch || !cand_row_ptr || (Ec > 0 && (!cand_edge_index || !cand_shift))) {
    set_error("m3g_verlet_update: null argument");
    return M3G_ERR_VALUE;
  }
  VerletScratch w = verlet_carve(N, Ec, scratch);
  M3G_HIP_CHECK(hipMemsetAsync(w.acc, 0, sizeof(unsigned long long) * 8, s));
  // cand_state == NULL: fresh candidates, no lists built from them yet -- everything counts as changed (and no row has to say so)
  if (!cand_state) M3G_HIP_CHECK(hipMemsetAsync(w.acc + 1, 1, 1, s));
  if (N <= kVerletFusedMaxAtoms) {   // small cells: the whole pass in one launch
    hipLaunchKernelGGL(k_verlet_update_small, g_for(N * 64), dim3(256), 0, s, N, S, Ec, pos, pos_ref, batch, lattice, cand_edge_index, cand_shift,
                       cand_row_ptr, cutoff, threebody_cutoff, cand_state, w.state, w.dist, w.row_keep, w.row_tri, w.acc);
    M3G_HIP_CHECK(hipMemcpyAsync(host_out, w.acc, sizeof(uint64_t) * 6, hipMemcpyDeviceToHost, s));
    return M3G_OK;
  }
  hipLaunchKernelGGL(k_verlet_prep, g_for(N), dim3(256), 0, s, N, S, pos, pos_ref, lattice, batch, w.pos_w, w.wrap, w.acc);
  hipLaunchKernelGGL(k_verlet_rows, g_for(N * 64), dim3(256), 0, s, N, S, Ec, batch, lattice, cand_edge_index, cand_shift, cand_row_ptr, w.pos_w,
                     w.wrap, cutoff, threebody_cutoff, cand_state, w.state, w.dist, w.row_keep, w.row_tri, w.acc);
  hipLaunchKernelGGL(k_verlet_totals, dim3(1), dim3(1024), 0, s, N, w.row_keep, w.row_tri, cand_row_ptr, w.acc);
  M3G_HIP_CHECK(hipMemcpyAsync(host_out, w.acc, sizeof(uint64_t) * 6, hipMemcpyDeviceToHost, s));
  return M3G_OK;
}

extern "C" int m3g_verlet_update(int64_t N, int64_t S, int64_t Ec, const double* pos, const double* pos_ref, const double* lattice,
                                 const int64_t* batch, const int64_t* cand_edge_index, const int32_t* cand_shift, const int32_t* cand_row_ptr,
                                 double cutoff, float threebody_cutoff, const uint8_t* cand_state, void* scratch, size_t scratch_bytes,
                                 double* host_max_disp, int32_t* host_changed, int64_t* host_n_edges, int64_t* host_n_triplets, void* stream_) {
  if (!host_max_disp || !host_changed || !host_n_edges || !host_n_triplets) { set_error("m3g_verlet_update: null argument"); return M3G_ERR_VALUE; }
  uint64_t h[6] = {0, 0, 0, 0, 0, 0};
  int rc = m3g_verlet_update_async(N, S, Ec, pos, pos_ref, lattice, batch, cand_edge_index, cand_shift, cand_row_ptr, cutoff, threebody_cutoff,
                                   cand_state, scratch, scratch_bytes, h, stream_);
  if (rc) return rc;
  M3G_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  double m2;
  memcpy(&m2, &h[0], sizeof(double));
  *host_max_disp = sqrt(m2);
  *host_changed = h[1] ? 1 : 0;
  *host_n_edges = (int64_t)h[2];
  *host_n_triplets = (int64_t)h[3];
  return M3G_OK;
}

extern "C" int m3g_verlet_fill(int64_t N, int64_t Ec, int64_t n_edges, void* scratch, const int64_t* cand_edge_index, const int32_t* cand_shift,
                               const int32_t* cand_row_ptr, int64_t* edge_index, int32_t* edge_cell_shift, double* distances,
                               uint8_t* cand_state, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (N == 0) return M3G_OK;
  if (!scratch || !cand_row_ptr || !cand_state || (n_edges > 0 && (!edge_index || !edge_cell_shift || !distances))) {
    set_error("m3g_verlet_fill: null argument");
    return M3G_ERR_VALUE;
  }
  VerletScratch w = verlet_carve(N, Ec, scratch);
  M3G_HIP_CHECK(hipMemsetAsync(w.row_keep + N, 0, sizeof(int32_t), s));
  M3G_HIP_CHECK(prims::exclusive_scan<int32_t>(w.row_keep, w.row_keep, N + 1, w.scan_tmp, s));
  hipLaunchKernelGGL(k_verlet_fill, g_for(N * 64), dim3(256), 0, s, N, Ec, n_edges, cand_edge_index, cand_shift, cand_row_ptr, w.row_keep, w.state,
                     w.dist, edge_index, edge_cell_shift, distances, cand_state);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}

// Refill in two launches: edges, shifts, membership bytes, triplets and the triplet counts straight from the candidates and the
// state the preceding m3g_verlet_update(_async) left in `scratch` (n_edges / n_triplets: its E and T).  Requires every candidate
// row to hold at most M3G_VERLET_FILL_LISTS_MAX_ROW entries and n_atoms <= 262,144 (M3G_ERR_UNSUPPORTED otherwise: use
// m3g_verlet_fill + m3g_threebody_build, which have no such limits).  The lists are identical to theirs.  No wait.
extern "C" int m3g_verlet_fill_lists(int64_t N, int64_t Ec, int64_t n_edges, int64_t n_triplets, int64_t max_cand_row, void* scratch,
                                     const int64_t* cand_edge_index, const int32_t* cand_shift, const int32_t* cand_row_ptr, int64_t* edge_index,
                                     int32_t* edge_cell_shift, uint8_t* cand_state, int64_t* triplet_edge_index, int64_t* num_triplet_i,
                                     int32_t* num_triplet_ij, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (N == 0) return M3G_OK;
  if (max_cand_row > kRefillList || N > kRefillMaxAtoms) { set_error("m3g_verlet_fill_lists: candidate rows or atom count beyond its limits"); return M3G_ERR_UNSUPPORTED; }
  if (!scratch || !cand_row_ptr || (Ec > 0 && !cand_state) || (n_edges > 0 && (!edge_index || !edge_cell_shift)) || (n_triplets > 0 && !triplet_edge_index)) {
    set_error("m3g_verlet_fill_lists: null argument");
    return M3G_ERR_VALUE;
  }
  VerletScratch w = verlet_carve(N, Ec, scratch);
  hipLaunchKernelGGL(k_verlet_offsets, dim3(1), dim3(kRefillScanThreads), 0, s, N, w.row_keep, w.row_tri, w.off_e, w.off_t);
  hipLaunchKernelGGL(k_verlet_fill_lists, g_for(N * 64), dim3(256), 0, s, N, Ec, n_edges, n_triplets, cand_edge_index, cand_shift, cand_row_ptr, w.off_e,
                     w.off_t, w.state, edge_index, edge_cell_shift, cand_state, triplet_edge_index, num_triplet_i, num_triplet_ij);
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
  hipLaunchKernelGGL(k_rank_valid, g_for(N * 64), dim3(256), 0, s, N, w.row_ptr, distances, threebody_cutoff, w.rank, w.deg, w.counts);
  M3G_HIP_CHECK(hipMemsetAsync(w.counts + E, 0, sizeof(int64_t), s));
  M3G_HIP_CHECK(prims::exclusive_scan<int64_t>(w.counts, w.counts, E + 1, w.scan_tmp, s));
  int h_flags = 0;
  M3G_HIP_CHECK(hipMemcpyAsync(host_n_triplets, w.counts + E, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipMemcpyAsync(&h_flags, flags, sizeof(int), hipMemcpyDeviceToHost, s));
  M3G_HIP_CHECK(hipStreamSynchronize(s));
  if (h_flags & 1) { set_error("edge_index must be sorted by centre atom (row 0)"); return M3G_ERR_VALUE; }
  return M3G_OK;
}

// count + fill in one call for a caller that already knows n_triplets (m3g_neighbor_count_triplets): nothing waits for the
// device here; a row-0 ordering violation is left to m3g_topology_build, which checks the same property.
extern "C" int m3g_threebody_build(int64_t N, int64_t E, const int64_t* edge_index, const float* distances, float threebody_cutoff,
                                   void* scratch, size_t scratch_bytes, int64_t n_triplets, int64_t* triplet_edge_index,
                                   int64_t* num_triplet_i, int32_t* num_triplet_ij, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  size_t need = 0;
  int rc = m3g_threebody_scratch_bytes(N, E, &need);
  if (rc) return rc;
  if (!scratch || scratch_bytes < need || (n_triplets > 0 && !triplet_edge_index)) { set_error("m3g_threebody_build: scratch too small or null argument"); return M3G_ERR_SIZE; }
  if (N == 0) return M3G_OK;
  TbScratch w = tb_carve(N, E, scratch);
  int* flags = (int*)((char*)scratch + w.total);
  M3G_HIP_CHECK(hipMemsetAsync(flags, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_rows_from_sorted, g_for(N + 1), dim3(256), 0, s, N, E, edge_index, w.row_ptr, flags);
  hipLaunchKernelGGL(k_rank_valid, g_for(N * 64), dim3(256), 0, s, N, w.row_ptr, distances, threebody_cutoff, w.rank, w.deg, w.counts);
  M3G_HIP_CHECK(hipMemsetAsync(w.counts + E, 0, sizeof(int64_t), s));
  M3G_HIP_CHECK(prims::exclusive_scan<int64_t>(w.counts, w.counts, E + 1, w.scan_tmp, s));
  hipLaunchKernelGGL(k_fill_triplets, g_for(N * 64), dim3(256), 0, s, N, E, n_triplets, w.row_ptr, w.rank, w.deg, w.counts, triplet_edge_index,
                     num_triplet_i, num_triplet_ij);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}

extern "C" int m3g_threebody_fill(int64_t N, int64_t E, const int64_t* edge_index, void* scratch, int64_t n_triplets,
                                  int64_t* triplet_edge_index, int64_t* num_triplet_i, int32_t* num_triplet_ij, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (N == 0) return M3G_OK;
  if (!scratch || (n_triplets > 0 && !triplet_edge_index)) { set_error("m3g_threebody_fill: null argument"); return M3G_ERR_VALUE; }
  TbScratch w = tb_carve(N, E, scratch);
  hipLaunchKernelGGL(k_fill_triplets, g_for(N * 64), dim3(256), 0, s, N, E, n_triplets, w.row_ptr, w.rank, w.deg, w.counts, triplet_edge_index,
                     num_triplet_i, num_triplet_ij);
  M3G_HIP_CHECK(hipGetLastError());
  return M3G_OK;
}

// ---- test hooks for the scan / sort primitives of m3g_prims.h (tests/test_gpu_graph_build.py) ----------------------------
// Temporary storage from hipMalloc, a wait at the end: diagnostics, not product paths.
extern "C" int m3g_debug_exclusive_scan(int32_t elem_bytes, int64_t n, const void* in, void* out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if ((elem_bytes != 4 && elem_bytes != 8) || n < 0 || (n > 0 && (!in || !out))) { set_error("m3g_debug_exclusive_scan: bad argument"); return M3G_ERR_VALUE; }
  void* tmp = nullptr;
  M3G_HIP_CHECK(hipMalloc(&tmp, elem_bytes == 4 ? prims::scan_tmp_bytes<int32_t>(n) : prims::scan_tmp_bytes<int64_t>(n)));
  hipError_t e = elem_bytes == 4 ? prims::exclusive_scan<int32_t>((const int32_t*)in, (int32_t*)out, n, tmp, s)
                                 : prims::exclusive_scan<int64_t>((const int64_t*)in, (int64_t*)out, n, tmp, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(tmp);
  if (e != hipSuccess) { set_error("m3g_debug_exclusive_scan: %s", hipGetErrorString(e)); return M3G_ERR_HIP; }
  return M3G_OK;
}
// keys (and vals, may be NULL) are sorted in place by the key bits [begin_bit, end_bit); stable
extern "C" int m3g_debug_radix_sort(int32_t key_bytes, int64_t n, void* keys, int32_t* vals, int32_t begin_bit, int32_t end_bit, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if ((key_bytes != 4 && key_bytes != 8) || n < 0 || (n > 0 && !keys) || begin_bit < 0 || end_bit > 8 * key_bytes || begin_bit > end_bit) {
    set_error("m3g_debug_radix_sort: bad argument");
    return M3G_ERR_VALUE;
  }
  if (n == 0) return M3G_OK;
  void *tmp = nullptr, *kb = nullptr, *vb = nullptr;
  if (hipMalloc(&tmp, prims::sort_tmp_bytes(n)) != hipSuccess || hipMalloc(&kb, (size_t)n * key_bytes) != hipSuccess ||
      (vals && hipMalloc(&vb, (size_t)n * sizeof(int32_t)) != hipSuccess)) {
    (void)hipFree(tmp); (void)hipFree(kb); (void)hipFree(vb);
    set_error("m3g_debug_radix_sort: out of device memory");
    return M3G_ERR_HIP;
  }
  const int where = key_bytes == 4 ? prims::radix_sort<uint32_t, int32_t>((uint32_t*)keys, (uint32_t*)kb, vals, (int32_t*)vb, n, begin_bit, end_bit, tmp, s)
                                   : prims::radix_sort<uint64_t, int32_t>((uint64_t*)keys, (uint64_t*)kb, vals, (int32_t*)vb, n, begin_bit, end_bit, tmp, s);
  hipError_t e = where < 0 ? hipErrorUnknown : hipSuccess;
  if (where == 1) {
    e = hipMemcpyAsync(keys, kb, (size_t)n * key_bytes, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && vals) e = hipMemcpyAsync(vals, vb, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(tmp); (void)hipFree(kb); if (vb) (void)hipFree(vb);
  if (e != hipSuccess) { set_error("m3g_debug_radix_sort failed"); return M3G_ERR_HIP; }
  return M3G_OK;
}
