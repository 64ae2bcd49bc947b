// Node-side stages on the matrix pipe: per-node tables TA / TB / v (k_node_pre_mfma, stage S2) and the readout with its
// reverse (k_readout_mfma, stage S5).  Split from m3g_edge_mfma.hip (shared device code: m3g_edge_common.h).
#include "m3g_edge_common.h"
#include "m3g_struct_sum.h"
#include "m3g_geometry_body.h"

namespace m3g {


// ---------------------------------------------------------------------------------------------- node tables
// S2 on the matrix pipe: [TA | TB | v]^T (528 rows) = W (528 x 64) . x^T (64 x 16 atoms) per 16-atom tile, bf16x3 chains
// (fp32 accumulate) like the edge kernels', the whole weight image (135 KB) resident in LDS.  Replaces the vector-ALU k_node_pre, which
// re-read 128 KB of weights from L2 for every 16 atoms.  x^b = x^(b-1) + the per-centre message sums of block b-1 is
// formed while the tile is loaded (x_prev != nullptr) and written back for the later stages.
constexpr int kNodeXPitch = 68;   // floats per staged x row: 64 + 4 keeps 16-byte alignment and spreads the 16 rows over the banks
// One of the three 11-row-block passes per WORKGROUP (blockIdx.x % 3): its third of the weight image (45 KB + the biases) is all
// the workgroup stages, two to three workgroups share a CU, and a small system's tiles spread over three times as many waves.
// (Until round 3 every workgroup staged the whole 135-KB image and walked all three passes -- or, for small systems, gave its
// waves one pass each but still staged everything: 35 MB of image reads per launch on the 10k-atom cell, 15 us per launch.)
constexpr int kNodePassFloats = 11 * 2 * 512;
template <int PREC>
__global__ void __launch_bounds__(256) k_node_pre_mfma(int C, int64_t N, const float* __restrict__ img, const float* __restrict__ x_prev,
                                                       const float* __restrict__ seg_head, const float* __restrict__ seg_first,
                                                       const int32_t* __restrict__ row_ptr, float* __restrict__ x,
                                                       float* __restrict__ v, float* __restrict__ TA, float* __restrict__ TB,
                                                       const int64_t* __restrict__ types, const float* __restrict__ emb, int num_types,
                                                       float w_inv) {
  constexpr int kBiasFloats = kNodeRowBlocks * 16;
  __shared__ __attribute__((aligned(16))) float lds[kNodePassFloats + kBiasFloats + 4 * 16 * kNodeXPitch];
  const int g = blockIdx.x % 3;   // this workgroup's pass (uniform)
  {  // this pass's image chunk and the biases -> LDS, every 16-byte load of a thread in flight at once
    constexpr int kVec = kNodePassFloats / 4, kPer = (kVec + 255) / 256;
    const float* src = img + (size_t)g * kNodePassFloats;
    f32x4 t[kPer];
    static_for<kPer>([&]<int j>() {
      const int i = j * 256 + (int)threadIdx.x;
      t[j] = i < kVec ? *(const f32x4*)(src + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
    });
    const f32x4 tb = (int)threadIdx.x < kBiasFloats / 4 ? *(const f32x4*)(img + kNodeRowBlocks * 16 * 64 + 4 * threadIdx.x) : f32x4{0.f, 0.f, 0.f, 0.f};
    static_for<kPer>([&]<int j>() {
      const int i = j * 256 + (int)threadIdx.x;
      if (i < kVec) *(f32x4*)(lds + 4 * i) = t[j];
    });
    if ((int)threadIdx.x < kBiasFloats / 4) *(f32x4*)(lds + kNodePassFloats + 4 * threadIdx.x) = tb;
  }
  __syncthreads();
  const float* bias = lds + kNodePassFloats;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, q = lane >> 4;
  float* xs = lds + kNodePassFloats + kBiasFloats + wave * 16 * kNodeXPitch;
  const int64_t tiles = (N + 15) / 16;
  for (int64_t tile = (int64_t)(blockIdx.x / 3) * 4 + wave; tile < tiles; tile += (int64_t)(gridDim.x / 3) * 4) {
    // stage the tile's x rows: lane (m, q) brings features 16q .. 16q+15 of atom m
    const int64_t atom = tile * 16 + m;
    const bool live = atom < N;
    f32x4 xr[4];
    static_for<4>([&]<int j>() { xr[j] = f32x4{0.f, 0.f, 0.f, 0.f}; });
    if (live) {
      const float* src = (x_prev ? x_prev : x) + atom * kDP + 16 * q;
      if (types) {   // block 0: x^0 = atom embedding row (nn/featurizer.py:99-103), formed and stored here
        int64_t ty = types[atom];
        ty = ty < 0 ? 0 : (ty >= num_types ? num_types - 1 : ty);
        src = emb + ty * kDP + 16 * q;
      }
      static_for<4>([&]<int j>() { xr[j] = *(const f32x4*)(src + 4 * j); });
      if (types && g == 0) static_for<4>([&]<int j>() { *(f32x4*)(x + atom * kDP + 16 * q + 4 * j) = xr[j]; });
      if (x_prev) {
        const int r0 = row_ptr[atom], r1 = row_ptr[atom + 1];
        if (r1 > r0) {
          if (r0 & 15) static_for<4>([&]<int j>() { xr[j] += *(const f32x4*)(seg_first + atom * (4 * kDP) + 16 * q + 4 * j); });
          for (int t = (r0 + 15) >> 4; t <= (r1 - 1) >> 4; ++t)
            static_for<4>([&]<int j>() { xr[j] += *(const f32x4*)(seg_head + (int64_t)t * (4 * kDP) + 16 * q + 4 * j); });
        }
        if (g == 0) static_for<4>([&]<int j>() { *(f32x4*)(x + atom * kDP + 16 * q + 4 * j) = xr[j]; });   // (the pass-0 workgroup writes x^b back)
      }
    }
    static_for<4>([&]<int j>() { *(f32x4*)(xs + m * kNodeXPitch + 16 * q + 4 * j) = xr[j]; });
    // (only this wave reads xs: LDS operations of a wave complete in order)
    // x as accumulator-layout blocks: lane (m, q) holds features blk*16 + 4q + {0..3} of atom m
    f32x4 xb[4];
    static_for<4>([&]<int blk>() { xb[blk] = *(const f32x4*)(xs + m * kNodeXPitch + blk * 16 + 4 * q); });
    int lv = lane;
    asm volatile("" : "+v"(lv));   // keep the image reads inside the tile loop
    static_for<3>([&]<int G>() {   // 11 row blocks: split-precision chains like the edge kernels' (fp32 accumulate)
      if (g != G) return;
      f32x4 acc[11];
      static_for<11>([&]<int j>() { acc[j] = *(const f32x4*)(bias + (11 * G + j) * 16 + 4 * q); });
      chain_p<PREC, 11, 2>(lds, xb, acc, lv, w_inv);
      if (live) {
        static_for<11>([&]<int j>() {
          constexpr int ob = 11 * G + j;
          if (ob < 16) *(f32x4*)(TA + atom * (4 * kDP) + ob * 16 + 4 * q) = acc[j];
          else if (ob < 32) *(f32x4*)(TB + atom * (4 * kDP) + (ob - 16) * 16 + 4 * q) = acc[j];
          else {
            f32x4 o;
            static_for<4>([&]<int r>() { o[r] = 4 * q + r < C ? fsigmoid(acc[j][r]) : 0.f; });
            *(f32x4*)(v + atom * kCP + 4 * q) = o;
          }
        });
      }
    });
  }
}

// Small systems: the node tables of one 16-atom tile and one of the three 11-row-block passes per WORKGROUP, the pass's row blocks
// dealt to its four waves (w, w + 4, w + 8), each wave's weight rows resident in registers (48): 176 exact-fp32 MFMAs per tile and
// pass take one wave 2.4 us behind a 45-KB image copy, here 48 per wave and no copy.  Rows are independent: no exchange, no barrier.
// Same chains, same bits (fp32 mode only: the split modes' images have another layout).
struct NodePreArgs {
  int C;
  int64_t N;
  const float *img, *x_prev, *seg_head, *seg_first;
  const int32_t* row_ptr;
  float *x, *v, *TA, *TB;
  const int64_t* types;
  const float* emb;
  int num_types;
};
__device__ __forceinline__ void node_pre_split_body(const NodePreArgs& args, int64_t vblock) {
  const int C = args.C;
  const int64_t N = args.N;
  const float* __restrict__ img = args.img;
  const float* __restrict__ x_prev = args.x_prev;
  const float* __restrict__ seg_head = args.seg_head;
  const float* __restrict__ seg_first = args.seg_first;
  const int32_t* __restrict__ row_ptr = args.row_ptr;
  float* __restrict__ x = args.x;
  float* __restrict__ v = args.v;
  float* __restrict__ TA = args.TA;
  float* __restrict__ TB = args.TB;
  const int64_t* __restrict__ types = args.types;
  const float* __restrict__ emb = args.emb;
  const int num_types = args.num_types;
  __shared__ __attribute__((aligned(16))) float xs_all[4 * 16 * kNodeXPitch];
  const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = (int)(vblock % 3);   // this workgroup's pass (uniform)
  const int64_t tile = vblock / 3;
  float* xs = xs_all + w * 16 * kNodeXPitch;
  // this wave's row blocks of the pass: j = w, w + 4, w + 8 (< 11); image [33 row blocks][16 k-steps][64] + [528] biases
  float a[3][16];
  f32x4 acc[3];
  static_for<3>([&]<int jj>() {
    const int j = w + 4 * jj;
    const int ob = 11 * g + (j < 11 ? j : 0);
    static_for<16>([&]<int k>() { a[jj][k] = img[(ob * 16 + k) * 64 + lane]; });
    acc[jj] = *(const f32x4*)(img + kNodeRowBlocks * 16 * 64 + ob * 16 + 4 * q);
  });
  const int64_t atom = tile * 16 + m;
  const bool live = atom < N;
  f32x4 xr[4];
  static_for<4>([&]<int j>() { xr[j] = f32x4{0.f, 0.f, 0.f, 0.f}; });
  if (live) {
    const float* src = (x_prev ? x_prev : x) + atom * kDP + 16 * q;
    if (types) {   // block 0: x^0 = atom embedding row (nn/featurizer.py:99-103), formed and stored here
      int64_t ty = types[atom];
      ty = ty < 0 ? 0 : (ty >= num_types ? num_types - 1 : ty);
      src = emb + ty * kDP + 16 * q;
    }
    static_for<4>([&]<int j>() { xr[j] = *(const f32x4*)(src + 4 * j); });
    if (types && g == 0 && w == 0) static_for<4>([&]<int j>() { *(f32x4*)(x + atom * kDP + 16 * q + 4 * j) = xr[j]; });
    if (x_prev) {
      const int r0 = row_ptr[atom], r1 = row_ptr[atom + 1];
      if (r1 > r0) {
        if (r0 & 15) static_for<4>([&]<int j>() { xr[j] += *(const f32x4*)(seg_first + atom * (4 * kDP) + 16 * q + 4 * j); });
        for (int t = (r0 + 15) >> 4; t <= (r1 - 1) >> 4; ++t)
          static_for<4>([&]<int j>() { xr[j] += *(const f32x4*)(seg_head + (int64_t)t * (4 * kDP) + 16 * q + 4 * j); });
      }
      if (g == 0 && w == 0) static_for<4>([&]<int j>() { *(f32x4*)(x + atom * kDP + 16 * q + 4 * j) = xr[j]; });   // (one wave writes x^b back)
    }
  }
  static_for<4>([&]<int j>() { *(f32x4*)(xs + m * kNodeXPitch + 16 * q + 4 * j) = xr[j]; });
  f32x4 xb[4];   // (only this wave reads its staging area)
  static_for<4>([&]<int blk>() { xb[blk] = *(const f32x4*)(xs + m * kNodeXPitch + blk * 16 + 4 * q); });
  static_for<4>([&]<int blk>() {
    static_for<4>([&]<int r>() {
      const float b = xb[blk][r];
      static_for<3>([&]<int jj>() { acc[jj] = mfma16(a[jj][blk * 4 + r], b, acc[jj]); });
    });
  });
  if (!live) return;
  static_for<3>([&]<int jj>() {
    const int j = w + 4 * jj;
    if (j >= 11) return;
    const int ob = 11 * g + j;
    if (ob < 16) *(f32x4*)(TA + atom * (4 * kDP) + ob * 16 + 4 * q) = acc[jj];
    else if (ob < 32) *(f32x4*)(TB + atom * (4 * kDP) + (ob - 16) * 16 + 4 * q) = acc[jj];
    else {
      f32x4 o;
      static_for<4>([&]<int r>() { o[r] = 4 * q + r < C ? fsigmoid(acc[jj][r]) : 0.f; });
      *(f32x4*)(v + atom * kCP + 4 * q) = o;
    }
  });
}

__global__ void __launch_bounds__(256) k_node_pre_split(NodePreArgs a) { node_pre_split_body(a, blockIdx.x); }
// Block 0's node tables need nothing from the geometry stage (x^0 is the species embedding) and the geometry stage nothing from
// them: for small systems the two run as the two workgroup ROLES of one launch (no dependency, no fence -- one launch boundary
// less).  Same device functions as the two kernels: same bits.
template <int L, int R>
__global__ void __launch_bounds__(256) k_geometry_node_pre(Consts c, GeomArgs ga, int n_geo, NodePreArgs na) {
  if ((int)blockIdx.x < n_geo) geometry_body<true, L, R>(c, ga, blockIdx.x);
  else node_pre_split_body(na, (int64_t)blockIdx.x - n_geo);
}

// ---------------------------------------------------------------------------------------------- readout
// S5 (nn/readout.py:39-58) and its reverse on the matrix pipe, per 16-atom tile: both layers of the dense and the gate
// branch as exact-fp32 MFMA chains (this stage seeds the reverse pass), the final 64 -> 1 products as lane-local dots + a lane-quarter sum, then (forces wanted) the
// transposed chains back to dE/dx.  All seven weight images (130 KB) resident in LDS; x^B = x^(B-1) + per-centre message
// sums of the last block is formed while the tile is loaded.  Replaces the vector-ALU k_readout on the MFMA path.
// per-structure sums formed by the launch's last workgroup (counter == nullptr: a separate k_struct_energy launch follows)
struct ReadoutSums {
  const int32_t *struct_ptr, *flags, *batch;
  float* total;
  int32_t* counter;
};
template <int PREC>   // kPrecF32: exact fp32 MFMA chains; kPrecF16x3: the same layers on scaled two-part fp16 operands (w_inv = 1 / weight scale)
__global__ void __launch_bounds__(256) k_readout_mfma(Consts c, int64_t N, const float* __restrict__ img, float w_inv, const float* __restrict__ elemental,
                                                      const int64_t* __restrict__ types, const float* __restrict__ x_prev,
                                                      const float* __restrict__ seg_head, const float* __restrict__ seg_first,
                                                      const int32_t* __restrict__ row_ptr, float* __restrict__ x,
                                                      float* scaled_atomic, float* __restrict__ dx,
                                                      float* __restrict__ scaled_total, int64_t S, ReadoutSums rs) {
  __shared__ __attribute__((aligned(16))) float lds[ReadoutImg::total + 4 * 16 * kNodeXPitch];
  // the per-structure sums are accumulated with atomics by the next kernel: cleared here instead of a memset launch
  if (blockIdx.x == 0) for (int64_t i = threadIdx.x; i < S; i += blockDim.x) scaled_total[i] = 0.f;
  {
    constexpr int kVec = ReadoutImg::total / 4, kBatch = 16;
    for (int base = 0; base < kVec; base += 256 * kBatch) {
      f32x4 t[kBatch];
      static_for<kBatch>([&]<int j>() {
        const int i = base + j * 256 + (int)threadIdx.x;
        t[j] = i < kVec ? *(const f32x4*)(img + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
      });
      static_for<kBatch>([&]<int j>() {
        const int i = base + j * 256 + (int)threadIdx.x;
        if (i < kVec) *(f32x4*)(lds + 4 * i) = t[j];
      });
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, q = lane >> 4;
  float* xs = lds + ReadoutImg::total + wave * 16 * kNodeXPitch;
  const int64_t tiles = (N + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t atom = tile * 16 + m;
    const bool live = atom < N;
    f32x4 xr[4];
    static_for<4>([&]<int j>() { xr[j] = f32x4{0.f, 0.f, 0.f, 0.f}; });
    if (live) {
      const float* src = (x_prev ? x_prev : x) + atom * kDP + 16 * q;
      static_for<4>([&]<int j>() { xr[j] = *(const f32x4*)(src + 4 * j); });
      if (x_prev) {
        const int r0 = row_ptr[atom], r1 = row_ptr[atom + 1];
        if (r1 > r0) {
          if (r0 & 15) static_for<4>([&]<int j>() { xr[j] += *(const f32x4*)(seg_first + atom * (4 * kDP) + 16 * q + 4 * j); });
          for (int t = (r0 + 15) >> 4; t <= (r1 - 1) >> 4; ++t)
            static_for<4>([&]<int j>() { xr[j] += *(const f32x4*)(seg_head + (int64_t)t * (4 * kDP) + 16 * q + 4 * j); });
        }
        static_for<4>([&]<int j>() { *(f32x4*)(x + atom * kDP + 16 * q + 4 * j) = xr[j]; });
      }
    }
    static_for<4>([&]<int j>() { *(f32x4*)(xs + m * kNodeXPitch + 16 * q + 4 * j) = xr[j]; });
    f32x4 xb[4];
    static_for<4>([&]<int blk>() { xb[blk] = *(const f32x4*)(xs + m * kNodeXPitch + blk * 16 + 4 * q); });
    int lv = lane;
    asm volatile("" : "+v"(lv));
    // layer 1 (dense blocks 0-3, gate 4-7): p1 -> hidden, p1 keeps SiLU'
    f32x4 p1[8], hid[8];
    static_for<8>([&]<int ob>() { p1[ob] = *(const f32x4*)(lds + ReadoutImg::b1 + ob * 16 + 4 * q); });
    chain_p<PREC, 8, 2>(lds + ReadoutImg::w1, xb, p1, lv, w_inv);
    static_for<8>([&]<int ob>() {
      static_for<4>([&]<int r>() {
        const float p = p1[ob][r], sg = fsigmoid(p);
        hid[ob][r] = p * sg;
        p1[ob][r] = sg * (1.f + p * (1.f - sg));
      });
    });
    // layer 2
    f32x4 p2[8];
    static_for<8>([&]<int ob>() { p2[ob] = *(const f32x4*)(lds + ReadoutImg::b2 + ob * 16 + 4 * q); });
    chain_p<PREC, 4, 2, 0, 0>(lds + ReadoutImg::w2d, hid, p2, lv, w_inv);
    chain_p<PREC, 4, 2, 4, 4>(lds + ReadoutImg::w2g, hid, p2, lv, w_inv);
    // final 64 -> 1 of both branches: lane-local dots over this lane's 16 features, then across the four lane quarters
    float od = 0.f, og = 0.f;
    f32x4 w3[8];
    static_for<8>([&]<int ob>() { w3[ob] = *(const f32x4*)(lds + ReadoutImg::w3 + ob * 16 + 4 * q); });
    static_for<4>([&]<int ob>() {
      static_for<4>([&]<int r>() {
        od += w3[ob][r] * fsilu(p2[ob][r]);
        og += w3[4 + ob][r] * fsilu(p2[4 + ob][r]);
      });
    });
    od = sum_lane_quarters(od) + lds[ReadoutImg::b3];
    og = sum_lane_quarters(og) + lds[ReadoutImg::b3 + 1];
    const float sg = fsigmoid(og);
    if (live && q == 0) {
      bool bad;
      const int64_t ty = species_index(types[atom], c.num_types, bad);
      scaled_atomic[atom] = bad ? __builtin_nanf("") : elemental[ty] / c.energy_scale + od * sg;
      if (bad) flag_bad_species(rs.flags);
    }
    if (dx == nullptr) continue;   // uniform
    // reverse: dL/d eps = energy_scale
    const float d_od = c.energy_scale * sg, d_og = c.energy_scale * od * sg * (1.f - sg);
    f32x4 d2[8];
    static_for<4>([&]<int ob>() {
      static_for<4>([&]<int r>() {
        d2[ob][r] = d_od * w3[ob][r] * fdsilu(p2[ob][r]);
        d2[4 + ob][r] = d_og * w3[4 + ob][r] * fdsilu(p2[4 + ob][r]);
      });
    });
    f32x4 dp1[8];
    zero(dp1);
    chain_p<PREC, 4, 2, 0, 0>(lds + ReadoutImg::w2dT, d2, dp1, lv, w_inv);
    chain_p<PREC, 4, 2, 4, 4>(lds + ReadoutImg::w2gT, d2, dp1, lv, w_inv);
    static_for<8>([&]<int ob>() { dp1[ob] *= p1[ob]; });
    f32x4 dxb[4];
    zero(dxb);
    chain_p<PREC, 4, 4>(lds + ReadoutImg::w1T, dp1, dxb, lv, w_inv);
    if (live) static_for<4>([&]<int blk>() { *(f32x4*)(dx + atom * kDP + blk * 16 + 4 * q) = dxb[blk]; });
  }
  if (!rs.counter) return;   // uniform
  // per-structure energy sums by the LAST workgroup of this launch (few structures: one launch less than k_struct_energy, same
  // fixed summation order -> bit-identical totals; no workgroup waits for another)
  __shared__ int s_last;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(rs.counter, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  float* part = lds + ReadoutImg::total;   // the x staging area (4 x 16 x kNodeXPitch floats >= kStructThreads)
  static_assert(4 * 16 * kNodeXPitch >= kStructThreads, "staging area too small for the energy sums");
  for (int sidx = 0; sidx < (int)S; ++sidx)
    struct_energy<256>(sidx, rs.struct_ptr, rs.flags, N, rs.batch, scaled_atomic, c.energy_scale, scaled_total, rs.total, part);
}

// Small systems: one 16-atom tile per WORKGROUP, its four waves each owning a quarter of every layer's output rows (dense
// block w, gate block 4 + w), the wave's weight operands resident in registers -- the scheme of m3g_edge_small.hip.  A tile is
// 512 exact-fp32 MFMAs (7 us for the one wave k_readout_mfma gives it, behind a 130-KB image copy); here 128 per wave.
// Every layer output is the same k-ordered chain, and the two 64 -> 1 sums are formed by every wave over ALL blocks in
// k_readout_mfma's order (the layer-2 pre-activations cross through LDS): energies and dE/dx are bit-identical.
template <bool GRAD>
__global__ void __launch_bounds__(256) k_readout_split(Consts c, int64_t N, const float* __restrict__ img, const float* __restrict__ elemental,
                                                       const int64_t* __restrict__ types, const float* __restrict__ x_prev,
                                                       const float* __restrict__ seg_head, const float* __restrict__ seg_first,
                                                       const int32_t* __restrict__ row_ptr, float* __restrict__ x, float* scaled_atomic,
                                                       float* __restrict__ dx, float* __restrict__ scaled_total, int64_t S, ReadoutSums rs) {
  __shared__ __attribute__((aligned(16))) float xs[4 * 16 * kNodeXPitch];   // per-wave x staging (also the energy sums' scratch)
  __shared__ __attribute__((aligned(16))) float hs[8 * 256];                // exchanged activations / gradients, two blocks per wave
  const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* xw = xs + w * 16 * kNodeXPitch;
  // this wave's rows of the images, once, into registers
  float a_w1[2][16], a_w2[2][16], a_w2t[2][16], a_w1t[32];
  static_for<2>([&]<int hf>() {
    static_for<16>([&]<int k>() {
      a_w1[hf][k] = img[ReadoutImg::w1 + ((hf * 4 + w) * 16 + k) * 64 + lane];
      a_w2[hf][k] = img[(hf == 0 ? ReadoutImg::w2d : ReadoutImg::w2g) + (w * 16 + k) * 64 + lane];
      if (GRAD) a_w2t[hf][k] = img[(hf == 0 ? ReadoutImg::w2dT : ReadoutImg::w2gT) + (w * 16 + k) * 64 + lane];
    });
  });
  if (GRAD) static_for<32>([&]<int k>() { a_w1t[k] = img[ReadoutImg::w1T + (w * 32 + k) * 64 + lane]; });
  f32x4 b1[2], b2[2], w3[8];
  static_for<2>([&]<int hf>() {
    b1[hf] = *(const f32x4*)(img + ReadoutImg::b1 + (hf * 4 + w) * 16 + 4 * q);
    b2[hf] = *(const f32x4*)(img + ReadoutImg::b2 + (hf * 4 + w) * 16 + 4 * q);
  });
  static_for<8>([&]<int ob>() { w3[ob] = *(const f32x4*)(img + ReadoutImg::w3 + ob * 16 + 4 * q); });
  const float b3d = img[ReadoutImg::b3], b3g = img[ReadoutImg::b3 + 1];
  const int64_t tiles = (N + 15) / 16;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t atom = tile * 16 + m;
    const bool live = atom < N;
    // x^B = x^(B-1) + per-centre message sums, formed by every wave for itself (wave 0 stores it)
    f32x4 xr[4];
    static_for<4>([&]<int j>() { xr[j] = f32x4{0.f, 0.f, 0.f, 0.f}; });
    if (live) {
      const float* src = (x_prev ? x_prev : x) + atom * kDP + 16 * q;
      static_for<4>([&]<int j>() { xr[j] = *(const f32x4*)(src + 4 * j); });
      if (x_prev) {
        const int r0 = row_ptr[atom], r1 = row_ptr[atom + 1];
        if (r1 > r0) {
          if (r0 & 15) static_for<4>([&]<int j>() { xr[j] += *(const f32x4*)(seg_first + atom * (4 * kDP) + 16 * q + 4 * j); });
          for (int t = (r0 + 15) >> 4; t <= (r1 - 1) >> 4; ++t)
            static_for<4>([&]<int j>() { xr[j] += *(const f32x4*)(seg_head + (int64_t)t * (4 * kDP) + 16 * q + 4 * j); });
        }
        if (w == 0) static_for<4>([&]<int j>() { *(f32x4*)(x + atom * kDP + 16 * q + 4 * j) = xr[j]; });
      }
    }
    static_for<4>([&]<int j>() { *(f32x4*)(xw + m * kNodeXPitch + 16 * q + 4 * j) = xr[j]; });
    f32x4 xb[4];   // (only this wave reads its staging area: LDS operations of a wave complete in order)
    static_for<4>([&]<int blk>() { xb[blk] = *(const f32x4*)(xw + m * kNodeXPitch + blk * 16 + 4 * q); });
    // layer 1, rows w (dense) and 4 + w (gate): p1 -> hidden, p1 keeps SiLU'
    f32x4 p1d = b1[0], p1g = b1[1];
    static_for<4>([&]<int blk>() {
      static_for<4>([&]<int r>() {
        p1d = mfma16(a_w1[0][blk * 4 + r], xb[blk][r], p1d);
        p1g = mfma16(a_w1[1][blk * 4 + r], xb[blk][r], p1g);
      });
    });
    f32x4 hd, hg;
    static_for<4>([&]<int r>() {
      float p = p1d[r], sg = fsigmoid(p);
      hd[r] = p * sg; p1d[r] = sg * (1.f + p * (1.f - sg));
      p = p1g[r]; sg = fsigmoid(p);
      hg[r] = p * sg; p1g[r] = sg * (1.f + p * (1.f - sg));
    });
    *(f32x4*)(hs + w * 256 + lane * 4) = hd;
    *(f32x4*)(hs + (4 + w) * 256 + lane * 4) = hg;
    __syncthreads();
    f32x4 hid[8];
    static_for<8>([&]<int ob>() { hid[ob] = *(const f32x4*)(hs + ob * 256 + lane * 4); });
    // layer 2, rows w of both branches
    f32x4 p2d = b2[0], p2g = b2[1];
    static_for<4>([&]<int blk>() {
      static_for<4>([&]<int r>() {
        p2d = mfma16(a_w2[0][blk * 4 + r], hid[blk][r], p2d);
        p2g = mfma16(a_w2[1][blk * 4 + r], hid[4 + blk][r], p2g);
      });
    });
    __syncthreads();   // every wave has read the hidden activations: hs is free
    *(f32x4*)(hs + w * 256 + lane * 4) = p2d;
    *(f32x4*)(hs + (4 + w) * 256 + lane * 4) = p2g;
    __syncthreads();
    f32x4 p2[8];
    static_for<8>([&]<int ob>() { p2[ob] = *(const f32x4*)(hs + ob * 256 + lane * 4); });
    // final 64 -> 1 of both branches over ALL blocks, in k_readout_mfma's order
    float od = 0.f, og = 0.f;
    static_for<4>([&]<int ob>() {
      static_for<4>([&]<int r>() {
        od += w3[ob][r] * fsilu(p2[ob][r]);
        og += w3[4 + ob][r] * fsilu(p2[4 + ob][r]);
      });
    });
    od = sum_lane_quarters(od) + b3d;
    og = sum_lane_quarters(og) + b3g;
    const float sg = fsigmoid(og);
    if (live && q == 0 && w == 0) {
      bool bad;
      const int64_t ty = species_index(types[atom], c.num_types, bad);
      scaled_atomic[atom] = bad ? __builtin_nanf("") : elemental[ty] / c.energy_scale + od * sg;
      if (bad) flag_bad_species(rs.flags);
    }
    if constexpr (GRAD) {
      // reverse: dL/d eps = energy_scale
      const float d_od = c.energy_scale * sg, d_og = c.energy_scale * od * sg * (1.f - sg);
      f32x4 d2d, d2g;
      static_for<4>([&]<int r>() {
        d2d[r] = d_od * w3[w][r] * fdsilu(p2d[r]);
        d2g[r] = d_og * w3[4 + w][r] * fdsilu(p2g[r]);
      });
      __syncthreads();   // every wave has read p2
      *(f32x4*)(hs + w * 256 + lane * 4) = d2d;
      *(f32x4*)(hs + (4 + w) * 256 + lane * 4) = d2g;
      __syncthreads();
      f32x4 d2[8];
      static_for<8>([&]<int ob>() { d2[ob] = *(const f32x4*)(hs + ob * 256 + lane * 4); });
      f32x4 dp1d = {0.f, 0.f, 0.f, 0.f}, dp1g = {0.f, 0.f, 0.f, 0.f};
      static_for<4>([&]<int blk>() {
        static_for<4>([&]<int r>() {
          dp1d = mfma16(a_w2t[0][blk * 4 + r], d2[blk][r], dp1d);
          dp1g = mfma16(a_w2t[1][blk * 4 + r], d2[4 + blk][r], dp1g);
        });
      });
      dp1d *= p1d;
      dp1g *= p1g;
      __syncthreads();   // every wave has read d2
      *(f32x4*)(hs + w * 256 + lane * 4) = dp1d;
      *(f32x4*)(hs + (4 + w) * 256 + lane * 4) = dp1g;
      __syncthreads();
      f32x4 dp1[8];
      static_for<8>([&]<int ob>() { dp1[ob] = *(const f32x4*)(hs + ob * 256 + lane * 4); });
      f32x4 dxb = {0.f, 0.f, 0.f, 0.f};
      static_for<8>([&]<int blk>() { static_for<4>([&]<int r>() { dxb = mfma16(a_w1t[blk * 4 + r], dp1[blk][r], dxb); }); });
      if (live) *(f32x4*)(dx + atom * kDP + w * 16 + 4 * q) = dxb;
    }
    __syncthreads();   // hs is rewritten by the next tile
  }
  if (!rs.counter) return;   // uniform
  __shared__ int s_last;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(rs.counter, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  for (int sidx = 0; sidx < (int)S; ++sidx)
    struct_energy<256>(sidx, rs.struct_ptr, rs.flags, N, rs.batch, scaled_atomic, c.energy_scale, scaled_total, rs.total, xs);
}

void launch_readout_mfma(const m3g_plan* plan, const Consts& c, const WeightLayout& wl, const Topo& t, const int64_t* types,
                         const float* x_prev, float* x, const Work& w, float* scaled_atomic, float* scaled_total, float* total,
                         bool want_grad, hipStream_t s, bool* energy_sums_deferred) {
  const bool may_defer = energy_sums_deferred && *energy_sums_deferred;
  if (energy_sums_deferred) *energy_sums_deferred = false;
  if (t.N == 0) (void)hipMemsetAsync(scaled_total, 0, sizeof(float) * t.S, s);
  bool sums_fused = false;
  if (t.N > 0) {
    const int64_t tiles = (t.N + 15) / 16;
    const int wgs = (int)std::min<int64_t>((tiles + 3) / 4, 256);
    sums_fused = plan->small_launches && w.sync && t.N <= kFusedSumsMaxAtoms && t.S > 0 && t.S <= kForceTailMaxStructs;
    const ReadoutSums rs{t.struct_ptr, t.flags, t.batch, total, sums_fused ? w.sync + kSyncReadout : nullptr};
    const bool f16_readout = plan->precision == kPrecF16x3 && plan->readout_f16;
    if (!f16_readout && plan->small_launches && tiles <= plan->split_node_tiles) {   // small systems: a tile over the four waves of a workgroup
      if (want_grad)
        hipLaunchKernelGGL(k_readout_split<true>, dim3((unsigned)tiles), dim3(256), 0, s, c, t.N, plan->d_readout_img, plan->d_weights + wl.elemental,
                           types, x_prev, w.seg_head, w.seg_first, t.row_ptr, x, scaled_atomic, w.dx, scaled_total, t.S, rs);
      else
        hipLaunchKernelGGL(k_readout_split<false>, dim3((unsigned)tiles), dim3(256), 0, s, c, t.N, plan->d_readout_img, plan->d_weights + wl.elemental,
                           types, x_prev, w.seg_head, w.seg_first, t.row_ptr, x, scaled_atomic, nullptr, scaled_total, t.S, rs);
    } else
    if (plan->precision == kPrecF16x3 && plan->readout_f16)   // (option; default: exact-fp32 readout in every mode)
      hipLaunchKernelGGL(k_readout_mfma<kPrecF16x3>, dim3(wgs), dim3(256), 0, s, c, t.N, plan->d_readout_img_h, plan->ro_w_scale_inv,
                         plan->d_weights + wl.elemental, types, x_prev, w.seg_head, w.seg_first, t.row_ptr, x, scaled_atomic,
                         want_grad ? w.dx : nullptr, scaled_total, t.S, rs);
    else   // every mode by default: this stage forms the energies and seeds the reverse pass (bf16x3 products here moved the Cu-32 virial from
           // 4.5e-5 to 1.2e-4 of its fp64 value; f16x3 products put a six-atom structure's ill-conditioned energy 5.3e-5 off instead of 9e-6)
      hipLaunchKernelGGL(k_readout_mfma<kPrecF32>, dim3(wgs), dim3(256), 0, s, c, t.N, plan->d_readout_img, 1.f,
                         plan->d_weights + wl.elemental, types, x_prev, w.seg_head, w.seg_first, t.row_ptr, x, scaled_atomic,
                         want_grad ? w.dx : nullptr, scaled_total, t.S, rs);
  }
  if (!sums_fused && may_defer && t.N > 0) *energy_sums_deferred = true;   // formed by the step's last launch (k_struct_stress)
  else if (!sums_fused) launch_energy_sums(c, t, scaled_atomic, scaled_total, total, s);
}

// types != nullptr (block 0): x is formed from the atom embedding `emb` ([num_types][kDP]) instead of being read
void launch_node_pre_mfma(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* x_prev, float* x,
                          float* v, float* TA, float* TB, const int64_t* types, const float* emb, hipStream_t s) {
  if (t.N == 0) return;
  const int64_t tiles = (t.N + 15) / 16;
  if (plan->precision == kPrecF32 && plan->small_launches && tiles <= plan->split_node_tiles) {   // small systems: a tile and pass per workgroup
    const NodePreArgs a{c.C, t.N, plan->d_node_img[kPrecF32] + (size_t)b * kNodeImgFloats, x_prev, w.seg_head, w.seg_first, t.row_ptr, x, v, TA, TB,
                        types, emb, c.num_types};
    hipLaunchKernelGGL(k_node_pre_split, dim3((unsigned)(3 * tiles)), dim3(256), 0, s, a);
    return;
  }
  const int wgs = 3 * (int)std::min<int64_t>((tiles + 3) / 4, 256);   // (pass, group of four tiles); groups beyond 256 loop
  M3G_PREC_SWITCH(plan->precision,
                  hipLaunchKernelGGL((k_node_pre_mfma<PREC>), dim3(wgs), dim3(256), 0, s, c.C, t.N,
                                     plan->d_node_img[plan->precision] + (size_t)b * kNodeImgFloats, x_prev, w.seg_head, w.seg_first,
                                     t.row_ptr, x, v, TA, TB, types, emb, c.num_types,
                                     plan->precision == kPrecF16x3 ? plan->w_scale_inv : 1.f));
}

// geometry stage + block 0's node tables in one launch (small systems, exact-fp32 mode); false: not a case it covers
bool launch_geometry_node_pre(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, const float* pos, const float* lattice,
                              const int32_t* shift, const int64_t* types, const float* emb, hipStream_t s) {
  const int64_t tiles = (t.N + 15) / 16;
  if (plan->precision != kPrecF32 || !plan->small_launches || c.B == 0 || t.E == 0 || t.N == 0 || tiles > plan->split_node_tiles) return false;
  const int n_geo = (int)((t.E + 255) / 256);
  const GeomArgs ga = geometry_args(t, pos, lattice, shift, w);
  const NodePreArgs na{c.C, t.N, plan->d_node_img[kPrecF32], nullptr, w.seg_head, w.seg_first, t.row_ptr, w.x[0], w.v[0], w.TAb[0], w.TBb[0], types, emb,
                       c.num_types};
  M3G_DISPATCH_LR(c.L, c.R, hipLaunchKernelGGL((k_geometry_node_pre<L, R>), dim3((unsigned)(n_geo + 3 * tiles)), dim3(256), 0, s, c, ga, n_geo, na));
  return true;
}

}  // namespace m3g
