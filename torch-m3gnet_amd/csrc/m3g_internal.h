// Internal declarations shared by the translation units of libm3gnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/m3gnet_hip.h"

namespace m3g {

constexpr int kDP = 64;      // padded feature width (embedding_dim <= kDP; zero padding is exact)
constexpr int kLCap = 4;     // l_max <= kLCap
constexpr int kRCap = 4;     // n_max <= kRCap
constexpr int kCP = 16;      // padded l_max*n_max
constexpr int kRP = 4;       // padded n_max
constexpr int kMaxBlocks = 8;
// arithmetic of the dense chains (plan option "precision"; m3g_edge_mfma.hip: chain_p)
constexpr int kPrecF32 = 0;      // v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate -- the reference's arithmetic (default)
constexpr int kPrecBf16x3 = 1;   // 3 v_mfma_f32_16x16x32_bf16 products of 2-way bf16 splits per fp32 product, fp32 accumulate
constexpr int kPrecF16x3 = 2;    // 3 v_mfma_f32_16x16x32_f16 products of 2-way fp16 splits of power-of-two SCALED operands: both parts of
                                 // an operand together carry 22-24 significant bits (an fp32 value to within its own rounding), fp32 accumulate
constexpr int kNumPrec = 3;

void set_error(const char* fmt, ...);
#define M3G_HIP_CHECK(expr)                                                               \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::m3g::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return M3G_ERR_HIP;                                                                 \
    }                                                                                     \
  } while (0)

// ---- small constants passed to kernels by value -------------------------------------------------
struct Consts {
  int L, R, C, D, B, num_types;
  float inv_len;        // 1 / length_scale is NOT used for pos (reference divides); kept for forces
  float length_scale, energy_scale;
  float rc, rc3;        // scaled cutoffs
  float a1[kRCap], a2[kRCap];          // (m+1)*pi/rc, (m+2)*pi/rc formed in fp32 like the reference
  float coeff[kRCap];
  float rec_mul[kRCap], rec_div[kRCap]; // sqrt(em[m]/dm[m-1]), sqrt(dm[m])
  float zeros[kLCap][kRCap];
  float factors[kLCap][kRCap];
  float ynorm[kLCap];   // sqrt((2l+1)/(4 pi))
};

// ---- packed weight blob: offsets in floats from the blob base -------------------------------------
struct MlpW {
  // forward orientation, k-major ("_t" = transposed w.r.t. torch's [out,in])
  size_t w1a_t, w1b_t, w1c_t;  // [kDP][2*kDP]  columns: dense | gate
  size_t b1;                   // [2*kDP]
  size_t w2d_t, w2g_t;         // [kDP][kDP]
  size_t b2d, b2g;             // [kDP]
  size_t wl_t;                 // [kRP][kDP]
  // reverse orientation, out-major (torch layout, padded)
  size_t w1a, w1b, w1c;        // [2*kDP][kDP]
  size_t w2d, w2g;             // [kDP][kDP]
  size_t wl;                   // [kDP][kRP]
};
struct BlockW {
  size_t tb_w1_t;   // [kDP][kCP]   v = sigmoid(x W1^T + b1)
  size_t tb_b1;     // [kCP]
  size_t tb_w1;     // [kCP][kDP]
  size_t tb_wd_t, tb_wg_t;  // [kCP][kDP]
  size_t tb_wd, tb_wg;      // [kDP][kCP]
  MlpW e, n;
};
struct ReadoutW {
  size_t w1d_t, w1g_t, w2d_t, w2g_t;  // [kDP][kDP] k-major
  size_t w1d, w1g, w2d, w2g;          // [kDP][kDP] out-major
  size_t b1d, b1g, b2d, b2g;          // [kDP]
  size_t w3d, w3g;                    // [kDP]
  size_t b3;                          // [2]  (dense, gate)
};
struct WeightLayout {
  size_t emb;      // [num_types][kDP]
  size_t adj_t;    // [kRP][kDP]
  size_t adj;      // [kDP][kRP]
  size_t elemental;  // [num_types]
  BlockW blk[kMaxBlocks];
  ReadoutW ro;
  size_t total;
};

// ---- MFMA edge-block weight images (one per block), copied verbatim into LDS by the kernels -----------
// v_mfma_f32_16x16x4_f32: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], D col = lane&15,
// row = 4*(lane>>4) + reg.  A tile is 16 edges (lane & 15); lane quarter qd = lane>>4 holds, in register `reg`
// of 16-feature block `blk`, feature blk*16 + 4*qd + reg.
// "chain" image of a layer with OB 16-row output blocks and KB 16-wide k blocks (accumulator feeds next layer):
//   img[((ob*KB + kb)*4 + reg)*64 + lane] = W[ob*16 + (lane&15)][kb*16 + 4*(lane>>4) + reg]
// "direct" image (k straight from memory, S steps of 4):
//   img[(ob*S + s)*64 + lane] = W[ob*16 + (lane&15)][4*s + (lane>>4)]
constexpr int kTbSteps = 4;             // three-body MLP: k = l_max*n_max padded to 16 -> 4 k-steps of 4
struct MfmaMlpFwd {                     // offsets in floats inside the forward image
  int w1c;   // chain [8][4]   rows: dense 0-63 | gate 64-127, k: edge features      (8192 floats)
  int w2d;   // chain [4][4]                                                         (4096)
  int w2g;   // chain [4][4]
  int b2;    // [2 (dense,gate)][4 ob][64]  bias as a k-step: lanes < 16 hold b[ob*16+lane], others 0
  int wl;    // direct [4 ob][1 step][64]   W_l [64][R<=4]
};
struct MfmaFwdLayout {
  int tb;    // direct [8 ob][kTbSteps][64]: dense ob0-3, gate ob4-7
  MfmaMlpFwd mlp[2];
  int adj;   // direct [4 ob][1][64]: edge embedding W_adj [64][R<=4] (block 0 forms e0 = SiLU(W_adj h) itself)
  int total;
};
// reverse images: one per conv MLP (two kernels per block: node MLP first, then edge MLP + three-body update)
struct MfmaMlpRev {
  int w1c;           // forward layer-1 image: nothing is saved by the forward pass, both layers are recomputed
  int w2d, w2g, b2;  // forward layer-2 images
  int w2dT;  // chain [4][4]   rows: hidden k, cols: out o
  int w2gT;
  int w1cT;  // chain [4][8]   rows: edge feature k, cols: layer-1 outputs (dense 0-63 | gate 64-127)
  int wl;    // [64][4] plain
  int total; // floats of the MLP part (the node-MLP kernel's whole image)
};
struct MfmaRevLayout {
  MfmaMlpRev mlp;   // offsets inside either image
  int tb;           // edge-MLP image only: forward three-body image
  int tbT;          // edge-MLP image only: chain [1][8] rows: c
  int total_n;      // node-MLP image size
  int total_e;      // edge-MLP image size (MLP part + tb + tbT)
  int per_block;    // total_e + total_n; block b: [edge image | node image]
};
// fused reverse kernel (one per block): dual-use images (m3g_dual_image.h) serve the recompute and the transposed
// products from one LDS copy, so both MLPs fit together
struct MfmaMlpFused {
  int w1c;        // dual image, 128 rows (dense 0-63 | gate 64-127 layer-1 outputs) x 64 edge features
  int w2d, w2g;   // dual images, 64 x 64
  int b2;         // bias images (as the forward kernel's)
  int wl;         // [64][4] plain
  int wld;        // direct [4 ob][64] image of the same W_l (A operand of the W_l h product)
};
struct MfmaRevFusedLayout {
  int tb;    // direct three-body image (forward recompute)
  int tbT;   // chain [1][8] rows: c
  MfmaMlpFused mlp[2];   // 0: edge update, 1: node message
  int adj;   // direct [4 ob][1][64] edge embedding W_adj (block 0 only: e0 and its reverse are formed in the kernel)
  int adjp;  // [64][4] plain copy of W_adj for the dL/dh accumulation
  int total;
};
// fused fp32 reverse kernel (m3g_edge_rev_f32.hip): it starts from the saved layer-1 pre-activations, so it needs W2 in both
// orientations (dual-use fp32 images, m3g_dual_f32.h) but W1c only transposed
struct MfmaMlpRevF32 {
  int w2d, w2g;   // dual fp32 images [64][64]
  int w1cT;       // f32 chain image [4 ob][32 k-steps]: rows = edge feature, k = layer-1 outputs (dense 0-63 | gate 64-127)
  int b2;         // bias images (as the forward kernel's)
  int wl;         // [64][4] plain
  int wld;        // direct [4 ob][64] image of W_l
};
struct MfmaRevF32Layout {
  int tb;    // direct three-body image (forward recompute)
  int tbT;   // f32 chain image [1][32 k-steps] rows: c
  MfmaMlpRevF32 mlp[2];   // 0: edge update, 1: node message
  int adj;   // direct edge-embedding image (block 0)
  int adjp;  // [64][4] plain copy
  int total;
};
MfmaRevF32Layout mfma_rev_f32_layout();
// k_node_pre_mfma image: the 528 x 64 matrix [W1a (TA columns 0-255) | W1b (TB columns) | W_sigmoid1 (v, 16 rows)] as three
// bf16x3 chain images of 11 row blocks (hi + lo parts, the size of an fp32 image), followed by the 528 row biases (b1 for
// TA, 0 for TB, b_sigmoid1)
constexpr int kNodeRowBlocks = 33;
constexpr int kNodeImgFloats = kNodeRowBlocks * 16 * 64 + kNodeRowBlocks * 16;
// k_readout_mfma image (floats): exact-fp32 chain images (f32_chain_image) of the readout GatedMLP (nn/readout.py:39-58)
// and its transposes, then the small vectors
struct ReadoutImg {
  static constexpr int w1 = 0;                 // [8 ob][16 k-steps]  rows: dense 0-63 | gate 64-127 first-layer outputs, k: x
  static constexpr int w2d = w1 + 8 * 2 * 512;   // chain [4][2]
  static constexpr int w2g = w2d + 4 * 2 * 512;
  static constexpr int w2dT = w2g + 4 * 2 * 512; // chain [4][2]  rows: hidden input k, cols: second-layer output
  static constexpr int w2gT = w2dT + 4 * 2 * 512;
  static constexpr int w1T = w2gT + 4 * 2 * 512; // chain [4][4]  rows: x feature, k: dense 0-63 | gate 64-127
  static constexpr int b1 = w1T + 4 * 4 * 512;   // [128]
  static constexpr int b2 = b1 + 128;            // [128] dense | gate
  static constexpr int w3 = b2 + 128;            // [128] dense | gate final weights
  static constexpr int b3 = w3 + 128;            // [2] (+2 pad)
  static constexpr int total = b3 + 4;
};
MfmaFwdLayout mfma_fwd_layout();
MfmaRevLayout mfma_rev_layout();
MfmaRevFusedLayout mfma_rev_fused_layout();

}  // namespace m3g

struct m3g_plan {
  m3g_config cfg;
  m3g::Consts consts;
  m3g::WeightLayout wl;
  std::map<std::string, std::vector<float>> params;  // raw state_dict tensors (host)
  std::map<std::string, std::vector<float>> cvals;   // raw constants (host)
  float* d_weights = nullptr;
  // MFMA weight images, one set per precision mode (same layouts and sizes: an fp32 image is as large as a bf16 hi + lo pair)
  float* d_mfma_fwd[m3g::kNumPrec] = {nullptr, nullptr, nullptr};   // [num_blocks][MfmaFwdLayout.total]
  float* d_mfma_rev[m3g::kNumPrec] = {nullptr, nullptr, nullptr};   // [num_blocks][MfmaRevLayout.total]
  float* d_mfma_revf = nullptr;  // [num_blocks][MfmaRevFusedLayout.total] (bf16x3 dual-use images)
  float* d_mfma_revf_h = nullptr;  // the same layout holding fp16 parts of the scaled weights (f16x3 mode)
  float* d_mfma_revf32 = nullptr;  // [num_blocks][MfmaRevF32Layout.total] (fused fp32 reverse kernel)
  float* d_node_img[m3g::kNumPrec] = {nullptr, nullptr, nullptr};   // [num_blocks][kNodeImgFloats]: node-table weights as MFMA A-operand images (k_node_pre_mfma)
  int precision = m3g::kPrecF32;   // option "precision" (default: exact fp32 MFMA products = the reference's arithmetic; 1 / 2 = the split modes, opt-in)
  float w_scale_inv = 1.f;       // f16x3 mode: 1 / (the power of two all chain-image weights were multiplied by), set by pack_mfma_images
  int save_p1 = 1;               // option "save_p1" (fp32 mode only): 0 = recompute layer 1 in the reverse kernels (A/B tests)
  int save_p2 = 1;               // option "save_p2" (fp32 mode, fused reverse): 0 = recompute layer 2 in the reverse kernel
  int device = -1;               // HIP device the plan's buffers live on (set by m3g_plan_commit)
  // generic path (m3g_generic.hip): raw state_dict tensors, unpadded, in one device blob; offsets by key
  bool generic = false;          // sizes beyond the MFMA kernels' tiles (or option "edge_kernel" = 2)
  float* d_generic = nullptr;
  std::map<std::string, size_t> generic_off;
  float* d_readout_img = nullptr;   // [ReadoutImg::total]: readout MLP weights as exact-fp32 chain images (k_readout_mfma; fp32 and bf16x3 modes)
  float* d_readout_img_h = nullptr; // the same layout as scaled two-part fp16 chain images (f16x3 mode), weights scaled by 1 / ro_w_scale_inv
  float ro_w_scale_inv = 1.f;
  int small_tiles_fwd = 3072;    // option "small_tiles_fwd": the same threshold for the forward kernel alone (measured: a gain up to ~900 atoms, equal at 1,372)
  int small_tiles = 1536;        // option "small_tiles": graphs of at most this many 16-edge tiles run the split-tile edge kernels
                                 // (m3g_edge_small.hip: a tile over the four SIMDs of a CU, operands in registers); 0 = never
  int dp1_by_dst = 0;            // option "dp1_by_dst": see dp1_rows_by_dst().  Measured on the 10k-atom cell: node reverse 205 -> 207 us, reverse edge kernels
                                 // 1.012 -> 1.017 ms per step -- the gather of whole 1-KB rows is not what bounds the node reverse; off by default
  int split_node_tiles = 128;    // option "split_node_tiles": 16-atom tiles (2,048 atoms) up to which the node tables and the readout take their split forms
                                 // (m3g_node_mfma.hip; measured at 625 tiles: node tables 58 -> 85 us, readout 28 -> 36 us per step -- not beyond)
  int fuse_node_tb = 1;          // option "fuse_node_tb": three-body reverse (moment path) + node reverse of a block as two workgroup roles of
                                 // one launch (k_node_tb_reverse, m3g_threebody.hip)
  int debug_node_tb_polls = 0;   // option "debug_node_tb_polls" (tests): see launch_node_tb_reverse
  int split_tail = 1;            // option "split_tail": see k_edge_rev_f32 (the tiles of a workgroup's last, part-filled round through the four-way split)
  int small_launches = 1;        // option "small_launches": small systems take fused launches (force tail, readout + energy sums, ...)
  int rev_kernel = 1;            // MFMA path: 1 = fused reverse kernel per block, 0 = node-MLP + edge-MLP kernel pair
  bool readout_f16 = false; // option "readout_f16": the readout layers on scaled two-part fp16 chains in the f16x3 mode (5 us faster at 10,000
                            // atoms); default: exact-fp32 chains in every mode -- the per-atom energy can be the ill-conditioned remainder of its
                            // last layer's terms, where 22 against 24 bits per product show (fuzz case 84)
  bool legendre_ref = false;   // option "legendre_backward" = 1: the reference's own (inexact) backward of P_l, list kernels only
  bool tb_moments = true;   // option "threebody_moments": per-atom moment sums where the partner lists are complete (m3g_threebody.hip)
  int stress_mode = 0;   // 0: reference formula sum pos (x) F / V; 1: pair virial (PBC consistent)
  int edge_kernel = 1;           // 0 = VALU baseline (m3g_edge_simple.hip), 1 = MFMA (m3g_edge_mfma.hip)
  int stamp_target = 0;          // which kernel runs its stamped variant: 0 forward edge block, 1 reverse edge-MLP kernel, 2 fused reverse (f16x3)
  unsigned long long* d_stamps = nullptr;  // option "stamps": diagnostic phase-cycle sums [256][16][12] of the fwd edge kernel
  bool committed = false;
  // opt-in stage profiler (m3g_profile_*): event pairs recorded around stage launches
  // side stream: the three-body reverse of a block (short, latency-bound, does not fill the chip) runs beside the node
  // reverse's dp1 gather (HBM-bound); fork/join with events, created on first use
  bool debug_force_move = false;        // option debug_force_move (tests): the next commit runs the device-move path
  mutable hipStream_t side_stream = nullptr;
  mutable hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int overlap = 0;               // option "overlap": 1 = use the side stream (measured 2 % SLOWER on the 10k-atom step: two fork/join
                                 // pairs of cross-stream event waits cost more than the ~40 us of kernel time they hide), default off
  // hipGraph replay (option "graph_replay"): the launch sequence of one m3g_energy_forces call is captured once per
  // distinct (io, workspace, stream, options) and replayed while those stay identical -- for small systems the ~36
  // launches of a step are launch-bound.  The caller must then keep every buffer of the call alive and at the same address.
  int graph_replay = 0;
  struct GraphEntry { std::vector<unsigned char> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
  mutable std::vector<GraphEntry> graphs;
  mutable bool capturing = false;
  mutable bool profile = false;
  mutable std::vector<hipEvent_t> ev_pool;
  mutable std::vector<int> ev_stage;   // stage id of pair k (events 2k, 2k+1)
  mutable size_t ev_used = 0;          // pairs used since the last read
};

namespace m3g {

// ---- topology view (device arrays carved from the caller's topo buffer) -----------------------------
// compile-time dispatch on (l_max, n_max) <= (4, 4): BODY sees constexpr int L, R
#define M3G_DISPATCH_LR(L_, R_, BODY)                         \
  switch ((L_) * 8 + (R_)) {                                  \
    case 1 * 8 + 1: { constexpr int L = 1, R = 1; BODY; } break; \
    case 1 * 8 + 2: { constexpr int L = 1, R = 2; BODY; } break; \
    case 1 * 8 + 3: { constexpr int L = 1, R = 3; BODY; } break; \
    case 1 * 8 + 4: { constexpr int L = 1, R = 4; BODY; } break; \
    case 2 * 8 + 1: { constexpr int L = 2, R = 1; BODY; } break; \
    case 2 * 8 + 2: { constexpr int L = 2, R = 2; BODY; } break; \
    case 2 * 8 + 3: { constexpr int L = 2, R = 3; BODY; } break; \
    case 2 * 8 + 4: { constexpr int L = 2, R = 4; BODY; } break; \
    case 3 * 8 + 1: { constexpr int L = 3, R = 1; BODY; } break; \
    case 3 * 8 + 2: { constexpr int L = 3, R = 2; BODY; } break; \
    case 3 * 8 + 3: { constexpr int L = 3, R = 3; BODY; } break; \
    case 3 * 8 + 4: { constexpr int L = 3, R = 4; BODY; } break; \
    case 4 * 8 + 1: { constexpr int L = 4, R = 1; BODY; } break; \
    case 4 * 8 + 2: { constexpr int L = 4, R = 2; BODY; } break; \
    case 4 * 8 + 3: { constexpr int L = 4, R = 3; BODY; } break; \
    case 4 * 8 + 4: { constexpr int L = 4, R = 4; BODY; } break; \
    default: break;                                           \
  }

#ifndef M3G_TB_ROWS
#define M3G_TB_ROWS 128
#endif
constexpr int kTbRows = M3G_TB_ROWS;
#ifndef M3G_TB_CAP
#define M3G_TB_CAP 255
#endif
constexpr int kTbCap = M3G_TB_CAP;   // staged three-body window: the rows of the workgroup + the other rows of the centres they belong to;
                                     // < 256 so a window-relative partner id fits a byte (rows beyond it are read from global memory)
// staged partner ids per row and list (one byte each; longer lists continue from global memory).  Two instantiations of the
// three-body kernels: the short lists of the usual 3-body cutoff (r_3 = 4 A: ~17 partners per active edge) and the long ones of
// dense neighbourhoods (BASELINE config 5, r_3 = 6 A: ~58 per edge) -- sweep in profiles/r02_config5_sweep.txt: staging 96
// ids per row takes the dense case from 61 to 24 us (forward) and 137 to 86 us (reverse) per launch, but costs the 10k-atom
// Cu cell 25 % of its three-body reverse (LDS footprint -> fewer resident workgroups), hence the choice per launch.
constexpr int kTbListShort = 32, kTbListLong = 96;
constexpr int kTopoFlags = 16;     // Topo::flags words
constexpr int kTbFastAtoms = 64;   // most centre atoms per workgroup window the moment path keeps sums for (more: the list path)
constexpr int kTbCapShort = kTbRows + 64 < kTbCap ? kTbRows + 64 : kTbCap;   // window rows the short-list instantiation stages
   // active edge rows per three-body workgroup (m3g_threebody.hip; windows precomputed in the topology)
struct Topo {
  int64_t N, E, T, S;
  int32_t* src;      // [E] centre of each edge
  int32_t* dst;      // [E] neighbour of each edge
  int32_t* row_ptr;  // [N+1] edges of centre i: row_ptr[i] .. row_ptr[i+1]
  int32_t* in_ptr;   // [N+1] incoming edges of atom j (dst == j)
  int32_t* in_edge;  // [E]
  int32_t* in_pos;   // [E] position of edge e in the by-neighbour list: in_edge[in_pos[e]] == e (dp1 rows are stored by that position in the
                     // exact-fp32 fused path, so the node reverse STREAMS the rows of an atom instead of gathering them)
  int32_t* in_pair;  // [E][2] (in_edge[k], act_id[in_edge[k]]): the k-th incoming edge and its compact three-body row (-1: none),
                     // one 8-byte load in the node reverse gather
  int32_t* t1_ptr;   // [E+1] triplets grouped by first edge e1
  int32_t* t1_e2;    // [T]
  int32_t* t2_ptr;   // [E+1] triplets grouped by second edge e2
  int32_t* t2_e1;    // [T]
  // "active" edges = edges that take part in at least one triplet (d <= three-body cutoff and a partner exists); the
  // three-body kernels run over this compacted list so no lane idles on the edges beyond the three-body cutoff
  int32_t* act_list;   // [A] active edge ids, ascending
  int32_t* act_scan;   // [E+1] number of active edges before e (compacted id of e when e is active)
  int32_t* act_id;     // [E] compacted id of edge e, -1 for an edge without triplets: the per-edge three-body arrays
                       // (q, q', m, dm, dg) hold rows for ACTIVE edges only, indexed by this id
  int32_t* arow_ptr;   // [N+1] compacted rows of centre i: arow_ptr[i] .. arow_ptr[i+1]
  int32_t* t1_e2c;     // [T] t1_e2 in compacted ids
  int32_t* t2_e1c;     // [T] t2_e1 in compacted ids
  uint8_t* t1_b;       // [T] the same partners as byte-sized ids relative to the LDS window of the row's workgroup
  uint8_t* t2_b;       //     (255: outside the staged window -> the kernels fall back to the int32 list)
  int32_t* act_dst;    // [A] neighbour atom of each active edge (saves a dependent load when staging)
  int32_t* tb_win;     // [6 * blocks] per three-body workgroup: compacted-row window [lo, hi) staged in LDS, then the
                       // ranges [t_lo, t_hi) of its rows' partner lists in t1_e2c and in t2_e1c
  int32_t* tb_fast;    // [2 * blocks] per three-body workgroup: {number of centre atoms spanned by its window, first of them} when every
                       // one of those atoms has COMPLETE partner lists (each active edge paired with every other active edge of its
                       // centre exactly once, in both roles) and there are at most kTbFastAtoms of them -- the workgroup may then use
                       // per-atom moment sums instead of walking the lists (m3g_threebody.hip); {0, x} otherwise
  int32_t* n_act;      // device scalar A (= flags + 2)
  int32_t* batch;    // [N]
  int32_t* struct_ptr;  // [S+1] atoms of structure s: struct_ptr[s] .. struct_ptr[s+1] (valid when `batch` is non-decreasing, flags[3] == 0)
  int32_t* flags;    // [kTopoFlags] [0] malformed-graph bits; [2] = A; [3] != 0: `batch` is not sorted (per-structure sums then use atomics);
                     // [4] three-body workgroups that may NOT use the moment path, [5] largest window (rows), [6] most atoms per window
                     // (m3g_topology_hints packs 4..6 for the caller); [7] the hints word m3g_topology_hints certified for this
                     // buffer (0 after a build); [8] sticky error bits set by the hot call (M3G_TOPO_ERR_*, m3g_topology_status)
  void* sort_tmp;    // scratch for the radix sorts
  size_t sort_tmp_bytes;
  size_t total_bytes;
};
Topo topo_carve(int64_t N, int64_t E, int64_t T, int64_t S, void* base);
size_t topo_sort_tmp_bytes(int64_t E, int64_t T);

// ---- workspace view ---------------------------------------------------------------------------------
// 24-bit dp1 rows (768 B per edge) leave the last quarter of the [E,4*kDP] fp32 buffer free: the f16x3 mode's per-row inverse
// scales ([E][4] floats) live at its start
inline float* dp1_scale_of(float* dp1, int64_t E) { return dp1 ? dp1 + (size_t)E * 192 : nullptr; }
// dp1 hand-over formats of k_node_reverse: fp32 rows, 24-bit floating rows (bf16x3 fused kernel), 24-bit fixed-point rows + scales
// (f16x3 fused kernel)
enum { kDp1F32 = 0, kDp1Packed = 1, kDp1Fixed = 2, kDp1F32ByDst = 3 /* fp32 rows stored by position in the by-neighbour list */ };

struct Work {
  // per-edge geometry / bases
  float *u, *d, *h, *hp, *q, *qp, *fc3, *fc3p;  // [E,3] [E] [E,kRP] [E,kRP] [E,kCP] [E,kCP] [E] [E]
  float* x[kMaxBlocks + 1];   // [N,kDP] node features before block b (x[B] = final)
  float* v[kMaxBlocks];       // [N,kCP]
  float *TA, *TB;             // [N,4*kDP]
  float* e;                   // [E,kDP] edge features (updated in place)
  float* m[kMaxBlocks];       // [E,kCP]
  float* act[kMaxBlocks];     // [E,8*kDP] saved pre-activations: e:{p1[2DP],p2d,p2g} n:{p1[2DP],p2d,p2g}
  // reverse pass
  float *dx, *dx2;            // [N,kDP]
  float* de;                  // [E,kDP]
  float* dm;                  // [E,kCP]
  float* g;                   // [E,kCP] q * v[dst] of the current block
  float* dg;                  // [E,kCP]
  float* dh;                  // [E,kRP]
  float* dd;                  // [E]
  float* du;                  // [E,3]
  float* dp1;                 // [E,4*kDP]  (packed 24-bit rows use the first 3/4 of it, dp1_scale_of() the start of the rest)
  float* dr;                  // [E,3]
  // MFMA path: tile-SoA images ([tile of 16 edges][4 blk][64 lanes][4]) of the edge features BEFORE each block
  // (e_blk[b]; e_blk[B] = final) and of dL/de, per-block node tables, row-major messages.  No activations saved.
  float* e_blk[kMaxBlocks + 1];
  float* p2_blk[kMaxBlocks];  // fp32 mode, saves_p2(): layer-2 pre-activations, same shape; p1_blk then holds SiLU'(p1)
  float* p1_blk[kMaxBlocks];  // fp32 mode: layer-1 pre-activations of both conv MLPs saved by the forward kernel,
                              // [tiles][2 mlp][8 blk][64 lanes][4] (the reverse kernels start from them: saves_p1())
  float* TAb[kMaxBlocks];     // [N,4*kDP] per block (the reverse pass recomputes layer 1 from them)
  float* TBb[kMaxBlocks];
  float* de_soa;
  float* dcn;                 // tile-SoA: node-MLP kernel's contribution to dL/de2 (store-only there, loaded by the edge-MLP kernel)
  float* dh_parts;            // [2B+1][E,kRP]: every reverse kernel stores its dL/dh share in its own slice (no read-modify-write)
  // per-centre sums formed inside the MFMA edge kernels (rows of a centre are consecutive edges): seg_head[t] = sum of the
  // tile's first run (the centre owning column 0), seg_first[i] = sum of the run in which centre i's row starts mid-tile
  float* seg_head;            // [tiles][4*kDP]
  float* seg_first;           // [N][4*kDP]
  int32_t* sync;              // [kSyncWords] "last workgroup" counters of the step's fused launches: cleared by k_geometry, the first
                              // kernel of every step (nullptr when carved without a base)
  size_t total_bytes;
};
constexpr int kSyncWords = 16;
constexpr int kSyncForceTail = 0, kSyncReadout = 1, kSyncNodeRev = 2;   // (+ block index for the per-block ones)
// fused launches: the per-structure sums (energies after the readout, virial after the force gather) are formed by the LAST
// workgroup of the producing launch when the batch has at most this many structures (it walks them one after the other)
constexpr int64_t kForceTailMaxStructs = 8;
// ... and at most this many atoms: the last workgroup's 256 threads then read <= 4 atoms each (measured on the 10,000-atom cell: 40
// dependent reads per thread of values other XCDs have just written cost 24 us after the readout and 65 us after the force gather,
// against 6 and 10 us for the stand-alone sum kernels)
constexpr int64_t kFusedSumsMaxAtoms = 1024;   // (2,000 atoms: readout + sums 23 us against ~20 separately, gather + virial 19 against 17: no gain any more)
constexpr int64_t kNodeTbFusedMaxAtoms = 128;   // k_node_tb_reverse (two roles in one launch): see launch_node_tb_reverse
Work work_carve(const Consts& c, bool mfma, int save_acts /* 0 none, 1 p1, 2 p1 + p2 */, int64_t N, int64_t E, int64_t T, int64_t S, void* base);

// ---- kernel launchers (each in its own .hip) -----------------------------------------------------------
// geometry.hip
void launch_geometry(const Consts& c, const Topo& t, const float* pos, const float* lattice, const int32_t* shift,
                     const Work& w, hipStream_t s);
bool launch_geometry_reverse(const Consts& c, const Topo& t, const Work& w, const float* dh, int dh_parts, float* forces,
                             float* stresses, hipStream_t s, bool fuse_stress = false, const float* pos = nullptr, const float* lattice = nullptr,
                             bool dr_done = false);
void launch_stress(const Consts& c, const Topo& t, const float* pos, const float* lattice, const float* forces,
                   float* stresses, hipStream_t s, const float* ea = nullptr, float* scaled_total = nullptr, float* total = nullptr);
void launch_stress_pair(const Topo& t, const Work& w, const float* lattice, float* stresses, hipStream_t s);
void launch_struct_energy(const Consts& c, const Topo& t, const float* ea, float* scaled_total, float* total, hipStream_t s);
void launch_force_gather(float length_scale, const Topo& t, const float* dr, float* forces, float* stresses, hipStream_t s);
// generic.hip: any-size path (embedding_dim, l_max, n_max beyond the MFMA kernels' tiles)
size_t generic_workspace_bytes(const m3g_plan* plan, int64_t N, int64_t E, int64_t T, int64_t S);
int generic_commit(m3g_plan* plan);
void generic_free(m3g_plan* plan);
int generic_energy_forces(const m3g_plan* plan, const m3g_io* io, void* workspace, size_t workspace_bytes, hipStream_t s);
void launch_triplet_angles(const Topo& t, const int64_t* tei, const float* u, float* out, hipStream_t s);
void launch_distance_only(float length_scale, const Topo& t, const float* pos, const float* lattice,
                          const int32_t* shift, float* u, float* d, hipStream_t s);
void launch_edge_featurizer(const Consts& c, int64_t E, const float* d, float* out, int out_stride, hipStream_t s);
// node.hip
void launch_embed(const Consts& c, const float* W, const WeightLayout& wl, const Topo& t, const int64_t* types,
                  const Work& w, hipStream_t s);
void launch_embed_reverse(const Consts& c, const float* W, const WeightLayout& wl, const Topo& t, const Work& w,
                          hipStream_t s);
void launch_node_pre(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w, const float* x_prev, float* x,
                     float* v, float* TA, float* TB, hipStream_t s);
void launch_node_pre_mfma(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* x_prev, float* x,
                          float* v, float* TA, float* TB, const int64_t* types, const float* emb, hipStream_t s);
bool launch_geometry_node_pre(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, const float* pos, const float* lattice,
                              const int32_t* shift, const int64_t* types, const float* emb, hipStream_t s);
void launch_energy_sums(const Consts& c, const Topo& t, const float* scaled_atomic, float* scaled_total, float* total, hipStream_t s);
// energy_sums_deferred (in/out): in true = the caller can form the per-structure energy sums later (k_struct_stress); out true =
// they are still to be formed (neither this launch's last workgroup nor k_struct_energy did)
void launch_readout_mfma(const m3g_plan* plan, const Consts& c, const WeightLayout& wl, const Topo& t, const int64_t* types,
                         const float* x_prev, float* x, const Work& w, float* scaled_atomic, float* scaled_total, float* total,
                         bool want_grad, hipStream_t s, bool* energy_sums_deferred = nullptr);
void launch_node_reverse(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w,
                         const float* v, const float* dx_new, float* dx_out, bool row_sums_in_seg, int dp1_packed, bool with_v_term,
                         hipStream_t s, bool small = false);
void launch_node_reverse_v_term(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w, const float* v,
                                float* dx_out, hipStream_t s);
void launch_readout(const Consts& c, const float* W, const WeightLayout& wl, const Topo& t, const int64_t* types,
                    const float* x_prev, float* x, const Work& w, float* scaled_atomic, float* scaled_total, float* total,
                    bool want_grad, hipStream_t s);
void launch_gather_rows(const float* table, int64_t n, int width, int table_stride, int table_rows, bool transposed,
                        const int64_t* idx, float* out, hipStream_t s);
void launch_copy_expand_rows(const int32_t* row_id, const float* in, int in_stride, float* out, int out_stride, int width, int64_t n,
                             hipStream_t s);
void launch_copy_strided(const float* in, int in_stride, float* out, int out_stride, int width, int64_t rows,
                         hipStream_t s);
// threebody.hip
void launch_threebody(const Consts& c, const Topo& t, const Work& w, const float* v, float* m, hipStream_t s, int topo_hints = 0);
void launch_threebody_reverse(const Consts& c, const Topo& t, const Work& w, const float* v, bool first, hipStream_t s, int topo_hints = 0,
                              bool ref_legendre = false);
bool launch_threebody_reverse_final(const Consts& c, const Topo& t, const Work& w, const float* v, bool first, const float* dh, int dh_parts,
                                    hipStream_t s, int topo_hints);
bool launch_node_tb_reverse(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w, const float* v, bool first,
                            const float* dx_new, float* dx_out, int dp1_packed, int block, hipStream_t s, int topo_hints, int debug_polls = 0);
// pack_mfma.hip / edge_mfma.hip
int pack_mfma_images(m3g_plan* plan);
void free_mfma_images(m3g_plan* plan);
// one fused reverse kernel per block: k_edge_rev_fused (bf16x3, dual-use bf16 images) or k_edge_rev_f32 (fp32, needs the saved
// layer-1 pre-activations); otherwise the node-MLP + edge-MLP kernel pair
// fp32 mode is bound by the matrix pipe: its forward kernel saves the layer-1 pre-activations of both MLPs (1 KB per edge and
// block) and the reverse kernels start from them instead of recomputing that layer (a quarter of their MFMAs)
inline bool saves_p1(const m3g_plan* plan) { return plan->edge_kernel == 1 && plan->precision == kPrecF32 && plan->save_p1 != 0; }
// ... and, with the fused fp32 reverse kernel, the layer-2 pre-activations as well (another 1 KB per edge and block): the
// reverse kernel then issues no recompute MFMA at all (576 instead of 832 per tile)
inline bool saves_p2(const m3g_plan* plan) { return saves_p1(plan) && plan->rev_kernel == 1 && plan->save_p2 != 0; }
inline int saved_activations(const m3g_plan* plan) { return saves_p2(plan) ? 2 : saves_p1(plan) ? 1 : 0; }
// exact-fp32 fused reverse kernels: dp1 rows stored by the edge's position in the by-neighbour list (option "dp1_by_dst")
inline bool dp1_rows_by_dst(const m3g_plan* plan) {
  return plan->dp1_by_dst && plan->edge_kernel == 1 && plan->rev_kernel == 1 && plan->precision == kPrecF32 && saves_p1(plan);
}
inline bool fused_reverse(const m3g_plan* plan) {
  return plan->edge_kernel == 1 && plan->rev_kernel == 1 && (plan->precision != kPrecF32 || saves_p1(plan));
}
void launch_edge_block_mfma(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, bool for_reverse, hipStream_t s);
void launch_edge_rev_node_mlp(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* dx_new,
                              hipStream_t s);
void launch_edge_rev_edge_mlp(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* dx_new,
                              bool de_is_zero, hipStream_t s);
void launch_edge_rev_fused(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* dx_new,
                           bool de_is_zero, hipStream_t s);
void launch_edge_rev_f32(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* dx_new,
                         bool de_is_zero, hipStream_t s);
// edge_small.hip: the exact-fp32 edge kernels with one tile split over the four waves of a workgroup (small systems); false: the
// configuration is not one they cover (another precision / A-B option / stamps) -- the caller then launches the persistent kernel
bool launch_edge_fwd_split(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, bool for_reverse, hipStream_t s);
bool launch_edge_rev_split(const m3g_plan* plan, const Consts& c, const Topo& t, const Work& w, int b, const float* dx_new, bool de_is_zero,
                           hipStream_t s);
void launch_embed_edges_soa(const Consts& c, const float* adj_t, const float* h, float* soa, int64_t E, hipStream_t s);
void launch_embed_edges_reverse_soa(const float* adj, const float* h, const float* de_soa, float* dh_slice, int64_t E, hipStream_t s);
void launch_embed_nodes_only(const Consts& c, const float* W, const WeightLayout& wl, const Topo& t, const int64_t* types,
                             const Work& w, hipStream_t s);
void launch_rows_to_soa(const float* rows, float* soa, int64_t E, hipStream_t s);
void launch_soa_to_rows(const float* soa, float* rows, int row_stride, int width, int64_t E, hipStream_t s);
// edge_simple.hip
void launch_edge_block(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w, int b,
                       float* x_new, hipStream_t s);
void launch_edge_block_reverse(const Consts& c, const float* W, const BlockW& bw, const Topo& t, const Work& w, int b,
                               const float* dx_new, hipStream_t s);

}  // namespace m3g
