/*
 * m3gnet_hip.h -- C ABI of libm3gnet_hip.so: MI355X (gfx950) M3GNet energy/force engine.
 *
 * The reference (lan496/torch-m3gnet) has no FFI: its operator boundary is the Python module
 * protocol `forward(graph) -> graph` over keyed tensors.  This header is the
 * plain-C boundary that protocol binds to in the MI355X build: every entry point names the
 * reference interface it replaces (paths relative to /root/reference/src/torch_m3gnet).
 *
 * Conventions
 *   - all `const float*` / `const int64_t*` / `void*` data arguments are DEVICE pointers unless
 *     the parameter name starts with `host_`;
 *   - buffers are caller-owned; the library never allocates inside a hot call -- scratch comes
 *     from the caller (sizes from m3g_workspace_bytes / m3g_topology_bytes);
 *   - every call takes the HIP stream to enqueue on (`hipStream_t` passed as void*); calls are
 *     asynchronous with respect to the host unless stated otherwise;
 *   - return value: 0 = M3G_OK, otherwise an m3g_status; m3g_last_error() gives a message
 *     (thread-local).  The Python host turns M3G_ERR_VALUE into ValueError (the reference raises
 *     ValueError for too-large l_max/n_max, nn/interaction.py:250-253) and the rest into RuntimeError;
 *   - thread-compatible: no global mutable state; one plan may be used from one thread at a time.
 *
 * Tensor layouts are those of the reference's MaterialGraph (data/material_graph.py:14-107):
 *   pos [N,3] f32, atom_types [N] i64 (Z-1), edge_index [2,E] i64 (row 0 centre i, row 1
 *   neighbour j, SORTED BY CENTRE), edge_cell_shift [E,3] i32, triplet_edge_index [2,T] i64
 *   (row 0 = edge ij, row 1 = edge ik, any order), lattice [S,3,3] f32 row-wise, batch [N] i64.
 */
#ifndef M3GNET_HIP_H
#define M3GNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  M3G_OK = 0,
  M3G_ERR_VALUE = 1,    /* bad hyper-parameter / malformed graph (Python: ValueError) */
  M3G_ERR_STATE = 2,    /* call order violated (e.g. parameters not committed) */
  M3G_ERR_SIZE = 3,     /* caller buffer too small */
  M3G_ERR_HIP = 4,      /* a HIP runtime call failed */
  M3G_ERR_UNSUPPORTED = 5
} m3g_status;

/* Hyper-parameters == arguments of build_model (model/build.py:16-28). */
typedef struct {
  double cutoff;            /* Angstrom, unscaled (Python float == double, as the reference holds it) */
  double threebody_cutoff;  /* Angstrom, unscaled */
  double energy_scale;
  double length_scale;
  int32_t l_max;
  int32_t n_max;
  int32_t num_types;
  int32_t embedding_dim;
  int32_t num_blocks;
  int32_t reserved;
} m3g_config;

typedef struct m3g_plan m3g_plan; /* opaque: packed weights + constants + kernel selection */

/* Library / device probe.  Returns M3G_OK and fills the fields when a gfx950 device is usable. */
typedef struct {
  int32_t abi_version;
  int32_t device_count;
  char arch[32]; /* gcnArchName of the current device, "" when none */
} m3g_info;
int m3g_get_info(m3g_info* out);
const char* m3g_last_error(void);

/* ---- plan: replaces module construction in build_model (model/build.py:37-81) -------------- */
int m3g_plan_create(const m3g_config* cfg, m3g_plan** out);
void m3g_plan_destroy(m3g_plan* plan);

/* Set one parameter by its reference state_dict key ("model.7.concat_edge_update.dense.0.weight",
 * ...).  HOST pointer, row-major as torch stores it, `numel` floats.  Unknown key
 * or wrong size -> M3G_ERR_VALUE. */
int m3g_plan_set_param(m3g_plan* plan, const char* key, const float* host_data, int64_t numel);

/* Constants the reference keeps as plain module attributes (not in state_dict):
 *   "elemental_energies" [num_types]   AtomRef            (nn/atom_ref.py:17-23)
 *   "em" "dm" "coeff"     [n_max]      EdgeFeaturizer     (nn/featurizer.py:61-79)
 *   "factors"             [l_max,n_max] NormalizedSphericalBessel (nn/interaction.py:256-266)
 *   "bessel_zeros"        [l_max,n_max] rows 0..l_max-1 of SPHERICAL_BESSEL_ZEROS (interaction.py:14-135)
 * HOST pointers. */
int m3g_plan_set_const(m3g_plan* plan, const char* name, const float* host_data, int64_t numel);

/* Engine options (not part of the reference):
 *   "precision"   arithmetic of the dense products of the gated MLPs (nn/core.py:61-62: fp32 Linear layers), fp32 accumulate in all modes:
 *                   0 (default) "fp32": every product on v_mfma_f32_16x16x4_f32 -- exact fp32 products accumulated in k order, bitwise
 *                   an fp32 fmaf chain: the reference's arithmetic;
 *                   2 "f16x3" (opt-in, narrower than fp32): every operand scaled by a power of two (weights: one for the model;
 *                   activations and gradients: one per edge and chain, chosen from the data) and split in two fp16 parts that
 *                   together carry 22-24 significant bits, three v_mfma_f32_16x16x32_f16 products per fp32 product (lo x lo
 *                   dropped) -- errors ~1.8 x those of an fp32 fmaf chain, parity inside north_star's tolerances on every case;
 *                   1 "bf16x3" (opt-in): two bf16 parts (16 significant bits), three v_mfma_f32_16x16x32_bf16 products (relative
 *                   product error ~2^-16; the fastest mode, parity within north_star's tolerances on near-linear weights only);
 *                   all weight image sets are resident after a commit, switching costs nothing;
 *   "edge_kernel" = 1 fused MFMA edge blocks (default), 0 = vector-ALU baseline kernels (also M3G_EDGE_KERNEL in the env),
 *                   2 = the any-size path (run-time-sized fp32 kernels; chosen automatically for embedding_dim > 64,
 *                   l_max or n_max > 4, more than 8 blocks -- up to the reference's own limits l_max <= 9, n_max <= 10);
 *   "save_p1" / "save_p2" (fp32 mode, default 1): the forward kernel saves SiLU'(p1) / the layer-2 pre-activations of both conv
 *                   MLPs for the reverse kernel (no recompute MFMAs) -- 0 for A/B measurements;
 *   "rev_kernel"  = 1 one fused reverse kernel per block (default; fp32 mode: k_edge_rev_f32 on the saved activations, bf16x3
 *                   mode: k_edge_rev_fused recomputing them), 0 = node-MLP + edge-MLP kernel pair (both modes, A/B tests);
 *   "stress_mode" = 0 the reference's sum pos (x) F / V (nn/gradient.py:39-62, default), 1 = pair virial
 *                   -(1/V) sum_e r_e (x) dE/dr_e (docs/gradient.md:47-84), invariant under lattice translations;
 *   "readout_f16" = 1 the readout layers run on scaled two-part fp16 chains in the f16x3 mode (5 us per step faster at 10,000 atoms);
 *                   default 0: exact-fp32 MFMA chains in every mode (energies that are the small remainder of larger terms keep
 *                   fp32's 24 bits per product);
 *   "threebody_moments" = 1 (default) the three-body sums run over per-atom moments when m3g_io.topo_hints says every centre's
 *                   triplet list is complete (m3g_topology_hints), 0 = always walk the lists (A/B tests);
 *   "legendre_backward" = 0 (default) d P_l / d cos(theta) is the true derivative; 1 = what the reference's
 *                   LegendreCosPolynomial.backward returns (nn/interaction.py:373-382 multiplies grad_output in at every level of
 *                   its recurrence: inexact for l >= 2), so forces and stresses reproduce the reference's own numbers rather than
 *                   the gradient of its energy.  Not linear in a triplet's incoming gradient: the list kernels run (as with
 *                   "threebody_moments" = 0).  Energies are unaffected;
 *   "overlap"     = 1 the three-body reverse of a block runs on an internal side stream beside the node reverse's gather
 *                   (fork/join with events on the caller's stream), 0 = everything on the caller's stream (default: the
 *                   cross-stream waits measured slower than the overlap gains on the benchmark workload);
 *   "graph_replay" = 1 m3g_energy_forces captures its launch sequence into a hipGraph the first time it sees a given
 *                   (m3g_io contents, workspace, stream, options) and replays it on later identical calls (launch-bound
 *                   small systems); every buffer of the call must stay alive at the same address.  Default 0;
 *   "stamps"      = 1 / 2 / 3: diagnostic builds with in-kernel cycle stamps (m3g_debug_read_stamps), 0 off;
 *   "debug_force_move" = 1 (tests) the next m3g_plan_commit takes the device-move path although the device is unchanged. */
int m3g_plan_set_option(m3g_plan* plan, const char* name, int32_t value);

/* Pack and upload everything set so far (synchronous; drains the device first).  Must be called before any compute
 * call and again after parameters/constants change.  Missing keys -> M3G_ERR_STATE.  The plan's device buffers live on the
 * HIP device that is current at commit; committing again under another device moves them there, and m3g_energy_forces
 * under a device other than the plan's returns M3G_ERR_STATE. */
int m3g_plan_commit(m3g_plan* plan);

/* ---- topology: replaces nothing in the reference nn (which re-gathers by index on every call);
 * it converts MaterialGraph index tensors into the receiver-sorted CSR form the kernels use.
 * Depends on the index tensors only -- reusable across calls while they are unchanged. ---------- */
int m3g_topology_bytes(int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs, size_t* bytes);
/* The call waits for the stream once (it reads back whether the triplet list is sorted and symmetric, which decides the
 * sorts); on return host_flags[0] != 0 means the graph is malformed (the remaining kernels of the build may still be running):
 *   bit 0: edge_index[0] not sorted; bit 1: index out of range; bit 2: triplet edges with different centres. */
int m3g_topology_build(int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs,
                       const int64_t* edge_index, const int64_t* triplet_edge_index, const int64_t* batch,
                       void* topo, size_t topo_bytes, int32_t* host_flags, void* stream);

/* What the build found out about the graph that lets m3g_energy_forces pick cheaper kernels: an opaque word for m3g_io.topo_hints
 * (0 is always valid).  Today: M3G_TOPO_TB_COMPLETE -- every centre atom's triplet list holds each ordered pair of its active
 * edges exactly once (what compute_threebody emits, data/material_graph.py:196-254), so the three-body sums may run over per-atom
 * moments instead of the lists -- plus the largest window sizes the kernels then need.  The certificate is formed by this call (a
 * few small kernels over the lists, then one wait for the stream: about what the moment kernels save in four steps on a 10k-atom
 * Cu cell, in one step on a dense cell), so ask once per topology that will be used repeatedly and keep the word with the buffer. */
#define M3G_TOPO_TB_COMPLETE 1
int m3g_topology_hints(int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs, const void* topo,
                       int32_t* host_hints, void* stream);
/* m3g_topology_build + m3g_topology_hints with ONE wait for the device: for the lists the graph builders emit (triplets sorted by
 * (e1, e2) and symmetric, symmetric edge list) the whole build and the certificate are queued on that assumption and a single
 * read-back confirms it; other lists redo the affected parts.  host_hints may be NULL (then exactly m3g_topology_build). */
int m3g_topology_build_hints(int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs,
                             const int64_t* edge_index, const int64_t* triplet_edge_index, const int64_t* batch,
                             void* topo, size_t topo_bytes, int32_t* host_flags, int32_t* host_hints, void* stream);

/* m3g_topology_build_hints for lists THIS LIBRARY's builders have just written (m3g_neighbor_fill / m3g_verlet_fill /
 * m3g_verlet_fill_lists + m3g_threebody_*), untouched since: their triplet lists are symmetric and complete by construction, so the
 * mirror check of the triplet list and the per-row completeness test of the certificate are skipped; index ranges, row order and
 * edge-list symmetry are still checked.  Handing it any other list is a caller error (the moment kernels would then sum over
 * partners the list does not hold).  Replaces the host loops of data/material_graph.py:196-254 on the trajectory path. */
int m3g_topology_build_canonical(int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs,
                                 const int64_t* edge_index, const int64_t* triplet_edge_index, const int64_t* batch,
                                 void* topo, size_t topo_bytes, int32_t* host_flags, int32_t* host_hints, void* stream);

/* m3g_topology_build_canonical in two calls, so that the host prepares its next call (workspace, outputs, the m3g_io block) while
 * the device builds: _begin queues the launches and the copy of the verdict words to `pinned_verdict` -- at least 8 int32 of PINNED
 * host memory (hipHostMalloc / torch pin_memory) the caller leaves untouched until _end; _end (same stream, same arguments) waits for
 * the stream and certifies the buffer, or -- when a check failed or _begin did not apply (no triplets, more atoms than its
 * one-workgroup scan takes) -- runs the general build there and then.  Results and errors are those of m3g_topology_build_canonical. */
int m3g_topology_build_canonical_begin(int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs,
                                       const int64_t* edge_index, const int64_t* triplet_edge_index, const int64_t* batch,
                                       void* topo, size_t topo_bytes, int32_t* pinned_verdict, void* stream);
int m3g_topology_build_canonical_end(int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs,
                                     const int64_t* edge_index, const int64_t* triplet_edge_index, const int64_t* batch,
                                     void* topo, size_t topo_bytes, const int32_t* pinned_verdict, int32_t* host_flags,
                                     int32_t* host_hints, void* stream);

/* Diagnostic / tests: the leading part of a topology buffer that holds the lists the kernels read (the rest is scratch of the build),
 * and which build m3g_topology_build_canonical took on this thread the last time: 1 = the six-launch build for canonical lists,
 * 0 = the general one (a failed check, no hints asked for, no triplets, or more atoms than its one-workgroup scan takes). */
int m3g_topology_data_bytes(int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs, size_t* bytes);
int m3g_topology_debug_last_path(int32_t* path);

/* Sticky error bits the hot call left on a topology buffer (0 = none).  M3G_TOPO_ERR_HINTS: m3g_energy_forces was handed a
 * non-zero m3g_io.topo_hints that is not the word m3g_topology_hints certified for THIS buffer (stale after a rebuild, or copied
 * from another topology): the three-body moment kernels then touch nothing (no out-of-bounds access) and the call's three-body
 * terms, hence its results, are INVALID.  m3g_energy_forces itself never reads the word back (that would stall the stream):
 * a caller that passes hints words around must poll this entry point -- once per topology is enough, the bits are sticky (the
 * Python engine does so at the second call with a topology).  Synchronises the stream. */
#define M3G_TOPO_ERR_HINTS 1
#define M3G_TOPO_ERR_SYNC 2    /* an in-launch wait between workgroup roles ran into its bound (never expected; the rows that were waited for
                                * are replaced by NaN, so the call's forces are NaN as well) */
#define M3G_TOPO_ERR_SPECIES 4 /* an atom_types entry outside [0, num_types): the reference raises there (IndexError at
                                * elemental_energies[atom_types], nn/atom_ref.py:27).  The hot call cannot return it without a wait, so it
                                * never indexes with such a value (clamped), stores NaN as that atom's energy -- its structure's energy is
                                * then NaN -- and sets this bit.  m3g_md_step checks the species itself and returns M3G_ERR_VALUE. */
int m3g_topology_status(int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs, const void* topo,
                        int32_t* host_status, void* stream);

/* Number of ACTIVE edges of a built topology: edges that appear in either column of triplet_edge_index (the three-body
 * arrays of the workspace hold one row per active edge).  Synchronises the stream. */
int m3g_topology_active_edges(int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs, const void* topo,
                              int64_t* host_count, void* stream);

/* ---- the hot call: replaces Gradient.forward over the whole Sequential (nn/gradient.py:25-64,
 * model/build.py:37-81): energies, forces, virial stresses ------------------------------------- */
int m3g_workspace_bytes(const m3g_plan* plan, int64_t n_atoms, int64_t n_edges, int64_t n_triplets,
                        int64_t n_structs, size_t* bytes);

typedef struct {
  /* inputs (MaterialGraph) */
  int64_t n_atoms, n_edges, n_triplets, n_structs;
  const float* pos;               /* [N,3] */
  const int64_t* atom_types;      /* [N] */
  const int32_t* edge_cell_shift; /* [E,3] */
  const float* lattice;           /* [S,3,3] */
  const void* topo;               /* from m3g_topology_build for the same index tensors */
  const int64_t* triplet_edge_index; /* [2,T] original order; only read when triplet_angles != NULL */
  /* required outputs */
  float* total_energy; /* [S]   MaterialGraphKey.TOTAL_ENERGY */
  float* forces;       /* [N,3] MaterialGraphKey.FORCES (may be NULL: energy only, no reverse pass) */
  /* optional outputs (NULL to skip) -- the other keys the reference writes */
  float* stresses;               /* [S,6] Voigt xx,yy,zz,yz,zx,xy (nn/gradient.py:39-62) */
  float* scaled_total_energy;    /* [S] */
  float* scaled_atomic_energies; /* [N] */
  float* node_features;          /* [N,D]  "x" after the last block */
  float* edge_attr;              /* [E,D]  after the last block */
  float* edge_distances;         /* [E] (scaled length units) */
  float* edge_weights;           /* [E,n_max] */
  float* triplet_angles;         /* [T] cos(theta_jik), original triplet order */
  float* mid_edge_features;      /* [num_blocks,E,l_max*n_max] three-body aggregate of every block */
  /* optional input */
  int32_t topo_hints;            /* from m3g_topology_hints for `topo` (0: none -- always valid, the general kernels) */
  int32_t reserved;              /* 0 */
} m3g_io;

int m3g_energy_forces(const m3g_plan* plan, const m3g_io* io, void* workspace, size_t workspace_bytes, void* stream);

/* ---- stage entry points: the reference modules a caller may run on their own ----------------- */
/* ScaleLength + DistanceAndAngle (nn/scale.py:24-29, nn/invariant.py:20-59) */
int m3g_distance_angle(double length_scale, int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs,
                       const float* pos, const float* lattice, const int32_t* edge_cell_shift, const void* topo,
                       const int64_t* triplet_edge_index, float* scratch_unit_vectors /* [E,3] */,
                       float* edge_distances, float* triplet_angles, void* stream);
/* EdgeFeaturizer.forward (nn/featurizer.py:81-100); host_em/dm/coeff are HOST arrays [n_max] */
int m3g_edge_featurizer(int32_t n_max, double scaled_cutoff, const float* host_em, const float* host_dm,
                        const float* host_coeff, int64_t n_edges, const float* edge_distances, float* edge_weights,
                        void* stream);
/* AtomFeaturizer.forward (nn/featurizer.py:33-38): x[a,:] = W[:, types[a]]; weight [D,num_types] DEVICE */
int m3g_atom_featurizer(int32_t num_types, int32_t dim, const float* weight, int64_t n_atoms,
                        const int64_t* atom_types, float* x, void* stream);
/* AtomRef.forward (nn/atom_ref.py:25-29) */
int m3g_atom_ref(int32_t num_types, const float* elemental_energies, int64_t n_atoms, const int64_t* atom_types,
                 float* out, void* stream);

/* ---- stand-alone forward of the block modules (any size; run-time-sized fp32 kernels, csrc/m3g_generic.hip).  The reference's
 * modules can be called one by one (tests/test_model.py:14-38 runs the bare Sequential); these are their kernels. -------------- */
/* torch.nn.Linear + activation: Y = act(X W^T + b); act 0 none, 1 SiLU, 2 sigmoid (nn/core.py:31-59, nn/featurizer.py:128-132) */
int m3g_linear(int64_t n, int32_t in_features, int32_t out_features, const float* x, const float* weight, const float* bias /* or NULL */,
               int32_t act, float* y, void* stream);
/* y = a * b: the dense(x) * gate(x) of GatedMLP.forward (nn/core.py:61-62) */
int m3g_multiply(int64_t n, const float* a, const float* b, float* y, void* stream);
/* NormalizedSphericalBessel.forward (nn/interaction.py:268-281): out [l_max, n_max, n]; host_zeros / host_factors HOST [l_max*n_max] */
int m3g_bessel_basis(int32_t l_max, int32_t n_max, double cutoff, const float* host_zeros, const float* host_factors, int64_t n,
                     const float* rs, float* out, void* stream);
/* ThreeBodyInteration.forward (nn/interaction.py:187-223): edge_attr [E,D] updated in place from the graph's edge_distances,
 * triplet_angles and node features; scratch (N*C + E*C + 2*E*D) floats, C = l_max*n_max; mid (or NULL) receives the aggregate [E,C] */
int m3g_three_body(int32_t l_max, int32_t n_max, int32_t embedding_dim, double scaled_cutoff, double scaled_threebody_cutoff,
                   const float* host_zeros, const float* host_factors, int64_t n_atoms, int64_t n_edges, int64_t n_triplets,
                   const int64_t* edge_index, const int64_t* triplet_edge_index, const float* edge_distances, const float* triplet_angles,
                   const float* x, const float* w_sigmoid, const float* b_sigmoid, const float* w_dense, const float* w_gate, float* scratch,
                   float* edge_attr, float* mid, void* stream);
/* M3GNetConv.forward (nn/conv.py:63-97): x [N,D] and edge_attr [E,D] updated in place.  host_params: HOST array of 18 DEVICE pointers,
 * edge MLP {dense.0.weight, gate.0.weight, dense.0.bias, gate.0.bias, dense.2.weight, gate.2.weight, dense.2.bias, gate.2.bias,
 * edge_linear.weight}, then the node MLP likewise (node_linear.weight last) */
int m3g_conv_block_scratch_bytes(int32_t embedding_dim, int64_t n_edges, size_t* bytes);
int m3g_conv_block(int32_t embedding_dim, int32_t n_max, int64_t n_atoms, int64_t n_edges, int64_t n_triplets, int64_t n_structs,
                   const void* topo, const float* const* host_params, const float* edge_weights, float* x, float* edge_attr, float* scratch,
                   size_t scratch_bytes, void* stream);
/* AtomWiseReadout.forward (nn/readout.py:39-58).  host_params: HOST array of 12 DEVICE pointers {dense.0.weight, dense.0.bias,
 * dense.2.weight, dense.2.bias, dense.4.weight, dense.4.bias, gate.0.weight, ...}; scratch (6*N*D + 2*N) floats */
int m3g_readout(int32_t embedding_dim, int64_t n_atoms, int64_t n_structs, const float* const* host_params, double energy_scale,
                const float* x, const float* elemental_energies_per_atom, const int64_t* batch, float* scaled_atomic_energies,
                float* scaled_total_energy, float* total_energy, float* scratch, void* stream);

/* ---- graph construction on the GPU -------------------------------------
 * Periodic neighbour list: replaces get_all_neighbors_with_cell_shifts (data/material_graph.py:168-193, pymatgen
 * Structure.get_all_neighbors).  pos [N,3] and lattice [S,3,3] are DEVICE fp64 (pymatgen works in double), batch
 * [N] int64 sorted.  Output order: centre atom, edge cell shift (sx, sy, sz) lexicographic, neighbour index
 * (the shift refers to the given coordinates: the order does not depend on which atoms sit outside the home cell).  Two phases because the
 * edge count is data dependent: *_count synchronises the stream and returns E, the caller allocates, *_fill writes.
 * max_images >= (2rx+1)(2ry+1)(2rz+1) of every structure, r_p = ceil((cutoff+1e-8) |a_q x a_r| / V). */
int m3g_neighbor_scratch_bytes(int64_t n_atoms, int64_t n_structs, int64_t max_images, size_t* bytes);
int m3g_neighbor_count(int64_t n_atoms, int64_t n_structs, int64_t max_images, const double* pos, const double* lattice,
                       const int64_t* batch, double cutoff, void* scratch, size_t scratch_bytes, int64_t* host_n_edges,
                       void* stream);
/* m3g_neighbor_count that also returns, from the same pass and the same wait, the number of triplets the list will have under
 * `threebody_cutoff` (valid edges: fp32 length <= it, as compute_threebody thresholds them): a caller building both lists sizes
 * every tensor after ONE wait for the device and follows up with m3g_neighbor_fill + m3g_threebody_build. */
int m3g_neighbor_count_triplets(int64_t n_atoms, int64_t n_structs, int64_t max_images, const double* pos, const double* lattice,
                                const int64_t* batch, double cutoff, float threebody_cutoff, void* scratch, size_t scratch_bytes,
                                int64_t* host_n_edges, int64_t* host_n_triplets, void* stream);
int m3g_neighbor_fill(int64_t n_atoms, int64_t n_structs, int64_t max_images, const int64_t* batch, double cutoff,
                      void* scratch, int64_t n_edges, int64_t* edge_index /* [2,E] */, int32_t* edge_cell_shift /* [E,3] */,
                      double* distances /* [E] */, void* stream);
/* Three-body index: replaces compute_threebody (data/material_graph.py:196-254), same triplet order.  distances are
 * the fp32 edge lengths the reference thresholds (material_graph.py:191,224). */
int m3g_threebody_scratch_bytes(int64_t n_atoms, int64_t n_edges, size_t* bytes);
int m3g_threebody_count(int64_t n_atoms, int64_t n_edges, const int64_t* edge_index, const float* distances,
                        float threebody_cutoff, void* scratch, size_t scratch_bytes, int64_t* host_n_triplets, void* stream);
int m3g_threebody_fill(int64_t n_atoms, int64_t n_edges, const int64_t* edge_index, void* scratch, int64_t n_triplets,
                       int64_t* triplet_edge_index /* [2,T] */, int64_t* num_triplet_i /* [N] or NULL */,
                       int32_t* num_triplet_ij /* [E] or NULL */, void* stream);
/* m3g_threebody_count + m3g_threebody_fill in one call when n_triplets is known already (m3g_neighbor_count_triplets): no wait
 * for the device; scratch from m3g_threebody_scratch_bytes. */
int m3g_threebody_build(int64_t n_atoms, int64_t n_edges, const int64_t* edge_index, const float* distances, float threebody_cutoff,
                        void* scratch, size_t scratch_bytes, int64_t n_triplets, int64_t* triplet_edge_index /* [2,T] */,
                        int64_t* num_triplet_i /* [N] */, int32_t* num_triplet_ij /* [E] */, void* stream);

/* ---- skin ("Verlet") list: repeated evaluation along an MD trajectory without a search per step --------------------
 * The reference rebuilds everything for every structure it sees, in Python (data/material_graph.py:133-254).  For a trajectory
 * the caller keeps CANDIDATES -- the list m3g_neighbor_count/fill return for cutoff + skin at reference positions -- on the
 * device, with their row pointers (m3g_verlet_rows) and one membership byte per candidate.  While no atom has moved by more than
 * skin / 2 since the reference, the candidates that pass d <= cutoff, in candidate order, are EXACTLY the list a fresh search at
 * the current positions returns (same edges, order, shifts, hence the same triplets): the canonical order does not depend on the
 * positions, and every distance is formed by the search's own arithmetic.
 *   m3g_verlet_update   one pass over atoms and candidates at the current positions (DEVICE fp64, unwrapped like pos_ref): the
 *                       largest displacement since pos_ref, whether any candidate's membership (bit 0: in the list; bit 1: fp32
 *                       length within the three-body cutoff) differs from cand_state -- the membership the caller's current lists
 *                       were filled with --, and the sizes E, T the lists have now.  Waits for the stream once.  changed == 0 and
 *                       max_disp < skin / 2: the current edge_index / edge_cell_shift / triplets / topology / hints stay valid,
 *                       only the positions of the next m3g_energy_forces call are new.
 *   m3g_verlet_fill     (changed != 0, max_disp < skin / 2) writes the new list from the candidates and updates cand_state; follow
 *                       with m3g_threebody_build (n_triplets from the update) and m3g_topology_build.  No wait.
 * max_disp >= skin / 2 (or a changed lattice): search again with cutoff + skin, pos_ref = pos, then update with cand_state = NULL
 * (fresh candidates: reported as changed) + fill.
 * cand_row_ptr: int32 [N + 2] (N + 1 pointers and one scratch word). */
int m3g_verlet_scratch_bytes(int64_t n_atoms, int64_t n_candidates, size_t* bytes);
int m3g_verlet_rows(int64_t n_atoms, int64_t n_candidates, const int64_t* cand_edge_index /* [2,Ec] */, int32_t* cand_row_ptr, void* stream);
int m3g_verlet_update(int64_t n_atoms, int64_t n_structs, int64_t n_candidates, const double* pos, const double* pos_ref,
                      const double* lattice, const int64_t* batch, const int64_t* cand_edge_index, const int32_t* cand_shift,
                      const int32_t* cand_row_ptr, double cutoff, float threebody_cutoff, const uint8_t* cand_state, void* scratch,
                      size_t scratch_bytes, double* host_max_disp, int32_t* host_changed, int64_t* host_n_edges,
                      int64_t* host_n_triplets, void* stream);
/* m3g_verlet_update without the wait: the pass and its 48-byte result copy are queued on the stream; host_out (PINNED host memory,
 * SIX words) receives {bits of max_disp^2 as a double, changed, E, T, (internal), the longest candidate row -- what
 * m3g_verlet_fill_lists wants as max_cand_row} when the stream reaches the copy.  A caller may queue the evaluation
 * behind it on the assumption that nothing changed and read the verdict afterwards. */
int m3g_verlet_update_async(int64_t n_atoms, int64_t n_structs, int64_t n_candidates, const double* pos, const double* pos_ref,
                            const double* lattice, const int64_t* batch, const int64_t* cand_edge_index, const int32_t* cand_shift,
                            const int32_t* cand_row_ptr, double cutoff, float threebody_cutoff, const uint8_t* cand_state, void* scratch,
                            size_t scratch_bytes, uint64_t* host_out /* [6] */, void* stream);
int m3g_verlet_fill(int64_t n_atoms, int64_t n_candidates, int64_t n_edges, void* scratch, const int64_t* cand_edge_index,
                    const int32_t* cand_shift, const int32_t* cand_row_ptr, int64_t* edge_index /* [2,E] */,
                    int32_t* edge_cell_shift /* [E,3] */, double* distances /* [E] */, uint8_t* cand_state /* [Ec] out */, void* stream);
/* m3g_verlet_fill + m3g_threebody_build in TWO launches: edge list, shifts, membership bytes, triplet list (the reference's
 * order, data/material_graph.py:239-248) and the per-centre / per-edge triplet counts straight from the candidates and the state the
 * preceding m3g_verlet_update left in `scratch` (n_edges, n_triplets: its E and T).  max_cand_row = the longest candidate row (the
 * caller knows it from cand_row_ptr); rows beyond M3G_VERLET_FILL_LISTS_MAX_ROW or more than 262,144 atoms return
 * M3G_ERR_UNSUPPORTED -- use the two calls above, which have no limits and return identical lists.  No wait.
 * Replaces, for a trajectory, the per-structure rebuild of data/material_graph.py:168-254. */
#define M3G_VERLET_FILL_LISTS_MAX_ROW 1024
int m3g_verlet_fill_lists(int64_t n_atoms, int64_t n_candidates, int64_t n_edges, int64_t n_triplets, int64_t max_cand_row, void* scratch,
                          const int64_t* cand_edge_index, const int32_t* cand_shift, const int32_t* cand_row_ptr,
                          int64_t* edge_index /* [2,E] */, int32_t* edge_cell_shift /* [E,3] */, uint8_t* cand_state /* [Ec] out */,
                          int64_t* triplet_edge_index /* [2,T] */, int64_t* num_triplet_i /* [N] or NULL */,
                          int32_t* num_triplet_ij /* [E] or NULL */, void* stream);

/* ---- one trajectory step per call: skin test, lists + topology when they changed, energies / forces / stresses ------------------
 * Replaces, for a structure followed along a trajectory, the reference's per-frame MaterialGraph.from_structure
 * (data/material_graph.py:132-254) + Gradient.forward (nn/gradient.py:25-64).  The caller searches the candidates (cutoff + skin;
 * m3g_neighbor_*, m3g_verlet_rows) and hands them over with list buffers of the candidates' capacity -- every pointer device memory
 * it owns and keeps alive until the next m3g_md_set_lists / m3g_md_destroy; the library then sequences m3g_verlet_update_async, (when
 * a pair crossed a cutoff) m3g_verlet_fill_lists + m3g_topology_build_canonical, and m3g_energy_forces itself.  Per step the host
 * waits twice at most (the verdict's sizes; the topology's certificate) and allocates nothing.  Same kernels on the same inputs as
 * the separate calls: identical lists, bit-identical results. */
typedef struct m3g_md m3g_md;
typedef struct {
  int64_t n_atoms, n_structs, n_cand;     /* N, S, number of candidate pairs Ec */
  int64_t cap_edges, cap_triplets;        /* capacity of the list buffers below (cap_edges >= n_cand; cap_triplets: an upper bound of
                                           * T for every configuration within skin / 2 of pos_ref, e.g. sum_i c_i (c_i - 1) over the
                                           * candidates within threebody_cutoff + skin) */
  double cutoff, threebody_cutoff, skin;
  const double* pos_ref;                  /* [N,3] positions the candidates were searched at */
  const double* lattice;                  /* [S,3,3] */
  const float* lattice32;                 /* [S,3,3] the same in fp32 (m3g_io.lattice) */
  const int64_t* batch;                   /* [N] */
  const int64_t* atom_types;              /* [N] */
  const int64_t* cand_edge_index;         /* [2,Ec] */
  const int32_t* cand_shift;              /* [Ec,3] */
  const int32_t* cand_row_ptr;            /* [N+2] (m3g_verlet_rows) */
  uint8_t* cand_state;                    /* [Ec+16] membership bytes (written by the library) */
  void* verlet_scratch; size_t verlet_scratch_bytes;   /* m3g_verlet_scratch_bytes(N, Ec) */
  int64_t* edge_index;                    /* [2 * cap_edges]      the lists the library re-derives: read as [2,E] / [E,3] / [2,T] / */
  int32_t* edge_cell_shift;               /* [3 * cap_edges]      [N] / [E] with the E, T of m3g_md_result */
  int64_t* triplet_edge_index;            /* [2 * cap_triplets] */
  int64_t* num_triplet_i;                 /* [N] */
  int32_t* num_triplet_ij;                /* [cap_edges] */
  float* pos32;                           /* [N,3] */
  void* topo; size_t topo_bytes;          /* m3g_topology_bytes(N, cap_edges, cap_triplets, S) */
  void* workspace; size_t workspace_bytes;/* m3g_workspace_bytes(plan, N, cap_edges, cap_triplets, S) */
} m3g_md_lists;
#define M3G_MD_REUSE 0        /* lists unchanged: the step ran on the standing lists */
#define M3G_MD_REFILL 1       /* a pair crossed a cutoff (or first step / asked for): lists and topology re-derived, then the step */
#define M3G_MD_NEED_SEARCH 2  /* an atom moved further than skin / 2: NOTHING was evaluated; search again, m3g_md_set_lists, call again */
#define M3G_MD_UNSUPPORTED 3  /* candidate rows beyond M3G_VERLET_FILL_LISTS_MAX_ROW, or lists beyond the buffers' capacity: NOTHING was
                               * evaluated; use the separate calls for this step */
typedef struct {
  int32_t path, topo_hints;
  int64_t n_edges, n_triplets;            /* of the lists the step ran on (path 0 / 1), of the verdict otherwise */
  double max_displacement;                /* largest |pos - pos_ref| */
} m3g_md_result;
int m3g_md_create(m3g_md** md);
void m3g_md_destroy(m3g_md* md);
int m3g_md_set_lists(m3g_md* md, const m3g_md_lists* lists);
int m3g_md_invalidate(m3g_md* md);   /* the caller has rewritten cand_state / the list buffers through other calls: re-derive at the next step */
/* pos: [N,3] fp64 device positions (unwrapped).  forces / stresses may be NULL (energies only).  force_refill != 0: re-derive the
 * lists whatever the verdict says (tests, timing).  Waits for `stream` (verdict) -- the outputs are queued, not waited for.
 * Behind that one wait the call also learns (ABI 6): whether atom_types lies in [0, num_types) of `plan` -- checked once per list set
 * and model; M3G_ERR_VALUE otherwise, as the reference raises IndexError (nn/atom_ref.py:25-29) -- and whether an EARLIER step left
 * sticky error bits on the topology buffer (M3G_TOPO_ERR_*): M3G_ERR_STATE, nothing evaluated, the next call re-derives lists and
 * topology.  M3G_MD_UNSUPPORTED also answers a workspace that has become too small for the plan's current options (the caller makes
 * buffers again).  cand_state and the list buffers may be NULL when n_cand == 0. */
int m3g_md_step(m3g_md* md, const m3g_plan* plan, const double* pos, float* total_energy, float* forces, float* stresses,
                int32_t force_refill, m3g_md_result* host_result, void* stream);

/* ---- measurement: per-stage device time from HIP events recorded on the call's own stream ---------
 * m3g_profile_enable(plan, 1) makes every following m3g_energy_forces record an event pair around each
 * stage launch; m3g_profile_read synchronises those events, returns per-stage totals since the last
 * read and resets.  `names[i]` points to static strings.  Off by default (no events, no overhead). */
#define M3G_MAX_STAGES 16
int m3g_profile_enable(m3g_plan* plan, int32_t enable);
int m3g_profile_read(m3g_plan* plan, int32_t* n_stages, const char** names /* [M3G_MAX_STAGES] */,
                     float* total_ms /* [M3G_MAX_STAGES] */, int32_t* launches /* [M3G_MAX_STAGES] */);

/* Diagnostic only: with option "stamps" = 1 the forward edge kernel runs a stamped variant (s_memtime per
 * phase), = 2 the reverse edge-MLP kernel of the two-kernel reverse, = 3 the fused reverse kernel (f16x3 mode; its waves ADD
 * their sums over the launches since the option was set); this copies the per-wave phase cycle sums [256 workgroups][16 wave
 * slots][12 phases] to the host. */
int m3g_debug_read_stamps(m3g_plan* plan, uint64_t* host_out);
/* Diagnostic only: number of HIP streams / events the plan currently owns (internal side stream, fork / join events, profiler
 * event pool).  They are bound to the plan's device: a commit that moves the plan to another device releases all of them. */
int m3g_debug_live_handles(const m3g_plan* plan, int32_t* out);

/* Test hooks for the library's own device-wide exclusive scan and stable radix sort (csrc/m3g_prims.h: the primitives behind the
 * neighbour search and the list / topology construction).  elem_bytes / key_bytes: 4 or 8 (integers); in == out allowed for the
 * scan; the sort works in place on `keys` (+ `vals`, int32, may be NULL) over the key bits [begin_bit, end_bit).  Both allocate their
 * temporary storage and wait for the stream: diagnostics only. */
int m3g_debug_exclusive_scan(int32_t elem_bytes, int64_t n, const void* in, void* out, void* stream);
int m3g_debug_radix_sort(int32_t key_bytes, int64_t n, void* keys, int32_t* vals, int32_t begin_bit, int32_t end_bit, void* stream);

/* Measurement: what ONE m3g_energy_forces call with these arguments puts on a stream -- kernel launches and other operations (memsets,
 * copies) -- counted by capturing the call's own launch sequence into a HIP graph on an internal stream (nothing executes, no buffer
 * is touched) and counting the graph's nodes.  The sequence counted is the un-profiled one (the stage profiler changes it). */
int m3g_count_launches(const m3g_plan* plan, const m3g_io* io, void* workspace, size_t workspace_bytes, int32_t* kernel_launches,
                       int32_t* other_operations);

#define M3G_ABI_VERSION 6   /* 2: m3g_io.topo_hints, m3g_topology_hints; 3: m3g_verlet_*, m3g_topology_status, hints word certified on the buffer,
                             * canonical edge order by the shift relative to the given coordinates, default precision fp32;
                             * 4: m3g_verlet_fill_lists, m3g_topology_build_canonical, M3G_TOPO_ERR_SYNC, options small_tiles / small_launches / fuse_node_tb;
                             * 5: m3g_topology_build_canonical_begin / _end, m3g_topology_data_bytes, option legendre_backward, m3g_md_*;
                             * 6: M3G_TOPO_ERR_SPECIES (species checked on the library side, m3g_md_step returns M3G_ERR_VALUE), m3g_count_launches,
                             *    m3g_debug_exclusive_scan / m3g_debug_radix_sort (the library's own scan and sort: no hipCUB) */

#ifdef __cplusplus
}
#endif
#endif /* M3GNET_HIP_H */
