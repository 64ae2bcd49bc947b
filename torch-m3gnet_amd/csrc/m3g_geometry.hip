// Per-edge geometry and bases (stage S0), its reverse (B0), force gather, virial stress.
// Reference: nn/scale.py:24-29, nn/invariant.py:20-59, nn/featurizer.py:81-100,
//            nn/interaction.py:268-350,389-400, nn/gradient.py:35-62.
// All HBM-bound elementwise work: one thread per edge / atom, coalesced SoA-ish rows.
#include "m3g_internal.h"
#include "m3g_struct_sum.h"
#include "m3g_geometry_body.h"

namespace m3g {

template <bool FULL, int L, int R>
__global__ void __launch_bounds__(256) k_geometry(Consts c, GeomArgs a) {
  geometry_body<FULL, L, R>(c, a, blockIdx.x);
}

__global__ void __launch_bounds__(256) k_geometry_reverse(GeomRev a, float* __restrict__ dr) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= a.E) return;
  float rx, ry, rz;
  edge_dr(a, e, rx, ry, rz);
  dr[e * 3 + 0] = rx;
  dr[e * 3 + 1] = ry;
  dr[e * 3 + 2] = rz;
}

// F_i = (sum_{e in row(i)} dr_e - sum_{e: dst(e)=i} dr_e) / length_scale   (no atomics: both CSR lists)
// 16 lanes per atom (a DPP row): each lane takes every 16th edge of both lists, then the row is reduced -- a thread per
// atom left 160 waves on the whole chip walking 84 dependent 12-byte gathers each.
__device__ __forceinline__ float inv_volume(const float* __restrict__ L) {
  const float cx = L[4] * L[8] - L[5] * L[7], cy = L[5] * L[6] - L[3] * L[8], cz = L[3] * L[7] - L[4] * L[6];
  return 1.f / fabsf(L[0] * cx + L[1] * cy + L[2] * cz);
}
// sum_a pos_a (x) F_a / V of structure s (nn/gradient.py:39-62), Voigt order
template <int REAL>
__device__ __forceinline__ void struct_stress(int s, const int32_t* __restrict__ struct_ptr, const int32_t* __restrict__ flags, int64_t n_atoms,
                                              const int32_t* __restrict__ batch, const float* __restrict__ pos, const float* __restrict__ lattice,
                                              const float* forces, float* __restrict__ stresses, float* part) {
  float tot[6];
  struct_reduce<6, REAL>(s, struct_ptr, flags, n_atoms, batch, tot, part, [&](int a, int ss, float* v) {
    const float inv = inv_volume(lattice + (int64_t)ss * 9);
    const float px = pos[a * 3], py = pos[a * 3 + 1], pz = pos[a * 3 + 2];
    const float fx = forces[a * 3], fy = forces[a * 3 + 1], fz = forces[a * 3 + 2];
    v[0] = px * fx * inv; v[1] = py * fy * inv; v[2] = pz * fz * inv;
    v[3] = py * fz * inv; v[4] = pz * fx * inv; v[5] = px * fy * inv;
  });
  if (threadIdx.x < 6) stresses[(int64_t)s * 6 + threadIdx.x] = tot[threadIdx.x];
}
// virial inputs of the launches that end with the reference stress (sum_a pos_a (x) F_a / V per structure) formed by their LAST
// workgroup: counter == nullptr -> no stress in this launch
struct StressTail {
  int64_t S;
  const int32_t *struct_ptr, *flags, *batch;
  const float *pos, *lattice;
  float* stresses;
  int32_t* counter;
};
__global__ void __launch_bounds__(256) k_force_gather(float length_scale, int64_t N, const int32_t* __restrict__ row_ptr,
                                                      const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_edge,
                                                      const float* __restrict__ dr, float* forces,
                                                      float* __restrict__ stresses, int64_t n_stress, StressTail st) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  // (a stress kernel that accumulates with atomics would need its output cleared: kept for the generic path's callers)
  if (blockIdx.x == 0) for (int64_t k = threadIdx.x; k < n_stress; k += blockDim.x) stresses[k] = 0.f;
  const int64_t i = gid >> 4;
  const int l = (int)(gid & 15);
  float fx = 0.f, fy = 0.f, fz = 0.f;
  if (i < N) {
    for (int e = row_ptr[i] + l; e < row_ptr[i + 1]; e += 16) { fx += dr[(int64_t)e * 3]; fy += dr[(int64_t)e * 3 + 1]; fz += dr[(int64_t)e * 3 + 2]; }
    for (int k = in_ptr[i] + l; k < in_ptr[i + 1]; k += 16) {
      const int64_t e = in_edge[k];
      fx -= dr[e * 3]; fy -= dr[e * 3 + 1]; fz -= dr[e * 3 + 2];
    }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {   // fixed-order tree inside the 16-lane group
    fx += __shfl_xor(fx, off, 16); fy += __shfl_xor(fy, off, 16); fz += __shfl_xor(fz, off, 16);
  }
  if (i < N && l == 0) {
    forces[i * 3 + 0] = fx / length_scale;
    forces[i * 3 + 1] = fy / length_scale;
    forces[i * 3 + 2] = fz / length_scale;
  }
  if (!st.counter) return;   // uniform
  // publish this workgroup's forces, then count it: the workgroup that counts last sees all forces and forms the virial of
  // every structure exactly as k_struct_stress does (no workgroup ever waits for another)
  __shared__ float part[kStructThreads * 6];
  __shared__ int s_last;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(st.counter, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  for (int s = 0; s < (int)st.S; ++s) struct_stress<256>(s, st.struct_ptr, st.flags, N, st.batch, st.pos, st.lattice, forces, st.stresses, part);
}

// ea != nullptr: the structure's energy sums as well (k_struct_energy's work, deferred to this launch: nothing in the reverse pass
// reads the totals, so a step that ends with the virial needs no launch of its own for them)
__global__ void __launch_bounds__(kStructThreads) k_struct_stress(const int32_t* __restrict__ struct_ptr, const int32_t* __restrict__ flags,
                                                       int64_t n_atoms, const int32_t* __restrict__ batch,
                                                       const float* __restrict__ pos, const float* __restrict__ lattice,
                                                       const float* __restrict__ forces, float* __restrict__ stresses,
                                                       const float* __restrict__ ea, float energy_scale, float* __restrict__ scaled_total,
                                                       float* __restrict__ total) {
  __shared__ float part[kStructThreads * 7];
  if (!ea) {
    struct_stress<kStructThreads>(blockIdx.x, struct_ptr, flags, n_atoms, batch, pos, lattice, forces, stresses, part);
    return;
  }
  // energy and virial in ONE pass over the structure's atoms (seven independent sums, each in struct_reduce's order: the same
  // bits as struct_energy followed by struct_stress, half the barriers and one wait per atom instead of two)
  float tot[7];
  struct_reduce<7, kStructThreads>(blockIdx.x, struct_ptr, flags, n_atoms, batch, tot, part, [&](int a, int ss, float* v) {
    const float inv = inv_volume(lattice + (int64_t)ss * 9);
    const float px = pos[a * 3], py = pos[a * 3 + 1], pz = pos[a * 3 + 2];
    const float fx = forces[a * 3], fy = forces[a * 3 + 1], fz = forces[a * 3 + 2];
    v[0] = px * fx * inv; v[1] = py * fy * inv; v[2] = pz * fz * inv;
    v[3] = py * fz * inv; v[4] = pz * fx * inv; v[5] = px * fy * inv;
    v[6] = ea[a];
  });
  if (threadIdx.x < 6) stresses[(int64_t)blockIdx.x * 6 + threadIdx.x] = tot[threadIdx.x];
  if (threadIdx.x == 0) {
    scaled_total[blockIdx.x] = tot[6];
    total[blockIdx.x] = energy_scale * tot[6];
  }
}
__global__ void __launch_bounds__(kStructThreads) k_struct_stress_pair(const int32_t* __restrict__ struct_ptr, const int32_t* __restrict__ flags,
                                                            int64_t n_atoms, const int32_t* __restrict__ batch,
                                                            const int32_t* __restrict__ row_ptr, const float* __restrict__ lattice,
                                                            const float* __restrict__ u, const float* __restrict__ dist,
                                                            const float* __restrict__ dr, float* __restrict__ stresses) {
  __shared__ float part[kStructThreads * 6];
  float tot[6];
  struct_reduce<6, kStructThreads>(blockIdx.x, struct_ptr, flags, n_atoms, batch, tot, part, [&](int a, int s, float* v) {
    const float inv = -inv_volume(lattice + (int64_t)s * 9);
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = 0.f;
    for (int e = row_ptr[a]; e < row_ptr[a + 1]; ++e) {
      const float d = dist[e];
      const float rx = d * u[e * 3], ry = d * u[e * 3 + 1], rz = d * u[e * 3 + 2];
      const float gx = dr[e * 3], gy = dr[e * 3 + 1], gz = dr[e * 3 + 2];
      v[0] += rx * gx; v[1] += ry * gy; v[2] += rz * gz;
      v[3] += 0.5f * (ry * gz + rz * gy); v[4] += 0.5f * (rz * gx + rx * gz); v[5] += 0.5f * (rx * gy + ry * gx);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] *= inv;
  });
  if (threadIdx.x < 6) stresses[(int64_t)blockIdx.x * 6 + threadIdx.x] = tot[threadIdx.x];
}

__global__ void __launch_bounds__(kStructThreads) k_struct_energy(const int32_t* __restrict__ struct_ptr, const int32_t* __restrict__ flags,
                                                       int64_t n_atoms, const int32_t* __restrict__ batch, const float* __restrict__ ea,
                                                       float energy_scale, float* __restrict__ scaled_total, float* __restrict__ total) {
  __shared__ float part[kStructThreads];
  struct_energy<kStructThreads>(blockIdx.x, struct_ptr, flags, n_atoms, batch, ea, energy_scale, scaled_total, total, part);
}
void launch_struct_energy(const Consts& c, const Topo& t, const float* ea, float* scaled_total, float* total, hipStream_t s) {
  if (t.S > 0)
    hipLaunchKernelGGL(k_struct_energy, dim3((unsigned)t.S), dim3(kStructThreads), 0, s, t.struct_ptr, t.flags, t.N, t.batch, ea, c.energy_scale,
                       scaled_total, total);
}

// virial: sum_a pos_a (x) F_a / V in Voigt order xx,yy,zz,yz,zx,xy (nn/gradient.py:39-62): k_struct_stress above.
// PBC-consistent virial (docs/gradient.md:47-84): with every pair vector r_e = d_e u_e
// straining as r -> (1 + eps) r, dE/d eps = sum_e r_e (x) dE/dr_e over the directed edges, so
//   sigma = -(1/V) sum_e r_e (x) dE/dr_e      (same sign convention as the reference's sum_a pos_a (x) F_a / V, to
// which it reduces when no edge crosses a cell boundary): k_struct_stress_pair above, each atom summing its own CSR row; the
// tensor is symmetric up to rounding and stored symmetrised.  d and dr are both in scaled length, so the scale cancels.

__global__ void __launch_bounds__(256) k_triplet_angles(int64_t T, const int64_t* __restrict__ tei, const float* __restrict__ u,
                                                        float* __restrict__ out) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= T) return;
  int64_t e1 = tei[t], e2 = tei[T + t];
  float c = u[e1 * 3] * u[e2 * 3] + u[e1 * 3 + 1] * u[e2 * 3 + 1] + u[e1 * 3 + 2] * u[e2 * 3 + 2];
  c = fminf(1.f, fmaxf(-1.f, c));   // torch.clamp(cos, -1, 1), nn/invariant.py:40
  // Collinear triplets (a neighbour and its mirror image, or two images of one neighbour) have |cos| = 1 exactly; in fp32
  // the quotient lands within 2 ulp of it, on either side.  The reference's clamp repairs the outside half of those cases;
  // the inside half is the same rounding accident, and arccos -- which the reference's own angle test applies,
  // tests/test_invariance.py:77-82 -- turns 1e-7 there into 4e-4 rad.  Reported angles therefore snap both halves to +-1.
  // (The energy path keeps its own value: cos enters it only through the polynomials P_l, where 1e-7 is rounding noise.)
  if (fabsf(c) > 1.f - 2.4e-7f) c = copysignf(1.f, c);
  out[t] = c;
}

__global__ void __launch_bounds__(256) k_edge_featurizer(Consts c, int64_t E, const float* __restrict__ d, float* __restrict__ out,
                                                         int out_stride) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= E) return;
  float h[kRCap], hp[kRCap];
  radial_basis(c, d[e], h, hp);
  for (int m = 0; m < c.R; ++m) out[e * out_stride + m] = h[m];
}

static inline dim3 grid_for(int64_t n, int tpb = 256) { return dim3((unsigned)((n + tpb - 1) / tpb)); }

void launch_geometry(const Consts& c, const Topo& t, const float* pos, const float* lattice, const int32_t* shift,
                     const Work& w, hipStream_t s) {
  if (t.E == 0) {   // no edge, no geometry kernel: the step's sync words are cleared by a memset instead
    if (w.sync) (void)hipMemsetAsync(w.sync, 0, sizeof(int32_t) * kSyncWords, s);
    return;
  }
  const GeomArgs a = geometry_args(t, pos, lattice, shift, w);
  M3G_DISPATCH_LR(c.L, c.R, hipLaunchKernelGGL((k_geometry<true, L, R>), grid_for(t.E), dim3(256), 0, s, c, a));
}

void launch_distance_only(float length_scale, const Topo& t, const float* pos, const float* lattice, const int32_t* shift,
                          float* u, float* d, hipStream_t s) {
  if (t.E == 0) return;
  Consts c{};
  c.length_scale = length_scale;
  const GeomArgs a{t.E, t.src, t.dst, t.batch, pos, lattice, shift, u, d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL((k_geometry<false, 1, 1>), grid_for(t.E), dim3(256), 0, s, c, a);
}

// Geometry reverse + force gather.  `fuse_stress`: the force-gather launch also forms the reference virial (its last workgroup,
// bit-identical to k_struct_stress) when the batch has few structures; returns whether it did (else the caller launches
// launch_stress / launch_stress_pair).
bool launch_geometry_reverse(const Consts& c, const Topo& t, const Work& w, const float* dh, int dh_parts, float* forces,
                             float* stresses, hipStream_t s, bool fuse_stress, const float* pos, const float* lattice, bool dr_done) {
  const GeomRev g{t.E, w.u, w.d, w.hp, dh, dh_parts, w.dd, w.du, (t.T > 0 && c.B > 0) ? t.act_id : nullptr};
  if (t.E > 0 && !dr_done) hipLaunchKernelGGL(k_geometry_reverse, grid_for(t.E), dim3(256), 0, s, g, w.dr);   // (dr_done: formed by the last three-body reverse)
  const bool fused = fuse_stress && stresses && w.sync && t.N > 0 && t.N <= kFusedSumsMaxAtoms && t.S > 0 && t.S <= kForceTailMaxStructs;
  if (t.N > 0) {
    StressTail st{t.S, t.struct_ptr, t.flags, t.batch, pos, lattice, stresses, fused ? w.sync + kSyncForceTail : nullptr};
    hipLaunchKernelGGL(k_force_gather, grid_for(t.N * 16), dim3(256), 0, s, c.length_scale, t.N, t.row_ptr, t.in_ptr, t.in_edge,
                       w.dr, forces, stresses, (stresses && !fused) ? 6 * t.S : 0, st);
  } else if (stresses) {
    (void)hipMemsetAsync(stresses, 0, sizeof(float) * 6 * t.S, s);
  }
  return fused;
}

// force gather on its own (generic path, m3g_generic.hip): F = -(d E / d r) summed through both CSR lists, / length_scale
void launch_force_gather(float length_scale, const Topo& t, const float* dr, float* forces, float* stresses, hipStream_t s) {
  if (t.N > 0)
    hipLaunchKernelGGL(k_force_gather, grid_for(t.N * 16), dim3(256), 0, s, length_scale, t.N, t.row_ptr, t.in_ptr, t.in_edge, dr, forces,
                       stresses, stresses ? 6 * t.S : 0, StressTail{});
  else if (stresses)
    (void)hipMemsetAsync(stresses, 0, sizeof(float) * 6 * t.S, s);
}

void launch_stress(const Consts& c, const Topo& t, const float* pos, const float* lattice, const float* forces,
                   float* stresses, hipStream_t s, const float* ea, float* scaled_total, float* total) {
  if (t.S > 0)
    hipLaunchKernelGGL(k_struct_stress, dim3((unsigned)t.S), dim3(kStructThreads), 0, s, t.struct_ptr, t.flags, t.N, t.batch, pos, lattice, forces,
                       stresses, ea, c.energy_scale, scaled_total, total);
}

void launch_stress_pair(const Topo& t, const Work& w, const float* lattice, float* stresses, hipStream_t s) {
  if (t.S > 0)
    hipLaunchKernelGGL(k_struct_stress_pair, dim3((unsigned)t.S), dim3(kStructThreads), 0, s, t.struct_ptr, t.flags, t.N, t.batch, t.row_ptr, lattice,
                       w.u, w.d, w.dr, stresses);
}

void launch_triplet_angles(const Topo& t, const int64_t* tei, const float* u, float* out, hipStream_t s) {
  if (t.T > 0) hipLaunchKernelGGL(k_triplet_angles, grid_for(t.T), dim3(256), 0, s, t.T, tei, u, out);
}

void launch_edge_featurizer(const Consts& c, int64_t E, const float* d, float* out, int out_stride, hipStream_t s) {
  if (E > 0) hipLaunchKernelGGL(k_edge_featurizer, grid_for(E), dim3(256), 0, s, c, E, d, out, out_stride);
}

}  // namespace m3g
