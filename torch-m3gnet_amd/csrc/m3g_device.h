// Device-side scalar helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace m3g {

// Accurate expf (ocml): the 1e-5 energy / 1e-4 force budget leaves no room for the ~1e-6 relative
// error of the hardware exp2 approximation accumulated over 3 blocks of gated MLPs.
__device__ __forceinline__ float sigmoid_f(float p) { return 1.f / (1.f + expf(-p)); }
__device__ __forceinline__ float silu_f(float p) { return p * sigmoid_f(p); }
__device__ __forceinline__ float dsilu_f(float p) {
  float s = sigmoid_f(p);
  return s * (1.f + p * (1.f - s));
}

// Species index of an atom as a table index.  The reference indexes elemental_energies / one_hot with it and raises on a value
// outside [0, num_types) (nn/atom_ref.py:27, nn/featurizer.py:33-38); a kernel cannot raise, so no table is ever indexed with such a
// value (clamped here) and the readout kernels turn `bad` into a NaN energy + the sticky M3G_TOPO_ERR_SPECIES bit.
__device__ __forceinline__ int64_t species_index(int64_t ty, int num_types, bool& bad) {
  bad = ty < 0 || ty >= num_types;
  return ty < 0 ? 0 : (ty >= num_types ? num_types - 1 : ty);
}
__device__ __forceinline__ void flag_bad_species(const int32_t* topo_flags) {
  atomicOr(const_cast<int32_t*>(topo_flags) + 8, 4 /* M3G_TOPO_ERR_SPECIES */);
}

}  // namespace m3g
