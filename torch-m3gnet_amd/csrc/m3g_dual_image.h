// Dual-use LDS weight image for v_mfma_f32_16x16x32_bf16 chains: ONE copy of W serves as the A operand of
// Y^T = W X^T (row reads, ds_read_b64) and of dX^T = W^T dY^T (ds_read_b64_tr_b16, gfx950's transposing LDS read),
// so a reverse kernel that recomputes the forward needs no second, transposed copy of each weight matrix.
//
// W is [ROWS][64] (64 input features = 16 chunks of 4); the image holds bf16(W) ("hi") followed by
// bf16(W - hi) ("lo"), ROWS*128 bytes each.  The 8-byte unit (row, chunk) lives at
//     row*128 + ((chunk ^ swz(row)) << 3),     swz(row) = 4*((row>>1)&3) ^ 2*((row>>3)&1).
// Bank check (64 banks x 4 B, 32-lane groups for ds_read_b64 and ds_read_b64_tr_b16; MI355X_MICROARCH.md, LDS):
//   row reads   lanes (m = lane&15, q = lane>>4) read row ob*16+m, chunk (2s+h)*4+q: the 16 m of a group differ in
//               bank half (row&1) and in the three swizzle bits, q in bit 0 -> 32 distinct bank pairs;
//   transposed  lane 4q'+p of 16-lane group q reads row sh*16+4q+q', chunk 4ob+p: the 8 rows of a 32-lane group
//               differ in bank half and in swizzle bits 2-3, p in bits 0-1 -> 32 distinct bank pairs.
// k order: k-step s of a chain consumes accumulator blocks 2s and 2s+1 (see split8), i.e. element j of the operand is
// feature (2s + (j>>2))*16 + 4q + (j&3) -- the same permutation on the A side falls out of the chunk addressing.
#pragma once
#include <cstdint>

namespace m3g {

__host__ __device__ inline int dual_swz(int row) { return (4 * ((row >> 1) & 3)) ^ (2 * ((row >> 3) & 1)); }
__host__ __device__ inline int dual_unit_byte(int row, int chunk) { return row * 128 + ((chunk ^ dual_swz(row)) << 3); }
inline size_t dual_image_floats(int rows) { return (size_t)rows * 64; }   // hi + lo = rows*256 bytes = rows*64 floats

}  // namespace m3g
