// Dual-use LDS weight image for v_mfma_f32_16x16x32_bf16 chains: ONE copy of W serves as the A operand of
// Y^T = W X^T (row reads, ds_read_b64) and of dX^T = W^T dY^T (ds_read_b64_tr_b16, gfx950's transposing LDS read),
// so a reverse kernel that recomputes the forward needs no second, transposed copy of each weight matrix.
//
// W is [ROWS][64] (64 input features = 16 chunks of 4).  A k-step s of a chain consumes accumulator blocks 2s and 2s+1,
// so lane quarter q needs chunk (2s)*4+q followed by chunk (2s+1)*4+q: chunks are therefore split by h = bit 2 of the
// chunk index into two half-rows of 64 bytes, each holding the 8 chunks cp = s*4+q of its h, and stored per group of
// 8 rows as [8 half-rows h=0 (512 B) | 8 half-rows h=1 (512 B)].  The image is [hi part | lo part], hi = bf16(W),
// lo = bf16(W - hi), ROWS*128 bytes each, and the 8-byte unit (row, chunk) lives at
//     part + (row>>3)*1024 + h*512 + (row&7)*64 + ((cp ^ swz(row)) << 3).
// The two halves of a row-read operand are exactly 512 bytes apart -- the closest pair of offsets from one base -- so
// the compiler fuses THEM into one ds_read2st64_b64 that fills the operand's four registers in order (with both halves
// in one 128-byte row it paired reads of DIFFERENT row blocks and needed four v_mov per operand to sort them out: 14 % of
// this kernel's VALU instructions).
// Bank check (MI355X_MICROARCH.md, LDS):
//   row reads   the 16 lanes (m) of a quarter read rows ob*16+m, same cp: bank pair (row&1)*8 + (cp ^ swz) -- swz is a
//               bijection of (row>>1)&7 onto 3 bits, so the 16 lanes cover the 16 bank pairs (ds_read2: (a/4) mod 32);
//   transposed  ds_read_b64_tr_b16, 32-lane groups, lane 4q'+p of quarter q reads row sh*16+4q+q', chunk 4ob+p: rows of
//               one quarter differ in row&3 (bank pairs (row&3)*8 + ..), the two quarters of a group differ in bit 2 of
//               swz, p in bits 0-1 -> 32 distinct bank pairs of the 64 banks.
#pragma once
#include <cstdint>

namespace m3g {

// swz: bit 0 <- row bit 1, bit 2 <- row bit 2, bit 1 <- row bit 3
__host__ __device__ inline int dual_swz(int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2) | (((row >> 3) & 1) << 1); }
// byte offset of unit (row, chunk) inside one part (hi or lo) of a ROWS-row image
__host__ __device__ inline int dual_unit_byte(int rows, int row, int chunk) {
  const int h = (chunk >> 2) & 1, cp = ((chunk >> 3) << 2) | (chunk & 3);
  (void)rows;
  return (row >> 3) * 1024 + h * 512 + (row & 7) * 64 + ((cp ^ dual_swz(row)) << 3);
}
inline size_t dual_image_floats(int rows) { return (size_t)rows * 64; }   // hi + lo = rows*256 bytes = rows*64 floats

}  // namespace m3g
