// Shared pieces of the attention kernels (attention.hip, attention64.hip, attention_bwd.hip).
#pragma once
#include "common.h"

namespace dseg {
namespace attn {

constexpr int KV_TILE_BYTES = 64 * 128;   // one [64 rows][64 bf16] slab = 8 KiB (128-byte rows)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

// K/V (and Q~/dO) tile swizzle: chunk ^ (bit1(row) << 2 | (row >> 2) & 3).  Like common.h's swizzle it gives every
// 16-row ds_read_b128 lane group 16 distinct 16-byte slots (it is a bit permutation of (row>>1)&7), and in addition
// the 4 rows of a transposing ds_read_b64_tr_b16 block fall on 4 distinct 64-byte quarters of the bank row (rows r
// and r+2 differ in chunk bit 2), so row reads and transposed reads of one image are both conflict-free
// (SQ_LDS_BANK_CONFLICT = 0 measured).
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)); }
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + (swz(row, chunk) << 4); }

// swap bits 2 and 3: rows of the A operand are read through this permutation so that accumulator registers
// 8s..8s+7 of a lane half h hold rows 16s+8h+0..7 in natural order = the B-operand fragment of the next product
__device__ __forceinline__ int sigma23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// A-operand fragment (8 rows of one column) from a row-major LDS image: two transposing reads of 4 rows each
__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
    const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)p0);
    const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)p1);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

}  // namespace attn
}  // namespace dseg
