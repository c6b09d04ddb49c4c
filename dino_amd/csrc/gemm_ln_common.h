// Shared pieces of the LayerNorm-fused GEMM kernels (gemm_ln.hip, gemm_ln12.hip).
#pragma once
#include "common.h"

namespace dseg {
namespace aln {

// [rows][32 k] bf16 slabs = 64-byte rows (4 chunks of 16 B): XOR the chunk with (row >> 2) & 3 -> the 16 rows of a ds_read_b128
// lane group fall on 16 distinct 16-byte slots of the 256-byte bank row
__device__ __forceinline__ int off64(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, moved);
}
// sum over the 16 lanes of a DPP row, result in all 16
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);      // row_half_mirror: the other quad of each 8
    v = dpp_add<0x140>(v);      // row_mirror: the other 8
    return v;
}

}  // namespace aln

// gemm_ln12.hip: the bf16 (one plane) configuration -- 12 waves, 128 x 384 tile
constexpr int LN12_BN = 384;
long gemm_ln12_slab_elems(int N, int K);
int launch_pack_slabs12(const float* src, int N, int K, bf16_t* dst, hipStream_t s, int fmt = 0);
struct LnGemmParams;
int launch_gemm_ln12(const LnGemmParams& p, hipStream_t s);

}  // namespace dseg
