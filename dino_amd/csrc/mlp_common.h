// Constants and inline-asm helpers of the fused MLP kernel (mlp_fused2.hip; round 3 had a second build on them): the packed
// weight format and the inline-asm helpers of the explicit MFMA-gap schedules.
#pragma once
#include <type_traits>
#include <utility>

#include "attn_common.h"
#include "kernels.h"

namespace dseg {

namespace mfc {
constexpr int D = 384, F = 1536, HT = 32, NT = F / HT, NKS = D / 16, NDB = D / 32;
constexpr int W_TILE = NKS * 1024;                 // bytes of one matrix's fragments of one hidden tile (24 KiB, both matrices)
constexpr int TILE_BYTES = 2 * W_TILE;             // packed copy: per tile the 24 W1 fragments, then the 24 W2 fragments
static_assert(NDB * 2 == NKS, "fragment counts");
}  // namespace mfc

// four 1-KiB LDS-DMA pieces: global sbase + voff + {0, 1, 2, 3} KiB -> LDS lds_dst + {0, 1, 2, 3} KiB (+ 16 * lane).  The
// instruction's immediate offset applies to the global AND the LDS address, so M0 is set once.
__device__ __forceinline__ void mf_dma4(uint32_t voff, uint64_t sbase, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

template <int... I, class Fn>
__device__ __forceinline__ void mf_for(std::integer_sequence<int, I...>, Fn&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// one 1-KiB LDS-DMA piece (the instruction's immediate offset applies to the global AND the LDS address)
template <int OFF>
__device__ __forceinline__ void mf_dma1(uint32_t voff, uint64_t sbase, uint32_t lds_dst) {
    // (M0 is not saved: nothing else in the tile loop uses it -- no LDS-DMA builtin, no movrel, no GWS -- and every statement that
    //  needs it sets it)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(OFF)
                 : "memory");
}
// fragment read / counted wait (the compiler must neither count nor move these: it would drain the read-ahead)
template <int OFF>
__device__ __forceinline__ void mf_rd(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void mf_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);      // (an MFMA is not a memory operation: only this keeps it behind the wait)
}

}  // namespace dseg
