// Shared by the row-stationary streaming kernels (gemm_rs.hip): the weight stream through a three-slot LDS ring and the step that
// multiplies one slot (see gemm_rs.hip's header comment).
#pragma once
#include <type_traits>

#include "mlp_common.h"

namespace dseg {

namespace rs {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // (a native vector: HIP's uint4 is a struct, which an asm register operand cannot be)
constexpr int NW = 4, BM = NW * 32, THREADS = NW * 64;
constexpr int RING = 3;
#ifndef RS_RA
#define RS_RA 4
#endif
constexpr int RA = RS_RA, NFR = RA + 1;      // fragment read-ahead; fragment registers
}  // namespace rs

#ifndef RS_ABL
#define RS_ABL 0      // timing ablations (wrong results): 1 no epilogue math, 2 no W DMA, 4 no MFMAs, 16 no fragment reads, 32 no global loads, 64 no global stores
#endif

// ---- shared: one step = NG MFMAs on the NG fragments of the slot at ring position rpos; this wave's NG / 4 pieces of the slot two steps
// ahead go out in the odd gaps; mma(gap tag, fragment) / valu(gap tag) as in mlp_fused3.hip
template <int NG, int VM, class Stream, class Pre, class Mma, class Valu>
__device__ __forceinline__ void rs_step(Stream& st, uint32_t frag_rd, uint32_t lane16, Pre&& pre, Mma&& mma, Valu&& valu) {
    using namespace rs;
    constexpr int SLOTB = NG * 1024, PW = NG / NW;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");      // this slot's pieces (issued two steps ago) have landed ...
    __builtin_amdgcn_s_barrier();                                   // ... every wave's; and every wave has left the slot refilled below
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t a = frag_rd + (uint32_t)st.rpos * SLOTB;
    st.rpos = st.rpos + 1 == RING ? 0 : st.rpos + 1;
    bf16x8 fr[NFR];
    auto issue_read = [&](auto j_tag) __attribute__((always_inline)) {
        constexpr int J = decltype(j_tag)::value;
        if (RS_ABL & 16) return;
        mf_rd<J * 1024>(fr[J % NFR], a);
    };
    // the first fragment reads go out before the step's scalar work (the LDS-DMA descriptors of the slot two steps ahead) and before pre():
    // their latency is the longest thing between the barrier and the first product
    mf_for(std::make_integer_sequence<int, RA>{}, issue_read);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NGRP = (PW + 3) / 4;
    uint64_t gsb[NGRP];
    uint32_t gld[NGRP];
    {
        const uint64_t sb = st.wp + (uint64_t)st.sn * SLOTB + st.piece0;
        const uint32_t ld = st.lds_base + (uint32_t)st.ipos * SLOTB + st.piece0;
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const uint64_t v = sb + g * 4096;
            gsb[g] = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v) |
                     ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
            gld[g] = __builtin_amdgcn_readfirstlane(ld + g * 4096);
        }
        st.sn = st.sn + 1 == st.nslots ? 0 : st.sn + 1;
        st.ipos = st.ipos + 1 == RING ? 0 : st.ipos + 1;
    }
    pre();                                                          // (the step's own set-up, e.g. the accumulator's bias: under the reads' latency)
    mf_for(std::make_integer_sequence<int, NG>{}, [&](auto j_tag) __attribute__((always_inline)) {
        constexpr int J = decltype(j_tag)::value;
        if constexpr (J + RA < NG) issue_read(std::integral_constant<int, J + RA>{});
        if (RS_ABL & 16) asm volatile("" : "=v"(fr[J % NFR]));
        else mf_wait<(NG - 1 - J < RA ? NG - 1 - J : RA)>();
        mma(j_tag, fr[J % NFR]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((J & 1) == 1 && J < 2 * PW) {
            constexpr int Q = J >> 1;
            if (!(RS_ABL & 2)) mf_dma1<(Q & 3) * 1024>(lane16, gsb[Q >> 2], gld[Q >> 2]);
        }
        valu(j_tag);
        __builtin_amdgcn_sched_barrier(0);
    });
}

struct RsStream {
    uint64_t wp;
    uint32_t lds_base, piece0;
    int sn, ipos, rpos, nslots;
};

template <int NG>
__device__ __forceinline__ void rs_prologue(RsStream& st, uint32_t lane16) {
    using namespace rs;
    constexpr int SLOTB = NG * 1024, PW = NG / NW;
#pragma unroll
    for (int k = 0; k < 2; ++k) {      // what the two steps before the first one would have issued
        const uint64_t sb = st.wp + (uint64_t)st.sn * SLOTB + st.piece0;
        const uint32_t ld = __builtin_amdgcn_readfirstlane(st.lds_base + (uint32_t)st.ipos * SLOTB + st.piece0);
        const uint64_t sbu = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb) |
                             ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32);
        if (!(RS_ABL & 2)) {
#pragma unroll
            for (int g = 0; g < PW / 4; ++g) mf_dma4(lane16, sbu + g * 4096, ld + g * 4096);
            if constexpr (PW % 4 != 0) {      // (the odd pieces of a slot that is not a multiple of 16 KiB)
                constexpr int G0 = PW / 4;
                mf_for(std::make_integer_sequence<int, PW % 4>{}, [&](auto q_tag) __attribute__((always_inline)) {
                    mf_dma1<decltype(q_tag)::value * 1024>(lane16, sbu + G0 * 4096, ld + G0 * 4096);
                });
            }
        }
        st.sn = st.sn + 1 == st.nslots ? 0 : st.sn + 1;
        st.ipos = st.ipos + 1 == RING ? 0 : st.ipos + 1;
    }
}

}  // namespace dseg
