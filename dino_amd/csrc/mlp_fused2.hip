// Fused transformer MLP, bf16 mode, ViT-S width -- role-split wave pairs (two waves per SIMD).
//     x += fc2(gelu(fc1(LayerNorm2(x))))      (vision_transformer.py:135 -> :59-65)
//
// One launch per block instead of LN+fc1+GELU and fc2+residual: the [M, 1536] hidden activation never exists in HBM (it was 31 % of a
// layer's bytes), and the residual stream is read once and written once (it is the accumulators' initial value).  Skeleton = the
// attention kernel's (attention_z.hip): a W1 tile of 32 hidden units plays K, the matching W2^T tile plays V, GELU plays exp, and
// the fc1 accumulator IS the B operand of the fc2 product.  Per 32-row block, resident in registers for a whole 128-row item:
//   xn[24]  LayerNorm2(x) as B-operand fragments (k = 16 s + 8 h + j on lane (row, h)): 96 registers
//   o[12]   out^T[384][32] accumulators, initialised with x + b2 (the residual): 192 registers
// per hidden tile t (48 of them): S^T[32 hid][32 rows] = W1_t . xn^T + b1_t (24 MFMAs), P = gelu(S) (16 values per lane),
// o^T += W2^T_t . P^T (24 MFMAs).  Weights: both matrices are re-packed once (launch_pack_mlp, below) in exactly the order the MFMAs
// consume them -- [tile][48 fragments][64 lanes][8], W1 rows and W2 output rows permuted by sigma23 (bits 2 <-> 3) so that accumulator
// registers 8s..8s+7 of a lane hold 8 consecutive k of the next product -- a fragment is one linear 1-KiB LDS-DMA piece and one
// conflict-free ds_read_b128.  Every tile step is written out as explicit MFMA "gaps" (template-unrolled): the compiler's own order,
// with or without sched_group_barrier, left the matrix pipe idle ~2/3 of the time.
//
// Round 3's first build of this (one wave per SIMD, 4 waves per workgroup with the whole 512-register file each; removed in round 4,
// `git log -- dino_amd/csrc/mlp_fused.hip`) showed (profiles/r03_mlp_ablation.md): with ONE wave per SIMD everything the wave issues
// besides its MFMAs -- 12 LDS-DMA pieces, 48 fragment reads, ~170 vector instructions of GELU per tile -- is serial with them (in-order
// issue: 55 cycles per MFMA gap instead of 32; the bare MFMA stream alone runs at the pipe's rate): 340-375 us per layer against this
// kernel's 305-310.  Here the work of a 32-row block is split between TWO waves that share a SIMD, so one wave's memory / vector
// issue overlaps the other's matrix work:
//   wave A ("fc1")  holds xn = LayerNorm2(x) of the 32 rows (96 registers).  Tile step s: S(s+1)^T = W1_{s+1} . xn^T + b1
//                   (24 MFMAs, fragments from the W1 ring) while the GELU of S(s) runs on the vector unit; P(s) = bf16(gelu(S(s)))
//                   goes to a 2-KiB LDS buffer in B-operand fragment order (two ds_write_b128 per lane).
//   wave B ("fc2")  holds the out^T[384][32] accumulators (192 registers), initialised with x + b2.  Tile step s: o^T += W2^T_{s-1} .
//                   P(s-1)^T (24 MFMAs, fragments from the W2 ring, P from the LDS buffer its partner filled one step earlier) and
//                   issues the workgroup's weight stream: 12 LDS-DMA pieces per step (W2(s), then W1(s+3)).
// A workgroup is four such pairs = 8 waves = 128 rows; one barrier per tile step (49 per item: wave B runs one step behind).  Wave A
// does the next item's LayerNorm prologue while wave B finishes the last tile and stores the rows: the item boundary overlaps too.
// LDS: W1 ring 3 x 24 KiB, W2 ring 2 x 24 KiB (read one step after it was issued, its pieces go first in the step), P 4 pairs x 2
// x 2 KiB, constants 12 KiB = 148.5 KiB.
//
// PROJ = true: the attention output projection of the same block runs in the same launch, ahead of the MLP:
//     x += proj(ctx) + b_proj;   x += fc2(gelu(fc1(LayerNorm2(x))))      (vision_transformer.py:123 -> :104-105, then :135)
// x_mid^T = x^T + Wproj . ctx^T has the shape of the fc2 product (A = weight fragments with the sigma23 row order, so that the
// accumulator registers 8 s2 .. 8 s2 + 7 of a lane are 8 consecutive features = the row layout LayerNorm and the stores use), with
// the ctx rows as the B operand.  Twelve projection steps (one per 32 input features) run ahead of the 48 tile steps in wave B, whose
// accumulators start from x + b_proj.  Wave B then normalises the rows from its registers (LayerNorm2: the statistics are sums
// over the lane's 192 accumulator registers and one cross-half shuffle) and hands xn to wave A as finished B-operand fragments:
// three rounds of eight fragments through an 8-KiB LDS window per pair.  Wave A reads no rows at all (the MLP-only kernel reads
// them twice, once per wave).  While the projection runs, the W1 ring carries the Wproj k-tiles (24 KiB each, prefetched two steps
// ahead; tiles 0..2 of the NEXT item take the place of the W1(0..2) prefetch at steps 46..48) and the W2 ring six 8-KiB slots of
// ctx k-tiles ([128 rows][32] bf16, chunk-swizzled 64-byte rows); W1(0..2) and W2(0) are fetched while wave B normalises.  b2 is
// added with the final stores.  (First version: both waves accumulated the projection and wave A normalised its own copy --
// 390 us per launch at 32 frames against 313 + 88 for the two kernels it replaces; this one: see profiles/r03_mlp_ablation.md.)  The
// separate proj launch (97 us at 32 frames: 442 MB of HBM traffic, of which 354 MB are the residual stream's round trip) is gone.
#include <stdio.h>

#include <vector>

#include "mlp_common.h"

namespace dseg {

namespace mf2 {
using namespace mfc;
constexpr int NP = 4, BM = NP * 32, THREADS = 2 * NP * 64;      // pairs per workgroup; rows; threads
constexpr int W1_OFF = 0, W2_OFF = 3 * W_TILE;
constexpr int P_OFF = W2_OFF + 2 * W_TILE;          // [pair][slot][fragment][64 lanes][16 B]
constexpr int B1_OFF = P_OFF + NP * 2 * 2048;       // b1 [F] fp32
constexpr int B2_OFF = B1_OFF + F * 4;              // b2, gamma, beta [D] fp32 each
constexpr int G_OFF = B2_OFF + D * 4, BE_OFF = G_OFF + D * 4;
constexpr int BP_OFF = BE_OFF + D * 4;              // PROJ: b2 (B2_OFF then holds b_proj, the initial value's bias)
constexpr int G1_OFF = BP_OFF + D * 4, BE1_OFF = G1_OFF + D * 4;      // QKV: norm1 weight / bias of the NEXT block
constexpr int BQ_OFF = BE1_OFF + D * 4;             // QKV: its qkv bias [1152]
constexpr int NQT = 3 * D / 32;                     // QKV: output tiles of 32 features (36)
constexpr int LDS_BYTES = BQ_OFF + 3 * D * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int NPT = D / 32;                         // PROJ: projection k-tiles
constexpr int CTX_OFF = W2_OFF, CTX_SLOT = BM * 64; // PROJ: six slots of [128 rows][32 k] bf16 in the (idle) W2 ring
static_assert(6 * CTX_SLOT == 2 * W_TILE, "ctx ring = W2 ring");
constexpr int H_OFF = W2_OFF + W_TILE;              // PROJ: xn hand-off windows, 8 KiB per pair (W2 slot 1 + the first half of P: idle then)
static_assert(H_OFF + NP * 8192 <= B1_OFF, "hand-off windows");
#ifndef MF2_RA
#define MF2_RA 4
#endif
constexpr int RA = MF2_RA, NFR = RA + 1;            // fragment read-ahead (gaps); fragment registers
}  // namespace mf2

#ifndef MF2_ABL
#define MF2_ABL 0      // timing ablations (wrong results): 1 no GELU, 2 no W DMA, 4 no fc1 MFMAs, 8 no fc2 MFMAs, 16 no fragment reads,
                       // 32 no row loads / stores, 64 no LDS-DMA in the qkv tail, 128 no MFMAs in the qkv tail
#endif

#ifndef MF2_NT
#define MF2_NT 0       // experiment (measured, off): 1 = the residual rows are loaded, 2 = stored with the non-temporal hint.  Both SLOWER here:
                       // 4.38 -> 4.74 / 4.59 / 5.30 ms per 32-frame step for 1 / 2 / 3 (profiles/r04_nt_experiments.md) -- the rows this
                       // launch stores are the rows the next launch (LayerNorm1 + qkv) loads: they should stay cached
#endif
template <class T>
__device__ __forceinline__ T mf2_ld(const T* p) {
    if (MF2_NT & 1) return __builtin_nontemporal_load(p);
    return *p;
}
template <class T>
__device__ __forceinline__ void mf2_st(T* p, const T& v) {
    if (MF2_NT & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}

#ifndef MF2_STAMP
#define MF2_STAMP 0      // diagnostic build only: lane 0 of waves 0 (A) and 4 (B) stamps s_memrealtime / s_memtime into p.queue
#endif
#define MF2_ST(role, item_k, idx)                                                                                         \
    do {                                                                                                                  \
        if (MF2_STAMP && (tid & 63) == 0 && (item_k) < 4) {                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            unsigned long long* d = reinterpret_cast<unsigned long long*>(p.queue) +                                      \
                                    ((((long)blockIdx.x * 2 + (role)) * 4 + (item_k)) * 16 + (idx)) * 2;                  \
            d[0] = __builtin_amdgcn_s_memrealtime();                                                                      \
            d[1] = __builtin_amdgcn_s_memtime();                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
        }                                                                                                                 \
    } while (0)

// One projection k-tile for one wave: acc^T[384][32] += Wproj(kt) . ctx(kt)^T -- the 24 MFMA gaps of an fc2 step, the weight
// fragments from a_w (W1-ring slot), the two ctx fragments from ap0 / ap1; piece(j), j = 0..11, is called in every other gap
// (wave B issues its LDS-DMA pieces there, wave A nothing).
template <int FMT, class PieceFn>
__device__ __forceinline__ void mf2_proj_tile(f32x16 (&acc)[mfc::NDB], uint32_t a_w, uint32_t ap0, uint32_t ap1, PieceFn&& piece) {
    using namespace mf2;
    bf16x8 p0, p1, fr[NFR];
    mf_rd<0>(p0, ap0);
    mf_rd<0>(p1, ap1);
    auto issue_read = [&](auto g_tag) __attribute__((always_inline)) {
        constexpr int G = decltype(g_tag)::value;
        if constexpr (G < 12) mf_rd<(2 * G) * 1024>(fr[G % NFR], a_w);
        else mf_rd<(2 * (G - 12) + 1) * 1024>(fr[G % NFR], a_w);
    };
    mf_for(std::make_integer_sequence<int, RA>{}, issue_read);
    mf_for(std::make_integer_sequence<int, 24>{}, [&](auto g_tag) __attribute__((always_inline)) {
        constexpr int G = decltype(g_tag)::value;
        if constexpr (G + RA < 24) issue_read(std::integral_constant<int, G + RA>{});
        mf_wait<(23 - G < RA ? 23 - G : RA)>();
        acc[G % 12] = mfma32f<FMT>(fr[G % NFR], G < 12 ? p0 : p1, acc[G % 12]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((G & 1) == 0) piece(std::integral_constant<int, G / 2>{});
        __builtin_amdgcn_sched_barrier(0);
    });
}

template <bool PROJ, bool QKV, int FMT>
__global__ __launch_bounds__(mf2::THREADS, 2) void mlp_fused2_kernel(MlpFusedParams p) {
    static_assert(PROJ || !QKV, "the qkv tail needs the hand-off machinery of the projection build");
    using namespace mf2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & 3;
    const bool is_b = wave >= 4;                  // wave-uniform role
    const int M = p.M;
    const int nitems = (M + BM - 1) / BM;
    if ((int)blockIdx.x >= nitems) return;

    // ---- constants into LDS: b1, b2, gamma, beta
    {
        float* const sB1w = reinterpret_cast<float*>(smem + B1_OFF);
        for (int i = tid; i < F / 4; i += THREADS) reinterpret_cast<f32x4*>(sB1w)[i] = reinterpret_cast<const f32x4*>(p.b1)[i];
        for (int i = tid; i < 3 * D / 4; i += THREADS) {
            const int which = i / (D / 4), j = i - which * (D / 4);
            const float* src = which == 0 ? p.b2 : (which == 1 ? p.gamma : p.beta);
            if (PROJ && which == 0) src = p.bproj;
            reinterpret_cast<f32x4*>(smem + B2_OFF + which * D * 4)[j] = reinterpret_cast<const f32x4*>(src)[j];
        }
        if (PROJ)
            for (int i = tid; i < D / 4; i += THREADS) reinterpret_cast<f32x4*>(smem + BP_OFF)[i] = reinterpret_cast<const f32x4*>(p.b2)[i];
        if (QKV) {
            for (int i = tid; i < 2 * D / 4; i += THREADS)
                reinterpret_cast<f32x4*>(smem + G1_OFF)[i] = reinterpret_cast<const f32x4*>(i < D / 4 ? p.gamma1 : p.beta1)[i < D / 4 ? i : i - D / 4];
            for (int i = tid; i < 3 * D / 4; i += THREADS) reinterpret_cast<f32x4*>(smem + BQ_OFF)[i] = reinterpret_cast<const f32x4*>(p.bqkv)[i];
        }
    }
    // Free desynchronisation (p.sleep_max > 0): with nitems = q * grid + r the workgroups >= r walk one item fewer than the others, so
    // they may start up to one item late without lengthening the launch -- spread over that window they leave the lock step in which
    // every CU stores / loads its rows at the same moment (the HBM bursts of the item boundaries)
    if (p.sleep_max > 0 && (int)blockIdx.x >= p.n_long) {
        const int n_short = (int)gridDim.x - p.n_long;
        const int reps = (int)(((long)p.sleep_max * ((int)blockIdx.x - p.n_long + 1)) / n_short);
        for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    const float* const sB1 = reinterpret_cast<const float*>(smem + B1_OFF);
    const float* const sB2 = reinterpret_cast<const float*>(smem + B2_OFF);
    const float* const sG = reinterpret_cast<const float*>(smem + G_OFF);
    const float* const sBe = reinterpret_cast<const float*>(smem + BE_OFF);
    auto uniform64 = [](uint64_t v) __attribute__((always_inline)) -> uint64_t {
        return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v) |
               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
    };
    const uint64_t wp = reinterpret_cast<uint64_t>(p.Wp);
    const uint64_t wpj = reinterpret_cast<uint64_t>(p.Wproj), ctxb = reinterpret_cast<uint64_t>(p.ctx);
    const uint64_t wq = reinterpret_cast<uint64_t>(p.Wqkv);
    // PROJ: a lane's two ctx fragments of a k-tile (row = pair * 32 + lr of the item, 16-byte chunk (2 s2 + lh) ^ ((row >> 2) & 3))
    auto ctx_frag_addr = [&](uint32_t lr_, uint32_t lh_) __attribute__((always_inline)) -> uint32_t {
        const uint32_t rin = (uint32_t)pair * 32 + lr_;
        return lds_base + CTX_OFF + rin * 64 + ((lh_ ^ ((rin >> 2) & 3)) << 4);      // s2 = 1: ^ 32
    };

    if (is_b) {
        // ================================================================================================= wave B: fc2 + weight stream
        // this wave's share of a tile's stream: fragments 6 pair .. 6 pair + 5 of each matrix
        const uint32_t frag0 = (uint32_t)pair * 6 * 1024;
        {
            // ring prologue = what the last steps of a previous item would have issued: W1(0), W1(1), W1(2)
            const uint32_t lane16 = (uint32_t)(tid & 63) * 16;
            if (!(MF2_ABL & 2))
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const uint64_t sb = uniform64(PROJ ? wpj + (uint32_t)t * W_TILE + frag0 : wp + (uint32_t)t * TILE_BYTES + frag0);
                    const uint32_t ld = __builtin_amdgcn_readfirstlane(lds_base + W1_OFF + t * W_TILE + frag0);
                    mf_dma4(lane16, sb, ld);
                    mf_dma1<0>(lane16, sb + 4096, ld + 4096);
                    mf_dma1<1024>(lane16, sb + 4096, ld + 4096);
                }
            if (PROJ && !(MF2_ABL & 2)) {      // ctx k-tiles 0..2 of the first item
                const uint32_t l = (uint32_t)(tid & 63);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint32_t rin = (uint32_t)(2 * pair + j) * 16 + (l >> 2);
                    int rg = (int)blockIdx.x * BM + (int)rin;
                    rg = rg < M ? rg : M - 1;
                    const uint32_t cv = (uint32_t)rg * (D * 2) + (((l & 3) ^ ((rin >> 2) & 3)) << 4);
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        mf_dma1<0>(cv, uniform64(ctxb + (uint32_t)t * 64),
                                   __builtin_amdgcn_readfirstlane(lds_base + CTX_OFF + t * CTX_SLOT + (2 * pair + j) * 1024));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();      // constants staged; W1(0..2) visible

        f32x16 o[NDB];
        int item_k = 0;
        for (int item = blockIdx.x; item < nitems; item += gridDim.x, ++item_k) {
            if (wave == 4) MF2_ST(1, item_k, 0);
            // per-lane constants from an opaque lane id, once per item (values that live across the item loop get spilled, and a
            // scratch reload inside the step loop drains the weight ring: round 3)
            uint32_t zero = 0;
            asm volatile("" : "+v"(zero));
            const uint32_t lane_i = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
            const uint32_t lane16_i = lane_i * 16, lr_i = lane_i & 31, lh_i = lane_i >> 5;
            const uint32_t frag_rd_i = lds_base + lane16_i;
            const uint32_t p_rd_i = lds_base + P_OFF + (uint32_t)pair * 4096 + lane16_i;
            const int row = item * BM + pair * 32 + (int)lr_i;
            const int row_c = row < M ? row : M - 1;
            float* const xrow = p.X + (long)row_c * D + lh_i * 8;
            // the residual rows of an item, straight into the accumulators (+ the fc2 bias): issued BEFORE the item's first
            // barrier -- for the first item at the kernel's start, for the others right behind the previous item's stores -- so
            // that they run beside wave A's LayerNorm prologue instead of after it
            auto load_rows = [&](const float* xr) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < NKS; ++k) {
                    f32x4 a = {1.f, 2.f, 3.f, (float)k}, b = a;
                    if (!(MF2_ABL & 32)) {
                        a = mf2_ld(reinterpret_cast<const f32x4*>(xr + k * 16));
                        b = mf2_ld(reinterpret_cast<const f32x4*>(xr + k * 16 + 4));
                    }
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(sB2 + k * 16 + lh_i * 8);
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(sB2 + k * 16 + lh_i * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[k >> 1][(k & 1) * 8 + e] = a[e] + c0[e];
                        o[k >> 1][(k & 1) * 8 + 4 + e] = b[e] + c1[e];
                    }
                }
            };
            auto load_rows_k = [&](const float* xr, auto k_tag) __attribute__((always_inline)) {
                constexpr int k = decltype(k_tag)::value;
                const f32x4 a = mf2_ld(reinterpret_cast<const f32x4*>(xr + k * 16)), b = mf2_ld(reinterpret_cast<const f32x4*>(xr + k * 16 + 4));
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(sB2 + k * 16 + lh_i * 8);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(sB2 + k * 16 + lh_i * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[k >> 1][(k & 1) * 8 + e] = a[e] + c0[e];
                    o[k >> 1][(k & 1) * 8 + 4 + e] = b[e] + c1[e];
                }
            };
            if (item == (int)blockIdx.x) load_rows(xrow);
            if (wave == 4) MF2_ST(1, item_k, 1);

            // the stream pieces of step s: W2(s) (s <= 47) first, then W1 of virtual tile s + 3 (48: none; 49.. = the next item's 0..)
            auto step_b = [&](int s, auto kind_tag) __attribute__((always_inline)) {
                constexpr int KIND = decltype(kind_tag)::value;      // 0: step 0 (row loads, no fc2); 1: steps 1..48
                __builtin_amdgcn_s_barrier();
                const int v1 = s + 3;
                // (PROJ: W2(0) was fetched behind the projection; the tiles past the item's end are the next item's Wproj k-tiles 0..2)
                const bool has_w2 = PROJ ? (s >= 1 && s <= NT - 1) : s <= NT - 1, has_w1 = v1 != NT;
                const int t1 = v1 < NT ? v1 : v1 - (NT + 1);
                const bool nxt = PROJ && v1 > NT;
                uint64_t gsb[4];
                uint32_t gld[4];
                {
                    const uint32_t so2 = (uint32_t)s * TILE_BYTES + W_TILE + frag0;
                    const uint32_t so1 = (nxt ? (uint32_t)t1 * W_TILE : (uint32_t)t1 * TILE_BYTES) + frag0;
                    const uint32_t d2 = lds_base + W2_OFF + (uint32_t)(s & 1) * W_TILE + frag0;
                    const uint32_t d1 = lds_base + W1_OFF + (uint32_t)(t1 % 3) * W_TILE + frag0;
                    const uint64_t w1src = nxt ? (QKV ? wq : wpj) : wp;      // (QKV: the qkv tail's first tiles; Wproj follows the tail)
                    gsb[0] = uniform64(wp + so2);
                    gsb[1] = uniform64(wp + so2 + 4096);
                    gsb[2] = uniform64(w1src + so1);
                    gsb[3] = uniform64(w1src + so1 + 4096);
                    gld[0] = __builtin_amdgcn_readfirstlane(d2);
                    gld[1] = __builtin_amdgcn_readfirstlane(d2 + 4096);
                    gld[2] = __builtin_amdgcn_readfirstlane(d1);
                    gld[3] = __builtin_amdgcn_readfirstlane(d1 + 4096);
                }
                // PROJ, step 48 (no W2 tile): the six free piece slots carry the next item's ctx k-tiles 0..2 (W2 slot 0 is idle: W2(46)
                // was read in step 47)
                const bool ctx_pre = PROJ && s == NT && item + (int)gridDim.x < nitems;
                uint32_t cvn[2] = {0u, 0u};
                if (ctx_pre) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint32_t rin = (uint32_t)(2 * pair + j) * 16 + (lane_i >> 2);
                        int rg = (item + (int)gridDim.x) * BM + (int)rin;
                        rg = rg < M ? rg : M - 1;
                        cvn[j] = (uint32_t)rg * (D * 2) + (((lane_i & 3) ^ ((rin >> 2) & 3)) << 4);
                    }
                }
                auto piece = [&](auto j_tag) __attribute__((always_inline)) {
                    constexpr int J = decltype(j_tag)::value;      // 0..5: W2, 6..11: W1
                    if (MF2_ABL & 2) return;
                    if constexpr (PROJ && J < 6) {
                        if (ctx_pre) {
                            constexpr int T = J >> 1, JJ = J & 1;
                            mf_dma1<0>(cvn[JJ], uniform64(ctxb + T * 64),
                                       __builtin_amdgcn_readfirstlane(lds_base + CTX_OFF + T * CTX_SLOT + (uint32_t)(2 * pair + JJ) * 1024));
                        }
                    }
                    if (J < 6 ? !has_w2 : !has_w1) return;
                    constexpr int Q = J % 6, G4 = (J / 6) * 2 + (Q >= 4 ? 1 : 0), OFF = (Q & 3) * 1024;
                    mf_dma1<OFF>(lane16_i, gsb[G4], gld[G4]);
                };
                if constexpr (KIND == 0) {
                    mf_for(std::make_integer_sequence<int, 12>{}, piece);
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // W2(0) (and the row loads / stores issued before it)
                } else {
                    const uint32_t a2 = frag_rd_i + W2_OFF + (uint32_t)((s - 1) & 1) * W_TILE;
                    const uint32_t ap = p_rd_i + (uint32_t)((s - 1) & 1) * 2048;
                    bf16x8 p0, p1, fr[NFR];
                    mf_rd<0>(p0, ap);
                    mf_rd<1024>(p1, ap);
                    // fragment of gap G: all k-step-0 products (they take p0), then the k-step-1 products
                    auto issue_read = [&](auto g_tag) __attribute__((always_inline)) {
                        constexpr int G = decltype(g_tag)::value;
                        if (MF2_ABL & 16) return;
                        if constexpr (G < 12) mf_rd<(2 * G) * 1024>(fr[G % NFR], a2);
                        else mf_rd<(2 * (G - 12) + 1) * 1024>(fr[G % NFR], a2);
                    };
                    mf_for(std::make_integer_sequence<int, RA>{}, issue_read);
                    mf_for(std::make_integer_sequence<int, 24>{}, [&](auto g_tag) __attribute__((always_inline)) {
                        constexpr int G = decltype(g_tag)::value;
                        if constexpr (G + RA < 24) issue_read(std::integral_constant<int, G + RA>{});
                        if (MF2_ABL & 16) fr[G % NFR] = p0;
                        else mf_wait<(23 - G < RA ? 23 - G : RA)>();
                        if (!(MF2_ABL & 8)) o[G % 12] = mfma32f<FMT>(fr[G % NFR], G < 12 ? p0 : p1, o[G % 12]);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr ((G & 1) == 0) piece(std::integral_constant<int, G / 2>{});
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    // what the next step reads must have landed: W2(s) (this wave's pieces; the barrier publishes them) and the W1
                    // tile issued the step before; the six W1 pieces issued last may stay in flight.  Step 45 issues no W1 tile (its
                    // six youngest pieces are W2(45): wait for all); step 48 no W2 tile (the six W1(2') pieces stay in flight across
                    // the item boundary: behind wave A's row loads they take microseconds, and nobody reads them before step 1)
                    if (s == NT - 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                }
            };
            // ---- LayerNorm of the rows in o, from the registers (register 8 s2 + j of block db = feature 32 db + 16 s2 + 8 lh + j), handed to
            // wave A as B-operand fragments in three rounds of eight through the pair's 8-KiB window (4 barriers inside, the caller adds
            // the one that publishes round 2)
            auto ln_handoff = [&](const float* sGam, const float* sBet) __attribute__((always_inline)) {
                float sum = 0.f;
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        sum += ((o[db][8 * h] + o[db][8 * h + 1]) + (o[db][8 * h + 2] + o[db][8 * h + 3])) +
                               ((o[db][8 * h + 4] + o[db][8 * h + 5]) + (o[db][8 * h + 6] + o[db][8 * h + 7]));
                sum += __shfl_xor(sum, 32);
                const float mean = sum * (1.0f / D);
                float qv = 0.f;
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        __builtin_amdgcn_sched_barrier(0);      // (192 live accumulators: no room for 192 differences at once)
                        float part = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float dlt = o[db][4 * q4 + e] - mean;
                            part = fmaf(dlt, dlt, part);
                        }
                        qv += part;
                    }
                qv += __shfl_xor(qv, 32);
                const float rstd = 1.0f / sqrtf(qv * (1.0f / D) + p.eps);
                float mean_n = mean;      // (opaque copy: otherwise the 192 differences of the variance pass are kept -- in scratch)
                asm volatile("" : "+v"(mean_n));
                char* const hw = smem + H_OFF + pair * 8192 + lane16_i;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if (r > 0) __builtin_amdgcn_s_barrier();      // wave A has read the previous round
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        __builtin_amdgcn_sched_barrier(0);      // (one k-step at a time: the accumulators leave no room for hoisted loads)
                        const int k = r * 8 + kk;
                        const float* gp = sGam + k * 16 + lh_i * 8;
                        const float* bp = sBet + k * 16 + lh_i * 8;
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
                        const f32x4 e0 = *reinterpret_cast<const f32x4*>(bp), e1 = *reinterpret_cast<const f32x4*>(bp + 4);
                        float y[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            y[e] = (o[k >> 1][(k & 1) * 8 + e] - mean_n) * rstd * g0[e] + e0[e];
                            y[4 + e] = (o[k >> 1][(k & 1) * 8 + 4 + e] - mean_n) * rstd * g1[e] + e1[e];
                        }
                        uint4 u;
                        u.x = pack2<FMT>(y[0], y[1]);
                        u.y = pack2<FMT>(y[2], y[3]);
                        u.z = pack2<FMT>(y[4], y[5]);
                        u.w = pack2<FMT>(y[6], y[7]);
                        *reinterpret_cast<uint4*>(hw + kk * 1024) = u;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (r < 2) __builtin_amdgcn_s_barrier();      // round r is in LDS
                }
            };

            if constexpr (PROJ) {
                // ---- projection phase: o^T += Wproj . ctx^T, twelve k-tiles (o already holds x + b_proj)
                uint32_t cv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint32_t rin = (uint32_t)(2 * pair + j) * 16 + (lane_i >> 2);
                    int rg = item * BM + (int)rin;
                    rg = rg < M ? rg : M - 1;
                    cv[j] = (uint32_t)rg * (D * 2) + (((lane_i & 3) ^ ((rin >> 2) & 3)) << 4);
                }
                const uint32_t cp0 = ctx_frag_addr(lr_i, lh_i);
#pragma unroll 1
                for (int kt = 0; kt < NPT; ++kt) {
                    __builtin_amdgcn_s_barrier();
                    // k-tile kt + 2 into the slots k-tile kt - 1 has left (tiles 0..2 were issued at the previous item's end)
                    const bool issue = kt >= 1 && kt + 2 < NPT;
                    const int t = kt + 2;
                    const uint32_t so = (uint32_t)t * W_TILE + frag0;
                    const uint32_t d = lds_base + W1_OFF + (uint32_t)(t % 3) * W_TILE + frag0;
                    const uint32_t dc = lds_base + CTX_OFF + (uint32_t)(t % 6) * CTX_SLOT + (uint32_t)(2 * pair) * 1024;
                    const uint64_t gb0 = uniform64(wpj + so), gb1 = uniform64(wpj + so + 4096), cb = uniform64(ctxb + (uint32_t)t * 64);
                    const uint32_t gl0 = __builtin_amdgcn_readfirstlane(d), gl1 = __builtin_amdgcn_readfirstlane(d + 4096);
                    const uint32_t cl0 = __builtin_amdgcn_readfirstlane(dc), cl1 = __builtin_amdgcn_readfirstlane(dc + 1024);
                    auto piece = [&](auto j_tag) __attribute__((always_inline)) {
                        constexpr int J = decltype(j_tag)::value;      // 0..5: Wproj fragments, 6..7: ctx pieces
                        if ((MF2_ABL & 2) || !issue) return;
                        if constexpr (J < 4) mf_dma1<J * 1024>(lane16_i, gb0, gl0);
                        else if constexpr (J < 6) mf_dma1<(J - 4) * 1024>(lane16_i, gb1, gl1);
                        else if constexpr (J == 6) mf_dma1<0>(cv[0], cb, cl0);
                        else if constexpr (J == 7) mf_dma1<0>(cv[1], cb, cl1);
                    };
                    const uint32_t cs = (uint32_t)(kt % 6) * CTX_SLOT;
                    mf2_proj_tile<FMT>(o, frag_rd_i + W1_OFF + (uint32_t)(kt % 3) * W_TILE, cp0 + cs, (cp0 ^ 32u) + cs, piece);
                    // k-tile kt + 1 (issued one step ago) has landed; the eight pieces of this step may stay in flight
                    if (issue) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (wave == 4) MF2_ST(1, item_k, 9);
                __builtin_amdgcn_s_barrier();      // every wave is done with the projection tiles: both rings are free
                if (!(MF2_ABL & 2)) {
                    // the MLP's first tiles, in flight while the rows are normalised: W1(0..2) and W2(0)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint64_t sb = uniform64(wp + (t < 3 ? (uint32_t)t * TILE_BYTES : (uint32_t)W_TILE) + frag0);
                        const uint32_t ld = __builtin_amdgcn_readfirstlane(lds_base + (t < 3 ? W1_OFF + t * W_TILE : W2_OFF) + frag0);
                        mf_dma4(lane16_i, sb, ld);
                        mf_dma1<0>(lane16_i, sb + 4096, ld + 4096);
                        mf_dma1<1024>(lane16_i, sb + 4096, ld + 4096);
                    }
                }
                ln_handoff(sG, sBe);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (wave == 4) MF2_ST(1, item_k, 10);
                __builtin_amdgcn_s_barrier();      // round 2 in LDS; W1(0..2), W2(0) visible
                if (wave == 4) MF2_ST(1, item_k, 11);
            }
            step_b(0, std::integral_constant<int, 0>{});
            if (wave == 4) MF2_ST(1, item_k, 2);
#pragma unroll 1
            for (int s = 1; s <= NT; ++s) {
                step_b(s, std::integral_constant<int, 1>{});
                if (wave == 4 && s == 1) MF2_ST(1, item_k, 3);
                if (wave == 4 && s == 24) MF2_ST(1, item_k, 4);
                if (wave == 4 && s == 47) MF2_ST(1, item_k, 5);
            }
            if (wave == 4) MF2_ST(1, item_k, 6);

            if constexpr (QKV) {
                // ---- qkv tail: LayerNorm1 of the NEXT block on the finished rows, handed to wave A; wave A multiplies (36 tiles of 32
                // output features, its own LDS-DMA stream through the W1 ring: a wave that stores must not wait on a vmcnt it shares
                // with a DMA ring) and leaves each tile's bf16 block in the pair's P buffer; this wave writes the blocks to Q / K / V one
                // step later, its own row stores and the next item's row loads in between
                const bool has_next = item + (int)gridDim.x < nitems;
                {
                    const float* fp = reinterpret_cast<const float*>(smem + BP_OFF) + lh_i * 8;      // b2: o becomes the block's output
#pragma unroll
                    for (int k = 0; k < NKS; ++k) {
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(fp + k * 16), c1 = *reinterpret_cast<const f32x4*>(fp + k * 16 + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[k >> 1][(k & 1) * 8 + e] += c0[e];
                            o[k >> 1][(k & 1) * 8 + 4 + e] += c1[e];
                        }
                    }
                }
                __builtin_amdgcn_s_barrier();      // every wave is done with step 48: W2 slot 1 and the P buffers are idle (the windows)
                ln_handoff(reinterpret_cast<const float*>(smem + G1_OFF), reinterpret_cast<const float*>(smem + BE1_OFF));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // Wqkv(0..2), issued at steps 46..48
                __builtin_amdgcn_s_barrier();      // round 2 in LDS; Wqkv(0..2) visible
                // destination of this lane's row: ((frame * heads) * npad + token) * 64 in each of Q, K, V
                const int fr_ = row_c / p.ntok, tok_ = row_c - fr_ * p.ntok;
                const long qkv_row = ((long)fr_ * p.heads * p.npad + tok_) * 64 + lh_i * 8;
                const float* const nxr = p.X + (long)((row + BM * (int)gridDim.x) < M ? row + BM * (int)gridDim.x : M - 1) * D + lh_i * 8;
                mf_for(std::make_integer_sequence<int, NQT + 1>{}, [&](auto t_tag) __attribute__((always_inline)) {
                    constexpr int T = decltype(t_tag)::value;
                    __builtin_amdgcn_s_barrier();
                    if constexpr (T >= 1) {      // tile T - 1 from the pair's P buffer to its place
                        constexpr int TT = T - 1, WHICH = TT / NDB, HB = TT % NDB;
                        const char* pr = smem + P_OFF + pair * 4096 + lane16_i + (TT & 1) * 2048;
                        const uint4 u0 = *reinterpret_cast<const uint4*>(pr), u1 = *reinterpret_cast<const uint4*>(pr + 1024);
                        bf16_t* dst = (WHICH == 0 ? p.q : (WHICH == 1 ? p.k : p.v)) + qkv_row + (long)(HB >> 1) * p.npad * 64 + (HB & 1) * 32;
                        if (row < M) {
                            *reinterpret_cast<uint4*>(dst) = u0;
                            *reinterpret_cast<uint4*>(dst + 16) = u1;
                        }
                    }
                    if constexpr (T >= 1 && T <= NKS) {      // this wave's own rows: two stores per step
                        constexpr int k = T - 1;
                        if (row < M) {
                            f32x4 a, b;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                a[e] = o[k >> 1][(k & 1) * 8 + e];
                                b[e] = o[k >> 1][(k & 1) * 8 + 4 + e];
                            }
                            mf2_st(reinterpret_cast<f32x4*>(xrow + k * 16), a);
                            mf2_st(reinterpret_cast<f32x4*>(xrow + k * 16 + 4), b);
                        }
                    }
                    if constexpr (T > NKS) {      // the next item's rows: two k-steps per step
                        if (has_next) {
                            load_rows_k(nxr, std::integral_constant<int, 2 * (T - NKS - 1)>{});
                            load_rows_k(nxr, std::integral_constant<int, 2 * (T - NKS - 1) + 1>{});
                        }
                    }
                });
                static_assert(NQT + 1 - NKS - 1 == NKS / 2, "row loads fill the steps behind the row stores");
                if (has_next && !(MF2_ABL & 2)) {      // the next item's first projection k-tiles (the W1 ring is free again)
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const uint64_t sb = uniform64(wpj + (uint32_t)t * W_TILE + frag0);
                        const uint32_t ld = __builtin_amdgcn_readfirstlane(lds_base + W1_OFF + t * W_TILE + frag0);
                        mf_dma4(lane16_i, sb, ld);
                        mf_dma1<0>(lane16_i, sb + 4096, ld + 4096);
                        mf_dma1<1024>(lane16_i, sb + 4096, ld + 4096);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (before the first projection barrier publishes them)
                continue;
            }
            // ---- epilogue: o = x + b2 + fc2(...) back to the residual stream
            if (MF2_ABL & 32) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) asm volatile("" ::"v"(o[db]));
            } else if (row < M) {
                // (through LDS as 48 linear 1-KiB instructions -- both W2 slots are idle here -- the rows leave no faster: 9.2 against
                //  10.5 us per item; 192 KiB per CU go out at ~12 bytes per clock whatever the pattern)
#pragma unroll
                for (int k = 0; k < NKS; ++k) {
                    f32x4 a, b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = o[k >> 1][(k & 1) * 8 + e];
                        b[e] = o[k >> 1][(k & 1) * 8 + 4 + e];
                    }
                    if constexpr (PROJ) {      // (the accumulators started from x + b_proj: LayerNorm2 sees x_mid without the fc2 bias)
                        const float* fp = reinterpret_cast<const float*>(smem + BP_OFF) + k * 16 + lh_i * 8;
                        a += *reinterpret_cast<const f32x4*>(fp);
                        b += *reinterpret_cast<const f32x4*>(fp + 4);
                    }
                    mf2_st(reinterpret_cast<f32x4*>(xrow + k * 16), a);
                    mf2_st(reinterpret_cast<f32x4*>(xrow + k * 16 + 4), b);
                }
            }
            if (wave == 4) MF2_ST(1, item_k, 7);
            if (item + (int)gridDim.x < nitems) {      // the next item's rows (same lane mapping, BM * gridDim rows further on)
                const int nrow = row + BM * (int)gridDim.x;
                load_rows(p.X + (long)(nrow < M ? nrow : M - 1) * D + lh_i * 8);
            }
            if (wave == 4) MF2_ST(1, item_k, 8);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the ring pieces issued past the last item's end
        return;
    }

    // ===================================================================================================== wave A: LayerNorm, fc1, GELU
    // (measured and not kept: s_setprio 1..3 for this wave -- the longer instruction stream of the pair, 1.4 us per step against
    //  wave B's 1.1 -- and two accumulator chains for fc1 instead of one: both +-0.5 %)
    __syncthreads();      // (pairs with wave B's: constants staged, W1(0..2) visible)
    bf16x8 xn[NKS];
    f32x16 sa, sb;

    // LayerNorm2 of this lane's half row (x[2k], x[2k+1] = features 16 k + 8 lh + 0..7) -> xn (B-operand fragments, k = 16 s + 8 lh + j)
    auto layernorm = [&](const f32x4 (&x)[2 * NKS], int lh) __attribute__((always_inline)) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * NKS; k += 2) sum += ((x[k][0] + x[k][1]) + (x[k][2] + x[k][3])) + ((x[k + 1][0] + x[k + 1][1]) + (x[k + 1][2] + x[k + 1][3]));
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / D);
        float qv = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * NKS; ++k) {
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dlt = x[k][e] - mean;
                part = fmaf(dlt, dlt, part);
            }
            qv += part;
        }
        qv += __shfl_xor(qv, 32);
        const float rstd = 1.0f / sqrtf(qv * (1.0f / D) + p.eps);
        int kc = lh * 8;
        asm volatile("" : "+v"(kc));
        float mean_n = mean;
        asm volatile("" : "+v"(mean_n));
#pragma unroll
        for (int k = 0; k < NKS; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            const int k0 = k * 16 + kc;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(sG + k0), g1 = *reinterpret_cast<const f32x4*>(sG + k0 + 4);
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(sBe + k0), e1 = *reinterpret_cast<const f32x4*>(sBe + k0 + 4);
            float y[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = (x[2 * k][e] - mean_n) * rstd * g0[e] + e0[e];
                y[4 + e] = (x[2 * k + 1][e] - mean_n) * rstd * g1[e] + e1[e];
            }
            uint4 u;
            u.x = pack2<FMT>(y[0], y[1]);
            u.y = pack2<FMT>(y[2], y[3]);
            u.z = pack2<FMT>(y[4], y[5]);
            u.w = pack2<FMT>(y[6], y[7]);
            xn[k] = __builtin_bit_cast(bf16x8, u);
        }
    };
    // S(0): accumulators start from b1 (tile 0), fragments from W1 ring slot 0 (compiler-scheduled)
    auto s_zero = [&](int lane_p, const float* sBias) __attribute__((always_inline)) {
        f32x16 s;
        const float* bp = sBias + (lane_p >> 5) * 8;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(bp), c1 = *reinterpret_cast<const f32x4*>(bp + 4);
        const f32x4 c2 = *reinterpret_cast<const f32x4*>(bp + 16), c3 = *reinterpret_cast<const f32x4*>(bp + 20);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[e] = c0[e];
            s[4 + e] = c1[e];
            s[8 + e] = c2[e];
            s[12 + e] = c3[e];
        }
        const char* w = smem + W1_OFF + lane_p * 16;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) s = mfma32f<FMT>(lds_frag(w + ks * 1024), xn[ks], s);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return s;
    };
    // (not PROJ) rows from global -> LayerNorm -> S(0)
    auto prologue = [&](int item) __attribute__((always_inline)) {
        int lane_p = tid & 63;
        asm volatile("" : "+v"(lane_p));
        const int lr = lane_p & 31, lh = lane_p >> 5;
        const int row = item * BM + pair * 32 + lr;
        const int row_c = row < M ? row : M - 1;
        const float* const xrow = p.X + (long)row_c * D + lh * 8;
        f32x4 x[2 * NKS];
#pragma unroll
        for (int k = 0; k < NKS; ++k) {
            if (MF2_ABL & 32) {
                x[2 * k] = f32x4{1.f, 2.f, (float)k, (float)lane_p};
                x[2 * k + 1] = x[2 * k];
            } else {
                x[2 * k] = mf2_ld(reinterpret_cast<const f32x4*>(xrow + k * 16));
                x[2 * k + 1] = mf2_ld(reinterpret_cast<const f32x4*>(xrow + k * 16 + 4));
            }
        }
        layernorm(x, lh);
        return s_zero(lane_p, sB1);
    };
    // ---- the GELU of S(s), spread over the 24 MFMA gaps of the step (+ a tail slot): element n (accumulator register n) starts
    // at slot (14 n) / 15 and issues instruction i of the GELU at slot start + i; odd elements pack a dword behind their last one.
    // (measured and not kept, round 4: the two-coefficient logistic form x / (1 + 2^(x (c1 + c3 x^2))) -- 7 instructions instead of 9, no
    // clamp, |error| <= 2.7e-4 against 2.6e-5 -- takes 19 % of this wave's vector instructions away and 1.5 % off the launch: 4.355 ->
    // 4.29 ms per 32-frame step, +0.3 % frames/s.  The step is not bound by wave A's instruction count.)
#define MF2_GELU2 0
    constexpr int GELU_LAST = MF2_GELU2 ? 8 : 10;      // slot offset of the pack
    float ex[16], ea[16], eb[16], ec[16];
    uint32_t pd[8];
    auto gelu_op = [&](auto n_tag, auto i_tag, const f32x16& s) __attribute__((always_inline)) {
        constexpr int N = decltype(n_tag)::value, I = decltype(i_tag)::value;
        if constexpr (I == 0) ex[N] = s[N];
        if (MF2_ABL & 1) {
            if constexpr (I == GELU_LAST && (N & 1)) pd[N >> 1] = pack2<FMT>(ex[N - 1], ex[N]);
            return;
        }
        if constexpr (MF2_GELU2) {
            if constexpr (I == 1) eb[N] = ex[N] * ex[N];
            if constexpr (I == 2) ec[N] = fmaf(-1.00125610e-1f, eb[N], -2.30876530f);      // -(1.60031416, 6.940179e-2) * log2(e)
            if constexpr (I == 3) ec[N] = ec[N] * ex[N];
            if constexpr (I == 4) ec[N] = __builtin_amdgcn_exp2f(ec[N]);
            if constexpr (I == 5) ec[N] = 1.0f + ec[N];
            if constexpr (I == 6) ec[N] = __builtin_amdgcn_rcpf(ec[N]);
            if constexpr (I == 7) ex[N] = ex[N] * ec[N];
        } else {
            if constexpr (I == 1) ea[N] = __builtin_amdgcn_fmed3f(ex[N], -8.0f, 8.0f);
            if constexpr (I == 2) eb[N] = ea[N] * ea[N];
            if constexpr (I == 3) ec[N] = fmaf(1.01537542e-3f, eb[N], -1.06782573e-1f);
            if constexpr (I == 4) ec[N] = fmaf(ec[N], eb[N], -2.30111381f);
            if constexpr (I == 5) ec[N] = ec[N] * ea[N];
            if constexpr (I == 6) ec[N] = __builtin_amdgcn_exp2f(ec[N]);
            if constexpr (I == 7) ec[N] = 1.0f + ec[N];
            if constexpr (I == 8) ec[N] = __builtin_amdgcn_rcpf(ec[N]);
            if constexpr (I == 9) ex[N] = ex[N] * ec[N];
        }
        if constexpr (I == GELU_LAST && (N & 1)) pd[N >> 1] = pack2<FMT>(ex[N - 1], ex[N]);
    };
    auto valu_slot = [&](auto g_tag, const f32x16& s) __attribute__((always_inline)) {
        constexpr int G = decltype(g_tag)::value;
        mf_for(std::make_integer_sequence<int, 16>{}, [&](auto n_tag) __attribute__((always_inline)) {
            constexpr int N = decltype(n_tag)::value;
            constexpr int SG = (14 * N) / 15;
            if constexpr (G >= SG && G - SG <= GELU_LAST) gelu_op(n_tag, std::integral_constant<int, G - SG>{}, s);
        });
    };

    if (wave == 0) MF2_ST(0, 0, 15);
    if constexpr (!PROJ) sa = prologue(blockIdx.x);
    int item_k = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x, ++item_k) {
        if (wave == 0) MF2_ST(0, item_k, 0);
        uint32_t zero = 0;
        asm volatile("" : "+v"(zero));
        const uint32_t lane_i = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
        const uint32_t lane16_i = lane_i * 16, lh_i = lane_i >> 5;
        const uint32_t frag_rd_i = lds_base + lane16_i;
        char* const p_wr = smem + P_OFF + pair * 4096 + lane16_i;
        const float* const b1_lane_p = sB1 + attn::sigma23((int)(lane_i & 31));     // + HT * tile: b1 of MFMA A row lr
        constexpr uint32_t ONE = FMT == FMT_FP16 ? 0x3C00u : 0x3F80u;
        const uint4 ones_u = {lh_i == 0 ? (ONE << 16 | ONE) : 0u, lh_i == 0 ? ONE : 0u, 0u, 0u};     // k = 0, 1, 2 are 1.0
        // normalised rows from wave B: three rounds of eight fragments through the pair's window (5 barriers)
        auto receive_xn = [&]() __attribute__((always_inline)) {
            const char* const hr = smem + H_OFF + pair * 8192 + lane16_i;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                __builtin_amdgcn_s_barrier();      // round r is in LDS (the last one: with the first weight tiles of what follows)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) xn[r * 8 + kk] = lds_frag(hr + kk * 1024);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (r < 2) __builtin_amdgcn_s_barrier();      // read: wave B may write the next round
            }
        };
        if constexpr (PROJ) {
            // ---- wave B projects and normalises (12 + 1 barriers); xn arrives in three rounds of eight fragments; then S(0)
#pragma unroll 1
            for (int kt = 0; kt <= NPT; ++kt) __builtin_amdgcn_s_barrier();
            receive_xn();
            sa = s_zero((int)lane_i, sB1);
            if (wave == 0) MF2_ST(0, item_k, 10);
        }

        // tile step s: S(s+1) (unless LAST) || GELU of S(s) -> P(s) into the pair's LDS buffer s & 1
        auto step_a = [&](f32x16& s_cur, f32x16& s_nxt, int s, auto last_tag) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(last_tag)::value;
            bf16x8 bias_frag;
            if (!LAST) {      // b1 of tile s+1 as an A fragment (hi + mid + lo = the fp32 value exactly) times a ones fragment
                const float bj = b1_lane_p[(s + 1) * HT];
                const uint32_t hi = pack2<FMT>(bj, 0.f);
                const float r1f = bj - lo_to_f32<FMT>(hi);
                const uint32_t mid = pack2<FMT>(r1f, 0.f);
                const uint32_t lo = pack2<FMT>(r1f - lo_to_f32<FMT>(mid), 0.f);
                const uint4 fu = {lh_i == 0 ? ((hi & 0xFFFFu) | (mid << 16)) : 0u, lh_i == 0 ? (lo & 0xFFFFu) : 0u, 0u, 0u};
                bias_frag = __builtin_bit_cast(bf16x8, fu);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if constexpr (!LAST) {
                const uint32_t a1 = frag_rd_i + W1_OFF + (uint32_t)((s + 1) % 3) * W_TILE;
                bf16x8 fr[NFR];
                auto issue_read = [&](auto g_tag) __attribute__((always_inline)) {
                    constexpr int G = decltype(g_tag)::value;
                    if (MF2_ABL & 16) return;
                    mf_rd<G * 1024>(fr[G % NFR], a1);
                };
                mf_for(std::make_integer_sequence<int, RA>{}, issue_read);
                {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    s_nxt = mfma32f<FMT>(bias_frag, __builtin_bit_cast(bf16x8, ones_u), z);
                }
                __builtin_amdgcn_sched_barrier(0);
                mf_for(std::make_integer_sequence<int, 24>{}, [&](auto g_tag) __attribute__((always_inline)) {
                    constexpr int G = decltype(g_tag)::value;
                    if constexpr (G + RA < 24) issue_read(std::integral_constant<int, G + RA>{});
                    if (MF2_ABL & 16) fr[G % NFR] = xn[(G + 1) % NKS];
                    else mf_wait<(23 - G < RA ? 23 - G : RA)>();
                    if (!(MF2_ABL & 4)) s_nxt = mfma32f<FMT>(fr[G % NFR], xn[G], s_nxt);
                    __builtin_amdgcn_sched_barrier(0);
                    valu_slot(g_tag, s_cur);
                    __builtin_amdgcn_sched_barrier(0);
                });
                valu_slot(std::integral_constant<int, 24>{}, s_cur);
            } else {
                mf_for(std::make_integer_sequence<int, 25>{}, [&](auto g_tag) __attribute__((always_inline)) { valu_slot(g_tag, s_cur); });
            }
            const uint4 u0 = {pd[0], pd[1], pd[2], pd[3]}, u1 = {pd[4], pd[5], pd[6], pd[7]};
            *reinterpret_cast<uint4*>(p_wr + (s & 1) * 2048) = u0;
            *reinterpret_cast<uint4*>(p_wr + (s & 1) * 2048 + 1024) = u1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // P(s) is in LDS before the next barrier
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll 1
        for (int s = 0; s < NT - 2; s += 2) {
            step_a(sa, sb, s, std::false_type{});
            if (wave == 0 && s == 0) MF2_ST(0, item_k, 1);
            step_a(sb, sa, s + 1, std::false_type{});
            if (wave == 0 && s == 0) MF2_ST(0, item_k, 2);
            if (wave == 0 && s == 22) MF2_ST(0, item_k, 3);
        }
        if (wave == 0) MF2_ST(0, item_k, 4);
        step_a(sa, sb, NT - 2, std::false_type{});
        step_a(sb, sa, NT - 1, std::true_type{});
        if (wave == 0) MF2_ST(0, item_k, 5);
        __builtin_amdgcn_s_barrier();      // step 48: wave B's last tile; this wave already works on the next item
        if (wave == 0) MF2_ST(0, item_k, 6);
        if constexpr (QKV) {
            // ---- qkv tail: LayerNorm1 of the next block arrives from wave B; Z(t)^T = Wqkv_t . xn^T + b (36 tiles of 32 output features)
            // with the step skeleton of fc1 -- Z(t+1) on the matrix pipe while Z(t) is scaled (Q: 64^-0.5 log2 e), packed and left in the
            // pair's P buffer for wave B to store -- and THIS wave's own LDS-DMA stream: tile t + 3 into the W1 ring slot tile t left
            __builtin_amdgcn_s_barrier();      // (wave B: step 48 done everywhere, the hand-off windows are free)
            receive_xn();
            const float* const sBq = reinterpret_cast<const float*>(smem + BQ_OFF);
            const float* const bq_lane_p = sBq + attn::sigma23((int)(lane_i & 31));
            const uint32_t frag0 = (uint32_t)pair * 6 * 1024;
            sa = s_zero((int)lane_i, sBq);
            auto step_q = [&](f32x16& s_cur, f32x16& s_nxt, int t, auto last_tag) __attribute__((always_inline)) {
                constexpr bool LAST = decltype(last_tag)::value;
                bf16x8 bias_frag;
                if (!LAST) {
                    const float bj = bq_lane_p[(t + 1) * 32];
                    const uint32_t hi = pack2<FMT>(bj, 0.f);
                    const float r1f = bj - lo_to_f32<FMT>(hi);
                    const uint32_t mid = pack2<FMT>(r1f, 0.f);
                    const uint32_t lo = pack2<FMT>(r1f - lo_to_f32<FMT>(mid), 0.f);
                    const uint4 fu = {lh_i == 0 ? ((hi & 0xFFFFu) | (mid << 16)) : 0u, lh_i == 0 ? (lo & 0xFFFFu) : 0u, 0u, 0u};
                    bias_frag = __builtin_bit_cast(bf16x8, fu);
                }
                const float sc = t < NDB ? p.qscale : 1.0f;
                const bool issue = t + 3 < NQT;
                const uint32_t so = (uint32_t)(t + 3) * W_TILE + frag0;
                const uint32_t d = lds_base + W1_OFF + (uint32_t)((t + 3) % 3) * W_TILE + frag0;
                const uint64_t gb0 = uniform64(wq + so), gb1 = uniform64(wq + so + 4096);
                const uint32_t gl0 = __builtin_amdgcn_readfirstlane(d), gl1 = __builtin_amdgcn_readfirstlane(d + 4096);
                auto piece = [&](auto j_tag) __attribute__((always_inline)) {
                    constexpr int J = decltype(j_tag)::value;
                    if ((MF2_ABL & (2 | 64)) || !issue) return;
                    if constexpr (J < 4) mf_dma1<J * 1024>(lane16_i, gb0, gl0);
                    else if constexpr (J < 6) mf_dma1<(J - 4) * 1024>(lane16_i, gb1, gl1);
                };
                auto q_slot = [&](auto g_tag) __attribute__((always_inline)) {
                    constexpr int G = decltype(g_tag)::value;
                    if constexpr (G < 16) ex[G] = s_cur[G] * sc;
                    else if constexpr (G < 24) {      // (V stays bf16 in the fp16 mode: kernels.h, AttnParams::fmt; t is wave-uniform)
                        if (FMT == FMT_BF16 || t >= 2 * NDB) pd[G - 16] = pack_bf16x2(ex[2 * (G - 16)], ex[2 * (G - 16) + 1]);
                        else pd[G - 16] = pack2_sat<FMT>(ex[2 * (G - 16)], ex[2 * (G - 16) + 1]);      // (Q / K: saturating, as gemm_ln12.hip)
                    }
                };
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                if constexpr (!LAST) {
                    const uint32_t a1 = frag_rd_i + W1_OFF + (uint32_t)((t + 1) % 3) * W_TILE;
                    bf16x8 fr[NFR];
                    auto issue_read = [&](auto g_tag) __attribute__((always_inline)) {
                        constexpr int G = decltype(g_tag)::value;
                        mf_rd<G * 1024>(fr[G % NFR], a1);
                    };
                    mf_for(std::make_integer_sequence<int, RA>{}, issue_read);
                    {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        s_nxt = mfma32f<FMT>(bias_frag, __builtin_bit_cast(bf16x8, ones_u), z);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mf_for(std::make_integer_sequence<int, 24>{}, [&](auto g_tag) __attribute__((always_inline)) {
                        constexpr int G = decltype(g_tag)::value;
                        if constexpr (G + RA < 24) issue_read(std::integral_constant<int, G + RA>{});
                        mf_wait<(23 - G < RA ? 23 - G : RA)>();
                        if (!(MF2_ABL & 128)) s_nxt = mfma32f<FMT>(fr[G % NFR], xn[G], s_nxt);
                        __builtin_amdgcn_sched_barrier(0);
                        q_slot(g_tag);
                        if constexpr ((G & 1) == 0 && G < 12) piece(std::integral_constant<int, G / 2>{});
                        __builtin_amdgcn_sched_barrier(0);
                    });
                } else {
                    mf_for(std::make_integer_sequence<int, 24>{}, q_slot);
                }
                const uint4 u0 = {pd[0], pd[1], pd[2], pd[3]}, u1 = {pd[4], pd[5], pd[6], pd[7]};
                *reinterpret_cast<uint4*>(p_wr + (t & 1) * 2048) = u0;
                *reinterpret_cast<uint4*>(p_wr + (t & 1) * 2048 + 1024) = u1;
                // tile t + 2 (issued one step ago) has landed before the next barrier publishes it; this step's six pieces may fly
                if (issue) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll 1
            for (int t = 0; t < NQT - 2; t += 2) {
                step_q(sa, sb, t, std::false_type{});
                step_q(sb, sa, t + 1, std::false_type{});
            }
            step_q(sa, sb, NQT - 2, std::false_type{});
            step_q(sb, sa, NQT - 1, std::true_type{});
            __builtin_amdgcn_s_barrier();      // wave B's last store step
        }
        if constexpr (!PROJ)
            if (item + (int)gridDim.x < nitems) sa = prologue(item + gridDim.x);
        if (wave == 0) MF2_ST(0, item_k, 7);
    }
}

// W1 [1536][384], W2 [384][1536] fp32 -> fragments [hidden tile][24 fc1 fragments, 24 fc2 fragments][64 lanes][8] in the operand format
__global__ __launch_bounds__(256) void pack_mlp_kernel(const float* __restrict__ W1, const float* __restrict__ W2,
                                                       bf16_t* __restrict__ dst, long total, int fmt) {
    using namespace mfc;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long t = idx;
        const int e = (int)(t & 7); t >>= 3;
        const int lane = (int)(t & 63); t >>= 6;
        const int frag = (int)(t % (2 * NKS));
        const int tile = (int)(t / (2 * NKS));
        const int i = attn::sigma23(lane & 31), h = lane >> 5;
        float v;
        if (frag < NKS) {        // fc1: A row = hidden unit, k = input feature
            v = W1[(long)(tile * HT + i) * D + frag * 16 + h * 8 + e];
        } else {                 // fc2: A row = output feature, k = hidden unit of this tile
            const int f2 = frag - NKS, db = f2 >> 1, s2 = f2 & 1;
            v = W2[(long)(db * 32 + i) * F + tile * HT + s2 * 16 + h * 8 + e];
        }
        dst[idx] = pack1(v, fmt);
    }
}

long mlp_fused_pack_elems(int Dm, int Fh) { return Dm == mfc::D && Fh == mfc::F ? (long)mfc::NT * mfc::TILE_BYTES / 2 : 0; }

int launch_pack_mlp(const float* W1, const float* W2, int Dm, int Fh, bf16_t* dst, hipStream_t s, int fmt) {
    const long total = mlp_fused_pack_elems(Dm, Fh);
    if (total <= 0) {
        dinoseg_set_error("pack_mlp: unsupported shape D=%d F=%d", Dm, Fh);
        return -1;
    }
    hipLaunchKernelGGL(pack_mlp_kernel, dim3(2048), dim3(256), 0, s, W1, W2, dst, total, fmt);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

bool mlp_fused_supported(int Dm, int Fh, int planes) { return Dm == mfc::D && Fh == mfc::F && planes == 1; }

// Wproj [384 out][384 in] fp32 -> bf16 fragments [k-tile kt][fragment db * 2 + s2][lane][8]: the fc2 fragment format of pack_mlp with
// the 32 input features of k-tile kt in the place of a hidden tile (A row = output feature 32 db + sigma23(lane & 31))
__global__ __launch_bounds__(256) void pack_proj_kernel(const float* __restrict__ W, bf16_t* __restrict__ dst, int fmt) {
    using namespace mf2;
    const int total = NPT * NKS * 512;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int t = idx;
        const int e = t & 7; t >>= 3;
        const int lane = t & 63; t >>= 6;
        const int frag = t % NKS, kt = t / NKS;
        const int i = attn::sigma23(lane & 31), h = lane >> 5, db = frag >> 1, s2 = frag & 1;
        dst[idx] = pack1(W[(long)(db * 32 + i) * D + kt * 32 + s2 * 16 + h * 8 + e], fmt);
    }
}
// Wqkv [1152 out][384 in] fp32 -> bf16 fragments [tile of 32 outputs][k-step][lane][8]: the fc1 fragment format of pack_mlp
__global__ __launch_bounds__(256) void pack_qkv_kernel(const float* __restrict__ W, bf16_t* __restrict__ dst, int fmt) {
    using namespace mf2;
    const int total = NQT * NKS * 512;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int t = idx;
        const int e = t & 7; t >>= 3;
        const int lane = t & 63; t >>= 6;
        const int frag = t % NKS, tile = t / NKS;
        const int i = attn::sigma23(lane & 31), h = lane >> 5;
        dst[idx] = pack1(W[(long)(tile * 32 + i) * D + frag * 16 + h * 8 + e], fmt);
    }
}
long mlp_fused_qkv_pack_elems(int Dm) { return Dm == mf2::D ? (long)mf2::NQT * mf2::W_TILE / 2 : 0; }
int launch_pack_qkv(const float* W, int Dm, bf16_t* dst, hipStream_t s, int fmt) {
    if (Dm != mf2::D) {
        dinoseg_set_error("pack_qkv: unsupported width %d", Dm);
        return -1;
    }
    hipLaunchKernelGGL(pack_qkv_kernel, dim3(432), dim3(256), 0, s, W, dst, fmt);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

long mlp_fused_proj_pack_elems(int Dm) { return Dm == mf2::D ? (long)mf2::NPT * mf2::W_TILE / 2 : 0; }
int launch_pack_proj(const float* W, int Dm, bf16_t* dst, hipStream_t s, int fmt) {
    if (Dm != mf2::D) {
        dinoseg_set_error("pack_proj: unsupported width %d", Dm);
        return -1;
    }
    hipLaunchKernelGGL(pack_proj_kernel, dim3(144), dim3(256), 0, s, W, dst, fmt);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int FMT>
static int launch_mlp_fused2_fmt(const MlpFusedParams& p, hipStream_t s);

int launch_mlp_fused2(const MlpFusedParams& p, hipStream_t s) {
    return p.fmt == FMT_FP16 ? launch_mlp_fused2_fmt<FMT_FP16>(p, s) : launch_mlp_fused2_fmt<FMT_BF16>(p, s);
}

template <int FMT>
static int launch_mlp_fused2_fmt(const MlpFusedParams& p, hipStream_t s) {
    static PerDeviceOnce once;
    if (once.first()) {
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused2_kernel<false, false, FMT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, mf2::LDS_BYTES));
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused2_kernel<true, false, FMT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, mf2::LDS_BYTES));
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused2_kernel<true, true, FMT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, mf2::LDS_BYTES));
        once.mark();
    }
    if (p.M <= 0 || p.ldx != mf2::D) {
        dinoseg_set_error("mlp_fused2: bad shape M=%d ldx=%d", p.M, p.ldx);
        return -1;
    }
    const int ncu = device_cu_count();
    if (ncu <= 0) return -2;
    const int nitems = (p.M + mf2::BM - 1) / mf2::BM;
    // persistent grid: as few workgroups as finish in the same number of rounds (901 items on 256 CUs take 4 rounds: 226 workgroups
    // do it too and leave 30 CUs to the other stream's kernels; measured +0.4 % frames/s); option mlp_grid overrides
    int grid;
    if (options().mlp_stagger > 0) {
        grid = nitems < ncu ? nitems : ncu;
    } else if (options().mlp_grid > 0) {
        grid = options().mlp_grid < ncu ? options().mlp_grid : ncu;
        if (grid > nitems) grid = nitems;
    } else {
        const int rounds = (nitems + ncu - 1) / ncu;
        grid = (nitems + rounds - 1) / rounds;
    }
    MlpFusedParams q = p;
    q.queue = nullptr;
    q.n_long = nitems % grid;      // workgroups 0 .. n_long-1 walk one item more than the others (0: all the same)
    q.sleep_max = q.n_long > 0 && nitems > grid ? options().mlp_stagger : 0;
#if MF2_STAMP
    static unsigned long long* sbuf = nullptr;
    const size_t sbytes = (size_t)grid * 2 * 4 * 16 * 2 * 8;
    if (!sbuf) DSEG_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&sbuf), 256 * 2 * 4 * 16 * 2 * 8));
    DSEG_CHECK_HIP(hipMemsetAsync(sbuf, 0, sbytes, s));
    q.queue = reinterpret_cast<int*>(sbuf);
#endif
    if (p.ctx) {
        if (!p.Wproj || !p.bproj) {
            dinoseg_set_error("mlp_fused2: ctx without the packed projection weight / bias");
            return -1;
        }
        if (p.Wqkv) {
            if (!p.bqkv || !p.gamma1 || !p.beta1 || !p.q || !p.k || !p.v || p.ntok <= 0 || p.npad < p.ntok || p.heads * 64 != mf2::D) {
                dinoseg_set_error("mlp_fused2: incomplete qkv tail (bias / norm1 / q / k / v / geometry)");
                return -1;
            }
            hipLaunchKernelGGL((mlp_fused2_kernel<true, true, FMT>), dim3(grid), dim3(mf2::THREADS), mf2::LDS_BYTES, s, q);
        } else {
            hipLaunchKernelGGL((mlp_fused2_kernel<true, false, FMT>), dim3(grid), dim3(mf2::THREADS), mf2::LDS_BYTES, s, q);
        }
    } else {
        if (p.Wqkv) {
            dinoseg_set_error("mlp_fused2: the qkv tail needs the projection in the same launch (ctx)");
            return -1;
        }
        hipLaunchKernelGGL((mlp_fused2_kernel<false, false, FMT>), dim3(grid), dim3(mf2::THREADS), mf2::LDS_BYTES, s, q);
    }
    DSEG_CHECK_HIP(hipGetLastError());
#if MF2_STAMP
    {
        static int calls = 0;
        if (++calls == 20) {      // one report, well after warm-up: per stamp the mean over workgroups of (time since the item's first stamp)
            DSEG_CHECK_HIP(hipStreamSynchronize(s));
            std::vector<unsigned long long> h(sbytes / 8);
            DSEG_CHECK_HIP(hipMemcpy(h.data(), sbuf, sbytes, hipMemcpyDeviceToHost));
            for (int role = 0; role < 2; ++role)
                for (int k = 0; k < 4; ++k) {
                    fprintf(stderr, "role %c item %d:", role ? 'B' : 'A', k);
                    for (int i = 0; i < 16; ++i) {
                        double sum = 0, sumc = 0;
                        int n = 0;
                        for (int b = 0; b < grid; ++b) {
                            const unsigned long long* d = &h[((((size_t)b * 2 + role) * 4 + k) * 16) * 2];
                            const unsigned long long* d0 = &h[((((size_t)b * 2 + 1) * 4 + 0) * 16) * 2];      // B, item 0, stamp 0
                            if (d[2 * i] == 0 || d0[0] == 0) continue;
                            sum += (double)(d[2 * i] - d0[0]) * 0.01;          // us (100 MHz)
                            sumc += (double)(d[2 * i + 1] - d0[1]);            // shader cycles
                            ++n;
                        }
                        if (n) fprintf(stderr, " [%d] %.1fus/%.0fkc", i, sum / n, sumc / n / 1e3);
                    }
                    fprintf(stderr, "\n");
                }
        }
    }
#endif
    return 0;
}

}  // namespace dseg
