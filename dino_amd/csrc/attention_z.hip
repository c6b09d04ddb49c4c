// Fused multi-head attention forward, "zero-reference" variant: four waves per SIMD.
//
// Same tiling, LDS image, operand orientation and loaders as attention.hip (read its header first); what changes is the softmax
// bookkeeping, to get from 165 to <= 128 registers per wave -- one more resident workgroup per CU.  Why that matters: a wave
// spends most of a tile waiting (LDS fragments, the tile barrier, v_exp results); with three waves per SIMD the matrix pipe
// is busy 53 % and the vector port 60 % -- the kernel is latency-bound, and the one thing that has raised its throughput
// every time is more resident waves.
//   * no running reference: probabilities are 2^S against the FIXED reference 0, S = q.k * scale * log2(e).  fp32 (and bf16, same
//     exponent range) hold 2^S for |S| < 126, i.e. raw logits up to +-87 * 8 / 1.44 -- far outside anything a trained ViT
//     produces -- so the score accumulators start from the inline constant 0 (no 16-register -m_run operand), there is no
//     per-tile overflow check, no rescale path in the loop;
//   * a row whose sum left the band [2^-60, 2^60] (overflow of a 2^S or of the O accumulators, or underflow) is detected ONCE, after the last tile; the
//     workgroup then recomputes exactly: one pass for the row maxima (scores only), one pass with S - max subtracted on the
//     VALU (32 extra v_sub per tile, only on this path).  Hit only by adversarial inputs (tests force it);
//   * the K fragments of a tile are read in two windows of four (16 registers instead of 32).
// lse = log2(sum) (+ max on the exact path), as before.
#include <type_traits>

#include "attn_common.h"
#include "kernels.h"

namespace dseg {

namespace az {
constexpr int QW = 32, KB = 64;
constexpr int KV_TILE = attn::KV_TILE_BYTES;
}  // namespace az

// NW waves per workgroup (32 query rows each); WPS = resident waves per SIMD the register budget is cut for.  The hi+lo
// instantiation (experiment, attn_variant bit 4) uses ONE 12-wave workgroup per CU around its 64 KiB K/V ring = three waves per
// SIMD (two 6-wave workgroups do not work: the dispatcher deals a workgroup's waves 2,2,1,1 over the SIMDs, twice).
// FMT (PLANES == 1 only): FMT_FP16 = Q, K (the QK^T product) and the ctx output are fp16; the probabilities and V -- the P.V product --
// stay bf16 (2^S against the fixed reference 0 lives on bf16's exponent range).
template <int PLANES, int WPS, int NW, int FMT = FMT_BF16>
__global__ __launch_bounds__(NW * 64, WPS) void attn_fwd_z_kernel(AttnParams p) {
    // (PLANES == 2 with FMT_FP16: Q / K fp16 hi + lo planes, V and the probabilities bf16 hi + lo, ctx fp16 hi + lo -- the arithmetic of
    //  attention_za.hip's hi + lo body; this instantiation is its bit-identity reference in the tests)
    using namespace az;
    constexpr int QB = NW * QW;
    using attn::sigma23;
    using attn::tr_frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = PLANES * 2 * KV_TILE;   // per plane: K slab + V slab
    static_assert(NW * 4096 <= 2 * STAGE_BYTES, "the O-store epilogue gives every wave a 4 KiB patch of the ring");
    int* const redo_flag = reinterpret_cast<int*>(smem + 2 * STAGE_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;

    // XCD-aware order: all q-tiles of one (batch, head) run on one XCD back-to-back (K/V stay in that L2).
    const int nq = (p.ntok + QB - 1) / QB;
    const int npairs = p.B * p.heads;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nq) * 8 + xcd, qt = slot % nq;
    if (pair >= npairs) return;
    if (tid == 0) *redo_flag = 0;

    const int ntok = p.ntok, npad = p.npad;
    const long pair_off = (long)pair * npad * 64;       // Q, K, V are [B*H][npad][64]
    const bf16_t* Qg = p.q + pair_off;
    const bf16_t* Kg = p.k + pair_off;
    const bf16_t* Vg = p.v + pair_off;

    // ---- Q fragments (B operand: k = d, col = query) straight from global into registers ----
    const int qrow = qt * QB + wave * QW + lr;
    const int qrow_c = qrow < ntok ? qrow : ntok - 1;
    bf16x8 qf[PLANES][4];
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[pl][s] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const uint4*>(Qg + pl * p.qkv_plane + (long)qrow_c * 64 + s * 16 + lh * 8));
    // the compiler must not carry "Q loads pending" into the tile loop (its vmcnt waits would drain the LDS-DMA prefetch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[pl][s]));

    // K/V tile loader (scalar base + constant lane offset; see attention.hip)
    constexpr int NPIECE = (16 + NW - 1) / NW;
    uint32_t soff[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int row = ((wave + i * NW) & 7) * 8 + (lane >> 3);
        soff[i] = (uint32_t)((row * 64 + attn::swz(row, lane & 7) * 8) * 2);
    }
    auto stage = [&](int st, int key0) __attribute__((always_inline)) {
        char* sbase = smem + st * STAGE_BYTES;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
            const char* kb = reinterpret_cast<const char*>(Kg + pl * p.qkv_plane + (long)key0 * 64);
            const char* vb = reinterpret_cast<const char*>(Vg + pl * p.qkv_plane + (long)key0 * 64);
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {
                const int piece = wave + i * NW;              // 0..7: K rows, 8..15: V rows
                if (16 % NW != 0 && piece >= 16) continue;    // (12 waves: 16 pieces do not divide evenly; wave-uniform)
                // (readfirstlane: all of this is wave-uniform, but the compiler's divergence analysis loses that in the two-plane
                //  instantiation and would hand the asm vector registers)
                const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(
                    (uint32_t)(size_t)(__attribute__((address_space(3))) char*)(sbase + pl * 2 * KV_TILE + piece * 1024));
                const uint64_t src = reinterpret_cast<uint64_t>(piece < 8 ? kb : vb);
                const uint64_t src_u = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)src) |
                                       ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(src >> 32)) << 32);
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(soff[i]), "s"(src_u), "s"(lds_dst)
                             : "memory");
            }
        }
    };

    const int ntiles = (ntok + KB - 1) / KB;
    const bool wave_active = qt * QB + wave * QW < ntok;     // wave-uniform
    const int krow_perm = sigma23(lr);
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_gl = (lane >> 4) & 1;
    int ka[4], va[2][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = attn::tile_off(krow_perm, s * 2 + lh);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            va[db][h] = KV_TILE + attn::tile_off(lh * 8 + tr_q + 4 * h, db * 4 + tr_gl * 2 + (tr_p >> 1)) + (tr_p & 1) * 8;

    f32x16 o[2];
    float l_run, m_ref = 0.f;

    // MODE 0: 2^S against the reference 0 (the fast pass).  MODE 1: row maxima only.  MODE 2: 2^(S - m_ref), exact.
    auto tile = [&](auto mode_tag, int t, auto slot_tag) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int SLOT = decltype(slot_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile t landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();      // everyone's pieces of tile t landed; everyone is done reading slot (t-1)&1
        if (t + 1 < ntiles) stage(SLOT ^ 1, (t + 1) * KB);
        if (!wave_active) return;
        const char* sb = smem + SLOT * STAGE_BYTES;
        // (the backend's MFMA / DS interleaving strategy for this region: 7.25 -> 7.19 ms of attention per 32-frame step; strategies
        //  2, 3: 7.30)
        if constexpr (PLANES == 1) __builtin_amdgcn_iglp_opt(0);

        // ---- S^T[key][q] = K . Q^T, the K fragments in two windows of four ----
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 kf[PLANES][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                kf[0][s] = lds_frag(sb + ka[s] + kb * 4096);
                if (PLANES == 2) kf[PLANES - 1][s] = lds_frag(sb + ka[s] + kb * 4096 + 2 * KV_TILE);
            }
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (PLANES == 2) {
                    z = mfma32f<FMT>(kf[PLANES - 1][s], qf[0][s], z);
                    z = mfma32f<FMT>(kf[0][s], qf[PLANES - 1][s], z);
                }
                z = mfma32f<FMT>(kf[0][s], qf[0][s], z);
            }
            sacc[kb] = z;
        }
        // lane (query lr, half lh): sacc[kb][8*s2 + j] is key  t*64 + kb*32 + s2*16 + lh*8 + j
        if ((t + 1) * KB > ntok) {   // ragged last tile: mask keys >= ntok (wave-uniform branch)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KB + kb * 32 + (r >> 3) * 16 + lh * 8 + (r & 7);
                    if (key >= ntok) sacc[kb][r] = -INFINITY;
                }
        }
        if (MODE == 1) {
            float mx = m_ref;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
            m_ref = mx;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            return;
        }
        // ---- P = 2^S (in place), this lane's partial row sum ----
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sv = MODE == 2 ? sacc[kb][r] - m_ref : sacc[kb][r];
                sacc[kb][r] = __builtin_amdgcn_exp2f(sv);
                ps += sacc[kb][r];
            }
        l_run += ps;

        // ---- P fragments (B operand: k = key, col = query): registers 8*s2..8*s2+7 of sacc[kb] ----
        bf16x8 pf[PLANES][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 hi, lo;
                split_bf16x2(sacc[kb][s2 * 8 + 0], sacc[kb][s2 * 8 + 1], hi.x, lo.x);
                split_bf16x2(sacc[kb][s2 * 8 + 2], sacc[kb][s2 * 8 + 3], hi.y, lo.y);
                split_bf16x2(sacc[kb][s2 * 8 + 4], sacc[kb][s2 * 8 + 5], hi.z, lo.z);
                split_bf16x2(sacc[kb][s2 * 8 + 6], sacc[kb][s2 * 8 + 7], hi.w, lo.w);
                pf[0][kb * 2 + s2] = __builtin_bit_cast(bf16x8, hi);
                if (PLANES == 2) pf[PLANES - 1][kb * 2 + s2] = __builtin_bit_cast(bf16x8, lo);
            }

        // ---- O^T[d][q] += V^T . P^T  (V^T fragments by transposing LDS reads) ----
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off0 = va[db][0] + ks * 2048, off1 = va[db][1] + ks * 2048;
                const bf16x8 vhi = tr_frag(sb + off0, sb + off1);
                if (PLANES == 2) {
                    const bf16x8 vlo = tr_frag(sb + 2 * KV_TILE + off0, sb + 2 * KV_TILE + off1);
                    o[db] = mfma32(vlo, pf[0][ks], o[db]);
                    o[db] = mfma32(vhi, pf[PLANES - 1][ks], o[db]);
                }
                o[db] = mfma32(vhi, pf[0][ks], o[db]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS reads of tile t are complete
    };
    auto pass = [&](auto mode_tag) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        if (MODE != 1) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
            l_run = 0.f;
        }
        stage(0, 0);
        int t = 0;
        for (; t + 1 < ntiles; t += 2) {
            tile(mode_tag, t, std::integral_constant<int, 0>{});
            tile(mode_tag, t + 1, std::integral_constant<int, 1>{});
        }
        if (t < ntiles) tile(mode_tag, t, std::integral_constant<int, 0>{});
    };

    pass(std::integral_constant<int, 0>{});
    float l_tot = l_run + __shfl_xor(l_run, 32);
    {
        // Exact recomputation for the whole workgroup (the tile loop is workgroup-synchronous) unless every row sum lies in a safe band
        // [2^-60, 2^60]: outside it a 2^S may have overflowed (also in the O accumulators: sum of P V reaches 2^128 before l does
        // when |V| > 1; a P in [2^127.99, 2^128) rounds to bf16 inf) or the row's large terms may sit in the flushed range of
        // v_exp (row maximum below ~-100: l stays > 0 but has lost entries).  The band costs the fast path nothing.
        const bool bad = wave_active && qrow < ntok && !(l_tot >= 0x1p-60f && l_tot <= 0x1p60f);
        if (__any(bad) && lane == 0) *redo_flag = 1;
        __syncthreads();                   // also: everyone is done reading the last tile
        if (__builtin_amdgcn_readfirstlane(*redo_flag) != 0) {             // workgroup-uniform (and uniform for the compiler)
            m_ref = -INFINITY;
            pass(std::integral_constant<int, 1>{});
            m_ref = fmaxf(m_ref, __shfl_xor(m_ref, 32));
            if (!(m_ref > -INFINITY)) m_ref = 0.f;      // (only for rows that have no key at all: cannot happen, ntok >= 1)
            __syncthreads();
            pass(std::integral_constant<int, 2>{});
            l_tot = l_run + __shfl_xor(l_run, 32);
            __syncthreads();               // the O patches below reuse the ring: everyone is done reading the last tile
        }
    }

    // ---- normalise and write ctx[b*ntok + q][head*64 + d] (whole-row stores through a wave-private LDS patch) ----
    const float inv = 1.0f / l_tot;
    const int b = pair / p.heads, head = pair - b * p.heads;
    const int dm = p.heads * 64;
    {
        // (every wave has passed the workgroup barrier behind the last tile: the whole ring is free)
        char* patch = smem + wave * 4096;
        const int q0 = qt * QB + wave * QW;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 hi, lo;
                    if constexpr (FMT == FMT_FP16 && PLANES == 2) {
                        split2<FMT>(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv, hi.x, lo.x);
                        split2<FMT>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv, hi.y, lo.y);
                    } else if constexpr (FMT == FMT_FP16) {
                        hi.x = pack2<FMT>(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv);
                        hi.y = pack2<FMT>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
                        lo = hi;
                    } else {
                        split_bf16x2(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv, hi.x, lo.x);
                        split_bf16x2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv, hi.y, lo.y);
                    }
                    *reinterpret_cast<uint2*>(patch + lr * 128 + (((db * 4 + g) ^ (lr & 7)) << 4) + lh * 8) = pl == 0 ? hi : lo;
                }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + (lane >> 3);
                const uint4 v = *reinterpret_cast<const uint4*>(patch + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
                if (q0 + row < ntok)
                    *reinterpret_cast<uint4*>(p.ctx + pl * p.ctx_plane + ((long)b * ntok + q0 + row) * dm + head * 64 + (lane & 7) * 8) = v;
            }
            asm volatile("" ::: "memory");
        }
        if (qrow < ntok && p.lse != nullptr && lh == 0) p.lse[(long)pair * ntok + qrow] = m_ref + __builtin_amdgcn_logf(l_tot);
    }
}

// ------------------------------------------------------------------------------------------------
// Small grids (a single frame: 6 heads x 29 q-tiles = 174 workgroups of four waves on 256 CUs -- one wave per SIMD at best, every LDS
// and MFMA latency exposed): the same kernel with the KEYS of a q-tile split over KS wave groups of one workgroup: group g (four waves =
// 128 queries) walks the g-th part of the K/V tiles with its own two-slot ring; with the fixed reference 0 the
// partial results need no rescaling -- O and the row sums of the parts simply add (through LDS, once, behind the tile loops).
// Two or three waves per SIMD cover each other's latencies: 42 -> 31 us per launch at one frame @480 (the outputs differ from the
// unsplit kernel's in the last place of a few elements per thousand: another summation order).  Single plane; the exact path
// (a row sum outside [2^-60, 2^60]) is run by group 0 alone over all tiles -- the unsplit kernel's order, bit for bit -- while the other
// groups keep it company at the barriers.
template <int FMT, int KS>
__global__ __launch_bounds__(KS * 256, KS) void attn_fwd_zs_kernel(AttnParams p) {
    using namespace az;
    constexpr int NWG = 4, QB = NWG * QW;      // waves per group, queries per workgroup
    using attn::sigma23;
    using attn::tr_frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = 2 * KV_TILE, RING_BYTES = 2 * STAGE_BYTES;       // per group
    int* const redo_flag = reinterpret_cast<int*>(smem + KS * RING_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, gw = wave & 3;          // key group, wave inside the group
    const int lr = lane & 31, lh = lane >> 5;

    const int nq = (p.ntok + QB - 1) / QB;
    const int npairs = p.B * p.heads;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nq) * 8 + xcd, qt = slot % nq;
    if (pair >= npairs) return;
    if (tid == 0) *redo_flag = 0;

    const int ntok = p.ntok, npad = p.npad;
    const long pair_off = (long)pair * npad * 64;
    const bf16_t* Qg = p.q + pair_off;
    const bf16_t* Kg = p.k + pair_off;
    const bf16_t* Vg = p.v + pair_off;

    const int qrow = qt * QB + gw * QW + lr;
    const int qrow_c = qrow < ntok ? qrow : ntok - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Qg + (long)qrow_c * 64 + s * 16 + lh * 8));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[s]));

    constexpr int NPIECE = 16 / NWG;
    uint32_t soff[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int row = ((gw + i * NWG) & 7) * 8 + (lane >> 3);
        soff[i] = (uint32_t)((row * 64 + attn::swz(row, lane & 7) * 8) * 2);
    }
    char* const ring = smem + grp * RING_BYTES;          // this group's two slots
    auto stage = [&](int st, int key0) __attribute__((always_inline)) {
        char* sbase = ring + st * STAGE_BYTES;
        const char* kb = reinterpret_cast<const char*>(Kg + (long)key0 * 64);
        const char* vb = reinterpret_cast<const char*>(Vg + (long)key0 * 64);
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int piece = gw + i * NWG;
            const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(
                (uint32_t)(size_t)(__attribute__((address_space(3))) char*)(sbase + piece * 1024));
            const uint64_t src = reinterpret_cast<uint64_t>(piece < 8 ? kb : vb);
            const uint64_t src_u = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)src) |
                                   ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(src >> 32)) << 32);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(soff[i]), "s"(src_u), "s"(lds_dst)
                         : "memory");
        }
    };

    const int ntiles = (ntok + KB - 1) / KB;
    const int nchunk = (ntiles + KS - 1) / KS;           // tiles per group (the last group takes what is left)
    const bool rows_active = qt * QB + gw * QW < ntok;   // wave-uniform
    const int krow_perm = sigma23(lr);
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_gl = (lane >> 4) & 1;
    int ka[4], va[2][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = attn::tile_off(krow_perm, s * 2 + lh);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            va[db][h] = KV_TILE + attn::tile_off(lh * 8 + tr_q + 4 * h, db * 4 + tr_gl * 2 + (tr_p >> 1)) + (tr_p & 1) * 8;

    f32x16 o[2];
    float l_run = 0.f, m_ref = 0.f;

    // iteration i of a pass over the tiles [t_begin, t_end): tile t_begin + i if it exists, the barrier in any case
    auto tile = [&](auto mode_tag, int i, auto slot_tag, int t_begin, int t_end, bool compute) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int SLOT = decltype(slot_tag)::value;
        const int t = t_begin + i;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < t_end) stage(SLOT ^ 1, (t + 1) * KB);
        if (!compute || t >= t_end) return;
        const char* sb = ring + SLOT * STAGE_BYTES;
        __builtin_amdgcn_iglp_opt(0);
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 kf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) kf[s] = lds_frag(sb + ka[s] + kb * 4096);
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) z = mfma32f<FMT>(kf[s], qf[s], z);
            sacc[kb] = z;
        }
        if ((t + 1) * KB > ntok) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KB + kb * 32 + (r >> 3) * 16 + lh * 8 + (r & 7);
                    if (key >= ntok) sacc[kb][r] = -INFINITY;
                }
        }
        if (MODE == 1) {
            float mx = m_ref;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
            m_ref = mx;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            return;
        }
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sv = MODE == 2 ? sacc[kb][r] - m_ref : sacc[kb][r];
                sacc[kb][r] = __builtin_amdgcn_exp2f(sv);
                ps += sacc[kb][r];
            }
        l_run += ps;
        bf16x8 pf[4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 hi, lo;
                split_bf16x2(sacc[kb][s2 * 8 + 0], sacc[kb][s2 * 8 + 1], hi.x, lo.x);
                split_bf16x2(sacc[kb][s2 * 8 + 2], sacc[kb][s2 * 8 + 3], hi.y, lo.y);
                split_bf16x2(sacc[kb][s2 * 8 + 4], sacc[kb][s2 * 8 + 5], hi.z, lo.z);
                split_bf16x2(sacc[kb][s2 * 8 + 6], sacc[kb][s2 * 8 + 7], hi.w, lo.w);
                pf[kb * 2 + s2] = __builtin_bit_cast(bf16x8, hi);
            }
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                o[db] = mfma32(tr_frag(sb + va[db][0] + ks * 2048, sb + va[db][1] + ks * 2048), pf[ks], o[db]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // niter iterations (the same count in both groups: the barrier is the workgroup's)
    auto pass = [&](auto mode_tag, int t_begin, int t_end, int niter, bool compute) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        if (MODE != 1) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
            l_run = 0.f;
        }
        if (t_begin < t_end) stage(0, t_begin * KB);
        int i = 0;
        for (; i + 1 < niter; i += 2) {
            tile(mode_tag, i, std::integral_constant<int, 0>{}, t_begin, t_end, compute);
            tile(mode_tag, i + 1, std::integral_constant<int, 1>{}, t_begin, t_end, compute);
        }
        if (i < niter) tile(mode_tag, i, std::integral_constant<int, 0>{}, t_begin, t_end, compute);
    };

    const int t0 = grp * nchunk < ntiles ? grp * nchunk : ntiles, t1 = t0 + nchunk < ntiles ? t0 + nchunk : ntiles;
    pass(std::integral_constant<int, 0>{}, t0, t1, nchunk, rows_active);

    // ---- merge: groups 1 .. KS-1 hand O and their row sums to group 0 (the rings are free behind the barrier) ----
    static_assert((KS - 1) * NWG * 33 * 64 * 4 <= KS * RING_BYTES, "exchange area inside the rings");
    float* const xch0 = reinterpret_cast<float*>(smem) + gw * (33 * 64);      // per (group, query wave): 32 accumulator registers + l, lane-major
    __syncthreads();
    if (grp != 0 && rows_active) {
        float* const xch = xch0 + (grp - 1) * (NWG * 33 * 64);
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[(d * 16 + r) * 64 + lane] = o[d][r];
        xch[32 * 64 + lane] = l_run;
    }
    __syncthreads();
    if (grp == 0 && rows_active) {
#pragma unroll
        for (int g = 1; g < KS; ++g) {
            const float* xch = xch0 + (g - 1) * (NWG * 33 * 64);
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] += xch[(d * 16 + r) * 64 + lane];
            l_run += xch[32 * 64 + lane];
        }
    }
    float l_tot = l_run + __shfl_xor(l_run, 32);
    {
        const bool bad = grp == 0 && rows_active && qrow < ntok && !(l_tot >= 0x1p-60f && l_tot <= 0x1p60f);
        if (__any(bad) && lane == 0) *redo_flag = 1;
        __syncthreads();                   // (also: group 0 is done reading the exchange area)
        if (__builtin_amdgcn_readfirstlane(*redo_flag) != 0) {
            // exact recomputation by group 0 over ALL tiles on its own ring; group 1 only meets the barriers
            const bool me = grp == 0 && rows_active;
            const int e1 = grp == 0 ? ntiles : 0;
            m_ref = -INFINITY;
            pass(std::integral_constant<int, 1>{}, 0, e1, ntiles, me);
            m_ref = fmaxf(m_ref, __shfl_xor(m_ref, 32));
            if (!(m_ref > -INFINITY)) m_ref = 0.f;
            __syncthreads();
            pass(std::integral_constant<int, 2>{}, 0, e1, ntiles, me);
            l_tot = l_run + __shfl_xor(l_run, 32);
            __syncthreads();
        }
    }
    if (grp != 0) return;

    // ---- normalise and write ctx (as attn_fwd_z_kernel) ----
    const float inv = 1.0f / l_tot;
    const int b = pair / p.heads, head = pair - b * p.heads;
    const int dm = p.heads * 64;
    {
        char* patch = smem + gw * 4096;
        const int q0 = qt * QB + gw * QW;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 hi, lo;
                if constexpr (FMT == FMT_FP16) {
                    hi.x = pack2<FMT>(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv);
                    hi.y = pack2<FMT>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
                } else {
                    split_bf16x2(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv, hi.x, lo.x);
                    split_bf16x2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv, hi.y, lo.y);
                }
                *reinterpret_cast<uint2*>(patch + lr * 128 + (((db * 4 + g) ^ (lr & 7)) << 4) + lh * 8) = hi;
            }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + (lane >> 3);
            const uint4 v = *reinterpret_cast<const uint4*>(patch + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
            if (q0 + row < ntok)
                *reinterpret_cast<uint4*>(p.ctx + ((long)b * ntok + q0 + row) * dm + head * 64 + (lane & 7) * 8) = v;
        }
        if (qrow < ntok && p.lse != nullptr && lh == 0) p.lse[(long)pair * ntok + qrow] = m_ref + __builtin_amdgcn_logf(l_tot);
    }
}

template <int FMT, int KS>
static int launch_zs(const AttnParams& p, hipStream_t s) {
    using namespace az;
    const int nq = (p.ntok + 127) / 128;
    const int npairs = p.B * p.heads;
    const int grid = ((npairs + 7) / 8) * 8 * nq;
    const size_t lds = (size_t)KS * 2 * 2 * KV_TILE + 16;
    static PerDeviceOnce once;          // (more than 64 KiB of dynamic LDS: say so once per device)
    if (once.first()) {
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_zs_kernel<FMT, KS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.mark();
    }
    hipLaunchKernelGGL((attn_fwd_zs_kernel<FMT, KS>), dim3(grid), dim3(KS * 256), lds, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int PLANES, int WPS, int NW, int FMT = FMT_BF16>
static int launch_z(const AttnParams& p, hipStream_t s) {
    using namespace az;
    constexpr int QB = NW * QW;
    const int nq = (p.ntok + QB - 1) / QB;
    const int npairs = p.B * p.heads;
    const int grid = ((npairs + 7) / 8) * 8 * nq;
    const size_t lds = (size_t)2 * PLANES * 2 * KV_TILE + 16;
    hipLaunchKernelGGL((attn_fwd_z_kernel<PLANES, WPS, NW, FMT>), dim3(grid), dim3(NW * 64), lds, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

bool attention_x3_za(int batch, int heads, int ntok) {
    // one 256-query workgroup per CU: worth it from two rounds on (below that the 128-query reference-based kernel fills the chip better);
    // attn_variant bit 10 = the assembly kernels at all, bit 6 = never the 256-query workgroups, bit 11 = at every grid size (tests)
    const int av = options().attn_variant;
    if (!(av & 1024) || (av & 64)) return false;
    if (av & 2048) return true;
    const int ncu = device_cu_count();
    const long wgs = (long)((batch * heads + 7) / 8) * 8 * ((ntok + 255) / 256);
    return ncu > 0 && wgs >= 2L * ncu;
}

int launch_attention_z(const AttnParams& p, hipStream_t s) {
    if (p.planes == 1) {
        // 256-query workgroups (8 waves share a K/V tile: 2 LDS-DMA pieces per wave and tile instead of 4, half the L2 -> LDS traffic)
        // once 128-query ones would queue two rounds deep on the chip anyway; small batches keep the finer grain.  Same outputs bit
        // for bit (a wave's 32 queries see the same tiles in the same order).
        // (32 frames: attention 7.53 -> 7.48 ms per step, 2352 -> 2384 frames/s; halves of 4..12 frames on two streams: +1..3 %; alone on
        //  the chip below one round of 128-query workgroups the finer grain wins: 1 frame 0.54 vs 0.69 ms, 4 frames 1.09 vs 1.16)
        const int ncu = device_cu_count();
        const long wgs4 = (long)((p.B * p.heads + 7) / 8) * 8 * ((p.ntok + 127) / 128);
        // (attn_variant bit 12: the 256-query workgroups at every grid size -- tests compare kernels at equal workgroup granularity: the
        //  exact recomputation is decided per workgroup)
        const bool wide = (ncu > 0 && (wgs4 >= 4L * ncu || (p.shared_gpu && wgs4 >= 2L * ncu))) || (options().attn_variant & 4096);
        // fewer 128-query workgroups than CUs and at least four K/V tiles: split the keys over wave groups (attn_fwd_zs_kernel;
        // attn_variant bit 9 keeps the unsplit kernel: A/B and tests)
        // (decided for the batch of the WHOLE call: the split changes the summation order, and the half-batches of a two-stream forward
        //  must run what one stream would.  Inference only -- lse == nullptr: the training forward keeps one arithmetic at every batch
        //  size, tests/test_train_gpu.py: a batch-8 step is the mean of eight single-frame steps to 6e-7.)
        const long wgs4_call = (long)(((p.dispatch_B > 0 ? p.dispatch_B : p.B) * p.heads + 7) / 8) * 8 * ((p.ntok + 127) / 128);
        const bool ksplit = ncu > 0 && wgs4_call < ncu && p.ntok > 3 * az::KB && p.lse == nullptr && !(options().attn_variant & (512 | 2048));
        if (ksplit) {      // (one frame @480, 57 tiles: 42 us unsplit, 33 with two groups, 31 with three, 37 with four -- 128 registers spill)
            if (p.ntok >= 32 * az::KB) {
                if (p.fmt == FMT_FP16) return launch_zs<FMT_FP16, 3>(p, s);
                return launch_zs<FMT_BF16, 3>(p, s);
            }
            if (p.fmt == FMT_FP16) return launch_zs<FMT_FP16, 2>(p, s);
            return launch_zs<FMT_BF16, 2>(p, s);
        }
        // attn_variant bit 10: the hand-scheduled assembly tile loop (attention_za.hip; bit-identical outputs)
        // (bit 11: at every grid size -- tests)
        if (((wide && !(options().attn_variant & 64)) || (options().attn_variant & 2048)) && (options().attn_variant & 1024))
            return launch_attention_za(p, s);
        if (p.fmt == FMT_FP16) {
            if (wide && !(options().attn_variant & 64)) return launch_z<1, 4, 8, FMT_FP16>(p, s);
            return launch_z<1, 4, 4, FMT_FP16>(p, s);
        }
        if (wide && !(options().attn_variant & 64)) return launch_z<1, 4, 8>(p, s);
        return launch_z<1, 4, 4>(p, s);
    }
    // hi + lo planes, zero-reference: V and the probabilities are bf16 hi + lo planes in both formats (FMT_FP16: Q, K and ctx fp16 hi + lo).
    // attn_variant bit 10: the assembly tile loop (attention_za.hip, eight waves, one workgroup per CU); else the compiled kernel
    // (168 registers, three waves per SIMD; the reference-based kernel of attention.hip: 213, two)
    if (options().attn_variant & 1024) return launch_attention_za(p, s);
    if (p.fmt == FMT_FP16) return launch_z<2, 3, 12, FMT_FP16>(p, s);
    return launch_z<2, 3, 12>(p, s);
}

}  // namespace dseg
