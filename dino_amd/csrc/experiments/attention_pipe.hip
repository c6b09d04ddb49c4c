// Fused attention forward, bf16 fast mode, software-pipelined across K/V tiles.
//
// Same mathematics, LDS images, fragment tricks and output as attention.hip (read that header first).  What changes is
// the order of work inside a wave.  attention.hip runs   S(t) -> exp(t) -> PV(t)   per tile: the exponentials have no
// matrix work of their own wave to sit beside, and PMC shows the SIMD's vector port active 59 % and the matrix pipe
// busy 49 % but together only 21 % of the time.  Here iteration t issues
//
//        PV(t-1)  [8 MFMAs]   and   S(t+1)  [8 MFMAs]        beside        exp / row-sum / pack of tile t,
//
// as 16 "gaps": one v_mfma_f32_32x32x16_bf16 followed by 2 v_exp + 2 v_add + 1 v_cvt_pk (one packed pair of the
// tile's probabilities), pinned in that order with sched_barrier.  All three streams are independent inside an
// iteration: S(t) was finished an iteration ago, P(t-1) is complete, K(t+1) and V(t-1) are resident.
// Costs: a second score accumulator set and a second P fragment set (+48 registers: 2 waves per SIMD instead of 3).
//
// The scores of tile t stay intact in their registers while their exponentials go to the P fragments, so the
// careful path of the overflow check (see attention.hip) needs no recomputation: row maximum of the kept scores,
// reference moved, O / l / negm and the already computed S(t+1) rescaled, tile t exponentiated again.
//
// Round 1 ran this with 2-slot K / V rings: V(t) was staged in iteration t and read in iteration t+1, one iteration (~0.4 us) of
// lead against ~1 us of L2 -> LDS latency, and the waves sat in s_waitcnt 44 % of the time (10 % slower than attention.hip).
// Now: K and V rings of RING = 4 slots each (64 KiB per workgroup, two workgroups per CU), tiles staged DEPTH = 3 iterations
// ahead, one constant counted `s_waitcnt vmcnt((DEPTH-1) * 4)` per iteration (each wave issues 2 K + 2 V pieces per iteration,
// nothing else touches vmcnt inside the loop), fragment reads LEAD = 5 MFMA gaps ahead of their use.
#include <type_traits>

#include "attn_common.h"
#include "kernels.h"

namespace dseg {

#ifndef APIPE_RING
#define APIPE_RING 4
#define APIPE_DEPTH 3
#endif
#ifndef APIPE_LEAD
#define APIPE_LEAD 5
#endif
#ifndef APIPE_SKEW
#define APIPE_SKEW 1
#endif
namespace apipe {
constexpr int QW = 32, KB = 64, NW = 4, QB = NW * QW;
constexpr int KV_TILE = attn::KV_TILE_BYTES;
constexpr int RING = APIPE_RING, DEPTH = APIPE_DEPTH;          // ring slots per operand; tiles in flight ahead of the one being read
constexpr int V_BASE = RING * KV_TILE;
constexpr int LDS_BYTES = 2 * RING * KV_TILE;
using attn::sigma23;
using attn::tile_off;
using attn::tr_frag;
}  // namespace apipe

__global__ __launch_bounds__(256, (apipe::LDS_BYTES > 80 * 1024 ? 1 : 2)) void attn_fwd_pipe_kernel(AttnParams p) {
    using namespace apipe;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;

    // XCD-aware order: all q-tiles of one (batch, head) run on one XCD back-to-back (K/V stay in that L2).
    const int nq = (p.ntok + QB - 1) / QB;
    const int npairs = p.B * p.heads;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nq) * 8 + xcd, qt = slot % nq;
    if (pair >= npairs) return;

    const int ntok = p.ntok, npad = p.npad;
    const long pair_off = (long)pair * npad * 64;       // Q, K, V are [B*H][npad][64]
    const bf16_t* Qg = p.q + pair_off;
    const bf16_t* Kg = p.k + pair_off;
    const bf16_t* Vg = p.v + pair_off;

    const int qrow = qt * QB + wave * QW + lr;
    const int qrow_c = qrow < ntok ? qrow : ntok - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Qg + (long)qrow_c * 64 + s * 16 + lh * 8));
    // the compiler must not carry "Q loads pending" into the tile loop: with the loader's uncounted LDS-DMA in flight its own
    // vmcnt waits in front of the first use of each fragment (re-inserted every iteration: the loop header joins the states)
    // would drain the ring every iteration
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[s]));

    // one 64-row slab = 8 pieces of 1 KiB, 2 per wave.  LDS-DMA by inline asm with a scalar base and a constant 32-bit lane
    // offset: besides saving the per-piece 64-bit vector address arithmetic of the builtin, the compiler does not know these
    // loads exist -- behind the builtin it put an `s_waitcnt vmcnt(0)` in front of the first LDS read of every iteration (the DMA
    // might be writing what is read), which drained the ring each time: the waves sat in s_waitcnt 38-44 % of the time (PMC).
    uint32_t soff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + (lane >> 3);
        soff[i] = (uint32_t)((row * 64 + attn::swz(row, lane & 7) * 8) * 2);
    }
    auto stage_slab = [&](const bf16_t* g, char* dst, int key0) {
        const char* base = reinterpret_cast<const char*>(g + (long)key0 * 64);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t lds_dst = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)(dst + (wave * 2 + i) * 1024);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(soff[i]), "s"(base), "s"(lds_dst)
                         : "memory");
        }
    };
    auto stage_k = [&](int t) { stage_slab(Kg, smem + (t & (RING - 1)) * KV_TILE, t * KB); };
    auto stage_v = [&](int t) { stage_slab(Vg, smem + V_BASE + (t & (RING - 1)) * KV_TILE, t * KB); };
    static_assert((RING & (RING - 1)) == 0 && DEPTH <= RING - 1, "ring: power of two, one slot being read");

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    constexpr float P_LIMIT = 65536.f;
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    float m_run = 0.f, l_run = 0.f;
    f32x16 sA[2], sB[2];        // scores of the even / odd tiles
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pA[4], pB[4];         // P fragments (bf16x8 as 4 packed words) of the even / odd tiles

    const int ntiles = (ntok + KB - 1) / KB;
    const bool wave_active = qt * QB + wave * QW < ntok;     // wave-uniform
    const int krow_perm = sigma23(lr);
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_gl = (lane >> 4) & 1;

    // LDS addresses: 4 + 4 per-lane byte offsets, everything else (half / key step / ring slot) is an immediate
    int ka[4], va[2][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = tile_off(krow_perm, s * 2 + lh);                     // + kb * 4096 (swizzle ignores row bit 5)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int h = 0; h < 2; ++h)                                                              // + ks * 2048 (swizzle ignores row bits 4, 5)
            va[db][h] = V_BASE + tile_off(lh * 8 + tr_q + 4 * h, db * 4 + tr_gl * 2 + (tr_p >> 1)) + (tr_p & 1) * 8;
    auto k_frag = [&](int slot_off, int kb, int s) { return lds_frag(smem + ka[s] + kb * 4096 + slot_off); };
    auto v_frag = [&](int slot_off, int ks, int db) {
        return tr_frag(smem + va[db][0] + ks * 2048 + slot_off, smem + va[db][1] + ks * 2048 + slot_off);
    };
    auto mask_tile = [&](int t, f32x16 (&s)[2]) {      // ragged last tile: keys >= ntok
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * KB + kb * 32 + (r >> 3) * 16 + lh * 8 + (r & 7);
                if (key >= ntok) s[kb][r] = -INFINITY;
            }
    };
    auto row_max = [&](const f32x16 (&s)[2]) {
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        return fmaxf(mx, __shfl_xor(mx, 32));
    };
    // one gap's VALU work: probabilities 2g, 2g+1 of the tile (registers 2(g&7), +1 of half g>>3) -> one packed word
    auto exp_pair = [&](const f32x16 (&s)[2], u32x4 (&pw)[4], int g, float& ps0, float& ps1) {
        const int kb = g >> 3, i = (g & 7) * 2;
        const float e0 = __builtin_amdgcn_exp2f(s[kb][i]), e1 = __builtin_amdgcn_exp2f(s[kb][i + 1]);
        ps0 += e0;
        ps1 += e1;
        uint32_t w = pack_bf16x2(e0, e1);
        // the results are consumed blocks later; without an anchor the compiler sinks all 32 exponentials below the MFMAs
        asm volatile("" : "+v"(ps0), "+v"(ps1), "+v"(w));
        pw[g >> 2][g & 3] = w;
    };

    // Iteration t: exp(t) from sc -> pc, beside PV(t-1) (pp = P fragments of tile t-1) and S(t+1) -> sn.
    // K(t+1) and V(t-1) were staged DEPTH iterations ago; this iteration stages K(t+1+DEPTH) and V(t-1+DEPTH) into the slots
    // that K(t) / V(t-2) left free an iteration ago.  Every iteration issues exactly 4 pieces per wave (clamped re-reads past
    // the last tile keep the count constant), so vmcnt((DEPTH-1)*4) retires exactly the two tiles read now.
    auto iteration = [&](int t, f32x16 (&sc)[2], f32x16 (&sn)[2], u32x4 (&pp)[4], u32x4 (&pc)[4], auto pv_tag, auto s_tag) {
        constexpr bool HAS_PV = decltype(pv_tag)::value, HAS_S = decltype(s_tag)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * 4) : "memory");   // this wave's pieces of K(t+1) and V(t-1) landed
        __builtin_amdgcn_s_barrier();      // ... and everyone's; everyone is done with K(t) and V(t-2)
        {
            const int tk = t + 1 + DEPTH, tv = t - 1 + DEPTH;
            stage_k(tk < ntiles ? tk : ntiles - 1);     // (past the end: harmless re-reads into slots nobody reads any more)
            stage_v(tv < ntiles ? tv : ntiles - 1);
        }
        if (!wave_active) return;
        const int k_off = ((t + 1) & (RING - 1)) * KV_TILE, v_off = ((t - 1) & (RING - 1)) * KV_TILE;

        // operand fragment of gap g's MFMA: V^T(ks = g>>1, db = g&1) for the PV gaps 0..7, K(kb, s) for the score gaps 8..15;
        // read LEAD gaps ahead of its use
        constexpr int LEAD = APIPE_LEAD;
        bf16x8 frag[16 + LEAD];
        auto read_frag = [&](int g) {
            if (g < 8) {
                if (HAS_PV) frag[g] = v_frag(v_off, g >> 1, g & 1);
            } else if (g < 16) {
                if (HAS_S) frag[g] = k_frag(k_off, (g - 8) >> 2, (g - 8) & 3);
            }
        };
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int g = 0; g < LEAD; ++g) read_frag(g);
        __builtin_amdgcn_sched_barrier(0);
        // The exponentials of gap g are consumed (row sums, packing) SKEW gaps later: v_exp's result takes tens of cycles to
        // come back, and with the consumer right behind it a wave alone on its SIMD spent 84 cycles per gap instead of ~36.
        constexpr int SKEW = APIPE_SKEW;
        float e0q[SKEW > 0 ? SKEW : 1], e1q[SKEW > 0 ? SKEW : 1];
#pragma unroll
        for (int g = 0; g < 16 + SKEW; ++g) {
            if (SKEW == 0) {
                if (g < 16) exp_pair(sc, pc, g, ps0, ps1);
            } else {
                float e0 = 0.f, e1 = 0.f;
                if (g < 16) {
                    const int kb = g >> 3, i = (g & 7) * 2;
                    e0 = __builtin_amdgcn_exp2f(sc[kb][i]);
                    e1 = __builtin_amdgcn_exp2f(sc[kb][i + 1]);
                    asm volatile("" : "+v"(e0), "+v"(e1));      // keep the pair in this gap
                }
                if (g >= SKEW) {
                    const int gp = g - SKEW;
                    const float f0 = e0q[gp % SKEW], f1 = e1q[gp % SKEW];
                    ps0 += f0;
                    ps1 += f1;
                    uint32_t w = pack_bf16x2(f0, f1);
                    asm volatile("" : "+v"(ps0), "+v"(ps1), "+v"(w));
                    pc[gp >> 2][gp & 3] = w;
                }
                if (g < 16) {
                    e0q[g % SKEW] = e0;
                    e1q[g % SKEW] = e1;
                }
                if (g >= 16) {
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
            }
            if (g < 8) {
                if (HAS_PV) o[g & 1] = mfma32(frag[g], __builtin_bit_cast(bf16x8, pp[g >> 1]), o[g & 1]);
            } else {
                if (HAS_S) {
                    const int kb = (g - 8) >> 2, s = (g - 8) & 3;
                    sn[kb] = mfma32(frag[g], qf[s], s == 0 ? negm : sn[kb]);
                }
            }
            read_frag(g + LEAD);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (HAS_S) {
            if ((t + 2) * KB > ntok) mask_tile(t + 1, sn);      // tile t+1 is the ragged last one (wave-uniform, once)
        }
        float ps = ps0 + ps1;
        if (__any(!(ps <= P_LIMIT))) {      // rare: some probability of tile t is too large against the current reference
            const float delta = fmaxf(row_max(sc), 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            m_run += delta;
            l_run *= alpha;                 // l, O (all finished PV products, P(t-1) included) and negm ...
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] -= delta;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sc[kb][r] -= delta;                 // ... the kept scores of tile t
                    if (HAS_S) sn[kb][r] -= delta;      // ... and those of tile t+1, taken against the old reference
                }
            ps0 = 0.f;
            ps1 = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) exp_pair(sc, pc, g, ps0, ps1);
            ps = ps0 + ps1;
        }
        l_run += ps;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS reads of the iteration are complete
    };

    // ---- prologue: K(0 .. DEPTH) and V(0 .. DEPTH-2) in flight = what iterations -1 .. -DEPTH+... would have staged; S(0);
    // the reference is tile 0's row maximum ----
    auto clampt = [&](int t) { return t < ntiles ? t : ntiles - 1; };
#pragma unroll
    for (int i = 0; i <= DEPTH; ++i) stage_k(clampt(i));
#pragma unroll
    for (int i = 0; i + 1 < DEPTH; ++i) stage_v(clampt(i));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave_active) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 kf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) kf[s] = k_frag(0, kb, s);     // K(0) is in slot 0
            sA[kb] = negm;
#pragma unroll
            for (int s = 0; s < 4; ++s) sA[kb] = mfma32(kf[s], qf[s], sA[kb]);
        }
        if (KB > ntok) mask_tile(0, sA);
        const float mx = row_max(sA);
        m_run = mx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            negm[r] = -mx;
            sA[0][r] -= mx;
            sA[1][r] -= mx;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    const std::true_type yes{};
    const std::false_type no{};
    // even t: scores sA -> fragments pA;  odd t: sB -> pB
    if (ntiles == 1) {
        iteration(0, sA, sB, pB, pA, no, no);
    } else {
        iteration(0, sA, sB, pB, pA, no, yes);
        int t = 1;
        for (; t + 2 < ntiles; t += 2) {        // steady state, two tiles per trip (register roles alternate)
            iteration(t, sB, sA, pA, pB, yes, yes);
            iteration(t + 1, sA, sB, pB, pA, yes, yes);
        }
        if (t + 1 < ntiles) {
            iteration(t, sB, sA, pA, pB, yes, yes);
            ++t;
        }
        // t == ntiles - 1
        if (t & 1) iteration(t, sB, sA, pA, pB, yes, no);
        else iteration(t, sA, sB, pB, pA, yes, no);
    }

    // ---- drain: PV(ntiles-1) ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // V(ntiles-1) landed for everyone; the K ring is free from here on (the clamped
                                           // re-reads all went to V / K slots of tiles >= ntiles - 1 - ... and have landed)
    if (wave_active) {
        const int t = ntiles - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const bf16x8 pfrag = __builtin_bit_cast(bf16x8, (t & 1) ? pB[ks] : pA[ks]);
                o[db] = mfma32(v_frag((t & (RING - 1)) * KV_TILE, ks, db), pfrag, o[db]);
            }
    }

    // ---- normalise and write ctx[b*ntok + q][head*64 + d] as whole 128-byte rows through a wave-private patch ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int b = pair / p.heads, head = pair - b * p.heads;
    const int dm = p.heads * 64;
    char* patch = smem + wave * 4096;       // K ring (16 KiB): no reads of it after the barrier above
    const int q0 = qt * QB + wave * QW;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 w = {pack_bf16x2(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv),
                             pack_bf16x2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv)};
            *reinterpret_cast<uint2*>(patch + lr * 128 + (((db * 4 + g) ^ (lr & 7)) << 4) + lh * 8) = w;
        }
    asm volatile("" ::: "memory");          // keep the compiler from crossing the 8-byte writes and the 16-byte reads
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + (lane >> 3);
        const uint4 v = *reinterpret_cast<const uint4*>(patch + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
        if (q0 + row < ntok)
            *reinterpret_cast<uint4*>(p.ctx + ((long)b * ntok + q0 + row) * dm + head * 64 + (lane & 7) * 8) = v;
    }
    if (qrow < ntok && p.lse != nullptr && lh == 0) p.lse[(long)pair * ntok + qrow] = m_run + __builtin_amdgcn_logf(l_tot);
}

int launch_attention_pipe(const AttnParams& p, hipStream_t s) {
    using namespace apipe;
    const int nq = (p.ntok + QB - 1) / QB;
    const int npairs = p.B * p.heads;
    const int grid = ((npairs + 7) / 8) * 8 * nq;
    static PerDeviceOnce once;
    if (once.first())
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_pipe_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(attn_fwd_pipe_kernel, dim3(grid), dim3(NW * 64), LDS_BYTES, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace dseg
