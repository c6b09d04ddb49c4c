// Fused multi-head attention forward (flash-style, head_dim = 64) for gfx950.
//
// Replaces the reference's materialised attention -- q@k^T, *scale, softmax, @v on a [B,H,N,N] fp32 tensor
// (vision_transformer.py:85,101,104; 297 MiB per frame per block @480) -- with a tiled kernel that never
// writes the N x N scores: per 128-query workgroup, K / V^T tiles of 64 keys stream through LDS (LDS-DMA,
// double-buffered, XOR-swizzled 128-byte rows), scores live in MFMA accumulators, softmax is online.
//
// Orientation ("keys on the MFMA rows"): S^T = K . Q^T with v_mfma_f32_32x32x16_bf16, so each lane owns one
// query column and 32 of the tile's 64 scores -> row max / row sum are in-lane plus one cross-half exchange.
// The K fragment rows are read through the bit-2<->bit-3 swap sigma(i): accumulator registers 8s..8s+7 of a
// lane half h then hold keys 16s+8h+0..7 in natural order, which is exactly the B-operand fragment of
// O^T += V^T . P^T -- the probabilities never leave registers (no LDS round trip, no permutes).
// V stays row-major [key][d] like K (same LDS-DMA loader, same swizzled image); the A operand of the PV product
// (rows = d, k = keys) is read column-wise with gfx950's transposing ds_read_b64_tr_b16 (two reads of 4 keys per
// fragment).  Q arrives pre-scaled by head_dim^-0.5 * log2(e); exp is v_exp_f32 (2^x).
//
// VALU diet (the first version was bound by VALU issue, 193 instructions per 64-key tile and wave, MFMA pipe
// 43 % busy): the running maximum rides in the score accumulator's initial value, the rescale is deferred behind a
// threshold (VAR bit 0: behind an overflow check on the row sums, no per-tile maximum at all) -- what is left per tile
// is 32 v_exp, 32 v_add and 16 v_cvt_pk.
//
// PLANES = 2 (parity mode): Q, K, V and P are bf16 hi+lo pairs; each product is 3 MFMAs.
#include <type_traits>

#include "attn_common.h"
#include "kernels.h"

namespace dseg {

constexpr int QW = 32;          // query rows per wave
constexpr int KB = 64;          // keys per tile
constexpr int KV_TILE = attn::KV_TILE_BYTES;   // [64][64] bf16 slab = 8 KiB

using attn::sigma23;
using attn::tr_frag;
__device__ __forceinline__ int swz2(int row, int chunk) { return attn::swz(row, chunk); }
__device__ __forceinline__ int tile_off2(int row, int chunk) { return attn::tile_off(row, chunk); }

// VAR bit 0: no per-tile row maximum.  The tile's probabilities are exponentiated against the current reference straight
//            away and the in-lane partial row sum (needed anyway) is the overflow detector: if any lane's sum exceeds
//            2^RESCALE_THR the tile is redone the careful way (scores recomputed from the K tile still in LDS, row
//            maximum, rescale).  Saves 16 v_max3 + a cross-half exchange per tile on the vector issue port.
//     bit 1: waves whose 32 query rows are all past ntok (last q-tile of a (batch, head)) only stage and synchronise
// FMT (PLANES == 2 only): FMT_FP16 = Q, K, V, the probabilities and ctx are fp16 hi + lo planes (~22 bits).  The probabilities are
//            2^(S - m_run) with the reference m_run trailing the row maximum by at most RESCALE_THR: fp16 holds them up to 2^15.
template <int PLANES, int NW, bool DBG, int VAR, int FMT = FMT_BF16>     // NW waves per workgroup, each 32 query rows; all share the K/V tiles
__device__ __forceinline__ void attn_fwd_body(const AttnParams& p) {
    static_assert(FMT == FMT_BF16 || PLANES == 2, "one fp16 plane: attention_z.hip");
    constexpr int QB = NW * QW;
    const int dbg = DBG ? p.dbg : 0;        // timing ablations are compiled out of the production instantiation
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = PLANES * 2 * KV_TILE;   // per plane: K slab + V slab
    static_assert(NW * 4096 <= STAGE_BYTES, "the O-store epilogue gives every wave a 4 KiB patch of one ring slot");
    // (a 3-slot ring with counted vmcnt measured 4 % slower: LDS 48 KiB per workgroup and a dynamic slot index)

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

    // ---- Q fragments (B operand: k = d, col = query) straight from global into registers ----
    int qrow = qt * QB + wave * QW + lr;
    const int qrow_c = qrow < ntok ? qrow : ntok - 1;
    bf16x8 qf[PLANES][4];
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[pl][s] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const uint4*>(Qg + pl * p.qkv_plane + (long)qrow_c * 64 + s * 16 + lh * 8));

    // the compiler must not carry "Q loads pending" into the tile loop: with the loader's uncounted LDS-DMA in flight its
    // own vmcnt waits there would drain the prefetch every tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[pl][s]));

    // K/V tile loader: per-lane byte offsets inside a 64-row slab are fixed for the whole kernel; a tile costs one scalar
    // base update per slab and no vector address arithmetic
    constexpr int NPIECE = (16 + NW - 1) / NW;
    uint32_t soff[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int row = ((wave + i * NW) & 7) * 8 + (lane >> 3);
        soff[i] = (uint32_t)((row * 64 + swz2(row, lane & 7) * 8) * 2);
    }
    auto stage = [&](int st, int key0) {
        char* sbase = smem + st * STAGE_BYTES;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
            const char* kb = reinterpret_cast<const char*>(Kg + pl * p.qkv_plane + (long)key0 * 64);
            const char* vb = reinterpret_cast<const char*>(Vg + pl * p.qkv_plane + (long)key0 * 64);
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {
                const int piece = wave + i * NW;              // 0..7: K rows, 8..15: V rows
                if (16 % NW == 0 || piece < 16) {               // (compile-time true for 4 and 8 waves: straight-line issue)
                    // LDS-DMA with a scalar base and a 32-bit lane offset (the builtin form keeps a 64-bit address per lane
                    // and piece and adds the tile offset to each on the vector port, which is the port this loop is bound by).
                    // M0 = LDS destination of the wave; saved and restored around the statement (compiler-reserved).
                    const uint32_t lds_dst = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)(sbase + pl * 2 * KV_TILE + piece * 1024);
                    uint32_t keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep)
                                 : "v"(soff[i]), "s"(piece < 8 ? kb : vb), "s"(lds_dst)
                                 : "memory");
                }
            }
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    // Softmax state.  m_run is the (log2-domain) reference each row's scores are measured against; it enters the
    // score MFMA chain as the accumulator's initial value (negm = -m_run in all 16 registers), so S' = S - m_run
    // costs no VALU.  m_run moves only when a row maximum exceeds it by more than RESCALE_THR (rare after the first
    // tiles), so the O / l rescale is off the steady-state path.
    constexpr float RESCALE_THR = FMT == FMT_FP16 ? 14.f : 16.f;     // P <= 2^16: far from fp32 / bf16 overflow (fp16: a lane's 32-key sum <= 2^15)
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    float m_run = 0.f, l_run = 0.f;

    // K/V tiles are double-buffered in LDS: tile t+1 is in flight (LDS-DMA) while tile t is multiplied.
    const int ntiles = (ntok + KB - 1) / KB;
    stage(0, 0);

    const bool wave_active = qt * QB + wave * QW < ntok;     // wave-uniform
    const int krow_perm = sigma23(lr);
    // transposed V read (ds_read_b64_tr_b16): within each 16-lane group, lane 4q+p addresses key row q, d columns
    // 4p..4p+3 of a 4-key x 16-d block and receives column (lane&15) of the 4 keys.  Group g = lane>>4 covers
    // d = 16*(g&1) + 0..15 of the 32-d block and keys 8*(g>>1) + 0..3 (second read: +4) of the 16-key step.
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_gl = (lane >> 4) & 1;
    // LDS addresses: 4 + 4 per-lane byte offsets; half (kb), key step (ks), plane and ring slot are immediates
    // (the swizzle ignores row bits 4 and 5, so +32 rows = +4096 B and +16 rows = +2048 B exactly)
    int ka[4], va[2][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = tile_off2(krow_perm, s * 2 + lh);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            va[db][h] = KV_TILE + tile_off2(lh * 8 + tr_q + 4 * h, db * 4 + tr_gl * 2 + (tr_p >> 1)) + (tr_p & 1) * 8;

    // one K/V tile; SLOT = t & 1 is a compile-time constant (two tiles per loop trip) so that every LDS address of the body
    // is a loop-invariant register plus an immediate
    auto tile = [&](int t, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile t landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();      // everyone's pieces of tile t landed; everyone is done reading slot (t-1)&1
        if (t + 1 < ntiles && !(dbg & 2)) stage(SLOT ^ 1, (t + 1) * KB);
        if ((VAR & 2) && !wave_active) return;
        const char* sb = smem + SLOT * STAGE_BYTES;

        // ---- S'^T[key][q] = K . Q^T - m_run: the tile's 8 K fragments are read up front (one LDS latency per tile) ----
        f32x16 sacc[2];
        auto scores = [&]() {
            bf16x8 kf[PLANES][2][4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    kf[0][kb][s] = lds_frag(sb + ka[s] + kb * 4096);
                    if (PLANES == 2) kf[PLANES - 1][kb][s] = lds_frag(sb + ka[s] + kb * 4096 + 2 * KV_TILE);
                }
            sacc[0] = negm;             // C input of the chains
            sacc[1] = negm;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (PLANES == 2) {
                        sacc[kb] = mfma32f<FMT>(kf[PLANES - 1][kb][s], qf[0][s], sacc[kb]);
                        sacc[kb] = mfma32f<FMT>(kf[0][kb][s], qf[PLANES - 1][s], sacc[kb]);
                    }
                    sacc[kb] = mfma32f<FMT>(kf[0][kb][s], qf[0][s], sacc[kb]);
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
        };
        // P = 2^S' in place; returns this lane's partial row sum (its 32 of the tile's 64 keys)
        auto exponentiate = [&]() {
            float ps = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[kb][r] = (dbg & 1) ? sacc[kb][r] * 0.01f : __builtin_amdgcn_exp2f(sacc[kb][r]);
                    ps += sacc[kb][r];
                }
            return ps;
        };
        // move the reference: everything still measured against the old one is rescaled exactly once (S', O, l, negm)
        auto rescale = [&](float mx) {
            const float delta = (t == 0) ? mx : fmaxf(mx, 0.f);
            // (first tile: O and l are still zero -- and 2^-delta overflows for a row whose scores all lie below -128: 0 * inf)
            const float alpha = (t == 0) ? 1.0f : __builtin_amdgcn_exp2f(-delta);
            m_run += delta;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] -= delta;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] -= delta;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        };
        auto row_max = [&]() {
            float mx = sacc[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
            return fmaxf(mx, __shfl_xor(mx, 32));
        };

        // ---- online softmax (log2 domain), deferred rescale ----
        scores();
        float ps;
        if (VAR & 1) {
            bool careful = (t == 0);
            if (!careful) {
                ps = exponentiate();
                // 2^RESCALE_THR bounds every P of the tile on the fast path; !(<=) also catches inf / NaN sums
                careful = __any(!(ps <= (FMT == FMT_FP16 ? 32768.f : 65536.f)));
                if (careful) scores();
            }
            if (careful) {
                rescale(row_max());
                ps = exponentiate();
            }
        } else {
            const float mx = (dbg & 1) ? sacc[0][0] : row_max();
            if (t == 0 || __any(mx > RESCALE_THR)) rescale(mx);
            ps = exponentiate();
        }

        // ---- P fragments (B operand: k = key, col = query): registers 8*s2..8*s2+7 of sacc[kb] ----
        bf16x8 pf[PLANES][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 hi, lo;
                split2<FMT>(sacc[kb][s2 * 8 + 0], sacc[kb][s2 * 8 + 1], hi.x, lo.x);
                split2<FMT>(sacc[kb][s2 * 8 + 2], sacc[kb][s2 * 8 + 3], hi.y, lo.y);
                split2<FMT>(sacc[kb][s2 * 8 + 4], sacc[kb][s2 * 8 + 5], hi.z, lo.z);
                split2<FMT>(sacc[kb][s2 * 8 + 6], sacc[kb][s2 * 8 + 7], hi.w, lo.w);
                pf[0][kb * 2 + s2] = __builtin_bit_cast(bf16x8, hi);
                if (PLANES == 2) pf[PLANES - 1][kb * 2 + s2] = __builtin_bit_cast(bf16x8, lo);
            }

        // ---- row sums on the VALU: in-lane partial sums of the fp32 probabilities, the two lane halves are joined at the end
        // (a ones-vector MFMA for the sums measured 5-7 % slower: the matrix pipe is the scarce unit)
        l_run += ps;

        // ---- O^T[d][q] += V^T . P^T  (V^T fragments by transposing LDS reads) ----
        if (!(dbg & 4))
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off0 = va[db][0] + ks * 2048, off1 = va[db][1] + ks * 2048;
                const bf16x8 vhi = tr_frag(sb + off0, sb + off1);
                if (PLANES == 2) {
                    const bf16x8 vlo = tr_frag(sb + 2 * KV_TILE + off0, sb + 2 * KV_TILE + off1);
                    o[db] = mfma32f<FMT>(vlo, pf[0][ks], o[db]);
                    o[db] = mfma32f<FMT>(vhi, pf[PLANES - 1][ks], o[db]);
                }
                o[db] = mfma32f<FMT>(vhi, pf[0][ks], o[db]);
            }
        }

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS reads of tile t are complete
    };
    {
        int t = 0;
        for (; t + 1 < ntiles; t += 2) {
            tile(t, std::integral_constant<int, 0>{});
            tile(t + 1, std::integral_constant<int, 1>{});
        }
        if (t < ntiles) tile(t, std::integral_constant<int, 0>{});
    }

    // ---- normalise and write ctx[b*ntok + q][head*64 + d] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int b = pair / p.heads, head = pair - b * p.heads;
    const int dm = p.heads * 64;
    {
        // Whole-row stores: the slot the last tile did NOT use is free for every wave (all passed the last barrier, nothing
        // was staged after it); each wave owns a [32 rows][128 B] patch of it, 16-byte chunks XORed with row & 7.
        // Row-per-lane 8-byte stores touch 32 lines per instruction and are store-issue bound; 8 lanes x 16 B per row are not.
        char* patch = smem + (ntiles & 1) * STAGE_BYTES + wave * 4096;
        const int q0 = qt * QB + wave * QW;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 hi, lo;
                    split2<FMT, true>(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv, hi.x, lo.x);
                    split2<FMT, true>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv, hi.y, lo.y);
                    *reinterpret_cast<uint2*>(patch + lr * 128 + (((db * 4 + g) ^ (lr & 7)) << 4) + lh * 8) = pl == 0 ? hi : lo;
                }
            // one wave's LDS requests execute in order; the compiler must keep them in order too (the 8-byte writes and the
            // 16-byte reads have unrelated types, so alias analysis alone would let them cross)
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
        if (qrow < ntok && p.lse != nullptr && lh == 0) p.lse[(long)pair * ntok + qrow] = m_run + __builtin_amdgcn_logf(l_tot);
    }
}

template <int PLANES, int NW, bool DBG, int VAR, int FMT = FMT_BF16>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(AttnParams p) {
    attn_fwd_body<PLANES, NW, DBG, VAR, FMT>(p);
}

template <int PLANES, int NW, bool DBG, int VAR, int FMT = FMT_BF16>
static int launch_attn(const AttnParams& p, hipStream_t s) {
    const int nq = (p.ntok + NW * QW - 1) / (NW * QW);
    const int npairs = p.B * p.heads;
    const int grid = ((npairs + 7) / 8) * 8 * nq;
    const size_t lds = (size_t)2 * PLANES * 2 * KV_TILE;
    hipLaunchKernelGGL((attn_fwd_kernel<PLANES, NW, DBG, VAR, FMT>), dim3(grid), dim3(NW * 64), lds, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int PLANES>
static int launch_attn_planes(const AttnParams& p, hipStream_t s) {
    switch (options().attn_variant & 3) {
        case 1: return launch_attn<PLANES, 4, false, 1>(p, s);
        case 2: return launch_attn<PLANES, 4, false, 2>(p, s);
        case 3: return launch_attn<PLANES, 4, false, 3>(p, s);
        default: return launch_attn<PLANES, 4, false, 0>(p, s);
    }
}

int launch_attention(const AttnParams& p0, hipStream_t s) {
    AttnParams p = p0;
    p.dbg = options().attn_dbg;
    if (p.npad % KB != 0 || p.npad < p.ntok) {
        dinoseg_set_error("attention: npad=%d must be a multiple of 64 and >= ntok=%d", p.npad, p.ntok);
        return -1;
    }
    if (p.fmt == FMT_FP16) {      // fp16 operands (inference: no log-sum-exp for a backward): one plane = the zero-reference kernel with bf16
        if (p.lse != nullptr) {   // P / V; hi + lo planes = this file's kernel, every operand fp16
            dinoseg_set_error("attention: the fp16 operand format is inference-only");
            return -1;
        }
        if (p.planes == 1) return launch_attention_z(p, s);
        if (p.planes == 2 && p.v_bf16) return launch_attention_z(p, s);      // V on bf16 planes: the zero-reference kernels
        if (p.planes == 2) return launch_attn<2, 4, false, 3, FMT_FP16>(p, s);
        dinoseg_set_error("attention: planes must be 1 or 2");
        return -1;
    }
    if (p.dbg != 0 && p.planes == 1) return launch_attn<1, 4, true, 0>(p, s);      // ablation build (tools/bench_ops.py)
    if (p.planes == 1 && (options().attn_variant & 8)) return launch_attention_z(p, s);      // zero-reference, 4 waves / SIMD
    // hi + lo planes: the zero-reference kernels from two rounds of 256-query workgroups on (attention_za.hip), or when asked for (bit 4)
    // (inference only -- lse == nullptr: the training forward keeps ONE arithmetic at every batch size: tests/test_train_gpu.py, a batch-8
    //  step is the mean of eight single-frame steps to 1e-5 of every gradient)
    if (p.planes == 2 && ((options().attn_variant & 16) ||
                          (p.lse == nullptr && attention_x3_za(p.dispatch_B > 0 ? p.dispatch_B : p.B, p.heads, p.ntok))))
        return launch_attention_z(p, s);
    if (p.planes == 1) return launch_attn_planes<1>(p, s);
    if (p.planes == 2) return launch_attn_planes<2>(p, s);
    dinoseg_set_error("attention: planes must be 1 or 2");
    return -1;
}

}  // namespace dseg
