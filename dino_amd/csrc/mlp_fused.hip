// Fused transformer MLP, bf16 mode, ViT-S width:   x += fc2(gelu(fc1(LayerNorm2(x))))      (vision_transformer.py:135 -> :59-65)
//
// One launch per block instead of LN+fc1+GELU and fc2+residual: the [M, 1536] hidden activation never exists in HBM (it was 31 %
// of a layer's bytes: written by fc1's epilogue, re-read by fc2), one LayerNorm prologue and one epilogue per layer disappear,
// and the residual stream is read once and written once (it is the accumulators' initial value).
//
// Skeleton = the attention kernel's (attention_z.hip): a W1 tile of 32 hidden units plays K, the matching W2^T tile plays V, GELU
// plays exp, and the fc1 accumulator IS the B operand of the fc2 product (no LDS round trip, no lane movement):
//   workgroup = 4 waves = 128 rows of x, ONE wave per SIMD with the whole 512-register file; wave = 32 rows.
//   per wave, resident in registers for the whole item:
//     xn[24]  LayerNorm2(x) of its 32 rows as bf16 B-operand fragments (k = 16 s + 8 h + j on lane (row, h)):   96 registers
//     o[12]   out^T[384][32] accumulators, initialised with x + b2 (the residual), k order = the same as xn's:     192 registers
//   per hidden tile t (32 units; 48 tiles):
//     S^T[32 hid][32 rows] = W1_t . xn^T + b1_t     24 MFMAs 32x32x16, A fragments from LDS, C initialised with the bias
//     P = bf16(gelu(S))                               16 values per lane, in registers
//     o^T += W2^T_t . P^T                             24 MFMAs, A fragments from LDS, B = P
//   One wave per SIMD has nobody to hide behind, so a tile step is written as 48 explicit MFMA "gaps" (the compiler's own order,
//   with or without sched_group_barrier, left the matrix pipe idle ~2/3 of the time: every MFMA waited for an LDS read issued one
//   instruction earlier, the GELU ran as one block): gap g issues the fragment read of gap g+6 (inline asm, hand-counted
//   lgkmcnt), its MFMA, every fourth gap one LDS-DMA piece of the weight stream, and 3-4 vector instructions of the tile's 16 GELU
//   evaluations, each from a different evaluation (see "the GELU stream" below).  Phase A (gaps 0-23) is fc1 of tile t+1, phase
//   B (24-47) fc2 of tile t.
// Weights: both matrices are re-packed once (launch_pack_mlp) in exactly the order the MFMAs consume them -- [tile][48 fragments]
// [64 lanes][8 bf16], W1 rows and W2 output rows permuted by sigma23 (bits 2 <-> 3) so that accumulator registers 8s..8s+7 of a lane
// hold 8 consecutive k of the next product -- so a fragment is one linear 1-KiB LDS-DMA piece and one conflict-free ds_read_b128.
// Ring: 3 slots per matrix (144 KiB); at tile t the workgroup issues W1(t+3) and W2(t+2) (12 pieces per wave), waits with a
// counted vmcnt(12) for what it issued one tile earlier... two tiles of lead; one barrier per tile; the stream is cyclic over the
// 48 tiles and runs on across the items of the persistent walk.
#include <type_traits>

#include "mlp_common.h"

namespace dseg {

namespace mf {
constexpr int D = 384, F = 1536, HT = 32, NT = F / HT, NKS = D / 16, NDB = D / 32;
constexpr int NW = 4, BM = NW * 32, THREADS = NW * 64;
constexpr int W_TILE = NKS * 1024;                 // bytes of one matrix's fragments of one hidden tile (24 KiB, both matrices)
constexpr int TILE_BYTES = 2 * W_TILE;             // packed copy: W1 fragments then W2 fragments
constexpr int RING = 3;
constexpr int W1_OFF = 0, W2_OFF = RING * W_TILE;
constexpr int B1_OFF = 2 * RING * W_TILE;          // b1 [F] fp32
constexpr int B2_OFF = B1_OFF + F * 4;             // b2, gamma, beta [D] fp32 each
constexpr int G_OFF = B2_OFF + D * 4, BE_OFF = G_OFF + D * 4;
constexpr int Q_OFF = BE_OFF + D * 4;              // next item of the persistent walk (one int, broadcast to the workgroup)
constexpr int LDS_BYTES = Q_OFF + 16;
static_assert(NDB * 2 == NKS && LDS_BYTES <= 160 * 1024, "fragment counts / LDS budget");
constexpr int PIECES = W_TILE / 1024 / 2;          // LDS-DMA pieces per wave and tile: waves 0,1 carry W1, waves 2,3 carry W2
static_assert(PIECES == 12, "three groups of four pieces");
}  // namespace mf

#ifndef MF_ABL
#define MF_ABL 0      // compile-time ablation bits for A/B builds (wrong results): 1 no GELU, 2 no W DMA, 4 no fc1 MFMAs, 8 no fc2 MFMAs,
#endif                // 16 no fragment reads, 32 no prologue loads / epilogue stores, 64 no epilogue stores, 128 no prologue loads

__global__ __launch_bounds__(256) void pack_mlp_kernel(const float* __restrict__ W1, const float* __restrict__ W2,
                                                       bf16_t* __restrict__ dst, long total, int fmt) {
    using namespace mf;
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

long mlp_fused_pack_elems(int Dm, int Fh) { return Dm == mf::D && Fh == mf::F ? (long)mf::NT * mf::TILE_BYTES / 2 : 0; }

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

__global__ __launch_bounds__(mf::THREADS, 1) void mlp_fused_kernel(MlpFusedParams p) {
    using namespace mf;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
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
            reinterpret_cast<f32x4*>(smem + B2_OFF + which * D * 4)[j] = reinterpret_cast<const f32x4*>(src)[j];
        }
    }

    // ---- weight stream.  Waves 0,1 carry W1 (fragments 12 w .. 12 w + 11 of a tile), waves 2,3 carry W2.
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    const bool carries_w1 = wave < 2;                              // wave-uniform
    const uint32_t my_frag0 = (uint32_t)(wave & 1) * PIECES;       // first fragment (within the matrix's 24) this wave copies
    const uint32_t src_off0 = (carries_w1 ? 0u : (uint32_t)W_TILE) + my_frag0 * 1024u;      // byte offset inside a packed tile
    const uint32_t dst_off0 = (carries_w1 ? (uint32_t)W1_OFF : (uint32_t)W2_OFF) + my_frag0 * 1024u;
    const uint32_t lane16 = (uint32_t)lane * 16;
    const uint64_t wp = reinterpret_cast<uint64_t>(p.Wp);
    auto uniform64 = [](uint64_t v) __attribute__((always_inline)) -> uint64_t {
        return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v) |
               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
    };
    // all 12 pieces of packed tile `tile` into ring slot `slot` of this wave's matrix at once (ring prologue only)
    auto issue_tile = [&](int tile, int slot) __attribute__((always_inline)) {
        if (MF_ABL & 2) return;
        const uint32_t so = (uint32_t)tile * (uint32_t)TILE_BYTES + src_off0;
        const uint32_t dof = lds_base + dst_off0 + (uint32_t)slot * (uint32_t)W_TILE;
#pragma unroll
        for (int g = 0; g < 3; ++g) mf_dma4(lane16, uniform64(wp + so + g * 4096), __builtin_amdgcn_readfirstlane(dof + g * 4096));
    };
    // ring prologue = what the last three tiles of a previous item would have issued: W1(0), W1(1), W1(2), W2(0), W2(1)
    if (carries_w1) issue_tile(0, 0);
    issue_tile(carries_w1 ? 1 : 0, carries_w1 ? 1 : 0);
    issue_tile(carries_w1 ? 2 : 1, carries_w1 ? 2 : 1);

    const char* const frag_rd = smem + lane16;          // + ring slot + 1024 * fragment
    const float* const sB1 = reinterpret_cast<const float*>(smem + B1_OFF);

    f32x16 o[NDB];
    bf16x8 xn[NKS];

    // S^T of hidden tile 0 from W1 ring slot 0 at an item's start (compiler-scheduled: once per item)
    auto fc1_first = [&]() __attribute__((always_inline)) -> f32x16 {
        f32x16 s;
        const float* bp = sB1 + lh * 8;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(bp), c1 = *reinterpret_cast<const f32x4*>(bp + 4);
        const f32x4 c2 = *reinterpret_cast<const f32x4*>(bp + 16), c3 = *reinterpret_cast<const f32x4*>(bp + 20);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[e] = c0[e];
            s[4 + e] = c1[e];
            s[8 + e] = c2[e];
            s[12 + e] = c3[e];
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) s = mfma32(lds_frag(frag_rd + W1_OFF + ks * 1024), xn[ks], s);
        return s;
    };

    // ---- the GELU stream.  One wave per SIMD: a dependent vector instruction issued right behind its producer stalls the wave
    // (and the MFMA behind it), so no gap ever holds two instructions of the same evaluation.  Element n of a tile step
    // (n = 0..7: registers 8..15 of the current S -> fragment p1 of this tile's fc2; n = 8..15: registers 0..7 of the next S ->
    // p0 of the next tile) starts at gap SG(n) and issues instruction i of gelu_fast (common.h; 10 instructions + a pack per pair)
    // at gap SG(n) + i: every gap carries 3-4 instructions of 3-4 different elements, dependent ones are >= 32 cycles apart.  The
    // lower half starts two gaps after the MFMA that completes S; the last elements run on into the next step's first gaps.
    float ex[16], ea[16], eb[16], ec[16];
    uint32_t pd[4] = {0u, 0u, 0u, 0u}, qd[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int n = 0; n < 16; ++n) ex[n] = ea[n] = eb[n] = ec[n] = 0.f;
    auto gelu_op = [&](auto n_tag, auto i_tag, const f32x16& s_up, const f32x16& s_lo) __attribute__((always_inline)) {
        constexpr int N = decltype(n_tag)::value, I = decltype(i_tag)::value;
        if constexpr (I == 0) ex[N] = N < 8 ? s_up[8 + N] : s_lo[N - 8];
        if (MF_ABL & 1) {
            if constexpr (I == 10 && (N & 1)) (N < 8 ? pd : qd)[(N & 7) >> 1] = pack_bf16x2(ex[N - 1], ex[N]);
            return;
        }
        if constexpr (I == 1) ea[N] = __builtin_amdgcn_fmed3f(ex[N], -8.0f, 8.0f);
        if constexpr (I == 2) eb[N] = ea[N] * ea[N];
        if constexpr (I == 3) ec[N] = fmaf(1.01537542e-3f, eb[N], -1.06782573e-1f);
        if constexpr (I == 4) ec[N] = fmaf(ec[N], eb[N], -2.30111381f);
        if constexpr (I == 5) ec[N] = ec[N] * ea[N];
        if constexpr (I == 6) ec[N] = __builtin_amdgcn_exp2f(ec[N]);
        if constexpr (I == 7) ec[N] = 1.0f + ec[N];
        if constexpr (I == 8) ec[N] = __builtin_amdgcn_rcpf(ec[N]);
        if constexpr (I == 9) ex[N] = ex[N] * ec[N];
        if constexpr (I == 10 && (N & 1)) (N < 8 ? pd : qd)[(N & 7) >> 1] = pack_bf16x2(ex[N - 1], ex[N]);
    };
    // vector work of gap G.  WHICH: 0 = a tile step (elements of this step + those the previous step left unfinished),
    // 1 = only this step's lower half (an item's start: what "step -1" would have done), 2 = no lower half (an item's last step)
    auto valu_gap = [&](auto g_tag, auto which_tag, const f32x16& s_up, const f32x16& s_lo) __attribute__((always_inline)) {
        constexpr int G = decltype(g_tag)::value, WHICH = decltype(which_tag)::value;
        mf_for(std::make_integer_sequence<int, 16>{}, [&](auto n_tag) __attribute__((always_inline)) {
            constexpr int N = decltype(n_tag)::value;
            constexpr int SG = N < 8 ? 3 * N : 26 + 3 * (N - 8);
            if constexpr (G >= SG && G - SG <= 10 && !(WHICH == 1 && N < 8) && !(WHICH == 2 && N >= 8))
                gelu_op(n_tag, std::integral_constant<int, G - SG>{}, s_up, s_lo);
            if constexpr (WHICH != 1 && N >= 8 && G + 48 - SG <= 10)         // carried over from the previous step
                gelu_op(n_tag, std::integral_constant<int, G + 48 - SG>{}, s_up, s_lo);
        });
    };
    constexpr int G_P0 = 10, G_P1 = 32;     // first gaps at which p0 (the previous step's lower half) / p1 are complete
    auto frag4 = [](const uint32_t (&d)[4]) __attribute__((always_inline)) -> bf16x8 {
        const uint4 u = {d[0], d[1], d[2], d[3]};
        return __builtin_bit_cast(bf16x8, u);
    };

    const float* const sB2 = reinterpret_cast<const float*>(smem + B2_OFF);
    const float* const sG = reinterpret_cast<const float*>(smem + G_OFF);
    const float* const sBe = reinterpret_cast<const float*>(smem + BE_OFF);

    // Persistent walk over the 128-row items.  With a work counter (p.queue, zeroed per launch) the next item is drawn when the
    // current one is finished -- every item costs the same, but workgroups drift apart (HBM phases, other kernels on the chip)
    // and the counter keeps the tail short; without one the items are dealt round-robin.
    for (int item = blockIdx.x; item < nitems;) {
        // ================= prologue: this lane's half of row (item*128 + wave*32 + lr): x[16 s + 8 lh + 0..7], s = 0..23, straight
        // into the accumulators (o[db][8 s2 + j] <-> s = 2 db + s2): they start from the residual
        const int row = item * BM + wave * 32 + lr;
        const int row_c = row < M ? row : M - 1;
        float* const xrow = p.X + (long)row_c * p.ldx + lh * 8;
        if (!(MF_ABL & (32 | 128))) {
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(xrow + s * 16);
                const f32x4 b = *reinterpret_cast<const f32x4*>(xrow + s * 16 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[s >> 1][(s & 1) * 8 + e] = a[e];
                    o[s >> 1][(s & 1) * 8 + 4 + e] = b[e];
                }
            }
        } else {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] = (float)(lane + r);
        }
        // LayerNorm statistics (two-pass, fp32; vision_transformer.py:303: eps 1e-6): the two lanes of a row hold half a row each
        float sum = 0.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 4) part += (o[db][r] + o[db][r + 1]) + (o[db][r + 2] + o[db][r + 3]);
            sum += part;
        }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / D);
        float qv = 0.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dlt = o[db][r] - mean;
                part = fmaf(dlt, dlt, part);
            }
            qv += part;
        }
        qv += __shfl_xor(qv, 32);
        const float rstd = 1.0f / sqrtf(qv * (1.0f / D) + p.eps);
        __syncthreads();        // (first item: gamma / beta / biases are in LDS; later items: nothing -- kept for simplicity)
        // (the constants are loop-invariant: without the opaque offset the compiler hoists all 576 of them out of the item loop
        //  and spills them)
        int kc = lh * 8;
        asm volatile("" : "+v"(kc));
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const int k0 = s * 16 + kc;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(sG + k0), g1 = *reinterpret_cast<const f32x4*>(sG + k0 + 4);
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(sBe + k0), e1 = *reinterpret_cast<const f32x4*>(sBe + k0 + 4);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(sB2 + k0), c1 = *reinterpret_cast<const f32x4*>(sB2 + k0 + 4);
            float y[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = o[s >> 1][(s & 1) * 8 + e];
                const float x1 = o[s >> 1][(s & 1) * 8 + 4 + e];
                y[e] = (x0 - mean) * rstd * g0[e] + e0[e];
                y[4 + e] = (x1 - mean) * rstd * g1[e] + e1[e];
                o[s >> 1][(s & 1) * 8 + e] = x0 + c0[e];        // the fc2 bias joins the residual
                o[s >> 1][(s & 1) * 8 + 4 + e] = x1 + c1[e];
            }
            uint4 u;
            u.x = pack_bf16x2(y[0], y[1]);
            u.y = pack_bf16x2(y[2], y[3]);
            u.z = pack_bf16x2(y[4], y[5]);
            u.w = pack_bf16x2(y[6], y[7]);
            xn[s] = __builtin_bit_cast(bf16x8, u);
        }

        // the ring's W1(0..2), W2(0..1) were issued by the previous item's last tiles (or the ring prologue): all of it has landed
        // (the row loads above were waited for with vmcnt(0): loads and LDS-DMA retire in order); make it visible to every wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        f32x16 sa = fc1_first(), sb;
        // what "tile step -1" would have left behind: the lower half of S(0) in the GELU stream, as far as gap 47
        mf_for(std::make_integer_sequence<int, 22>{}, [&](auto i_tag) __attribute__((always_inline)) {
            valu_gap(std::integral_constant<int, 26 + decltype(i_tag)::value>{}, std::integral_constant<int, 1>{}, sa, sa);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        // Per-lane constants of the tile loop, recomputed per item from an opaque zero: values that live across the item loop are
        // what the register allocator spills at the prologue's pressure peak, and a reload inside the tile loop is a scratch
        // load -- the compiler then waits vmcnt(0) for it at the loop head and drains the weight ring every iteration.
        uint32_t zero = 0;
        asm volatile("" : "+v"(zero));
        const uint32_t lane_i = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
        const uint32_t lane16_i = lane_i * 16, lh_i = lane_i >> 5;
        const uint32_t frag_rd_i = lds_base + lane16_i;
        const float* const b1_lane_p = sB1 + attn::sigma23((int)(lane_i & 31));     // + HT * tile: b1 of MFMA A row lr
        const uint4 ones_u = {lh_i == 0 ? 0x3F803F80u : 0u, lh_i == 0 ? 0x00003F80u : 0u, 0u, 0u};     // k = 0, 1, 2 are 1.0

        int r1 = 1, r2 = 0;            // ring slots: W1 of tile t+1, W2 of tile t
#ifndef MF_RA
#define MF_RA 6
#endif
        constexpr int RA = MF_RA, NFR = RA + 1;      // fragment read-ahead in gaps; fragment registers
        auto step = [&](f32x16& s_cur, f32x16& s_nxt, int t, auto last_tag) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(last_tag)::value;
            // ---- ahead of the barrier: b1 of tile t+1 as an A fragment (hi + mid + lo = the fp32 value exactly; times a ones
            // fragment it is the accumulators' initial value)
            bf16x8 bias_frag;
            if (!LAST) {
                const float bj = b1_lane_p[(t + 1) * HT];
                const uint32_t hi = pack_bf16x2(bj, 0.f);
                const float r1f = bj - bf16_lo_to_f32(hi);
                const uint32_t mid = pack_bf16x2(r1f, 0.f);
                const uint32_t lo = pack_bf16x2(r1f - bf16_lo_to_f32(mid), 0.f);
                const uint4 fu = {lh_i == 0 ? ((hi & 0xFFFFu) | (mid << 16)) : 0u, lh_i == 0 ? (lo & 0xFFFFu) : 0u, 0u, 0u};
                bias_frag = __builtin_bit_cast(bf16x8, fu);
            }
            __builtin_amdgcn_sched_barrier(0);
            // pieces issued during tile t-1 may stay in flight; those of tile t-2 (W1(t+1), W2(t)) have landed
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            __builtin_amdgcn_s_barrier();      // ... for every wave; and every wave is done reading the slots refilled below
            // W1(t+3) -> the slot fc1(t) read (slot of t+1 minus one); W2(t+2) -> the slot fc2(t-1) read: one piece every 4th gap
            uint64_t gsb[3];
            uint32_t gld[3];
            {
                const int t1 = t + 3 >= NT ? t + 3 - NT : t + 3, t2 = t + 2 >= NT ? t + 2 - NT : t + 2;
                const int s1 = r1 == 0 ? 2 : r1 - 1, s2 = r2 == 0 ? 2 : r2 - 1;
                const uint32_t so = (uint32_t)(carries_w1 ? t1 : t2) * (uint32_t)TILE_BYTES + src_off0;
                const uint32_t dof = lds_base + dst_off0 + (uint32_t)(carries_w1 ? s1 : s2) * (uint32_t)W_TILE;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    gsb[g] = uniform64(wp + so + g * 4096);
                    gld[g] = __builtin_amdgcn_readfirstlane(dof + g * 4096);
                }
            }
            const uint32_t a1 = frag_rd_i + W1_OFF + (uint32_t)r1 * W_TILE, a2 = frag_rd_i + W2_OFF + (uint32_t)r2 * W_TILE;
            bf16x8 fr[NFR];
            bf16x8 p0, p1;
            // fragment of gap G (fc2: all k-step-0 products first -- they need only p0 -- then the k-step-1 products)
            auto issue_read = [&](auto g_tag) __attribute__((always_inline)) {
                constexpr int G = decltype(g_tag)::value;
                if (MF_ABL & 16) return;
                if constexpr (G < 24) mf_rd<G * 1024>(fr[G % NFR], a1);
                else if constexpr (G < 36) mf_rd<(2 * (G - 24)) * 1024>(fr[G % NFR], a2);
                else mf_rd<(2 * (G - 36) + 1) * 1024>(fr[G % NFR], a2);
            };
            constexpr int G0 = LAST ? 24 : 0;
            mf_for(std::make_integer_sequence<int, RA>{}, [&](auto i_tag) __attribute__((always_inline)) {
                issue_read(std::integral_constant<int, G0 + decltype(i_tag)::value>{});
            });
            if (!LAST) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                s_nxt = mfma32(bias_frag, __builtin_bit_cast(bf16x8, ones_u), z);
            } else {
                // no next tile: the previous step's last two GELU stages, then this tile's upper half, with nothing to hide behind
                // (and the six stream pieces the missing gaps 0..23 would have issued)
                if (!(MF_ABL & 2)) {
                    mf_dma1<0>(lane16_i, gsb[0], gld[0]);
                    mf_dma1<1024>(lane16_i, gsb[0], gld[0]);
                    mf_dma1<2048>(lane16_i, gsb[0], gld[0]);
                    mf_dma1<3072>(lane16_i, gsb[0], gld[0]);
                    mf_dma1<0>(lane16_i, gsb[1], gld[1]);
                    mf_dma1<1024>(lane16_i, gsb[1], gld[1]);
                }
                mf_for(std::make_integer_sequence<int, G_P1>{}, [&](auto i_tag) __attribute__((always_inline)) {
                    valu_gap(i_tag, std::integral_constant<int, 2>{}, s_cur, s_cur);
                });
                p0 = frag4(qd);
                p1 = frag4(pd);
            }
            __builtin_amdgcn_sched_barrier(0);
            mf_for(std::make_integer_sequence<int, 48 - G0>{}, [&](auto i_tag) __attribute__((always_inline)) {
                constexpr int G = G0 + decltype(i_tag)::value;
                if constexpr (G + RA < 48) issue_read(std::integral_constant<int, G + RA>{});
                if (MF_ABL & 256) asm volatile("" : "+v"(fr[G % NFR]));       // timing only: reads issued, never waited for
                else if (!(MF_ABL & 16)) mf_wait<(47 - G < RA ? 47 - G : RA)>();
                else fr[G % NFR] = xn[G % NKS];
                if constexpr (G < 24) {
                    if (!(MF_ABL & 4)) s_nxt = mfma32(fr[G % NFR], xn[G], s_nxt);
                } else if constexpr (G < 36) {
                    if (!(MF_ABL & 8)) o[G - 24] = mfma32(fr[G % NFR], p0, o[G - 24]);
                } else {
                    if (!(MF_ABL & 8)) o[G - 36] = mfma32(fr[G % NFR], p1, o[G - 36]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (G % 4 == 1) {
                    constexpr int J = G / 4;
                    if (!(MF_ABL & 2)) mf_dma1<(J & 3) * 1024>(lane16_i, gsb[J >> 2], gld[J >> 2]);
                }
                if constexpr (!LAST) {
                    if constexpr (G == G_P0) p0 = frag4(qd);      // complete: the lower half of s_cur (the previous step's stream)
                    if constexpr (G == G_P1) p1 = frag4(pd);      // complete: the upper half of s_cur
                    valu_gap(std::integral_constant<int, G>{}, std::integral_constant<int, 0>{}, s_cur, s_nxt);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            r1 = r1 == 2 ? 0 : r1 + 1;
            r2 = r2 == 2 ? 0 : r2 + 1;
        };
#pragma unroll 1
        for (int t = 0; t < NT - 2; t += 2) {
            step(sa, sb, t, std::false_type{});
            step(sb, sa, t + 1, std::false_type{});
        }
        step(sa, sb, NT - 2, std::false_type{});
        step(sb, sa, NT - 1, std::true_type{});

        // ================= epilogue: o = x + b2 + fc2(...) back to the residual stream
        if (!(MF_ABL & (32 | 64)) && row < M) {
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                f32x4 a, b;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = o[s >> 1][(s & 1) * 8 + e];
                    b[e] = o[s >> 1][(s & 1) * 8 + 4 + e];
                }
                *reinterpret_cast<f32x4*>(xrow + s * 16) = a;
                *reinterpret_cast<f32x4*>(xrow + s * 16 + 4) = b;
            }
        }
        if (MF_ABL & (32 | 64)) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) asm volatile("" ::"v"(o[db]));
        }
        if (p.queue != nullptr) {
            int* const nxt = reinterpret_cast<int*>(smem + Q_OFF);
            if (tid == 0) *nxt = atomicAdd(p.queue, 1) + (int)gridDim.x;
            __syncthreads();
            item = __builtin_amdgcn_readfirstlane(*nxt);
            __syncthreads();
        } else {
            item += gridDim.x;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the ring units issued past the last item's end
}

bool mlp_fused_supported(int Dm, int Fh, int planes) { return Dm == mf::D && Fh == mf::F && planes == 1; }

int launch_mlp_fused(const MlpFusedParams& p, hipStream_t s) {
    static PerDeviceOnce once;
    if (once.first()) {
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           mf::LDS_BYTES));
        once.mark();
    }
    if (p.M <= 0 || p.ldx < mf::D || (p.ldx & 3)) {
        dinoseg_set_error("mlp_fused: bad shape M=%d ldx=%d", p.M, p.ldx);
        return -1;
    }
    const int ncu = device_cu_count();
    if (ncu <= 0) return -2;
    const int nitems = (p.M + mf::BM - 1) / mf::BM;
    const int grid = nitems < ncu ? nitems : ncu;
    MlpFusedParams q = p;
    q.queue = nullptr;
    if (nitems > grid) {
        // the work counter: one int per device, zeroed ahead of every launch (a memset node when captured).  Launches on two
        // streams of one device may overlap (the two-stream forward): eight counters per device, taken in turn
        static int* qbuf[64] = {};
        static std::atomic<unsigned> turn{0};
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            if (!qbuf[dev]) {
                int* b = nullptr;
                if (hipMalloc(reinterpret_cast<void**>(&b), 8 * 64) == hipSuccess) qbuf[dev] = b;
            }
            if (qbuf[dev]) {
                q.queue = qbuf[dev] + (turn.fetch_add(1) & 7) * 16;
                DSEG_CHECK_HIP(hipMemsetAsync(q.queue, 0, 16, s));
            }
        }
    }
    hipLaunchKernelGGL(mlp_fused_kernel, dim3(grid), dim3(mf::THREADS), mf::LDS_BYTES, s, q);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace dseg
