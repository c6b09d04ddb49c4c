// Attention output projection + MLP half of a transformer block on hi + lo operand planes (the parity modes fp16x3 / bf16x3), ViT-S
// width, ONE launch:
//     x += proj(ctx) + b_proj;   x += fc2(gelu(fc1(LayerNorm2(x))))      (vision_transformer.py:123 -> :104-105, then :135 -> :59-65)
//
// Round 5's parity mode ran this as four GEMM launches and a LayerNorm (proj 286 + LN2 57 + fc1 496 + fc2 286 us per block at 32 frames):
// fc1's epilogue wrote 708 MB of hidden activations that fc2 read straight back, and the stores WERE the epilogue
// (profiles/r05_gemm_epilogue_stores.md).  Here the hidden activation, the projected rows and LayerNorm2's output never leave the
// register file: a workgroup is FOUR waves, one per SIMD with the whole 512-register file, and a wave owns 32 rows for a whole item --
//   o[12]          out^T[384][32] fp32 accumulators: x + b_proj, then + Wproj . ctx^T, then (+ b2) + W2^T . P^T            192 registers
//   xh[24], xl[24] B-operand fragments (k = 16 s + 8 h + j on lane (row, h)), hi and lo plane: first the ctx rows (projection),
//                  then LayerNorm2 of o, computed in registers (statistics = sums over the lane's 192 accumulators + one
//                  cross-half shuffle)                                                                                       192 registers
//   S[2]           fc1 accumulators of two consecutive hidden tiles (32 units each)                                           32 registers
//   P              gelu(S) split into hi + lo B-operand fragments: the fc1 accumulator layout IS the fc2 operand layout       16 registers
// (mlp_fused2.hip, the single-plane kernel, splits these roles over two waves of a SIMD; with two planes per operand the state of 32 rows
// fills one wave's file, and three MFMAs per product leave three times the matrix time per fragment read, LDS-DMA piece and GELU
// instruction to hide them in: one wave per SIMD is enough.)
//
// Everything the matrix pipe does is one kind of STEP: 72 MFMAs 32x32x16 on one 48-KiB slot of packed weights = 24 (hi, lo) pairs of
// A-operand fragments, three products per pair (lo . hi, hi . lo, hi . hi: the small terms first, as gemm_big.hip).  Per item of 128 rows:
//   12 projection steps    slot kt = Wproj columns 32 kt .. 32 kt + 31:   o[db] += W(db, s2) . ctx(2 kt + s2)
//   fc1 step t (48)        slot = W1 rows 32 t .. 32 t + 31 (sigma23 order):   S[t & 1] = b1 + sum_ks W(ks) . xn(ks)
//   fc2 step t (48)        slot = W2 columns 32 t .. 32 t + 31:   o[db] += W(db, s2) . P(s2)
// in the order  P0 .. P11, LayerNorm2, F1(0), F1(1), F2(0), F1(2), F2(1), ..., F1(47), F2(46), F2(47):  the GELU of tile t rides in the
// MFMA gaps of F1(t + 1) (and, for its second half, in the first gaps of F2(t), whose products are ordered s2 = 0 first).  The packed
// weights (pack_mlp3_kernel) are these 108 slots in exactly this order, so the weight stream is linear: at the start of a step the
// workgroup has waited for the slot it reads (its own twelve 1-KiB LDS-DMA pieces, then the barrier) and issues the slot two steps
// ahead into the ring position the previous step has left -- three ring positions, 144 KiB, one barrier per 72 MFMAs.  The stream runs on
// across the items of the persistent walk.
// Item boundary: the rows are stored from the accumulators and the next item's x / ctx rows are loaded into the registers that have just
// become free.
//
// QKV = true: LayerNorm1 + the qkv projection of the NEXT block run at the end of the same launch (vision_transformer.py:122 -> :75): the
// finished rows are normalised from the accumulators into the fragment registers like LayerNorm2, then 36 more steps of the fc1 kind
// (slot = 32 rows of the next block's Wqkv: Q tiles 0 .. 11, K 12 .. 23, V 24 .. 35) follow; tile q's 32 x 32 block is scaled (Q: 64^-0.5 log2 e),
// split into hi + lo planes and stored to Q / K / V [B, heads, npad, 64] in the gaps of step q + 1.  The next block then has no
// LayerNorm launch, no qkv GEMM launch, and the normalised planes never exist in HBM.
#include <stdio.h>

#include "mlp_common.h"

namespace dseg {

namespace mf3 {
using namespace mfc;
constexpr int NW = 4, BM = NW * 32, THREADS = NW * 64;
constexpr int SLOT = 2 * W_TILE;                    // 48 KiB: 24 pairs of (lo, hi) fragments, 2 KiB per pair
constexpr int NPT = D / 32;                         // projection k-tiles
constexpr int NSLOT = NPT + 2 * NT;                 // slots per item (108)
constexpr int RING = 3;
constexpr int NQT = 3 * D / 32;                     // QKV tail: output tiles of 32 features (36)
constexpr int B1_OFF = RING * SLOT;                 // b1' [F] fp32 (b1 + W1 beta2: the LayerNorm in front of fc1 is folded into the packed copy)
constexpr int BQ_OFF = B1_OFF + F * 4;              // QKV tail: the next block's folded qkv bias [1152] fp32
constexpr int BP_OFF = BQ_OFF + 3 * D * 4;          // the two residual biases: b_proj, b2 [D] fp32 each
constexpr int B2_OFF = BP_OFF + D * 4;
constexpr int LDS_BYTES = B2_OFF + D * 4;
// the packed copy of a block: NSLOT (+ NQT) slots, then the folded biases (fp32): b1' [F], bq' [3 D]
constexpr long BIAS_OFF_BYTES = (long)(NSLOT + NQT) * SLOT;
[[maybe_unused]] constexpr long PACK_BYTES = BIAS_OFF_BYTES + (F + 3 * D) * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int PIECES = SLOT / 1024 / NW;            // LDS-DMA pieces per wave and step
static_assert(PIECES == 12, "three groups of four pieces");
#ifndef MF3_RA
#define MF3_RA 2
#endif
constexpr int RA = MF3_RA, NF = RA + 1;             // fragment pairs read ahead; pair registers
constexpr int LO = 0, HI = 1024;                    // byte offset of a pair's lo / hi fragment
}  // namespace mf3

#ifndef MF3_ABL
#define MF3_ABL 0      // timing ablations (wrong results): 1 no GELU, 2 no W DMA, 4 no fc1 MFMAs, 8 no fc2 / proj MFMAs, 16 no fragment reads,
                       // 32 no row loads / stores
#endif

// stream slot n (0 .. 107) of a block's packed weights -> which matrix tile it holds
__host__ __device__ inline void mf3_slot_kind(int n, int& kind, int& t) {      // kind 0 proj (t = k-tile), 1 fc1, 2 fc2 (t = hidden tile), 3 qkv (t = tile)
    using namespace mf3;
    if (n >= NSLOT) { kind = 3; t = n - NSLOT; return; }
    if (n < NPT) { kind = 0; t = n; return; }
    const int m = n - NPT;
    if (m == 0) { kind = 1; t = 0; }
    else if (m == 2 * NT - 1) { kind = 2; t = NT - 1; }
    else if (m & 1) { kind = 1; t = (m + 1) >> 1; }
    else { kind = 2; t = (m - 2) >> 1; }
}

#if !defined(MF3_PART) || MF3_PART == 0
// Wproj [384][384], W1 [1536][384], W2 [384][1536] fp32 -> [slot][pair][lo, hi][64 lanes][8] in the operand format
// LayerNorm2's weight g2 is folded into the columns of W1 and the next block's LayerNorm1 weight g1n into those of Wqkv (their biases into b1 / bqkv:
// fold_bias3_kernel): LayerNorm(x) W^T + b = ((x - mean) rstd) (W diag(g))^T + (b + W beta) -- the kernel's LayerNorms need no per-feature constants
__global__ __launch_bounds__(256) void pack_mlp3_kernel(const float* __restrict__ Wpr, const float* __restrict__ W1, const float* __restrict__ W2,
                                                        const float* __restrict__ Wqkv, const float* __restrict__ g2, const float* __restrict__ g1n,
                                                        bf16_t* __restrict__ dst, int fmt) {
    using namespace mf3;
    const long total = (long)(NSLOT + (Wqkv ? NQT : 0)) * NKS * 512;      // (slot, pair, lane, e) tuples; each writes a lo and a hi element
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long q = idx;
        const int e = (int)(q & 7); q >>= 3;
        const int lane = (int)(q & 63); q >>= 6;
        const int pair = (int)(q % NKS);
        const int slot = (int)(q / NKS);
        const int i = attn::sigma23(lane & 31), h = lane >> 5;
        int kind, t;
        mf3_slot_kind(slot, kind, t);
        float v;
        if (kind == 1 || kind == 3) {      // fc1 / qkv: A row = hidden unit / output feature, k = input feature, pair = k-step
            v = (kind == 1 ? W1 : Wqkv)[(long)(t * HT + i) * D + pair * 16 + h * 8 + e] * (kind == 1 ? g2 : g1n)[pair * 16 + h * 8 + e];
        } else {              // proj / fc2: A row = output feature, k = the tile's 32 inputs, pair = (db, s2)
            const int db = pair >> 1, s2 = pair & 1;
            v = kind == 0 ? Wpr[(long)(db * 32 + i) * D + t * 32 + s2 * 16 + h * 8 + e] : W2[(long)(db * 32 + i) * F + t * HT + s2 * 16 + h * 8 + e];
        }
        bf16_t vh, vl;
        split1(v, fmt, vh, vl);
        const long base = ((long)slot * NKS + pair) * 1024 + lane * 8 + e;      // elements: a pair is 2 x 512
        dst[base] = vl;
        dst[base + 512] = vh;
    }
}

long mlp_fused3_pack_elems(int Dm, int Fh) { return Dm == mf3::D && Fh == mf3::F ? mf3::PACK_BYTES / 2 : 0; }

// out[n] = bias[n] + sum_k W[n][k] beta[k]  (fp32; one wave per output feature)
__global__ __launch_bounds__(256) void fold_bias3_kernel(const float* __restrict__ W, const float* __restrict__ beta, const float* __restrict__ bias,
                                                         int N, int K, float* __restrict__ out) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(W[(long)n * K + k], beta[k], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[n] = bias[n] + acc;
}
bool mlp_fused3_supported(int Dm, int Fh, int planes) { return Dm == mf3::D && Fh == mf3::F && planes == 2; }

int launch_pack_mlp3(const MlpFused3Weights& w, int Dm, int Fh, bf16_t* dst, hipStream_t s, int fmt) {
    if (mlp_fused3_pack_elems(Dm, Fh) <= 0 || !w.Wproj || !w.W1 || !w.b1 || !w.W2 || !w.gamma2 || !w.beta2 || !dst ||
        (w.Wqkv_next && (!w.bqkv_next || !w.gamma1_next || !w.beta1_next))) {
        dinoseg_set_error("pack_mlp3: null pointer or unsupported shape D=%d F=%d", Dm, Fh);
        return -1;
    }
    hipLaunchKernelGGL(pack_mlp3_kernel, dim3(2048), dim3(256), 0, s, w.Wproj, w.W1, w.W2, w.Wqkv_next, w.gamma2, w.gamma1_next, dst, fmt);
    float* fb = reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + mf3::BIAS_OFF_BYTES);
    hipLaunchKernelGGL(fold_bias3_kernel, dim3((Fh + 3) / 4), dim3(256), 0, s, w.W1, w.beta2, w.b1, Fh, Dm, fb);
    if (w.Wqkv_next)
        hipLaunchKernelGGL(fold_bias3_kernel, dim3((3 * Dm + 3) / 4), dim3(256), 0, s, w.Wqkv_next, w.beta1_next, w.bqkv_next, 3 * Dm, Dm, fb + Fh);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

#endif

template <int FMT, bool PROJ, bool QKV, bool VBF>
__global__ __launch_bounds__(mf3::THREADS, 1) void mlp_fused3_kernel(MlpFused3Params p) {
    static_assert(PROJ || !QKV, "the qkv tail comes with the projection build");
    using namespace mf3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M;
    const int nitems = (M + BM - 1) / BM;
    if ((int)blockIdx.x >= nitems) return;
    constexpr bool has_proj = PROJ;

    // ---- constants into LDS: b1 (and the next block's qkv bias)
    {
        const float* fb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.Wp) + BIAS_OFF_BYTES);      // folded b1', then bq'
        for (int i = tid; i < F / 4; i += THREADS) reinterpret_cast<f32x4*>(smem + B1_OFF)[i] = reinterpret_cast<const f32x4*>(fb)[i];
        if constexpr (QKV)
            for (int i = tid; i < 3 * D / 4; i += THREADS) reinterpret_cast<f32x4*>(smem + BQ_OFF)[i] = reinterpret_cast<const f32x4*>(fb + F)[i];
        for (int i = tid; i < D / 4; i += THREADS) {
            if constexpr (PROJ) reinterpret_cast<f32x4*>(smem + BP_OFF)[i] = reinterpret_cast<const f32x4*>(p.bproj)[i];
            reinterpret_cast<f32x4*>(smem + B2_OFF)[i] = reinterpret_cast<const f32x4*>(p.b2)[i];
        }
    }
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    const float* const sB1 = reinterpret_cast<const float*>(smem + B1_OFF);
    const float* const sBq = reinterpret_cast<const float*>(smem + BQ_OFF);
    auto uniform64 = [](uint64_t v) __attribute__((always_inline)) -> uint64_t {
        return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v) |
               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
    };
    const uint64_t wp = reinterpret_cast<uint64_t>(p.Wp);
    const uint32_t piece0 = (uint32_t)wave * PIECES * 1024;      // this wave's share of a slot: bytes piece0 .. piece0 + 12 KiB
    constexpr int first_slot = has_proj ? 0 : NPT, end_slot = QKV ? NSLOT + NQT : NSLOT;

    // the weight stream: sn = stream slot the next step issues, ipos = the ring position it goes to, rpos = the position the next step reads
    int sn = first_slot, ipos = 0, rpos = 0;
    auto next_slot = [&]() __attribute__((always_inline)) {
        sn = sn + 1 == end_slot ? first_slot : sn + 1;
        ipos = ipos + 1 == RING ? 0 : ipos + 1;
    };
    {
        // ring prologue = what the two steps before the first one would have issued
        const uint32_t lane16 = (uint32_t)(tid & 63) * 16;
        if (!(MF3_ABL & 2))
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint64_t sb = uniform64(wp + (uint64_t)sn * SLOT + piece0);
                const uint32_t ld = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)ipos * SLOT + piece0);
#pragma unroll
                for (int g = 0; g < 3; ++g) mf_dma4(lane16, sb + g * 4096, ld + g * 4096);
                next_slot();
            }
        else { next_slot(); next_slot(); }
    }
    __syncthreads();      // constants staged

    f32x16 o[NDB];
    bf16x8 xh[NKS], xl[NKS];
    f32x16 S0, S1;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        // per-lane constants from an opaque lane id, once per item (values that live across the item loop would be spilled)
        uint32_t zero = 0;
        asm volatile("" : "+v"(zero));
        const uint32_t lane_i = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
        const uint32_t lane16_i = lane_i * 16, lh_i = lane_i >> 5;
        const uint32_t frag_rd_i = lds_base + lane16_i;
        // this lane's row of item `it` (clamped), from a fresh opaque lane id: the row pointers are recomputed where they are used -- kept
        // alive across the item they are spilled, and a scratch reload in the store gaps waits for every operation in flight
        auto lane_row = [&](int it) __attribute__((always_inline)) -> long {
            uint32_t z = 0;
            asm volatile("" : "+v"(z));
            const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
            const int r = it * BM + wave * 32 + (int)(l & 31);
            return (long)(r < M ? r : M - 1) * D + (l >> 5) * 8;
        };

        // rows of an item: x into the accumulators (k-step k = two 16-byte loads), the ctx planes into the fragment registers (one load each)
        auto load_x = [&](const float* xr, auto k_tag) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value;
            f32x4 a = {1.f, 2.f, 3.f, (float)k}, b = a;
            if (!(MF3_ABL & 32)) {
                a = *reinterpret_cast<const f32x4*>(xr + k * 16);
                b = *reinterpret_cast<const f32x4*>(xr + k * 16 + 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[k >> 1][(k & 1) * 8 + e] = a[e];
                o[k >> 1][(k & 1) * 8 + 4 + e] = b[e];
            }
        };
        // ctx fragments of projection k-tile kt: four 16-byte loads (i = 2 s2 + plane) into buffer kt % 3 of the (still unused) fragment
        // registers: xh / xl[2 (kt % 3) + s2]
        auto load_ctx = [&](const bf16_t* cr, auto kt_tag, auto i_tag) __attribute__((always_inline)) {
            constexpr int KT = decltype(kt_tag)::value, I = decltype(i_tag)::value, S2 = I >> 1, PL = I & 1, R = 2 * (KT % 3) + S2;
            uint4 u = {0x3c003c00u, 0x3c003c00u, (uint32_t)KT, 0u};
            if (!(MF3_ABL & 32)) u = *reinterpret_cast<const uint4*>(cr + (PL ? p.ctx_plane : 0) + (2 * KT + S2) * 16);
            if constexpr (PL) xl[R] = __builtin_bit_cast(bf16x8, u);
            else xh[R] = __builtin_bit_cast(bf16x8, u);
        };
        const bool has_next = item + (int)gridDim.x < nitems;
        if (item == (int)blockIdx.x) {      // (later items: loaded in the gaps of the previous item's last two steps)
            const long ro = lane_row(item);
            mf_for(std::make_integer_sequence<int, NKS>{}, [&](auto k_tag) __attribute__((always_inline)) { load_x(p.X + ro, k_tag); });
            if constexpr (has_proj)
                mf_for(std::make_integer_sequence<int, 8>{}, [&](auto i_tag) __attribute__((always_inline)) {
                    constexpr int I = decltype(i_tag)::value;
                    load_ctx(p.ctx + ro, std::integral_constant<int, I / 4>{}, std::integral_constant<int, I % 4>{});
                });
        }

        // ---- one step: 72 MFMAs on the slot at ring position rpos; pieces of the slot two steps ahead into the position before it
        // mma(pair tag J, product 0 / 1 / 2, fragment): the product; valu(gap tag G): vector work of MFMA gap G (0 .. 71)
        // VM: vector-memory operations known to have been issued AFTER the pieces this step reads (normally the previous step's twelve pieces;
        // in the first two steps of an item also at least 48 of the row loads the previous item's last steps issued behind their pieces)
        // pre(): runs between the step's scalar work and its first product -- behind the first fragment reads, under their latency
        auto step = [&](auto order_tag, auto vm_tag, auto&& pre, auto&& mma, auto&& valu) __attribute__((always_inline)) {
            constexpr int ORDER = decltype(order_tag)::value;      // 0: pair J reads fragment pair J; 1: s2-major (J -> 2 (J % 12) + J / 12)
            constexpr int VM = decltype(vm_tag)::value;
            // what this step reads has landed: every wave's pieces of two steps ago
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t a = frag_rd_i + (uint32_t)rpos * SLOT;
            rpos = rpos + 1 == RING ? 0 : rpos + 1;
            bf16x8 fh[NF], fl[NF];
            auto issue_read = [&](auto j_tag) __attribute__((always_inline)) {
                constexpr int J = decltype(j_tag)::value;
                constexpr int IDX = ORDER == 0 ? J : 2 * (J % 12) + J / 12;
                if (MF3_ABL & 16) return;
                mf_rd<IDX * 2048 + LO>(fl[J % NF], a);
                mf_rd<IDX * 2048 + HI>(fh[J % NF], a);
            };
            // the first fragment reads go out before the step's scalar work (the LDS-DMA descriptors of the slot two steps ahead): their latency is
            // the longest thing between the barrier and the first product
            mf_for(std::make_integer_sequence<int, RA>{}, issue_read);
            __builtin_amdgcn_sched_barrier(0);
            uint64_t gsb[3];
            uint32_t gld[3];
            {
                const uint64_t sb = wp + (uint64_t)sn * SLOT + piece0;
                const uint32_t ld = lds_base + (uint32_t)ipos * SLOT + piece0;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    gsb[g] = uniform64(sb + g * 4096);
                    gld[g] = __builtin_amdgcn_readfirstlane(ld + g * 4096);
                }
                next_slot();
            }
            pre();
            auto gap = [&](auto g_tag) __attribute__((always_inline)) {
                constexpr int G = decltype(g_tag)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr ((G & 1) == 1 && G < 2 * PIECES) {
                    constexpr int Q = G >> 1;
                    if (!(MF3_ABL & 2)) mf_dma1<(Q & 3) * 1024>(lane16_i, gsb[Q >> 2], gld[Q >> 2]);
                }
                valu(g_tag);
                __builtin_amdgcn_sched_barrier(0);
            };
            mf_for(std::make_integer_sequence<int, NKS>{}, [&](auto j_tag) __attribute__((always_inline)) {
                constexpr int J = decltype(j_tag)::value;
                if constexpr (J + RA < NKS) issue_read(std::integral_constant<int, J + RA>{});
                constexpr int AHEAD = (NKS - 1 - J < RA ? NKS - 1 - J : RA);
                if (MF3_ABL & 16) {
                    fl[J % NF] = xh[J];
                    fh[J % NF] = xl[J];
                } else {
                    mf_wait<2 * AHEAD + 1>();      // the lo fragment of pair J
                }
                mma(j_tag, std::integral_constant<int, 0>{}, fl[J % NF]);
                gap(std::integral_constant<int, 3 * J>{});
                if (!(MF3_ABL & 16)) mf_wait<2 * AHEAD>();      // ... and the hi fragment
                mma(j_tag, std::integral_constant<int, 1>{}, fh[J % NF]);
                gap(std::integral_constant<int, 3 * J + 1>{});
                mma(j_tag, std::integral_constant<int, 2>{}, fh[J % NF]);
                gap(std::integral_constant<int, 3 * J + 2>{});
            });
        };

        // ---- projection: o^T += Wproj . ctx^T, twelve k-tiles
        if (has_proj) {
            // (the ctx fragments of k-tile kt + 2 are loaded in the first gaps of step kt; tiles 0 and 1 came with the rows)
            const bf16_t* const cr = p.ctx + lane_row(item);
            mf_for(std::make_integer_sequence<int, NPT>{}, [&](auto kt_tag) __attribute__((always_inline)) {
                constexpr int KT = decltype(kt_tag)::value;
                step(std::integral_constant<int, 0>{}, std::integral_constant<int, (KT == 0 || (KT == 1 && !QKV) ? 60 : 12)>{}, []() {},
                     [&](auto j_tag, auto w_tag, const bf16x8& fr) __attribute__((always_inline)) {
                         constexpr int J = decltype(j_tag)::value, W = decltype(w_tag)::value, DB = J >> 1, R = 2 * (KT % 3) + (J & 1);
                         if (!(MF3_ABL & 8)) o[DB] = mfma32f<FMT>(fr, W == 1 ? xl[R] : xh[R], o[DB]);
                     },
                     [&](auto g_tag) __attribute__((always_inline)) {
                         constexpr int G = decltype(g_tag)::value;
                         if constexpr (KT + 2 < NPT && G >= 2 && G <= 8 && (G & 1) == 0)
                             load_ctx(cr, std::integral_constant<int, KT + 2>{}, std::integral_constant<int, (G - 2) / 2>{});
                     });
            });
        }

        // ---- (x - mean) rstd of the rows in o (register 8 s2 + j of block db = feature 32 db + 16 s2 + 8 lh + j) -> xh / xl as hi + lo fragments: the
        // LayerNorm's weight and bias ride in the packed weights (pack_mlp3_kernel).  PRE / POST: the residual biases (LDS) added to o before the
        // statistics / after the normalised copy has been taken
        auto layer_norm = [&](auto pre_tag, auto post_tag) __attribute__((always_inline)) {
            constexpr bool PRE = decltype(pre_tag)::value, POST = decltype(post_tag)::value;
            uint32_t zz = 0;
            asm volatile("" : "+v"(zz));
            const uint32_t lo8 = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zz)) >> 5) * 8;
            const float* const pre = reinterpret_cast<const float*>(smem + BP_OFF) + lo8;
            const float* const post = reinterpret_cast<const float*>(smem + B2_OFF) + lo8;
            if constexpr (PRE) {
#pragma unroll
                for (int k = 0; k < NKS; ++k) {
                    if (k % 4 == 0) __builtin_amdgcn_sched_barrier(0);      // (four k-steps of constants in flight: hoisted, the 48 reads would need 192 registers)
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(pre + k * 16);
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(pre + k * 16 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[k >> 1][(k & 1) * 8 + e] += c0[e];
                        o[k >> 1][(k & 1) * 8 + 4 + e] += c1[e];
                    }
                }
            }
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
                    __builtin_amdgcn_sched_barrier(0);
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
            float nmr = -mean * rstd;
            asm volatile("" : "+v"(nmr));
#pragma unroll
            for (int k = 0; k < NKS; ++k) {
                if (k % 4 == 0) __builtin_amdgcn_sched_barrier(0);
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
                if constexpr (POST) {
                    c0 = *reinterpret_cast<const f32x4*>(post + k * 16);
                    c1 = *reinterpret_cast<const f32x4*>(post + k * 16 + 4);
                }
                float y[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = o[k >> 1][(k & 1) * 8 + e], x1 = o[k >> 1][(k & 1) * 8 + 4 + e];
                    y[e] = fmaf(x0, rstd, nmr);
                    y[4 + e] = fmaf(x1, rstd, nmr);
                    if constexpr (POST) {
                        o[k >> 1][(k & 1) * 8 + e] = x0 + c0[e];
                        o[k >> 1][(k & 1) * 8 + 4 + e] = x1 + c1[e];
                    }
                }
                uint4 uh, ul;
                split2<FMT>(y[0], y[1], uh.x, ul.x);
                split2<FMT>(y[2], y[3], uh.y, ul.y);
                split2<FMT>(y[4], y[5], uh.z, ul.z);
                split2<FMT>(y[6], y[7], uh.w, ul.w);
                xh[k] = __builtin_bit_cast(bf16x8, uh);
                xl[k] = __builtin_bit_cast(bf16x8, ul);
            }
            if constexpr (PRE || POST) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        // LayerNorm2: sees x + proj(ctx) + b_proj; b2 joins the residual afterwards
        layer_norm(std::integral_constant<bool, has_proj>{}, std::true_type{});

        // ---- the GELU of a tile's 16 accumulator values per lane, one instruction per MFMA gap and element: element n starts at gap
        // 5 n of its carrier step (F1 of the next tile) and runs on into the first gaps of its own fc2 step (gaps 72 ..); the
        // exact-erf form of common.h's gelu_erf (Abramowitz & Stegun 7.1.26) as max(x, 0) - |x| (P(t) / 2) exp(-x^2 / 2),
        // t = 1 / (1 + 0.3275911 |x| / sqrt 2), then the hi + lo split of each pair of values
        constexpr int GELU_OPS = 14;                                  // value complete after op 13
        constexpr int OP_SAT = GELU_OPS, OP_HI = OP_SAT + 1, OP_R = OP_HI + 1, OP_LO = OP_R + 1;      // (pairs: at the odd element)
        constexpr int GELU_STRIDE = 5, GELU_END = 15 * GELU_STRIDE + OP_LO;      // last gap with GELU work (92)
        static_assert(7 * GELU_STRIDE + OP_LO < 72 && GELU_END < 72 + 36, "P fragment s2 is complete before the fc2 products that read it");
        float ex[16], ea[16], eb[16], ec[16];
        uint32_t pdh[8], pdl[8];
        auto gelu_op = [&](auto n_tag, auto i_tag, const f32x16& s) __attribute__((always_inline)) {
            constexpr int N = decltype(n_tag)::value, I = decltype(i_tag)::value;
            if (MF3_ABL & 1) {
                if constexpr (I == 0) ex[N] = s[N];
            } else {
                if constexpr (I == 0) ea[N] = fmaf(0.3275911f * 0.70710678118654752440f, __builtin_fabsf(s[N]), 1.0f);
                if constexpr (I == 1) eb[N] = s[N] * s[N];
                if constexpr (I == 2) ea[N] = __builtin_amdgcn_rcpf(ea[N]);
                if constexpr (I == 3) eb[N] = eb[N] * (-0.5f * 1.44269504088896340736f);
                if constexpr (I == 4) ec[N] = fmaf(0.5f * 1.061405429f, ea[N], 0.5f * -1.453152027f);
                if constexpr (I == 5) eb[N] = __builtin_amdgcn_exp2f(eb[N]);
                if constexpr (I == 6) ec[N] = fmaf(ec[N], ea[N], 0.5f * 1.421413741f);
                if constexpr (I == 7) ec[N] = fmaf(ec[N], ea[N], 0.5f * -0.284496736f);
                if constexpr (I == 8) ec[N] = fmaf(ec[N], ea[N], 0.5f * 0.254829592f);
                if constexpr (I == 9) ea[N] = ea[N] * eb[N];
                if constexpr (I == 10) ec[N] = ec[N] * ea[N];
                if constexpr (I == 11) ex[N] = fmaxf(s[N], 0.f);
                if constexpr (I == 12) ec[N] = __builtin_fabsf(s[N]) * ec[N];
                if constexpr (I == 13) ex[N] = ex[N] - ec[N];
            }
            if constexpr (FMT == FMT_FP16 && I == OP_SAT) ex[N] = __builtin_amdgcn_fmed3f(ex[N], -65504.0f, 65504.0f);
            if constexpr ((N & 1) == 1) {
                // the pair (N - 1, N): hi, the two residuals, lo (element N - 1 finished GELU_STRIDE gaps ago)
                if constexpr (I == OP_HI) pdh[N >> 1] = pack2<FMT>(ex[N - 1], ex[N]);
                if constexpr (I == OP_R) {
                    ex[N - 1] -= lo_to_f32<FMT>(pdh[N >> 1]);
                    ex[N] -= hi_to_f32<FMT>(pdh[N >> 1]);
                }
                if constexpr (I == OP_LO) pdl[N >> 1] = pack2<FMT>(ex[N - 1], ex[N]);
            }
        };
        // gap G of the GELU program of one tile (G = 0 .. 71: the carrier step; 72 ..: the first gaps of the tile's fc2 step)
        auto gelu_gap = [&](auto g_tag, const f32x16& s) __attribute__((always_inline)) {
            constexpr int G = decltype(g_tag)::value;
            mf_for(std::make_integer_sequence<int, 16>{}, [&](auto n_tag) __attribute__((always_inline)) {
                constexpr int N = decltype(n_tag)::value;
                constexpr int SG = N * GELU_STRIDE;
                if constexpr (G >= SG && G - SG <= OP_LO) gelu_op(n_tag, std::integral_constant<int, G - SG>{}, s);
            });
        };
        // S = b1 of hidden tile t (register j of lane half h = unit (j & 7) + 8 h + 16 (j >> 3): the sigma23 row order)
        // (the four reads go out in front of the step's barrier, the accumulator is written behind the step's first fragment reads: s_bias_set as its pre())
        f32x4 bc0, bc1, bc2, bc3;
        auto s_bias_load = [&](const float* table, int t) __attribute__((always_inline)) {
            const float* bp = table + t * HT + lh_i * 8;
            bc0 = *reinterpret_cast<const f32x4*>(bp); bc1 = *reinterpret_cast<const f32x4*>(bp + 4);
            bc2 = *reinterpret_cast<const f32x4*>(bp + 16); bc3 = *reinterpret_cast<const f32x4*>(bp + 20);
        };
        auto s_bias_set = [&](f32x16& s) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[e] = bc0[e];
                s[4 + e] = bc1[e];
                s[8 + e] = bc2[e];
                s[12 + e] = bc3[e];
            }
        };
        // F1(t): s_nxt = b1(t) + W1(t) . xn^T, with the first 72 gaps of the GELU of s_cur (GELU = false: none)
        auto step_f1 = [&](f32x16& s_nxt, const f32x16& s_cur, int t, auto gelu_tag, auto vm_tag) __attribute__((always_inline)) {
            constexpr bool GELU = decltype(gelu_tag)::value;
            s_bias_load(sB1, t);
            step(std::integral_constant<int, 0>{}, vm_tag, [&]() __attribute__((always_inline)) { s_bias_set(s_nxt); },
                 [&](auto j_tag, auto w_tag, const bf16x8& fr) __attribute__((always_inline)) {
                     constexpr int J = decltype(j_tag)::value, W = decltype(w_tag)::value;
                     if (!(MF3_ABL & 4)) s_nxt = mfma32f<FMT>(fr, W == 1 ? xl[J] : xh[J], s_nxt);
                 },
                 [&](auto g_tag) __attribute__((always_inline)) {
                     if constexpr (GELU) gelu_gap(g_tag, s_cur);
                 });
        };
        // F2(t): o^T += W2(t)^T . P^T, products ordered s2 = 0 first, with the last gaps of the GELU of s_cur = S(t).
        // TAIL 2 (F2(47)): the next item's ctx fragments of projection k-tiles 0 and 1 into the fragment registers (dead since F1(47)); block db of the accumulators is final after product 12 + db: its rows are stored and the next item's rows loaded in
        // the gaps behind it (boundary operation b = 3 db + {0, 1: two stores each; 2: four loads} at gap 39 + b; those past gap 71 follow the step)
        float* xrow = nullptr;            // set by the steps that use them (lane_row)
        const float* nxr = nullptr;
        const bf16_t* ncr = nullptr;
        auto boundary_op = [&](auto b_tag) __attribute__((always_inline)) {
            constexpr int B_ = decltype(b_tag)::value, DB = B_ / 3, W = B_ % 3;
            if (MF3_ABL & 32) {
                if constexpr (W == 2) asm volatile("" ::"v"(o[DB]));
                if constexpr (W == 2 && !QKV) {
                    if (has_next) {
                        load_x(nxr, std::integral_constant<int, 2 * DB>{});
                        load_x(nxr, std::integral_constant<int, 2 * DB + 1>{});
                    }
                }
                return;
            }
            if constexpr (W < 2) {
                // (no row guard: a lane past the last row works on a copy of row M - 1 -- the clamped loads -- and every output column of
                // an MFMA is computed alike, so it stores the same bits to the same place as that row's own lane)
                constexpr int k = 2 * DB + W;
                f32x4 a, b;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = o[DB][W * 8 + e];
                    b[e] = o[DB][W * 8 + 4 + e];
                }
                *reinterpret_cast<f32x4*>(xrow + k * 16) = a;
                *reinterpret_cast<f32x4*>(xrow + k * 16 + 4) = b;
            } else if (!QKV && has_next) {      // (QKV: the accumulators are still needed -- LayerNorm1 -- the next rows come in the last qkv steps)
                load_x(nxr, std::integral_constant<int, 2 * DB>{});
                load_x(nxr, std::integral_constant<int, 2 * DB + 1>{});
            }
        };
        auto step_f2 = [&](const f32x16& s_cur, auto tail_tag) __attribute__((always_inline)) {
            constexpr int TAIL = decltype(tail_tag)::value;
            if constexpr (TAIL == 2) {
                xrow = p.X + lane_row(item);
                const long nro = lane_row(item + (int)gridDim.x);
                nxr = p.X + nro;
                if constexpr (has_proj) ncr = p.ctx + nro;
            }
            step(std::integral_constant<int, 1>{}, std::integral_constant<int, 12>{}, []() {},
                 [&](auto j_tag, auto w_tag, const bf16x8& fr) __attribute__((always_inline)) {
                     constexpr int J = decltype(j_tag)::value, W = decltype(w_tag)::value, DB = J % 12, S2 = J / 12;
                     const uint4 uh = {pdh[4 * S2], pdh[4 * S2 + 1], pdh[4 * S2 + 2], pdh[4 * S2 + 3]};
                     const uint4 ul = {pdl[4 * S2], pdl[4 * S2 + 1], pdl[4 * S2 + 2], pdl[4 * S2 + 3]};
                     const bf16x8 bh = __builtin_bit_cast(bf16x8, uh), bl = __builtin_bit_cast(bf16x8, ul);
                     if (!(MF3_ABL & 8)) o[DB] = mfma32f<FMT>(fr, W == 1 ? bl : bh, o[DB]);
                 },
                 [&](auto g_tag) __attribute__((always_inline)) {
                     constexpr int G = decltype(g_tag)::value;
                     if constexpr (72 + G <= GELU_END) gelu_gap(std::integral_constant<int, 72 + G>{}, s_cur);
                     if constexpr (TAIL == 2 && has_proj && !QKV && G >= 24 && G < 32) {
                         if (has_next) load_ctx(ncr, std::integral_constant<int, (G - 24) / 4>{}, std::integral_constant<int, (G - 24) % 4>{});
                     }
                     if constexpr (TAIL == 2 && G >= 39) boundary_op(std::integral_constant<int, G - 39>{});
                 });
            if constexpr (TAIL == 2)
                mf_for(std::make_integer_sequence<int, 3 * NDB - (72 - 39)>{}, [&](auto i_tag) __attribute__((always_inline)) {
                    boundary_op(std::integral_constant<int, 72 - 39 + decltype(i_tag)::value>{});
                });
        };

        using VM12 = std::integral_constant<int, 12>;
        using VMB = std::integral_constant<int, has_proj ? 12 : 60>;      // (no projection: F1(0), F1(1) are the item's first two steps)
        using T0 = std::integral_constant<int, 0>;
        step_f1(S0, S0, 0, std::false_type{}, VMB{});
        step_f1(S1, S0, 1, std::true_type{}, VMB{});           // F1(1) + GELU(0)
        step_f2(S0, T0{});                                     // F2(0)
        step_f1(S0, S1, 2, std::true_type{}, VM12{});          // F1(2) + GELU(1)
        step_f2(S1, T0{});                                     // F2(1)
#pragma unroll 1
        for (int t = 2; t < NT - 2; t += 2) {
            step_f1(S1, S0, t + 1, std::true_type{}, VM12{});  // F1(t + 1) + GELU(t)
            step_f2(S0, T0{});                                 // F2(t)
            step_f1(S0, S1, t + 2, std::true_type{}, VM12{});  // F1(t + 2) + GELU(t + 1)
            step_f2(S1, T0{});                                 // F2(t + 1)
        }
        step_f1(S1, S0, NT - 1, std::true_type{}, VM12{});     // F1(47) + GELU(46)
        step_f2(S0, T0{});                                     // F2(46)
        // GELU(47): its first 72 gaps have no step to ride in
        mf_for(std::make_integer_sequence<int, 72>{}, [&](auto g_tag) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            gelu_gap(g_tag, S1);
        });
        step_f2(S1, std::integral_constant<int, 2>{});         // F2(47): the rows out, the next item's rows in

        if constexpr (QKV) {
            // ---- LayerNorm1 of the next block on the finished rows, then Z(q)^T = Wqkv_q . xn^T + b: 36 steps of the fc1 kind; the block of tile
            // q - 1 is scaled, split and stored in the gaps of step q (behind the step's pieces: stores count in the same vmcnt)
            layer_norm(std::false_type{}, std::false_type{});
            // destination of this lane's row in each of Q, K, V: ((frame * heads) * npad + token) * 64 (+ 8 elements for the upper lane half)
            long qrow;
            {
                uint32_t z = 0;
                asm volatile("" : "+v"(z));
                const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
                const int r = item * BM + wave * 32 + (int)(l & 31);
                const int rc = r < M ? r : M - 1;
                const int fr_ = rc / p.ntok, tok_ = rc - fr_ * p.ntok;
                qrow = ((long)fr_ * p.heads * p.npad + tok_) * 64 + (l >> 5) * 8;
            }
            float ez[16];
            uint32_t zh[8], zl[8];
            // epilogue program of tile tq (values in z), gap G of the step that carries it.  VCONV: the V planes are bf16 whatever FMT is
            auto epi_gap = [&](auto g_tag, auto vconv_tag, const f32x16& z, int tq) __attribute__((always_inline)) {
                constexpr int G = decltype(g_tag)::value;
                constexpr bool VCONV = decltype(vconv_tag)::value;
                constexpr int EF = VCONV ? (int)FMT_BF16 : FMT;
                if constexpr (G >= 24 && G < 40) ez[G - 24] = z[G - 24] * (tq < NDB ? p.qscale : 1.0f);
                if constexpr (G >= 41 && G < 65) {
                    constexpr int MM = (G - 41) / 3, W = (G - 41) % 3;
                    if constexpr (W == 0) {
                        if constexpr (EF == FMT_FP16) {
                            ez[2 * MM] = __builtin_amdgcn_fmed3f(ez[2 * MM], -65504.0f, 65504.0f);
                            ez[2 * MM + 1] = __builtin_amdgcn_fmed3f(ez[2 * MM + 1], -65504.0f, 65504.0f);
                        }
                        zh[MM] = pack2<EF>(ez[2 * MM], ez[2 * MM + 1]);
                    }
                    if constexpr (W == 1) {
                        ez[2 * MM] -= lo_to_f32<EF>(zh[MM]);
                        ez[2 * MM + 1] -= hi_to_f32<EF>(zh[MM]);
                    }
                    if constexpr (W == 2) zl[MM] = pack2<EF>(ez[2 * MM], ez[2 * MM + 1]);
                }
                if constexpr (G >= 66 && G < 70) {
                    // (no row guard: duplicates of row M - 1 store the same bits to the same place, as the residual rows above)
                    constexpr int PL = (G - 66) >> 1, GG = (G - 66) & 1;
                    const int which = tq / NDB, hb = tq - which * NDB;
                    bf16_t* base = which == 0 ? p.q : (which == 1 ? p.k : p.v);
                    bf16_t* dst = base + (PL ? p.qkv_plane : 0) + qrow + (long)(hb >> 1) * p.npad * 64 + (hb & 1) * 32 + GG * 16;
                    const uint32_t* src = PL ? zl : zh;
                    const uint4 u = {src[4 * GG], src[4 * GG + 1], src[4 * GG + 2], src[4 * GG + 3]};
                    if (!(MF3_ABL & 32)) *reinterpret_cast<uint4*>(dst) = u;
                }
            };
            // TQ: 0 plain; 1 (step 34): the next item's rows into the accumulators, one k-step per gap; 2 (step 35): its first ctx fragments
            auto step_q = [&](f32x16& z_nxt, const f32x16& z_cur, int q, auto epi_tag, auto vconv_tag, auto tq_tag) __attribute__((always_inline)) {
                constexpr bool EPI = decltype(epi_tag)::value;
                constexpr int TQ = decltype(tq_tag)::value;
                s_bias_load(sBq, q);
                if constexpr (TQ != 0) {
                    const long nro = lane_row(item + (int)gridDim.x);
                    nxr = p.X + nro;
                    ncr = p.ctx + nro;
                }
                step(std::integral_constant<int, 0>{}, std::integral_constant<int, 12>{}, [&]() __attribute__((always_inline)) { s_bias_set(z_nxt); },
                     [&](auto j_tag, auto w_tag, const bf16x8& fr) __attribute__((always_inline)) {
                         constexpr int J = decltype(j_tag)::value, W = decltype(w_tag)::value;
                         if (!(MF3_ABL & 4)) z_nxt = mfma32f<FMT>(fr, W == 1 ? xl[J] : xh[J], z_nxt);
                     },
                     [&](auto g_tag) __attribute__((always_inline)) {
                         constexpr int G = decltype(g_tag)::value;
                         if constexpr (EPI) epi_gap(g_tag, vconv_tag, z_cur, q - 1);
                         if constexpr (TQ == 1 && G >= 24 && G < 24 + NKS) {
                             if (has_next) load_x(nxr, std::integral_constant<int, G - 24>{});
                         }
                         if constexpr (TQ == 2 && G >= 12 && G < 20) {      // (fragments 0 .. 3 of xh / xl: last read by pair 3, gaps 9 .. 11)
                             if (has_next) load_ctx(ncr, std::integral_constant<int, (G - 12) / 4>{}, std::integral_constant<int, (G - 12) % 4>{});
                         }
                     });
            };
            using VC = std::integral_constant<bool, (VBF && FMT == FMT_FP16)>;
            step_q(S0, S0, 0, std::false_type{}, std::false_type{}, T0{});
#pragma unroll 1
            for (int q = 1; q < 2 * NDB + 1; q += 2) {          // steps 1 .. 24 carry the Q and K tiles 0 .. 23
                step_q(S1, S0, q, std::true_type{}, std::false_type{}, T0{});
                step_q(S0, S1, q + 1, std::true_type{}, std::false_type{}, T0{});
            }
#pragma unroll 1
            for (int q = 2 * NDB + 1; q < NQT - 2; q += 2) {    // steps 25 .. 33 (+ 34 below): V tiles 24 .. 32
                step_q(S1, S0, q, std::true_type{}, VC{}, T0{});
                if (q + 1 < NQT - 2) step_q(S0, S1, q + 1, std::true_type{}, VC{}, T0{});
            }
            step_q(S0, S1, NQT - 2, std::true_type{}, VC{}, std::integral_constant<int, 1>{});      // step 34
            step_q(S1, S0, NQT - 1, std::true_type{}, VC{}, std::integral_constant<int, 2>{});      // step 35
            mf_for(std::make_integer_sequence<int, 72>{}, [&](auto g_tag) __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
                epi_gap(g_tag, VC{}, S1, NQT - 1);
            });
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the ring pieces issued past the last item's end
}

template <int FMT>
int launch_mlp_fused3_fmt(const MlpFused3Params& p, hipStream_t s) {
    static PerDeviceOnce once;
    if (once.first()) {
        auto opt_in = [](const void* fn) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, mf3::LDS_BYTES); };
        DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&mlp_fused3_kernel<FMT, false, false, false>)));
        DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&mlp_fused3_kernel<FMT, true, false, false>)));
        DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&mlp_fused3_kernel<FMT, true, true, false>)));
        if (FMT == FMT_FP16) DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&mlp_fused3_kernel<FMT, true, true, FMT == FMT_FP16>)));
        once.mark();
    }
    const int ncu = device_cu_count();
    if (ncu <= 0) return -2;
    const int nitems = (p.M + mf3::BM - 1) / mf3::BM;
    // persistent grid: as few workgroups as finish in the same number of rounds (the rest of the chip is the other stream's)
    int grid;
    if (options().mlp_grid > 0) {
        grid = options().mlp_grid < ncu ? options().mlp_grid : ncu;
        if (grid > nitems) grid = nitems;
    } else {
        const int rounds = (nitems + ncu - 1) / ncu;
        grid = (nitems + rounds - 1) / rounds;
    }
    const dim3 g(grid), b(mf3::THREADS);
    if (p.q) {
        if (FMT == FMT_FP16 && p.v_bf16) hipLaunchKernelGGL((mlp_fused3_kernel<FMT, true, true, FMT == FMT_FP16>), g, b, mf3::LDS_BYTES, s, p);
        else hipLaunchKernelGGL((mlp_fused3_kernel<FMT, true, true, false>), g, b, mf3::LDS_BYTES, s, p);
    } else if (p.ctx) {
        hipLaunchKernelGGL((mlp_fused3_kernel<FMT, true, false, false>), g, b, mf3::LDS_BYTES, s, p);
    } else {
        hipLaunchKernelGGL((mlp_fused3_kernel<FMT, false, false, false>), g, b, mf3::LDS_BYTES, s, p);
    }
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// two translation units (the seven instantiations take minutes to compile): MF3_PART 0 = everything above + the fp16 kernels, 1 = the bf16 kernels
#ifndef MF3_PART
#define MF3_PART 0
#endif
#if MF3_PART == 0
template int launch_mlp_fused3_fmt<FMT_FP16>(const MlpFused3Params&, hipStream_t);
extern template int launch_mlp_fused3_fmt<FMT_BF16>(const MlpFused3Params&, hipStream_t);

int launch_mlp_fused3(const MlpFused3Params& p, hipStream_t s) {
    if (p.M <= 0 || !p.X || !p.Wp || !p.b2 || (p.ctx && (!p.bproj || p.ctx_plane <= 0))) {
        dinoseg_set_error("mlp_fused3: null pointer or bad shape (M=%d)", p.M);
        return -1;
    }
    if (p.q && (!p.ctx || !p.k || !p.v || p.qkv_plane <= 0 || p.ntok <= 0 || p.npad < p.ntok ||
                p.heads * 64 != mf3::D || p.M % p.ntok != 0)) {
        dinoseg_set_error("mlp_fused3: incomplete qkv tail (needs ctx, q / k / v, whole frames of ntok rows)");
        return -1;
    }
    return p.fmt == FMT_FP16 ? launch_mlp_fused3_fmt<FMT_FP16>(p, s) : launch_mlp_fused3_fmt<FMT_BF16>(p, s);
}
#else
template int launch_mlp_fused3_fmt<FMT_BF16>(const MlpFused3Params&, hipStream_t);
#endif

}  // namespace dseg
