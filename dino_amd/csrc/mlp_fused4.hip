// Attention output projection + MLP half of a transformer block on ONE operand plane (the benchmark modes fp16 / bf16), ViT-S width, one
// launch, one wave per SIMD:
//     x += proj(ctx) + b_proj;   x += fc2(gelu(fc1(LayerNorm2(x))))      (vision_transformer.py:123 -> :104-105, then :135 -> :59-65)
//
// mlp_fused2.hip computes the same thing with two waves per SIMD (256 registers each) that split the roles and hand P = gelu(fc1) over
// through LDS; this kernel is the single-plane form of mlp_fused3.hip's structure, which runs the hi + lo modes at the power limit: a
// workgroup is FOUR waves, one per SIMD with the whole 512-register file, and a wave owns 32 rows for a whole item of 128 --
//   o[12]          out^T[384][32] fp32 accumulators: x, then + Wproj . ctx^T, then (+ b_proj, LayerNorm2, + b2) + W2^T . P^T     192 registers
//   xn[24]         B-operand fragments (k = 16 s + 8 h + j on lane (row, h)): the ctx row, then LayerNorm2 of o                   96 registers
//   SA[2], SB[2]   fc1 accumulators of the two hidden tiles of a step, double-buffered (the GELU of a tile runs a step behind)     64 registers
//   pdA, pdB       gelu(S) packed as B-operand fragments (the fc1 accumulator layout IS the fc2 operand layout)                   16 registers
// Everything the matrix pipe does is one kind of STEP: 48 MFMAs 32x32x16 on one 48-KiB slot of packed weights = 48 A-operand fragments of
// 1 KiB (one LDS-DMA piece, one conflict-free ds_read_b128).  Per item:
//   6 projection steps     slot u = Wproj columns 64 u .. 64 u + 63 (two k-tiles):   o[db] += W(db, s2) . ctx(2 kt + s2)
//   fc1 step u (24)        slot = W1 rows 64 u .. 64 u + 63 (two hidden tiles, sigma23 order):   S(2u), S(2u+1) = b1 + sum_ks W(ks) . xn(ks), alternating
//   fc2 step u (24)        slot = W2 columns 64 u .. 64 u + 63:   o[db] += W(db, s2) . P(2u), then P(2u+1)   (s2-major)
// in the order  P0 .. P5, LayerNorm2, F1(0), F1(1), F2(0), F1(2), F2(1), ..., F1(23), F2(22), F2(23).  The packed weights (pack_mlp4_kernel)
// are these 54 slots in this order: the weight stream is linear, three ring positions of 48 KiB, a slot is issued two steps ahead of its
// use, one barrier per 48 MFMAs; the stream runs on across the items of the persistent walk.
// The GELU (the logistic form of common.h's gelu_fast: 9 instructions, two of them transcendental) is a PROGRAM of one instruction per MFMA
// gap and element: element n of a tile starts at program index 3 n, so a gap carries three or four elements in different phases -- about 15
// issue cycles, never two transcendentals -- next to its fragment read and, every fourth gap, one LDS-DMA piece (MI355X_MICROARCH.md: 24
// cycles of other issue hide behind a 32x32x16 MFMA with one wave per SIMD).  A tile's program is 55 gaps long: it starts in gaps 24 .. 47
// of one step and finishes in gaps 0 .. 30 of the next (tile 2u: F2(u-1) -> F1(u+1); tile 2u+1: F1(u+1) -> F2(u)).
// Item boundary: the rows are stored from the accumulators as their last products finish, and the next item's x / ctx rows are loaded into
// the registers that have just become free.
#include <stdio.h>

#include "mlp_common.h"

namespace dseg {

namespace mf4 {
using namespace mfc;
constexpr int NW = 4, BM = NW * 32, THREADS = NW * 64;
constexpr int SLOT = 2 * W_TILE;                    // 48 KiB: 48 fragments
constexpr int NFRAG = SLOT / 1024;
constexpr int NPS = D / 64;                         // projection steps (two k-tiles each)
constexpr int NU = NT / 2;                          // fc1 / fc2 steps (two hidden tiles each)
constexpr int NSLOT = NPS + 2 * NU;                 // slots per item (54)
constexpr int NQS = 3 * D / 64;                     // QKV tail: steps of two 32-feature tiles of the next block's Wqkv (18)
constexpr int RING = 3;
constexpr int B1_OFF = RING * SLOT;                 // b1' [F] fp32 (b1 + W1 beta2: LayerNorm2 is folded into the packed copy)
constexpr int BP_OFF = B1_OFF + F * 4;              // the two residual biases: b_proj, b2 [D] fp32 each
constexpr int B2_OFF = BP_OFF + D * 4;
constexpr int BQ_OFF = B2_OFF + D * 4;              // QKV tail: the next block's folded qkv bias [1152] fp32
constexpr int LDS_BYTES = BQ_OFF + 3 * D * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
// the packed copy of a block: NSLOT (+ NQS) slots, then the folded biases (fp32): b1' [F], bq' [3 D]
constexpr long BIAS_OFF_BYTES = (long)(NSLOT + NQS) * SLOT;
[[maybe_unused]] constexpr long PACK_BYTES = BIAS_OFF_BYTES + (F + 3 * D) * 4;
constexpr int PIECES = SLOT / 1024 / NW;            // LDS-DMA pieces per wave and step
static_assert(PIECES == 12, "three groups of four pieces");
#ifndef MF4_RA
#define MF4_RA 4
#endif
constexpr int RA = MF4_RA, NFR = RA + 2;            // fragments read ahead (even: they are read in pairs, one counted wait per pair); fragment registers
static_assert(RA % 2 == 0 && RA >= 2, "pairs");
}  // namespace mf4

#ifndef MF4_ABL
#define MF4_ABL 0      // timing ablations (wrong results): 1 no GELU, 2 no W DMA, 4 no fc1 MFMAs, 8 no fc2 / proj MFMAs, 16 no fragment reads,
                       // 32 no row loads / stores
#endif

// stream slot n (0 .. 53) of a block's packed weights -> which matrix tiles it holds
__host__ __device__ inline void mf4_slot_kind(int n, int& kind, int& u) {      // kind 0 proj, 1 fc1, 2 fc2, 3 qkv tail; u = step index of that kind
    using namespace mf4;
    if (n >= NSLOT) { kind = 3; u = n - NSLOT; return; }
    if (n < NPS) { kind = 0; u = n; return; }
    const int m = n - NPS;
    if (m == 0) { kind = 1; u = 0; }
    else if (m == 2 * NU - 1) { kind = 2; u = NU - 1; }
    else if (m & 1) { kind = 1; u = (m + 1) >> 1; }
    else { kind = 2; u = (m - 2) >> 1; }
}

// Wproj [384][384], W1 [1536][384], W2 [384][1536] fp32 -> [slot][48 fragments][64 lanes][8] in the operand format, in consumption order
// LayerNorm2's weight g2 is folded into the columns of W1 (its bias into b1: fold_bias4_kernel): LayerNorm(x) W^T + b = ((x - mean) rstd) (W diag(g))^T +
// (b + W beta) -- the kernel's LayerNorm needs no per-feature constants
__global__ __launch_bounds__(256) void pack_mlp4_kernel(const float* __restrict__ Wpr, const float* __restrict__ W1, const float* __restrict__ W2,
                                                        const float* __restrict__ g2, const float* __restrict__ Wqkv, const float* __restrict__ g1n,
                                                        bf16_t* __restrict__ dst, int fmt) {
    using namespace mf4;
    const long total = (long)(NSLOT + (Wqkv ? NQS : 0)) * NFRAG * 512;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long q = idx;
        const int e = (int)(q & 7); q >>= 3;
        const int lane = (int)(q & 63); q >>= 6;
        const int frag = (int)(q % NFRAG);
        const int slot = (int)(q / NFRAG);
        const int h = lane >> 5;
        const int half = frag / NKS, r = frag % NKS;
        int kind, u;
        mf4_slot_kind(slot, kind, u);
        float v;
        if (kind == 1 || kind == 3) {      // fc1 / qkv tail: A row = hidden unit / output feature (sigma23 order: the accumulator layout is fc2's operand layout), fragment = (k-step, tile
                                     // half): the two tiles' products alternate.  k runs in the order LayerNorm2 leaves the row in a lane's registers:
                                     // element e of lane half h = feature 16 ks + 8 (e >> 2) + 4 h + (e & 3)
            const int i = attn::sigma23(lane & 31);
            const int f = (frag >> 1) * 16 + 8 * (e >> 2) + 4 * h + (e & 3);
            v = (kind == 1 ? W1 : Wqkv)[(long)((2 * u + (frag & 1)) * HT + i) * D + f] * (kind == 1 ? g2 : g1n)[f];
        } else if (kind == 0) {      // proj: A row = output feature (natural order: register 4 g + e of lane half h = feature 32 db + 8 g + 4 h + e, the
                                     // two lanes of a row hold adjacent 16-byte pieces), k = the k-tile's 32 inputs, fragment = (k-tile half, db, s2)
            const int db = r >> 1, s2 = r & 1, i = lane & 31;
            v = Wpr ? Wpr[(long)(db * 32 + i) * D + (2 * u + half) * 32 + s2 * 16 + h * 8 + e] : 0.f;
        } else {                     // fc2: A row = output feature (natural order), k = the hidden tile's 32 units, fragment = (tile half, s2, db)
            const int s2 = r / NDB, db = r % NDB, i = lane & 31;
            v = W2[(long)(db * 32 + i) * F + (2 * u + half) * HT + s2 * 16 + h * 8 + e];
        }
        dst[idx] = pack1(v, fmt);
    }
}

long mlp_fused4_pack_elems(int Dm, int Fh) { return Dm == mf4::D && Fh == mf4::F ? mf4::PACK_BYTES / 2 : 0; }
bool mlp_fused4_supported(int Dm, int Fh, int planes) { return Dm == mf4::D && Fh == mf4::F && planes == 1; }

// out[n] = bias[n] + sum_k W[n][k] beta[k]  (fp32; one wave per output feature)
__global__ __launch_bounds__(256) void fold_bias4_kernel(const float* __restrict__ W, const float* __restrict__ beta, const float* __restrict__ bias,
                                                         int N, int K, float* __restrict__ out) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(W[(long)n * K + k], beta[k], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[n] = bias[n] + acc;
}

int launch_pack_mlp4(const MlpFused3Weights& w, int Dm, int Fh, bf16_t* dst, hipStream_t s, int fmt) {
    if (mlp_fused4_pack_elems(Dm, Fh) <= 0 || !w.W1 || !w.b1 || !w.W2 || !w.gamma2 || !w.beta2 || !dst ||
        (w.Wqkv_next && (!w.bqkv_next || !w.gamma1_next || !w.beta1_next))) {
        dinoseg_set_error("pack_mlp4: null pointer or unsupported shape D=%d F=%d", Dm, Fh);
        return -1;
    }
    hipLaunchKernelGGL(pack_mlp4_kernel, dim3(2048), dim3(256), 0, s, w.Wproj, w.W1, w.W2, w.gamma2, w.Wqkv_next, w.gamma1_next, dst, fmt);
    float* fb = reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + mf4::BIAS_OFF_BYTES);
    hipLaunchKernelGGL(fold_bias4_kernel, dim3((Fh + 3) / 4), dim3(256), 0, s, w.W1, w.beta2, w.b1, Fh, Dm, fb);
    if (w.Wqkv_next)
        hipLaunchKernelGGL(fold_bias4_kernel, dim3((3 * Dm + 3) / 4), dim3(256), 0, s, w.Wqkv_next, w.beta1_next, w.bqkv_next, 3 * Dm, Dm, fb + Fh);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// QKV: LayerNorm1 + the qkv projection of the NEXT block at the end of the same launch (vision_transformer.py:122 -> :75): the finished rows are
// normalised from the accumulators, then 18 steps of the fc1 kind on the next block's Wqkv (two 32-feature tiles per step: Q tiles 0 .. 11, K 12 .. 23,
// V 24 .. 35); a step's two tiles are scaled (Q: 64^-0.5 log2 e), packed and stored to Q / K / V [B, heads, npad, 64] from the gaps of the next step
template <int FMT, bool PROJ, bool QKV>
__global__ __launch_bounds__(mf4::THREADS, 1) void mlp_fused4_kernel(MlpFused3Params p) {
    static_assert(PROJ || !QKV, "the qkv tail comes with the projection build");
    using namespace mf4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M;
    const int nitems = (M + BM - 1) / BM;
    if ((int)blockIdx.x >= nitems) return;

    // ---- constants into LDS: the folded fc1 bias (behind the slots of the packed copy), the two residual biases
    {
        const float* fb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.Wp) + BIAS_OFF_BYTES);
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
    auto uniform64 = [](uint64_t v) __attribute__((always_inline)) -> uint64_t {
        return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v) |
               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
    };
    const uint64_t wp = reinterpret_cast<uint64_t>(p.Wp);
    const uint32_t piece0 = (uint32_t)wave * PIECES * 1024;      // this wave's share of a slot: bytes piece0 .. piece0 + 12 KiB
    constexpr int first_slot = PROJ ? 0 : NPS, end_slot = QKV ? NSLOT + NQS : NSLOT;

    // the weight stream: sn = stream slot the next step issues, ipos = the ring position it goes to, rpos = the position the next step reads
    int sn = first_slot, ipos = 0, rpos = 0;
    auto next_slot = [&]() __attribute__((always_inline)) {
        sn = sn + 1 == end_slot ? first_slot : sn + 1;
        ipos = ipos + 1 == RING ? 0 : ipos + 1;
    };
    {
        // ring prologue = what the two steps before the first one would have issued
        const uint32_t lane16 = (uint32_t)(tid & 63) * 16;
        if (!(MF4_ABL & 2))
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
    bf16x8 xn[NKS];
    f32x16 SA0, SA1, SB0, SB1;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        // per-lane constants from an opaque lane id, once per item (values that live across the item loop would be spilled)
        uint32_t zero = 0;
        asm volatile("" : "+v"(zero));
        const uint32_t lane_i = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
        const uint32_t lane16_i = lane_i * 16, lh_i = lane_i >> 5;
        const uint32_t frag_rd_i = lds_base + lane16_i;
        // this lane's row of item `it` (clamped: an item past the end is row M - 1 for every lane), from a fresh opaque lane id: the row
        // pointers are recomputed where they are used -- kept alive across the item they are spilled
        // (HS: elements between the two lanes of a row -- 8 in a ctx fragment, 4 in the fp32 row, whose 16-byte pieces the two lanes hold side by side)
        auto lane_row_hs = [&](int it, int hs) __attribute__((always_inline)) -> long {
            uint32_t z = 0;
            asm volatile("" : "+v"(z));
            const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
            const long r = (long)it * BM + wave * 32 + (int)(l & 31);
            return (r < M ? r : (long)M - 1) * D + (l >> 5) * hs;
        };
        auto lane_row = [&](int it) __attribute__((always_inline)) -> long { return lane_row_hs(it, 8); };
        auto lane_row_x = [&](int it) __attribute__((always_inline)) -> long { return lane_row_hs(it, 4); };

        // rows of an item.  PROJ: the projection accumulates from zero while the item's x row arrives in xs (48 loads of 16 bytes issued in the
        // gaps of the six projection steps, into registers that are dead there: the fc1 accumulators, the GELU's, the ctx fragments already
        // consumed) and joins the accumulators in front of LayerNorm2 -- its latency hides behind the projection; the ctx row of the NEXT item is
        // loaded into the fragment registers in the last step's gaps.  Not PROJ: x straight into the accumulators at the item's start
        f32x4 xs[2 * NKS];
        auto load_xs = [&](const float* xr, auto j_tag) __attribute__((always_inline)) {
            constexpr int J = decltype(j_tag)::value;
            xs[J] = f32x4{1.f, 2.f, 3.f, (float)J};
            if (!(MF4_ABL & 32)) xs[J] = *reinterpret_cast<const f32x4*>(xr + (J >> 1) * 16 + (J & 1) * 8);
        };
        auto load_ctx = [&](const bf16_t* cr, auto k_tag) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value;
            uint4 u = {0x3c003c00u, 0x3c003c00u, (uint32_t)k, 0u};
            if (!(MF4_ABL & 32)) u = *reinterpret_cast<const uint4*>(cr + k * 16);
            xn[k] = __builtin_bit_cast(bf16x8, u);
        };
        const float* const xr_item = p.X + lane_row_x(item);
        const bf16_t* const cr_item = PROJ ? p.ctx + lane_row(item) : nullptr;
        if constexpr (PROJ) {
            // (the ctx fragments of a projection step are loaded two steps ahead: those of P0 and P1 here / in the previous item's last step)
            if (item == (int)blockIdx.x)
                mf_for(std::make_integer_sequence<int, 8>{}, [&](auto k_tag) __attribute__((always_inline)) { load_ctx(cr_item, k_tag); });
        } else {
            mf_for(std::make_integer_sequence<int, 2 * NKS>{}, [&](auto j_tag) __attribute__((always_inline)) { load_xs(xr_item, j_tag); });
        }

        // ---- one step: 48 MFMAs on the slot at ring position rpos; the pieces of the slot two steps ahead go into the position before it
        // (every fourth gap one piece; M0 = the LDS destination of a group of four, set in the gap before the group's first piece)
        // mma(gap tag G, fragment): the product; valu(gap tag G): vector work of MFMA gap G (0 .. 47)
        // VM: vector-memory operations known to have been issued AFTER the pieces this step reads (at least the previous step's twelve pieces)
        // pre(): runs between the step's scalar work and its first product -- behind the first fragment reads, under their latency
        auto step = [&](auto vm_tag, auto&& pre, auto&& mma, auto&& valu) __attribute__((always_inline)) {
            // what this step reads has landed: every wave's pieces of two steps ago
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(vm_tag)::value) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t a = frag_rd_i + (uint32_t)rpos * SLOT;
            rpos = rpos + 1 == RING ? 0 : rpos + 1;
            bf16x8 fr[NFR];
            auto issue_read = [&](auto g_tag) __attribute__((always_inline)) {
                constexpr int G = decltype(g_tag)::value;
                if (MF4_ABL & 16) return;
                mf_rd<G * 1024>(fr[G % NFR], a);
            };
            // the first fragment reads go out before the step's scalar work (the LDS-DMA descriptors of the slot two steps ahead)
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
                if (!(MF4_ABL & 2)) {
                    if constexpr ((G & 15) == 0) asm volatile("s_mov_b32 m0, %0" ::"s"(gld[G >> 4]) : "memory");
                    if constexpr ((G & 3) == 2) {
                        constexpr int Q = G >> 2;
                        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(lane16_i), "s"(gsb[Q >> 2]), "n"((Q & 3) * 1024) : "memory");
                    }
                }
                valu(g_tag);
                __builtin_amdgcn_sched_barrier(0);
            };
            mf_for(std::make_integer_sequence<int, NFRAG>{}, [&](auto g_tag) __attribute__((always_inline)) {
                constexpr int G = decltype(g_tag)::value;
                if constexpr ((G & 1) == 0) {
                    if constexpr (G + RA < NFRAG) {
                        issue_read(std::integral_constant<int, G + RA>{});
                        issue_read(std::integral_constant<int, G + RA + 1>{});
                    }
                    // fragments G and G + 1 have landed: at most the RA read behind them are in flight
                    if (!(MF4_ABL & 16)) mf_wait<(NFRAG - 2 - G < RA ? NFRAG - 2 - G : RA)>();
                }
                if (MF4_ABL & 16) fr[G % NFR] = xn[G % NKS];
                mma(g_tag, fr[G % NFR]);
                gap(g_tag);
            });
        };

        // ---- projection: o^T = Wproj . ctx^T, six steps of two k-tiles; eight of the x row's loads and the four ctx fragments of the step after the
        // next in the gaps of each.  Behind the pieces a step reads came at least: P0 -- the previous step's pieces and the 8 ctx loads of the boundary;
        // P1 -- those 8, P0's pieces and its 12 row loads; then a step's pieces and 8 loads
        if constexpr (PROJ) {
            mf_for(std::make_integer_sequence<int, NPS>{}, [&](auto u_tag) __attribute__((always_inline)) {
                constexpr int U = decltype(u_tag)::value;
                step(std::integral_constant<int, (U == 0 ? 20 : U == 1 ? 32 : 20)>{}, []() {},
                     [&](auto g_tag, const bf16x8& fr) __attribute__((always_inline)) {
                         constexpr int G = decltype(g_tag)::value, KT = 2 * U + G / NKS, R = G % NKS, DB = R >> 1, S2 = R & 1;
                         if (!(MF4_ABL & 8)) {
                             if constexpr (KT == 0 && S2 == 0) {
                                 f32x16 z;
#pragma unroll
                                 for (int r = 0; r < 16; ++r) z[r] = 0.f;
                                 o[DB] = mfma32f<FMT>(fr, xn[2 * KT + S2], z);
                             } else {
                                 o[DB] = mfma32f<FMT>(fr, xn[2 * KT + S2], o[DB]);
                             }
                         }
                     },
                     [&](auto g_tag) __attribute__((always_inline)) {
                         constexpr int G = decltype(g_tag)::value;
                         if constexpr ((G & 3) == 1 && G < 32) load_xs(xr_item, std::integral_constant<int, 8 * U + (G >> 2)>{});
                         if constexpr ((G & 3) == 3 && G < 16 && U + 2 < NPS) load_ctx(cr_item, std::integral_constant<int, 4 * (U + 2) + (G >> 2)>{});
                     });
            });
        }

        // ---- (v - mean) rstd of the rows v = o + x + b_proj (register 4 g + e of block db = feature 32 db + 8 g + 4 lh + e) -> xn as B-operand fragments
        // (fragment k = registers 8 (k & 1) .. + 7 of block k >> 1: fc1's weights are packed in that k order, with LayerNorm2's weight in their columns
        // and its bias in b1').  v is built once in the x row's registers (one read of the accumulators), the statistics and the normalised copy come from
        // there, and the accumulators are written once: v + b2
        {
            uint32_t zz = 0;
            asm volatile("" : "+v"(zz));
            const uint32_t lo8 = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zz)) >> 5) * 4;      // (this lane's pieces: + 0 and + 8)
            const float* const sBp = reinterpret_cast<const float*>(smem + BP_OFF) + lo8;
            const float* const sB2 = reinterpret_cast<const float*>(smem + B2_OFF) + lo8;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * NKS; ++j) {
                if constexpr (PROJ) {
                    if (j % 8 == 0) __builtin_amdgcn_sched_barrier(0);      // (a few pieces of constants in flight, not all 48)
                    const f32x4 c = *reinterpret_cast<const f32x4*>(sBp + (j >> 1) * 16 + (j & 1) * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xs[j][e] += o[j >> 2][((j >> 1) & 1) * 8 + (j & 1) * 4 + e] + c[e];
                }
                sum += (xs[j][0] + xs[j][1]) + (xs[j][2] + xs[j][3]);
            }
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.0f / D);
            float qv = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * NKS; ++j) {
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dlt = xs[j][e] - mean;
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
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(sB2 + k * 16), c1 = *reinterpret_cast<const f32x4*>(sB2 + k * 16 + 8);
                float y[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = xs[2 * k][e], x1 = xs[2 * k + 1][e];
                    y[e] = fmaf(x0, rstd, nmr);
                    y[4 + e] = fmaf(x1, rstd, nmr);
                    o[k >> 1][(k & 1) * 8 + e] = x0 + c0[e];
                    o[k >> 1][(k & 1) * 8 + 4 + e] = x1 + c1[e];
                }
                uint4 u;
                u.x = pack2<FMT>(y[0], y[1]);
                u.y = pack2<FMT>(y[2], y[3]);
                u.z = pack2<FMT>(y[4], y[5]);
                u.w = pack2<FMT>(y[6], y[7]);
                xn[k] = __builtin_bit_cast(bf16x8, u);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }

        // ---- the GELU program of one tile's 16 accumulator values per lane: element n runs instruction i at program index 3 n + i
        // (gelu_fast of common.h; the pair (n - 1, n) is packed behind the odd element's last instruction)
        constexpr int GELU_OPS = 10, GELU_STRIDE = 3, GELU_LEN = 15 * GELU_STRIDE + GELU_OPS + 1;      // program indices 0 .. 55
        constexpr int CARRY = 24;      // a tile's program starts at gap CARRY of one step and goes on at gap 0 of the next with index 48 - CARRY
        static_assert(GELU_LEN - (NFRAG - CARRY) <= 36 - 4, "P (s2 = 1) is complete before the fc2 products of the second tile that read it");
        float exA[16], eaA[16], ebA[16], ecA[16], exB[16], eaB[16], ebB[16], ecB[16];
        uint32_t pdA[8], pdB[8];
        auto gelu_op = [&](auto n_tag, auto i_tag, const f32x16& s, float(&ex)[16], float(&ea)[16], float(&eb)[16], float(&ec)[16],
                           uint32_t(&pd)[8]) __attribute__((always_inline)) {
            constexpr int N = decltype(n_tag)::value, I = decltype(i_tag)::value;
            // (instruction 0 brings the accumulator value into a vector register: left to the compiler, the 32 reads of a step's two tiles stand in
            // front of its first product)
            if constexpr (I == 0) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(ex[N]) : "a"(s[N]));
            if (!(MF4_ABL & 1)) {
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
            if constexpr (I == GELU_OPS && (N & 1)) pd[N >> 1] = pack2<FMT>(ex[N - 1], ex[N]);
        };
        auto gelu_idx = [&](auto x_tag, const f32x16& s, float(&ex)[16], float(&ea)[16], float(&eb)[16], float(&ec)[16],
                            uint32_t(&pd)[8]) __attribute__((always_inline)) {
            constexpr int X = decltype(x_tag)::value;
            mf_for(std::make_integer_sequence<int, 16>{}, [&](auto n_tag) __attribute__((always_inline)) {
                constexpr int N = decltype(n_tag)::value;
                constexpr int SG = N * GELU_STRIDE;
                if constexpr (X >= SG && X - SG <= GELU_OPS) gelu_op(n_tag, std::integral_constant<int, X - SG>{}, s, ex, ea, eb, ec, pd);
            });
        };
        // gap G of a carrier step: the program of the A-stream tile (even tiles: registers exA .., pdA) and of the B-stream tile (odd tiles).
        // CONT_A: the A stream continues here (index G + 48 - CARRY) and the B stream starts at gap CARRY (index G - CARRY); otherwise the
        // other way round.  DO_C / DO_S: there is a continuing / a starting program at all
        auto gelu_gap = [&](auto g_tag, auto conta_tag, auto doc_tag, auto dos_tag, const f32x16& sa, const f32x16& sb) __attribute__((always_inline)) {
            constexpr int G = decltype(g_tag)::value;
            constexpr bool CONT_A = decltype(conta_tag)::value, DO_C = decltype(doc_tag)::value, DO_S = decltype(dos_tag)::value;
            constexpr int XC = G + NFRAG - CARRY, XS = G - CARRY;
            if constexpr (CONT_A) {
                if constexpr (DO_C && XC < GELU_LEN) gelu_idx(std::integral_constant<int, XC>{}, sa, exA, eaA, ebA, ecA, pdA);
                if constexpr (DO_S && XS >= 0) gelu_idx(std::integral_constant<int, XS>{}, sb, exB, eaB, ebB, ecB, pdB);
            } else {
                if constexpr (DO_C && XC < GELU_LEN) gelu_idx(std::integral_constant<int, XC>{}, sb, exB, eaB, ebB, ecB, pdB);
                if constexpr (DO_S && XS >= 0) gelu_idx(std::integral_constant<int, XS>{}, sa, exA, eaA, ebA, ecA, pdA);
            }
        };
        // S = b1 of hidden tile t (register j of lane half h = unit (j & 7) + 8 h + 16 (j >> 3): the sigma23 row order)
        // (the eight reads go out in front of the step's barrier, the accumulators are written behind the step's first fragment reads)
        f32x4 bca[4], bcb[4];
        auto s_bias_load = [&](f32x4 (&c)[4], int t) __attribute__((always_inline)) {
            const float* bp = sB1 + t * HT + lh_i * 8;
            c[0] = *reinterpret_cast<const f32x4*>(bp); c[1] = *reinterpret_cast<const f32x4*>(bp + 4);
            c[2] = *reinterpret_cast<const f32x4*>(bp + 16); c[3] = *reinterpret_cast<const f32x4*>(bp + 20);
        };
        auto s_bias_set = [&](f32x16& s, const f32x4 (&c)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[4 * q4 + e] = c[q4][e];
        };
        // F1(u): sa_n = S(2u), sb_n = S(2u+1); its gaps carry the end of tile 2u-2's program (from sa_c) and the start of tile 2u-1's (from sb_c)
        auto step_f1 = [&](f32x16& sa_n, f32x16& sb_n, const f32x16& sa_c, const f32x16& sb_c, int u, auto gelu_tag) __attribute__((always_inline)) {
            s_bias_load(bca, 2 * u);
            s_bias_load(bcb, 2 * u + 1);
            step(std::integral_constant<int, 12>{},
                 [&]() __attribute__((always_inline)) {
                     s_bias_set(sa_n, bca);
                     s_bias_set(sb_n, bcb);
                 },
                 [&](auto g_tag, const bf16x8& fr) __attribute__((always_inline)) {
                     // (the two tiles' accumulation chains alternate: a product never waits for the one issued just before it)
                     constexpr int G = decltype(g_tag)::value, KS = G >> 1;
                     if (!(MF4_ABL & 4)) {
                         if constexpr ((G & 1) == 0) sa_n = mfma32f<FMT>(fr, xn[KS], sa_n);
                         else sb_n = mfma32f<FMT>(fr, xn[KS], sb_n);
                     }
                 },
                 [&](auto g_tag) __attribute__((always_inline)) { gelu_gap(g_tag, std::true_type{}, gelu_tag, gelu_tag, sa_c, sb_c); });
        };
        // F2(u): o^T += W2^T . P(2u)^T (pdA), then P(2u+1)^T (pdB); its gaps carry the end of tile 2u+1's program (from sb_c) and the start of
        // tile 2u+2's (from sa_c).  TAIL (F2(23)): the next item's ctx row goes into the fragment registers (dead since F1(23)) in gaps 0 .. 23;
        // the rows are stored behind the step (behind its last piece: the next item's first steps wait for the pieces, not for the stores)
        const bf16_t* ncr = nullptr;
        auto step_f2 = [&](const f32x16& sb_c, const f32x16& sa_c, auto tail_tag) __attribute__((always_inline)) {
            constexpr bool TAIL = decltype(tail_tag)::value;
            if constexpr (TAIL && PROJ && !QKV) ncr = p.ctx + lane_row(item + (int)gridDim.x);
            step(std::integral_constant<int, 12>{}, []() {},
                 [&](auto g_tag, const bf16x8& fr) __attribute__((always_inline)) {
                     constexpr int G = decltype(g_tag)::value, R = G % NKS, S2 = R / NDB, DB = R % NDB;
                     uint4 u;
                     if constexpr (G < NKS) u = uint4{pdA[4 * S2], pdA[4 * S2 + 1], pdA[4 * S2 + 2], pdA[4 * S2 + 3]};
                     else u = uint4{pdB[4 * S2], pdB[4 * S2 + 1], pdB[4 * S2 + 2], pdB[4 * S2 + 3]};
                     if (!(MF4_ABL & 8)) o[DB] = mfma32f<FMT>(fr, __builtin_bit_cast(bf16x8, u), o[DB]);
                 },
                 [&](auto g_tag) __attribute__((always_inline)) {
                     constexpr int G = decltype(g_tag)::value;
                     gelu_gap(g_tag, std::false_type{}, std::true_type{}, std::integral_constant<bool, !TAIL>{}, sa_c, sb_c);
                     if constexpr (TAIL && PROJ && !QKV && G < 8) load_ctx(ncr, g_tag);
                 });
            if constexpr (TAIL) {
                // (no row guard: a lane past the last row works on a copy of row M - 1 -- the clamped loads -- and every output column of an
                // MFMA is computed alike, so it stores the same bits to the same place as that row's own lane)
                float* const xrow = p.X + lane_row_x(item);
                mf_for(std::make_integer_sequence<int, NKS>{}, [&](auto k_tag) __attribute__((always_inline)) {
                    constexpr int k = decltype(k_tag)::value;
                    f32x4 a, b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = o[k >> 1][(k & 1) * 8 + e];
                        b[e] = o[k >> 1][(k & 1) * 8 + 4 + e];
                    }
                    if (MF4_ABL & (32 | 64)) {      // (64: only the stores off)
                        asm volatile("" ::"v"(a), "v"(b));
                    } else {
                        *reinterpret_cast<f32x4*>(xrow + k * 16) = a;
                        *reinterpret_cast<f32x4*>(xrow + k * 16 + 8) = b;
                    }
                });
            }
        };
        // VALU-only stretches where a program has no carrier step: indices [X0, X1) of the A tile in sa and / or [Y0, Y1) of the B tile in sb
        auto filler = [&](auto x0_tag, auto n_tag, auto y0_tag, const f32x16& sa, const f32x16& sb) __attribute__((always_inline)) {
            constexpr int X0 = decltype(x0_tag)::value, Y0 = decltype(y0_tag)::value;
            mf_for(std::make_integer_sequence<int, decltype(n_tag)::value>{}, [&](auto i_tag) __attribute__((always_inline)) {
                constexpr int I = decltype(i_tag)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (X0 >= 0 && X0 + I < GELU_LEN) gelu_idx(std::integral_constant<int, X0 + I>{}, sa, exA, eaA, ebA, ecA, pdA);
                if constexpr (Y0 >= 0 && Y0 + I < NFRAG - CARRY) gelu_idx(std::integral_constant<int, Y0 + I>{}, sb, exB, eaB, ebB, ecB, pdB);
            });
        };
        using I0 = std::integral_constant<int, 0>;
        using NONE = std::integral_constant<int, -1>;

        step_f1(SA0, SB0, SA0, SB0, 0, std::false_type{});                              // F1(0): tiles 0, 1
        filler(I0{}, std::integral_constant<int, NFRAG - CARRY>{}, NONE{}, SA0, SB0);   // tile 0's first indices have no carrier
        step_f1(SA1, SB1, SA0, SB0, 1, std::true_type{});                               // F1(1) + tile 0 (end), tile 1 (start)
        step_f2(SB0, SA1, std::false_type{});                                           // F2(0) + tile 1 (end), tile 2 (start)
#pragma unroll 1
        for (int u = 2; u < NU; u += 2) {
            step_f1(SA0, SB0, SA1, SB1, u, std::true_type{});                           // F1(u) + tile 2u-2 (end), tile 2u-1 (start)
            step_f2(SB1, SA0, std::false_type{});                                       // F2(u-1) + tile 2u-1 (end), tile 2u (start)
            step_f1(SA1, SB1, SA0, SB0, u + 1, std::true_type{});                       // F1(u+1) + tile 2u (end), tile 2u+1 (start)
            step_f2(SB0, SA1, std::false_type{});                                       // F2(u) + tile 2u+1 (end), tile 2u+2 (start)
        }
        // (the last F2 above started tile 46 = SA1; tile 47 = SB1 has not started)
        filler(std::integral_constant<int, NFRAG - CARRY>{}, std::integral_constant<int, GELU_LEN - (NFRAG - CARRY)>{}, I0{}, SA1, SB1);
        step_f2(SB1, SA1, std::true_type{});                                            // F2(23) + tile 47 (end): the rows out, the next item's rows in

        if constexpr (QKV) {
            // ---- LayerNorm1 of the next block on the finished rows: (v - mean) rstd from a copy in the x registers -> xn (its weight and bias ride in
            // Wqkv' / bq')
            {
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < 2 * NKS; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xs[j][e] = o[j >> 2][((j >> 1) & 1) * 8 + (j & 1) * 4 + e];
                    sum += (xs[j][0] + xs[j][1]) + (xs[j][2] + xs[j][3]);
                }
                sum += __shfl_xor(sum, 32);
                const float mean = sum * (1.0f / D);
                float qv = 0.f;
#pragma unroll
                for (int j = 0; j < 2 * NKS; ++j) {
                    float part = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dlt = xs[j][e] - mean;
                        part = fmaf(dlt, dlt, part);
                    }
                    qv += part;
                }
                qv += __shfl_xor(qv, 32);
                const float rstd = 1.0f / sqrtf(qv * (1.0f / D) + p.eps);
                const float nmr = -mean * rstd;
#pragma unroll
                for (int k = 0; k < NKS; ++k) {
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        y[e] = fmaf(xs[2 * k][e], rstd, nmr);
                        y[4 + e] = fmaf(xs[2 * k + 1][e], rstd, nmr);
                    }
                    uint4 u;
                    u.x = pack2<FMT>(y[0], y[1]);
                    u.y = pack2<FMT>(y[2], y[3]);
                    u.z = pack2<FMT>(y[4], y[5]);
                    u.w = pack2<FMT>(y[6], y[7]);
                    xn[k] = __builtin_bit_cast(bf16x8, u);
                }
            }
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
            const float* const sBq = reinterpret_cast<const float*>(smem + BQ_OFF);
            float ezA[16], ezB[16];
            uint32_t zpA[8], zpB[8];
            // epilogue program of the two tiles 2 uq (values in za) and 2 uq + 1 (zb), gap G of the step that carries it: element n of the first tile is
            // read out of its accumulator at gap n and scaled at n + 1, a pair packed at n + 2, the tile's two 16-byte stores at gaps 19, 20; the second
            // tile 21 gaps later.  V tiles (24 ..) are bf16 whatever FMT is (the one-plane attention's P.V product); Q / K saturate like gemm_big.hip
            auto epi_tile = [&](auto x_tag, const f32x16& z, float(&ez)[16], uint32_t(&zp)[8], int tq) __attribute__((always_inline)) {
                constexpr int X = decltype(x_tag)::value;
                mf_for(std::make_integer_sequence<int, 16>{}, [&](auto n_tag) __attribute__((always_inline)) {
                    constexpr int N = decltype(n_tag)::value;
                    if constexpr (X == N) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(ez[N]) : "a"(z[N]));
                    if constexpr (X == N + 1) ez[N] = ez[N] * (tq < NDB ? p.qscale : 1.0f);
                    if constexpr ((N & 1) == 1 && X == N + 2) {
                        if (FMT == FMT_BF16 || tq >= 2 * NDB) zp[N >> 1] = pack_bf16x2(ez[N - 1], ez[N]);
                        else zp[N >> 1] = pack2_sat<FMT>(ez[N - 1], ez[N]);
                    }
                });
                if constexpr (X == 19 || X == 20) {
                    // (no row guard: duplicates of row M - 1 store the same bits to the same place, as the residual rows)
                    constexpr int GG = X - 19;
                    const int which = tq / NDB, hb = tq - which * NDB;
                    bf16_t* dst = (which == 0 ? p.q : (which == 1 ? p.k : p.v)) + qrow + (long)(hb >> 1) * p.npad * 64 + (hb & 1) * 32 + GG * 16;
                    const uint4 u = {zp[4 * GG], zp[4 * GG + 1], zp[4 * GG + 2], zp[4 * GG + 3]};
                    if (!(MF4_ABL & (32 | 64))) *reinterpret_cast<uint4*>(dst) = u;
                    else asm volatile("" ::"v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w));
                }
            };
            auto epi_gap = [&](auto g_tag, const f32x16& za, const f32x16& zb, int uq) __attribute__((always_inline)) {
                constexpr int G = decltype(g_tag)::value;
                if constexpr (G <= 20) epi_tile(g_tag, za, ezA, zpA, 2 * uq);
                if constexpr (G >= 21 && G <= 41) epi_tile(std::integral_constant<int, G - 21>{}, zb, ezB, zpB, 2 * uq + 1);
            };
            // LASTQ: the next item's first ctx fragments into xn[0 .. 7] (last read by product 15) in gaps 16 .. 23
            const bf16_t* ncq = nullptr;
            auto step_q = [&](f32x16& za_n, f32x16& zb_n, const f32x16& za_c, const f32x16& zb_c, int uq, auto epi_tag, auto last_tag) __attribute__((always_inline)) {
                constexpr bool EPI = decltype(epi_tag)::value, LASTQ = decltype(last_tag)::value;
                {
                    const float* bp = sBq + (2 * uq) * 32 + lh_i * 8;
                    bca[0] = *reinterpret_cast<const f32x4*>(bp); bca[1] = *reinterpret_cast<const f32x4*>(bp + 4);
                    bca[2] = *reinterpret_cast<const f32x4*>(bp + 16); bca[3] = *reinterpret_cast<const f32x4*>(bp + 20);
                    bcb[0] = *reinterpret_cast<const f32x4*>(bp + 32); bcb[1] = *reinterpret_cast<const f32x4*>(bp + 36);
                    bcb[2] = *reinterpret_cast<const f32x4*>(bp + 48); bcb[3] = *reinterpret_cast<const f32x4*>(bp + 52);
                }
                if constexpr (LASTQ) ncq = p.ctx + lane_row(item + (int)gridDim.x);
                step(std::integral_constant<int, 12>{},
                     [&]() __attribute__((always_inline)) {
                         s_bias_set(za_n, bca);
                         s_bias_set(zb_n, bcb);
                     },
                     [&](auto g_tag, const bf16x8& fr) __attribute__((always_inline)) {
                         constexpr int G = decltype(g_tag)::value, KS = G >> 1;
                         if (!(MF4_ABL & 4)) {
                             if constexpr ((G & 1) == 0) za_n = mfma32f<FMT>(fr, xn[KS], za_n);
                             else zb_n = mfma32f<FMT>(fr, xn[KS], zb_n);
                         }
                     },
                     [&](auto g_tag) __attribute__((always_inline)) {
                         constexpr int G = decltype(g_tag)::value;
                         if constexpr (EPI) epi_gap(g_tag, za_c, zb_c, uq - 1);
                         if constexpr (LASTQ && G >= 16 && G < 24) load_ctx(ncq, std::integral_constant<int, G - 16>{});
                     });
            };
            step_q(SA0, SB0, SA0, SB0, 0, std::false_type{}, std::false_type{});
#pragma unroll 1
            for (int uq = 1; uq < NQS - 1; uq += 2) {
                step_q(SA1, SB1, SA0, SB0, uq, std::true_type{}, std::false_type{});
                step_q(SA0, SB0, SA1, SB1, uq + 1, std::true_type{}, std::false_type{});
            }
            step_q(SA1, SB1, SA0, SB0, NQS - 1, std::true_type{}, std::true_type{});
            mf_for(std::make_integer_sequence<int, 42>{}, [&](auto g_tag) __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
                epi_gap(g_tag, SA1, SB1, NQS - 1);
            });
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the ring pieces issued past the last item's end
}

template <int FMT>
static int launch_mlp_fused4_fmt(const MlpFused3Params& p, hipStream_t s) {
    static PerDeviceOnce once;
    if (once.first()) {
        auto opt_in = [](const void* fn) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, mf4::LDS_BYTES); };
        DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&mlp_fused4_kernel<FMT, false, false>)));
        DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&mlp_fused4_kernel<FMT, true, false>)));
        DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&mlp_fused4_kernel<FMT, true, true>)));
        once.mark();
    }
    const int ncu = device_cu_count();
    if (ncu <= 0) return -2;
    const int nitems = (p.M + mf4::BM - 1) / mf4::BM;
    // persistent grid: as few workgroups as finish in the same number of rounds (the rest of the chip is the other stream's)
    int grid;
    if (options().mlp_grid > 0) {
        grid = options().mlp_grid < ncu ? options().mlp_grid : ncu;
        if (grid > nitems) grid = nitems;
    } else {
        const int rounds = (nitems + ncu - 1) / ncu;
        grid = (nitems + rounds - 1) / rounds;
    }
    const dim3 g(grid), b(mf4::THREADS);
    if (p.q) hipLaunchKernelGGL((mlp_fused4_kernel<FMT, true, true>), g, b, mf4::LDS_BYTES, s, p);
    else if (p.ctx) hipLaunchKernelGGL((mlp_fused4_kernel<FMT, true, false>), g, b, mf4::LDS_BYTES, s, p);
    else hipLaunchKernelGGL((mlp_fused4_kernel<FMT, false, false>), g, b, mf4::LDS_BYTES, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_mlp_fused4(const MlpFused3Params& p, hipStream_t s) {
    if (p.M <= 0 || !p.X || !p.Wp || !p.b2 || (p.ctx && !p.bproj)) {
        dinoseg_set_error("mlp_fused4: null pointer or bad shape (M=%d)", p.M);
        return -1;
    }
    if (p.q && (!p.ctx || !p.k || !p.v || p.ntok <= 0 || p.npad < p.ntok || p.heads * 64 != mf4::D || p.M % p.ntok != 0)) {
        dinoseg_set_error("mlp_fused4: incomplete qkv tail (needs ctx, q / k / v, whole frames of ntok rows)");
        return -1;
    }
    return p.fmt == FMT_FP16 ? launch_mlp_fused4_fmt<FMT_FP16>(p, s) : launch_mlp_fused4_fmt<FMT_BF16>(p, s);
}

}  // namespace dseg
