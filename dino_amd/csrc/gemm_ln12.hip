// LayerNorm-fused, A-stationary MFMA GEMM, bf16 configuration: 12 waves (three per SIMD), 128 x 384 tile.
//     qkv = LN1(x) Wqkv^T + b  (vision_transformer.py:123 -> :75,82)      fc1: gelu(LN2(x) W1^T + b)  (:135 -> :60-61)
//
// Same design as gemm_ln.hip (K-resident bf16 image of the normalised 128-row panel in LDS, only W streams, private per-wave
// LDS-DMA rings, no workgroup barrier in the k-loop, register-only epilogue, bias through an MFMA) with one more wave per SIMD.
// Why: in the 8-wave kernel the non-MFMA work of a half-step (fragment reads, ring bookkeeping -- ~27 scalar and ~10 vector
// instructions per four MFMAs -- stream waits, prologue, epilogue) was not hidden behind matrix work: every piece added to
// the run time (MFMAs + loop skeleton 74 us = 48 us of pipe time + 28 us of skeleton).  tools/mfma_peak.py shows a pure MFMA
// stream needs only one wave, but 32 scalar instructions per four MFMAs hold the pipe at ~75 % whatever the occupancy (one
// scalar unit per CU): a third wave per SIMD gives the scheduler more to interleave (core loop 103 -> 84 us).
// What changes to make 12 waves fit:
//   * 168 registers per wave: accumulators 64 (128 rows x 32 columns), TWO fragment sets (reads one half-step ahead of their
//     MFMAs: the other waves cover the LDS latency), immediate epilogue;
//   * LDS: 96 KiB image + 12 rings of five 1-KiB units (one unit = the wave's 32 W rows x 16 k = one HALF k-step, one LDS-DMA
//     instruction per wave and half-step): 156 KiB.  Three units are in flight while one is multiplied and one is read;
//   * the tile is 384 columns wide: qkv (1152) is exactly 3 tiles and fc1 (1536) exactly 4 -- no padded half tile.
// Packed W layout (launch_pack_slabs12): [tile][half-step 0..23][wave 0..11][32 rows][16 k], the two 16-byte chunks of a row
// swapped for rows 8..15 and 24..31 (32-byte rows: without the swap rows r and r+8 of a ds_read_b128 lane group share banks).
#include "gemm_ln_common.h"
#include "kernels.h"

namespace dseg {

namespace ln12 {
constexpr int KD = 384, BM = 128, NW = 12, THREADS = NW * 64, BN = NW * 32, MI = 4;
constexpr int NK = KD / 32, NH = 2 * NK;            // k-steps / half-steps per tile
constexpr int RING = 5, UNIT = 1024;                // ring slots per wave; bytes per wave and half-step
constexpr int HSTAGE = NW * UNIT;                   // one half-step of one column tile in the packed copy
constexpr int A_SLAB = BM * 64, A_BYTES = NK * A_SLAB;
constexpr int GB_OFF = A_BYTES + NW * RING * UNIT;      // gamma [KD] and beta [KD] fp32 (3 KiB)
constexpr int LDS_BYTES = GB_OFF + 2 * KD * 4;
static_assert(BN == LN12_BN && LDS_BYTES <= 160 * 1024, "tile width / LDS budget");
constexpr int PASSES = BM / 4, PPW = (PASSES + NW - 1) / NW;     // LayerNorm: 4 rows per pass; passes per wave (3; waves 8-11: 2)
}  // namespace ln12

__global__ __launch_bounds__(256) void pack_slabs12_kernel(const float* __restrict__ src, int N, int K, bf16_t* __restrict__ dst,
                                                           long total, int fmt) {
    const int nh = K / 16;
    const long nstages = (long)((N + ln12::BN - 1) / ln12::BN) * nh;      // + RING more: the first RING stages again (see the kernel)
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long t = idx;
        const int e = (int)(t & 7); t >>= 3;
        const int phys = (int)(t & 1); t >>= 1;
        const int r = (int)(t & 31); t >>= 5;
        const int rb = (int)(t % ln12::NW); t /= ln12::NW;
        if (t >= nstages) t -= nstages;
        const int h = (int)(t % nh);
        const int tile = (int)(t / nh);
        const int n = tile * ln12::BN + rb * 32 + r, k = h * 16 + ((phys ^ ((r >> 3) & 1)) << 3) + e;
        const float v = n < N ? src[(long)n * K + k] : 0.f;
        dst[idx] = pack1(v, fmt);
    }
}

// (the walk over the nbn * NH half-step stages of a panel is followed by a copy of its first RING stages: the stream runs RING
//  units ahead of the multiplication, so it crosses into the next panel's first stages before the panel ends -- with the copy
//  it does so linearly and the wrap-around is one subtraction per panel instead of a compare + select per half-step)
long gemm_ln12_slab_elems(int N, int K) {
    return (long)((N + ln12::BN - 1) / ln12::BN) * ln12::BN * K + (long)ln12::RING * ln12::HSTAGE / 2;
}

int launch_pack_slabs12(const float* src, int N, int K, bf16_t* dst, hipStream_t s, int fmt) {
    const long total = gemm_ln12_slab_elems(N, K);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_slabs12_kernel, dim3(grid), dim3(256), 0, s, src, N, K, dst, total, fmt);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

#ifndef LN12_NT
#define LN12_NT 1           // 1 = the residual rows (read once per launch) are loaded with the non-temporal hint: the packed W, which every
#endif                      // panel re-streams, keeps its place in the XCD's L2 (qkv 1.83 -> 1.76 ms per 32-frame step, two boxes); 2 = the
                            // outputs stored non-temporal too: 2.9 ms (attention reads them next: they should stay in the Infinity Cache)
#ifndef ALN_ABL
#define ALN_ABL 0           // compile-time ablation bits for A/B builds (tools/build_variant.sh): 1 skip epilogue, 2 skip W DMA,
#endif                      // 4 skip LayerNorm, 8 skip MFMAs, 32 skip fragment reads

template <int EPI, bool DBG, int FMT>
__global__ __launch_bounds__(ln12::THREADS, 3) void gemm_ln12_kernel(LnGemmParams p) {
    using namespace ln12;
    using aln::off64;
    using aln::row16_sum;
    const int dbg = DBG ? p.dbg : ALN_ABL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    char* const sWw = smem + A_BYTES + wave * (RING * UNIT);        // this wave's private W ring

    const int M = p.M, N = p.N;
    const int nbn = (N + BN - 1) / BN, npanels = (M + BM - 1) / BM;
    const int my_panels = ((int)blockIdx.x < npanels) ? (npanels - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (my_panels == 0) return;
    const bool wave_cols_valid_last = (nbn - 1) * BN + wave * 32 < N;     // does this wave own real columns in the last tile?

    // ---- W stream: a linear walk over nbn * NH contiguous half-step stages, restarted for every panel; one 1-KiB LDS-DMA per
    // wave and half-step.  Bookkeeping is kept to a few SCALAR instructions per half-step (the CU has one scalar unit for its
    // four SIMDs: tools/mfma_peak.py -- 32 scalar instructions per four MFMAs hold the matrix pipe at ~75 %, and the first
    // version of this loop carried ~27 scalar + ~10 vector ones): the stream position is one 32-bit scalar byte offset (a packed
    // W is < 4 GiB) next to a constant per-lane offset, ring positions are byte offsets with one wrap each, the refill is
    // unconditional (the stream wraps inside W, so the RING units issued past this workgroup's last half-step read valid
    // memory into a ring nobody reads any more; they are drained before the kernel ends).
    const uint32_t w_panel_bytes = (uint32_t)(nbn * NH * HSTAGE);
    const uint32_t lane16 = (uint32_t)lane * 16;
    uint32_t is_off = (uint32_t)wave * UNIT;                    // scalar: this wave's unit of the half-step stage being issued
    uint32_t ring_off = 0;                                      // scalar: byte offset (in the wave's ring) of the unit being multiplied
    auto issue_next = [&](uint32_t slot_off) {
        glds16(reinterpret_cast<const char*>(p.W) + (size_t)(is_off + lane16), sWw + slot_off);
        is_off += HSTAGE;       // wraps once per panel (below): the packed copy repeats its first RING stages at the end
    };
    static_assert(NH >= RING, "the ring prologue assumes a panel has at least RING half-steps");
#pragma unroll
    for (int s = 0; s < RING; ++s)
        if (!(dbg & 2)) issue_next(s * UNIT);

    f32x16 acc[MI];
    uint32_t row_off[MI];       // element offset of this lane's output row in each row block of the current panel
    float bias_cur, bias_nxt;   // lane = column; loaded by asm (the compiler must not wait for them with the ring in flight)
    constexpr uint32_t ONE = FMT == FMT_FP16 ? 0x3C00u : 0x3F80u;
    const uint4 ones_u = {lh == 0 ? (ONE << 16 | ONE) : 0u, lh == 0 ? ONE : 0u, 0u, 0u};     // k = 0, 1, 2 are 1.0
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);
    auto load_bias = [&](int bn, float& dst) {
        int n = bn * BN + wave * 32 + lr;
        n = n < N ? n : N - 1;
        asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p.bias + n) : "memory");
    };
    // bias as the accumulators' initial value: b = hi + mid + lo exactly (three bf16 terms) times a ones fragment, one MFMA per
    // row block with C = 0
    auto init_acc = [&](float& b) {
        asm volatile("" : "+v"(b));
        const float bj = b;
        const uint32_t hi = pack2<FMT>(bj, 0.f);
        const float r1 = bj - lo_to_f32<FMT>(hi);
        const uint32_t mid = pack2<FMT>(r1, 0.f);
        const uint32_t lo = pack2<FMT>(r1 - lo_to_f32<FMT>(mid), 0.f);
        const uint4 fu = {lh == 0 ? ((hi & 0xFFFFu) | (mid << 16)) : 0u, lh == 0 ? (lo & 0xFFFFu) : 0u, 0u, 0u};
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        bf16x8 one_frag = ones;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            asm volatile("" : "+v"(one_frag));      // opaque: MI separate MFMAs, not one result copied MI times
            acc[i] = mfma32f<FMT>(__builtin_bit_cast(bf16x8, fu), one_frag, z);
        }
    };

    // one 32x32 block of the finished tile: activation, bf16 packing, two 16-byte stores (+ the pre-activation).  Accumulator
    // register r of lane (lr, lh) is output row lr, column (r&3) + 8(r>>2) + 4 lh: two v_permlane32_swap per quad pair give every
    // lane 8 consecutive columns.
    auto drain_block = [&](auto blk_tag, int n0) {
        constexpr int i = decltype(blk_tag)::value;
        if ((dbg & 1) && acc[0][0] != 12345.678f) return;
        if (n0 >= N) return;                            // padding columns of the last tile (wave-uniform)
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[i][r];
        bf16_t* base;
        int which = 0;
        if (EPI == EPI_QKV) {
            which = n0 / p.dmodel;                      // the block lies inside one head of one of Q / K / V
            const int hcol = n0 - which * p.dmodel;
            base = (which == 0 ? p.q : (which == 1 ? p.k : p.v)) + row_off[i] + ((long)(hcol >> 6) * p.npad * 64 + (hcol & 63));
        } else {
            base = p.out_bf16 + row_off[i] + n0;
        }
        auto store_block = [&](const float* val, bf16_t* dst, bool as_bf16 = false) {
            uint2 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (FMT == FMT_BF16 || as_bf16) {        // (as_bf16 is wave-uniform: V of the fp16 mode)
                    w[q].x = pack_bf16x2(val[4 * q], val[4 * q + 1]);
                    w[q].y = pack_bf16x2(val[4 * q + 2], val[4 * q + 3]);
                } else {
                    // GEMM outputs are unbounded: saturate at the fp16 range like gemm.hip / gemm_big.hip (an inf in Q or K turns
                    // into NaN probabilities; the small-batch route of the same weights would clamp)
                    w[q].x = pack2_sat<FMT>(val[4 * q], val[4 * q + 1]);
                    w[q].y = pack2_sat<FMT>(val[4 * q + 2], val[4 * q + 3]);
                }
            }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const uint2 a = w[2 * qq], b = w[2 * qq + 1];
                const auto sx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
                const uint4 o = {sx[0], sy[0], sx[1], sy[1]};
                if ((DBG || ALN_ABL) && (dbg & 64)) asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));      // ablation: stores issued from the k-loop instead
                else if (LN12_NT & 2) {
                    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(dst + qq * 16 + lh * 8));
                }
                else *reinterpret_cast<uint4*>(dst + qq * 16 + lh * 8) = o;
            }
        };
        if (EPI == EPI_GELU && p.aux_out != nullptr) store_block(v, p.aux_out + row_off[i] + n0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (EPI == EPI_GELU) v[r] = gelu_fast(v[r]);
            if (EPI == EPI_QKV && which == 0) v[r] *= p.qscale;
        }
        store_block(v, base, EPI == EPI_QKV && which == 2);
    };

    // ---- fragment pipeline: two register sets, the reads of half-step U+1 are issued during half-step U (inline asm: the
    // compiler's wait-count pass would put s_waitcnt vmcnt(0) in front of LDS reads while LDS-DMA is in flight)
    struct Half {
        bf16x8 w;
        bf16x8 a[MI];
    };
    Half hs[2];
    const uint32_t lds_a = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)sA;
    const uint32_t lds_w = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)sWw;
    const uint32_t la0 = lds_a + off64(lr, lh), la1 = lds_a + off64(lr, 2 + lh);        // the two halves of a 32-k slab
    const uint32_t lw = lds_w + lr * 32 + ((lh ^ ((lr >> 3) & 1)) << 4);
    constexpr int TILE_STORES = MI * 2;     // 16-byte stores of one tile's epilogue (without aux_out: with it the waits are merely stricter)
    bool prev_stored = false;               // did this wave's previous tile end with stores (wave-uniform)?
    auto issue_reads = [&](auto set_tag, uint32_t a_off, int kk, uint32_t slot_off) {
        constexpr int S = decltype(set_tag)::value;
        Half& f = hs[S];
        if ((DBG || ALN_ABL) && (dbg & 32)) return;
        const uint32_t aw = lw + slot_off;
        const uint32_t ap = (kk ? la1 : la0) + a_off;
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.w) : "v"(aw));
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.a[0]) : "v"(ap));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(f.a[1]) : "v"(ap));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.a[2]) : "v"(ap));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(f.a[3]) : "v"(ap));
    };
    auto await_set = [&](auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        Half& f = hs[S];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(f.w));
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(f.a[i]));
    };
    // one half-step (slab byte offset a_off, half kk) = the ring unit at ring_off, fragments in set KK.  The reads one half-step
    // ahead are issued unconditionally: at a panel's last half-step they fetch stale image rows into set 0, which the next
    // panel's fill overwrites (LDS requests of one wave complete in order).  gfx9 retires loads, stores and LDS-DMA through one
    // in-order vmcnt: after the refill the three youngest units may stay in flight; the unit read at the NEXT half-step is older.
    // LAX (the first RING - 2 half-steps of a tile): the previous tile's epilogue stores sit between units in the queue and are
    // YOUNGER than the awaited unit -- they may stay in flight too (without them in the count every tile began by waiting for its
    // predecessor's stores to be acknowledged, all twelve waves at once)
    auto half_step = [&](auto valid_tag, auto kk_tag, auto lax_tag, uint32_t a_off_next) {
        constexpr bool VALID = decltype(valid_tag)::value;
        constexpr int KK = decltype(kk_tag)::value;
        constexpr bool LAX = decltype(lax_tag)::value;
        using Cur = std::integral_constant<int, KK>;
        using Nxt = std::integral_constant<int, KK ^ 1>;
        const Half& f = hs[KK];
        await_set(Cur{});
        const bool mm = VALID && !((DBG || ALN_ABL) && (dbg & 8));
        if (mm) acc[0] = mfma32f<FMT>(f.w, f.a[0], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t ring_next = ring_off + UNIT == RING * UNIT ? 0 : ring_off + UNIT;
        issue_reads(Nxt{}, a_off_next, KK ^ 1, ring_next);
        __builtin_amdgcn_sched_barrier(0);
        if (mm) acc[1] = mfma32f<FMT>(f.w, f.a[1], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (!(dbg & 2)) issue_next(ring_off);      // this unit's fragments are in registers: its slot takes unit + RING
        __builtin_amdgcn_sched_barrier(0);
        if (mm) acc[2] = mfma32f<FMT>(f.w, f.a[2], acc[2]);
        if (mm) acc[3] = mfma32f<FMT>(f.w, f.a[3], acc[3]);
        __builtin_amdgcn_sched_barrier(0);
        if (!(dbg & 2)) {
            if (LAX && prev_stored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING - 2 + TILE_STORES) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING - 2) : "memory");
        }
        asm volatile("" ::: "memory");
        ring_off = ring_next;
    };
    auto run_tile = [&](auto valid_tag, int bn) {
        load_bias(bn + 1 < nbn ? bn + 1 : 0, bias_nxt);      // of the NEXT tile (wraps to the next panel's first)
        using K0 = std::integral_constant<int, 0>;
        using K1 = std::integral_constant<int, 1>;
        static_assert(RING - 2 == 3 && NK % 2 == 0, "three lax half-steps, the k-loop unrolled by two");
        // k-steps 0 and 1 (the first three half-steps lax), then pairs of k-steps; the reads run one half-step ahead
        half_step(valid_tag, K0{}, std::true_type{}, 0);
        half_step(valid_tag, K1{}, std::true_type{}, A_SLAB);
        half_step(valid_tag, K0{}, std::true_type{}, A_SLAB);
        half_step(valid_tag, K1{}, std::false_type{}, 2 * A_SLAB);
        uint32_t a_off = 2 * A_SLAB;
#pragma unroll 1
        for (int kt = 2; kt < NK; kt += 2) {
            half_step(valid_tag, K0{}, std::false_type{}, a_off);
            half_step(valid_tag, K1{}, std::false_type{}, a_off + A_SLAB);
            if ((DBG || ALN_ABL) && (dbg & 64) && EPI == EPI_GELU && bn > 0) {
                // timing experiment (wrong results): the PREVIOUS tile's 8 stores spread over this tile's k-loop, two per pair of
                // k-steps, the way a deferred epilogue would issue them -- is the epilogue's cost the burst or the bytes?
                const int blk = (kt - 2) >> 1;
                if (blk < MI) {
                    const uint4 junk = {0u, 0u, 0u, 0u};
                    bf16_t* dst = p.out_bf16 + row_off[blk & (MI - 1)] + ((bn - 1) * BN + wave * 32);
                    *reinterpret_cast<uint4*>(dst + lh * 8) = junk;
                    *reinterpret_cast<uint4*>(dst + 16 + lh * 8) = junk;
                }
            }
            half_step(valid_tag, K0{}, std::false_type{}, a_off + A_SLAB);
            a_off = a_off + 2 * A_SLAB == NK * A_SLAB ? 0 : a_off + 2 * A_SLAB;
            half_step(valid_tag, K1{}, std::false_type{}, a_off);
        }
    };

    // LayerNorm gain / shift live in LDS (written here, read after the first panel barrier): as global loads inside the row
    // loop the compiler hoisted them out of the panel loop -- 48 registers carried, and spilled, across the whole kernel
    float* const sG = reinterpret_cast<float*>(smem + GB_OFF);
    if (tid < 2 * KD / 4) {
        const int j = tid < KD / 4 ? tid : tid - KD / 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>((tid < KD / 4 ? p.gamma : p.beta) + j * 4);
        *reinterpret_cast<f32x4*>(sG + (tid < KD / 4 ? 0 : KD) + j * 4) = v;
    }
    load_bias(0, bias_cur);
#ifndef LN12_PHASES
#define LN12_PHASES 1
#define LN12_STAGGER 0
#endif
    if (LN12_PHASES > 1) {      // experiment: start the CUs of an XCD in LN12_PHASES groups, LN12_STAGGER x 3.5 us apart
        const int ph = ((int)blockIdx.x >> 3) & (LN12_PHASES - 1);
        for (int i = 0; i < ph * LN12_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // once: the first tile's bias and the ring's first RING units
    for (int pi = 0; pi < my_panels; ++pi) {
        const int panel = blockIdx.x + pi * gridDim.x;
        // nothing of the previous panel's pipeline is live here (tells the register allocator so: the conditional reads / MFMAs
        // above would otherwise carry 104 registers through the prologue and spill it)
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" : "=v"(acc[i]));
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            asm volatile("" : "=v"(hs[st].w));
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" : "=v"(hs[st].a[i]));
        }
        // ================= LayerNorm prologue: rows of this panel -> bf16 in LDS (vision_transformer.py:303, eps 1e-6).  16 lanes
        // per row, 4 rows per pass, 32 passes dealt round-robin to the 12 waves.  The only two workgroup barriers per panel.
        __builtin_amdgcn_s_barrier();
        if (!(dbg & 4)) {
            constexpr int NT = KD / 64;                     // 16-byte loads per lane and row
            const int g = lane >> 4, l = lane & 15;
            f32x4 x[PPW][NT];
#pragma unroll
            // (waves 8-11 have two passes: their third repeats pass 31 and rewrites the same bytes -- a conditional pass would
            //  make x loop-carried for the register allocator, which then spills all of it around the main loop)
            for (int ps = 0; ps < PPW; ++ps) {
                int pass = wave + ps * NW;
                pass = pass < PASSES ? pass : PASSES - 1;
                int gm = panel * BM + pass * 4 + g;
                gm = gm < M ? gm : M - 1;
                const float* xr = p.X + (long)gm * p.ldx + l * 4;
#pragma unroll
                for (int it = 0; it < NT; ++it)
                    x[ps][it] = (LN12_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + it * 64))
                                              : *reinterpret_cast<const f32x4*>(xr + it * 64);
            }
#pragma unroll
            for (int ps = 0; ps < PPW; ++ps) {
                int pass = wave + ps * NW;
                pass = pass < PASSES ? pass : PASSES - 1;
                {
                    float s = 0.f;
#pragma unroll
                    for (int it = 0; it < NT; ++it) s += (x[ps][it][0] + x[ps][it][1]) + (x[ps][it][2] + x[ps][it][3]);
                    const float mean = row16_sum(s) * (1.0f / KD);
                    float qv = 0.f;
#pragma unroll
                    for (int it = 0; it < NT; ++it)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            x[ps][it][e] -= mean;
                            qv = fmaf(x[ps][it][e], x[ps][it][e], qv);
                        }
                    const float rstd = 1.0f / sqrtf(row16_sum(qv) * (1.0f / KD) + p.eps);
                    const int row = pass * 4 + g;
                    int gm = panel * BM + row;
                    gm = gm < M ? gm : M - 1;
#pragma unroll
                    for (int it = 0; it < NT; ++it) {
                        const f32x4 gam = *reinterpret_cast<const f32x4*>(sG + it * 64 + l * 4);
                        const f32x4 bet = *reinterpret_cast<const f32x4*>(sG + KD + it * 64 + l * 4);
                        float y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = x[ps][it][e] * rstd * gam[e] + bet[e];
                        uint2 hi;
                        hi.x = pack2<FMT>(y[0], y[1]);
                        hi.y = pack2<FMT>(y[2], y[3]);
                        // column c = it*64 + l*4: k-slab c >> 5, 16-byte chunk (c & 31) >> 3, byte (c & 7) * 2
                        char* dst = sA + (it * 2 + (l >> 3)) * A_SLAB + off64(row, (l >> 1) & 3) + (l & 1) * 8;
                        *reinterpret_cast<uint2*>(dst) = hi;
                        if (p.a_out) *reinterpret_cast<uint2*>(p.a_out + (long)gm * KD + it * 64 + l * 4) = hi;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's rows are in the image
        }
        __builtin_amdgcn_s_barrier();

        // output row offsets of this panel (rows past M are copies of row M-1: their stores rewrite identical bytes)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int gm = panel * BM + i * 32 + lr;
            gm = gm < M ? gm : M - 1;
            if (EPI == EPI_QKV) {
                const int bq = gm / p.ntok, tok = gm - bq * p.ntok;
                row_off[i] = (uint32_t)((bq * p.heads * p.npad + tok) * 64);
            } else {
                row_off[i] = (uint32_t)gm * (uint32_t)p.ldo;
            }
        }
#ifdef LN12_SKEW      // experiment: wave group g (= wave / 4, one wave per SIMD each) starts the panel's tiles g * LN12_SKEW * 64 cycles late
        for (int i = 0; i < wave / 4; ++i) __builtin_amdgcn_s_sleep(LN12_SKEW);
#endif
        // fill the fragment pipeline: this panel's first two units were waited for already (kernel start / the last half-steps of
        // the previous panel)
        issue_reads(std::integral_constant<int, 0>{}, 0, 0, ring_off);
        for (int bn = 0; bn < nbn; ++bn) {
            const bool cols_valid = bn + 1 < nbn || wave_cols_valid_last;      // wave-uniform
            init_acc(bias_cur);
            if (cols_valid) run_tile(std::true_type{}, bn);
            else run_tile(std::false_type{}, bn);
            // ---- the tile is complete: epilogue straight from the accumulators
            if (cols_valid) {
                const int n0 = bn * BN + wave * 32;
                drain_block(std::integral_constant<int, 0>{}, n0);
                drain_block(std::integral_constant<int, 1>{}, n0);
                drain_block(std::integral_constant<int, 2>{}, n0);
                drain_block(std::integral_constant<int, 3>{}, n0);
            }
            prev_stored = cols_valid && bn * BN + wave * 32 < N && !(dbg & 1);
            bias_cur = bias_nxt;
        }
        is_off -= w_panel_bytes;       // the stream has walked nbn * NH stages (it is RING stages into the repeated head)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the RING units issued past the end (and the last stores)
}

template <int EPI, int FMT>
static int launch_ln12(const LnGemmParams& p, hipStream_t s) {
    static PerDeviceOnce once;
    if (once.first()) {
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln12_kernel<EPI, false, FMT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, ln12::LDS_BYTES));
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln12_kernel<EPI, true, FMT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, ln12::LDS_BYTES));
        once.mark();
    }
    const int ncu = device_cu_count();
    if (ncu <= 0) return -2;
    const int npanels = (p.M + ln12::BM - 1) / ln12::BM;
#ifdef LN12_MAXGRID      // experiment: the same per-workgroup work on fewer CUs
    const int grid = npanels < LN12_MAXGRID ? npanels : LN12_MAXGRID;
#else
    // as few workgroups as finish in the same number of rounds (901 panels: 226 instead of 256): the rest of the chip is the other
    // stream's (mlp_fused2.hip does the same)
    const int rounds = (npanels + ncu - 1) / ncu;
    const int grid = (npanels + rounds - 1) / rounds;
#endif
    if (p.dbg) hipLaunchKernelGGL((gemm_ln12_kernel<EPI, true, FMT>), dim3(grid), dim3(ln12::THREADS), ln12::LDS_BYTES, s, p);
    else hipLaunchKernelGGL((gemm_ln12_kernel<EPI, false, FMT>), dim3(grid), dim3(ln12::THREADS), ln12::LDS_BYTES, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// (shape checks: launch_gemm_ln in gemm_ln.hip)
int launch_gemm_ln12(const LnGemmParams& p, hipStream_t s) {
    if (p.fmt == FMT_FP16) {
        if (p.a_out != nullptr || p.aux_out != nullptr) {
            dinoseg_set_error("gemm_ln12: the fp16 operand format is inference-only (no a_out / aux_out)");
            return -1;
        }
        if (p.epi == EPI_QKV) return launch_ln12<EPI_QKV, FMT_FP16>(p, s);
        return launch_ln12<EPI_GELU, FMT_FP16>(p, s);
    }
    if (p.epi == EPI_QKV) return launch_ln12<EPI_QKV, FMT_BF16>(p, s);
    return launch_ln12<EPI_GELU, FMT_BF16>(p, s);
}

}  // namespace dseg
