// LayerNorm-fused, A-stationary MFMA GEMM for the two LN-fed layers of a block (K = embed_dim):
//     qkv = LN1(x) Wqkv^T + b  (vision_transformer.py:123 -> :75,82)      fc1: gelu(LN2(x) W1^T + b)  (:135 -> :60-61)
// This file: the design, the generic kernel template, the packed-W layout and the dispatch; it RUNS the hi+lo (bf16x3)
// configuration (64 x 128 tile, 4 waves: small batches of the parity mode and the bf16x3 training forward).  The bf16
// configuration is the 12-wave kernel of gemm_ln12.hip (its header says what the measurements on this one led to); the 8-wave
// 128 x 256 bf16 instantiation of this template (Cfg384x1) is kept for reference builds only.
//
// Why: the separate LayerNorm kernel wrote bf16 A (88 MB per launch at B=32) that the 256x384 persistent GEMM then fetched
// 3-4x from the fabric (one pass per column tile: its 32 workgroups per XCD keep 6 MB of A panels in flight against a 4 MB
// L2) -- fc1 and qkv ran at 25-27 % of the MFMA peak while moving 5.6 TB/s.  Here a workgroup owns a row panel of X:
//   * prologue: the panel's fp32 rows are read ONCE (16 B per lane, 16 lanes per row), normalised in registers -- two-pass
//     statistics, the 16-lane sums by 4 DPP adds (the first version used wave-wide ds_bpermute butterflies: 12 serial LDS
//     round trips per row made the prologue 37 % of the kernel) -- and written as bf16 (hi[/lo]) into a K-resident LDS image
//     [K/32 slabs][BM rows][32 k] (96 KiB): no A in HBM, no LayerNorm launch;
//   * main loop: only W streams, from a slab-major, pre-swizzled packed copy (launch_pack_slabs: one k-step of one column
//     tile = one contiguous 16 KiB block, so every LDS-DMA piece is 1 KiB of consecutive bytes and the whole stream is a linear
//     walk) through a 4-slot LDS ring; W is < 1.3 MB and L2-resident, every workgroup walks it in the same order;
//   * operand roles are swapped (W rows on the MFMA rows): accumulator register r of lane (lr, lh) is output row lr, column
//     (r&3) + 8(r>>2) + 4 lh -- four CONSECUTIVE columns per register quad -- so bf16 outputs are packed in the accumulator
//     layout and two v_permlane32_swap per quad pair give every lane 8 consecutive columns = one 16-byte store, with no LDS
//     round trip (the ring needs all the LDS that is left: 96 KiB image + 4 x 16 KiB = 160 KiB);
//   * the bias enters as the accumulators' initial value through one extra MFMA per block (bias as a 3-term bf16 split times
//     a ones fragment: exact), because with this layout it varies along the registers, not the lanes;
//   * the epilogue of tile t (activation, packing, stores) is deferred: it runs in pieces between the k-steps of tile t+1,
//     from a copy of the accumulators, so its VALU / store work overlaps the partner wave's MFMAs instead of serialising
//     with the main loop;
//   * waves are laid out 1 x WN: each owns 32 output columns of every row block, DMAs exactly the W rows it multiplies into a
//     PRIVATE ring, and reads the shared (read-only) image -- so the k-loop has NO workgroup barrier: waves run free between
//     panel boundaries and the two waves of a SIMD drift into complementary phases (one in its MFMA burst, the other waiting
//     for LDS reads or draining an epilogue block).  The 2 x 4 layout with a shared ring and a barrier per k-step measured
//     0.6 us per k-step for 0.24 us of matrix work: both waves of a SIMD read LDS together, then multiply together;
//   * LDS-DMA completion needs one constant counted wait per k-step (see wait_next_stage): gfx9 retires loads, stores and
//     LDS-DMA through one in-order vmcnt, and RING-1 younger stages are always behind the awaited one.
// Training forwards get the normalised planes (a_out) and the fc1 pre-activation (aux_out) as by-products.
#include "gemm_ln_common.h"
#include "kernels.h"

namespace dseg {

namespace aln {
constexpr int BK = 32;
// KD = K (compile-time: the A image is K-resident); PL = operand planes; WN waves side by side, each MI row blocks x 32 columns
template <int KD_, int PL_, int MI_, int WN_, int RING_>
struct Cfg {
    static constexpr int KD = KD_, PL = PL_, WN = WN_, MI = MI_, RING = RING_;
    static constexpr int NWAVES = WN, THREADS = NWAVES * 64;
    static constexpr int BM = MI * 32, BN = WN * 32;
    static constexpr int NK = KD / BK;
    static constexpr int A_SLAB = BM * BK * 2;                       // one [BM][32] k-slab of one plane
    static constexpr int A_PLANE = NK * A_SLAB, A_BYTES = PL * A_PLANE;
    static constexpr int W_BLOCK = 32 * BK * 2;                      // [32 rows][32 k] of one plane = 2 KiB
    static constexpr int W_WAVE = PL * W_BLOCK;                      // what one wave consumes per k-step
    static constexpr int W_STAGE = WN * W_WAVE;                      // one k-step of one column tile, contiguous in the packed copy
    static constexpr int PIECES = W_WAVE / 1024;                     // LDS-DMA pieces per wave and k-step
    static constexpr int LDS_BYTES = A_BYTES + NWAVES * RING * W_WAVE;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static constexpr int RPW = BM / NWAVES;                          // rows each wave normalises in the prologue
    static_assert(BM % NWAVES == 0 && RPW % 4 == 0 && KD % 64 == 0, "prologue shape: 4 rows per pass, 16 lanes x 4 columns per load");
    static constexpr int NBLK = MI;                                  // 32x32 accumulator blocks per wave
    static_assert(NK >= 3 * NBLK - 1 && NBLK <= 4, "the deferred epilogue takes one block every third k-step");
};
using Cfg384x1 = Cfg<384, 1, 4, 8, 4>;    // bf16:   128 x 256 tile, 8 waves of 128 x 32, private 4 x 2 KiB W rings
using Cfg384x2 = Cfg<384, 2, 2, 4, 4>;    // bf16x3:  64 x 128 tile, 4 waves of  64 x 32, private 4 x 4 KiB W rings (hi+lo)

}  // namespace aln

// ------------------------------------------------------------------------------------------------
// fp32 W [N][K] -> slab-major bf16 planes for gemm_ln: dst[tile][kt][32-row block][plane][32 rows][32 k], rows >= N zero, the
// four 16-byte chunks of each 64-byte row stored in the XOR-swizzled order of the LDS image (so the image is a linear copy
// and the [plane][32][32] block of one wave is contiguous).
__global__ __launch_bounds__(256) void pack_slabs_kernel(const float* __restrict__ src, int N, int K, int BN, int planes,
                                                         bf16_t* __restrict__ dst, long total) {
    const int nk = K / 32;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long t = idx;
        const int e = (int)(t & 7); t >>= 3;
        const int phys = (int)(t & 3); t >>= 2;
        const int r = (int)(t & 31); t >>= 5;
        const int pl = (int)(t % planes); t /= planes;
        const int rb = (int)(t % (BN / 32)); t /= (BN / 32);
        const int kt = (int)(t % nk);
        const int tile = (int)(t / nk);
        const int n = tile * BN + rb * 32 + r, k = kt * 32 + ((phys ^ ((r >> 2) & 3)) << 3) + e;
        const float v = n < N ? src[(long)n * K + k] : 0.f;
        const uint32_t hi = pack_bf16x2(v, 0.f);
        dst[idx] = pl == 0 ? (bf16_t)(hi & 0xFFFF) : (bf16_t)(pack_bf16x2(v - bf16_lo_to_f32(hi), 0.f) & 0xFFFF);
    }
}

// planes == 1 runs the 12-wave kernel of gemm_ln12.hip, which has its own packed layout
int gemm_ln_tile_cols(int planes) { return planes == 1 ? LN12_BN : aln::Cfg384x2::BN; }

long gemm_ln_slab_elems(int N, int K, int planes) {
    if (planes == 1) return gemm_ln12_slab_elems(N, K);
    const int bn = gemm_ln_tile_cols(planes);
    return (long)((N + bn - 1) / bn) * bn * K * planes;
}

int launch_pack_slabs(const float* src, int N, int K, int planes, bf16_t* dst, hipStream_t s, int fmt) {
    if (planes == 1) return launch_pack_slabs12(src, N, K, dst, s, fmt);
    if (fmt != FMT_BF16) {
        dinoseg_set_error("pack_slabs: the fp16 operand format is single-plane only");
        return -1;
    }
    const long total = gemm_ln_slab_elems(N, K, planes);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_slabs_kernel, dim3(grid), dim3(256), 0, s, src, N, K, gemm_ln_tile_cols(planes), planes, dst, total);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
template <int EPI, class C, bool DBG = false>
__global__ __launch_bounds__(C::THREADS, (C::NWAVES + 3) / 4) void gemm_ln_kernel(LnGemmParams p) {
    using namespace aln;
    constexpr int KD = C::KD, PL = C::PL, BM = C::BM, BN = C::BN, MI = C::MI, NK = C::NK, RING = C::RING;
    constexpr int PIECES = C::PIECES, NBLK = C::NBLK, NI = 1;
#ifndef ALN_DEFER_PL1
#define ALN_DEFER_PL1 0
#endif
    // the deferred epilogue needs a second accumulator set: affordable at 4 waves per workgroup (512 registers per wave), not at 8
    constexpr bool DEFER = PL == 2 || ALN_DEFER_PL1;
#ifndef ALN_ABL
#define ALN_ABL 0           // compile-time ablation bits for A/B builds (tools/build_variant.sh): exact, unlike the runtime ones
#endif
    const int dbg = DBG ? p.dbg : ALN_ABL;      // ablations are compiled out of the production instantiation
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;
    char* const sW = smem + C::A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave;                        // this wave's 32-column block of every tile
    const int lr = lane & 31, lh = lane >> 5;
    char* const sWw = sW + wave * (RING * C::W_WAVE);       // this wave's private W ring

    const int M = p.M, N = p.N;
    const int nbn = (N + BN - 1) / BN, npanels = (M + BM - 1) / BM;
    const int my_panels = ((int)blockIdx.x < npanels) ? (npanels - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total_steps = my_panels * nbn * NK;
    if (total_steps == 0) return;
    const bool wave_cols_valid_last = (nbn - 1) * BN + wc * 32 < N;     // does this wave own real columns in the last (partial) tile?

    // ---- W stream: a linear walk over nbn * NK contiguous stages, restarted for every panel
    const uint32_t voff = (uint32_t)(lane * 16);
    int is_step = 0;
    long is_off = 0;
    const long w_panel_bytes = (long)nbn * NK * C::W_STAGE;
    int is_slot = 0;
    auto issue_next = [&]() {
        const char* wbase = reinterpret_cast<const char*>(p.W) + is_off + wave * C::W_WAVE;
        char* sbase = sWw + is_slot * C::W_WAVE;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) glds16(wbase + i * 1024 + voff, sbase + i * 1024);
        ++is_step;
        is_off += C::W_STAGE;
        if (is_off == w_panel_bytes) is_off = 0;
        is_slot = is_slot + 1 == RING ? 0 : is_slot + 1;
    };
    // Wait for the stage whose first half is read next (two stages after the one just multiplied).  gfx9 retires loads, stores
    // and LDS-DMA through one in-order vmcnt, so the wait names how many YOUNGER operations may stay in flight: the
    // (RING-2) * PIECES pieces of the two later stages, plus -- for the two waits after an epilogue -- that epilogue's stores.
    // Without the stores in the count the wait also retires the stage issued one k-step ago and most of the stores; an exact
    // count (ALN_STORE_AWARE_WAIT) measured no faster, so the plain count is used.  The count must never exceed what was
    // really issued after the awaited stage; `stores_young` is armed only by a drain that did store.
    //   drain position               stores are younger than the awaited stage at the waits of
    //   after step d's wait (tile end, or between the halves of step d+1)      steps d+1 and d+2
    constexpr int TILE_STORES = DEFER ? 2 * PL : NBLK * 2 * PL;      // 16-byte stores one drain issues (without aux_out)
    (void)TILE_STORES;
    int stores_young = 0;
    auto wait_next_stage = [&](bool refilled) {
        if (!refilled) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last RING k-steps of the kernel
#ifdef ALN_STORE_AWARE_WAIT      // measured (tools/ab_ops.sh, one box): no faster than the plain count -- off
        else if (stores_young > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * PIECES + TILE_STORES) : "memory");
#endif
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * PIECES) : "memory");
        stores_young = stores_young > 0 ? stores_young - 1 : 0;
    };

    // ---- ring prologue: the first RING stages (fragments are read one k-step ahead, so a stage's slot is refilled while that
    // stage is being multiplied from registers: RING-1 stages are in flight during every k-step)
#pragma unroll
    for (int s = 0; s < RING; ++s)
        if (s < total_steps && !(dbg & 2)) issue_next();

    // ---- accumulators.  acc = the tile being multiplied; accp = the finished previous tile, drained between the k-steps
    f32x16 acc[MI][NI], accp[MI][NI];
    bool have_prev = false;
    int prev_n0 = 0;                          // first column of the previous tile's wave block
    uint32_t row_off[MI], prev_row_off[MI];   // element offset of this lane's output row in each row block (current / drained panel)
    bool prev_valid = false;                  // previous tile: does this wave own real columns?
    // bias values of the CURRENT / NEXT tile (lane = column): loaded by asm (the compiler must not wait for them while the
    // LDS-DMA ring is in flight: it would drain the ring); they are old by the time the next tile starts
    float bias_cur[NI], bias_nxt[NI];
    const uint4 ones_u = {lh == 0 ? 0x3F803F80u : 0u, lh == 0 ? 0x00003F80u : 0u, 0u, 0u};     // k = 0, 1, 2 are 1.0
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);
    auto load_bias = [&](int bn, float* dst) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            int n = bn * BN + wc * 32 + j * 32 + lr;
            n = n < N ? n : N - 1;
            asm volatile("global_load_dword %0, %1, off" : "=v"(dst[j]) : "v"(p.bias + n) : "memory");
        }
    };
    auto init_acc = [&](float* b) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            // b = hi + mid + lo exactly (three bf16 terms of an fp32); rows of the fragment = output columns
            asm volatile("" : "+v"(b[j]));
            const float bj = b[j];
            const uint32_t hi = pack_bf16x2(bj, 0.f);
            const float r1 = bj - bf16_lo_to_f32(hi);
            const uint32_t mid = pack_bf16x2(r1, 0.f);
            const uint32_t lo = pack_bf16x2(r1 - bf16_lo_to_f32(mid), 0.f);
            const uint4 fu = {lh == 0 ? ((hi & 0xFFFFu) | (mid << 16)) : 0u, lh == 0 ? (lo & 0xFFFFu) : 0u, 0u, 0u};
            // one MFMA per row block with C = 0 (an inline constant): computing the block once and copying it cost 16 v_mov for
            // the zeros + 16 per copy -- 80 VALU issues per tile and wave behind an MFMA result, with the matrix pipe idle
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            bf16x8 one_frag = ones;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                asm volatile("" : "+v"(one_frag));      // opaque: MI separate MFMAs, not one result copied MI times
                acc[i][j] = mfma32(__builtin_bit_cast(bf16x8, fu), one_frag, z);
            }
        }
    };

    // one 32x32 block of the previous tile: activation, bf16 packing, two 16-byte stores per plane (+ the pre-activation)
    auto drain_block = [&](auto blk_tag) {
        constexpr int blk = decltype(blk_tag)::value;
        if (!prev_valid || ((dbg & 1) && accp[0][0][0] != 12345.678f)) return;
        constexpr int i = blk, j = 0;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = accp[i][j][r];
        const int nb = prev_n0 + j * 32;                // first column of the block (wave-uniform)
        if (nb >= N) return;                            // padding columns of the last tile
        bf16_t* base;
        long pstride;
        int which = 0;
        // this lane's output row offset was worked out once per panel (prev_row_off: one integer division per row block and
        // panel instead of one per block and tile); the column part is wave-uniform
        if (EPI == EPI_QKV) {
            which = nb / p.dmodel;                      // the block lies inside one head of one of Q / K / V
            const int hcol = nb - which * p.dmodel;
            base = (which == 0 ? p.q : (which == 1 ? p.k : p.v)) + prev_row_off[i] + ((long)(hcol >> 6) * p.npad * 64 + (hcol & 63));
            pstride = p.qkv_plane;
        } else {
            base = p.out_bf16 + prev_row_off[i] + nb;
            pstride = p.out_plane;
        }
        auto store_planes = [&](const float* val, bf16_t* dst, long plane_stride) {
            uint2 hi[4], lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                split_bf16x2(val[4 * q], val[4 * q + 1], hi[q].x, lo[q].x);
                split_bf16x2(val[4 * q + 2], val[4 * q + 3], hi[q].y, lo[q].y);
            }
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) {
                uint2* w = pl == 0 ? hi : lo;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    // quads 2qq (columns 16qq + 4lh + 0..3) and 2qq+1 (+8): after the half swaps the lower lanes hold columns
                    // 16qq + 0..7 and the upper lanes 16qq + 8..15
                    const uint2 a = w[2 * qq], b = w[2 * qq + 1];
                    const auto sx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
                    const uint4 o = {sx[0], sy[0], sx[1], sy[1]};
                    *reinterpret_cast<uint4*>(dst + pl * plane_stride + qq * 16 + lh * 8) = o;
                }
            }
        };
        // (DEFER: one block per call; otherwise the NBLK calls of a tile end re-arm the same two waits.  With aux_out there are
        //  more stores than counted: the waits are then merely stricter)
        stores_young = 2;
        if (EPI == EPI_GELU && p.aux_out != nullptr) store_planes(v, p.aux_out + prev_row_off[i] + nb, p.aux_plane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (EPI == EPI_GELU) v[r] = PL == 1 ? gelu_fast(v[r]) : gelu_erf(v[r]);
            if (EPI == EPI_QKV && which == 0) v[r] *= p.qscale;
        }
        store_planes(v, base, pstride);
    };

    // ---- fragment pipeline.  LDS reads run TWO half-steps (one half = 16 of a stage's 32 k) ahead of their MFMAs through three
    // rotating register sets: with one half-step of lead (two sets) every MFMA group still waited ~400 cycles for its reads
    // (8 waves read at once; measured: the bare loop ran the matrix pipe 45 % of the time, and reading a quarter of the bytes
    // changed nothing -- latency, not bandwidth).  The reads are inline asm with hand-counted lgkmcnt (LDS returns in order): the
    // compiler's own wait-count pass knows nothing of them and cannot put an lgkmcnt(0) in front of every MFMA group.
    struct Half {
        bf16x8 w[PL];
        bf16x8 a[PL][MI];
    };
    Half hs[3];
    constexpr int RD = PL * (1 + MI);       // ds_read_b128 per half-step
    static_assert(2 * RD <= 15, "lgkmcnt is a 4-bit counter");
    static_assert(NK % 3 == 0, "the k-loop is unrolled by three k-steps = six half-steps = two turns of the three register sets");
    const uint32_t lds_a = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)sA;
    const uint32_t lds_w = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)sWw;
    uint32_t la[2], lw[2];                  // per-lane fragment addresses of the two halves (image slab 0 / ring slot 0)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        la[kk] = lds_a + off64(lr, kk * 2 + lh);
        lw[kk] = lds_w + off64(lr, kk * 2 + lh);
    }
    int cs_slot = 0;                        // ring slot of the stage being multiplied
    auto issue_reads = [&](auto set_tag, int kt, int kk, int slot) {
        constexpr int S = decltype(set_tag)::value;
        Half& f = hs[S];
        if ((DBG || ALN_ABL) && (dbg & 32)) return;      // ablation: no fragment reads (stale registers)
        const uint32_t aw = lw[kk] + slot * C::W_WAVE;
        const uint32_t aa = la[kk] + kt * C::A_SLAB;
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(f.w[pl]) : "v"(aw + pl * C::W_BLOCK));
            const uint32_t ap = aa + pl * C::A_PLANE;
            asm volatile("ds_read_b128 %0, %1" : "=v"(f.a[pl][0]) : "v"(ap));
            if (MI > 1) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(f.a[pl][1 % MI]) : "v"(ap));
            if (MI > 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.a[pl][2 % MI]) : "v"(ap));
            if (MI > 3) asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(f.a[pl][3 % MI]) : "v"(ap));
        }
        static_assert(MI <= 4, "extend the read list");
    };
    // the fragments of set S have arrived once at most `later` younger reads are outstanding; every register of the set passes
    // through the statement so that no consumer can be scheduled above it
    auto await_set = [&](auto set_tag, auto later_tag) {
        constexpr int S = decltype(set_tag)::value, LATER = decltype(later_tag)::value;
        Half& f = hs[S];
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LATER) : "memory");
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            asm volatile("" : "+v"(f.w[pl]));
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(f.a[pl][i]));
        }
    };
    // the MFMAs of row block I of one half-step (rows of the MFMA = W rows = output columns, columns = token rows)
    auto mfma_block = [&](auto set_tag, auto i_tag) {
        constexpr int S = decltype(set_tag)::value, I = decltype(i_tag)::value;
        const Half& f = hs[S];
        if ((DBG || ALN_ABL) && (dbg & 8)) return;
        if (PL == 2) {
            acc[I][0] = mfma32(f.w[PL - 1], f.a[0][I], acc[I][0]);
            acc[I][0] = mfma32(f.w[0], f.a[PL - 1][I], acc[I][0]);
        }
        acc[I][0] = mfma32(f.w[0], f.a[0][I], acc[I][0]);
    };
    // one half-step: MFMAs of half (kt, kk) from set U % 3 while the reads of the half two half-steps later go to set (U+2) % 3.
    // `tail` = the last two half-steps of a panel issue no reads (the next panel's fill does, from the new image).
    // Everything that is not an MFMA is placed BETWEEN the MFMAs of the half-step (pinned by scheduling barriers).  A wave issues
    // in order, at most one instruction every four cycles, and its next MFMA waits for the pipe: with the half-step laid out as
    // [reads, wait, MFMA x MI, bookkeeping + LDS-DMA issue] the ~30 non-MFMA instructions ran only after the wave's last MFMA had
    // issued, the partner wave of the SIMD was in the same place of the same code, and the matrix pipe idled ~90 of every ~350
    // cycles (ablation: MFMAs + loop skeleton 74 us against 53 us of pure pipe time).  Between two MFMAs of one wave the pipe
    // is busy with the partner's MFMA for 32 cycles = 8 issue slots, which is where these instructions now go.
    // VALID = false: a wave whose 32 columns lie past N in the last tile keeps the stream, the reads and the waits, without MFMAs.
    auto half_step = [&](auto valid_tag, auto u_tag, int bn, int kt, int kk, bool tail) {
        constexpr int U = decltype(u_tag)::value;
        constexpr bool VALID = decltype(valid_tag)::value;
        using Cur = std::integral_constant<int, U % 3>;
        using Nxt = std::integral_constant<int, (U + 2) % 3>;
        // sets U and U+1 are outstanding: U has arrived once at most the RD reads of U+1 are left
        if (!tail) await_set(Cur{}, std::integral_constant<int, RD>{});
        else await_set(Cur{}, std::integral_constant<int, 0>{});
        if (VALID) mfma_block(Cur{}, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        if (!tail) issue_reads(Nxt{}, kt + 1 == NK ? 0 : kt + 1, kk, cs_slot + 1 == RING ? 0 : cs_slot + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (VALID && MI > 1) mfma_block(Cur{}, std::integral_constant<int, 1 % MI>{});
        __builtin_amdgcn_sched_barrier(0);
        bool refill = false;
        if (kk == 1 && !(dbg & 2)) {
            // this stage's fragments are all in registers (set U was its second half): refill its slot with stage step + RING;
            // the private slot needs no barrier
            refill = is_step < total_steps;
            if (refill) issue_next();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (VALID && MI > 2) mfma_block(Cur{}, std::integral_constant<int, 2 % MI>{});
        if (VALID && MI > 3) mfma_block(Cur{}, std::integral_constant<int, 3 % MI>{});
        static_assert(MI <= 4, "extend the MFMA list");
        __builtin_amdgcn_sched_barrier(0);
        if (kk == 1) {
            // make sure stage step + 2 has landed (its first half is read at the next half-step)
            if (!(dbg & 2)) wait_next_stage(refill);
            else ++is_step;
            asm volatile("" ::: "memory");
            cs_slot = cs_slot + 1 == RING ? 0 : cs_slot + 1;
        }
    };
    // three k-steps = six half-steps = two turns of the register sets; j-th group of a tile
    auto kgroup = [&](auto valid_tag, int bn, int j) {
        const int kt0 = 3 * j;
        const bool last_group = bn + 1 == nbn && j + 1 == NK / 3;     // of the panel
        half_step(valid_tag, std::integral_constant<int, 0>{}, bn, kt0, 0, false);
        if (j == 0) {                       // bias of the NEXT tile (wraps to the first tile of the next panel)
            const int bn_next = bn + 1 < nbn ? bn + 1 : 0;
            load_bias(bn_next, bias_nxt);
        }
        half_step(valid_tag, std::integral_constant<int, 1>{}, bn, kt0, 1, false);
        half_step(valid_tag, std::integral_constant<int, 2>{}, bn, kt0 + 1, 0, false);
        // deferred epilogue of the previous tile: block j between the halves of the group's middle k-step
        if (DEFER && have_prev) {
            if (j == 0) drain_block(std::integral_constant<int, 0>{});
            if (NBLK > 1 && j == 1) drain_block(std::integral_constant<int, 1 % NBLK>{});
            if (NBLK > 2 && j == 2) drain_block(std::integral_constant<int, 2 % NBLK>{});
            if (NBLK > 3 && j == 3) drain_block(std::integral_constant<int, 3 % NBLK>{});
        }
        half_step(valid_tag, std::integral_constant<int, 3>{}, bn, kt0 + 1, 1, false);
        half_step(valid_tag, std::integral_constant<int, 4>{}, bn, kt0 + 2, 0, last_group);
        half_step(valid_tag, std::integral_constant<int, 5>{}, bn, kt0 + 2, 1, last_group);
    };
    load_bias(0, bias_cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // once: the first tile's bias and the ring's first RING stages
    for (int pi = 0; pi < my_panels; ++pi) {
        const int panel = blockIdx.x + pi * gridDim.x;
        // ================= LayerNorm prologue: rows of this panel -> bf16 planes in LDS (vision_transformer.py:303, eps 1e-6)
        // (the only two workgroup barriers per panel: the image is rewritten once nobody reads the old one, and read once
        //  everybody has written its rows; LDS-DMA into the private rings stays in flight across both)
        __builtin_amdgcn_s_barrier();
        if (!(dbg & 4)) {
            constexpr int NT = KD / 64;                     // 16-byte loads per lane and row
            const int g = lane >> 4, l = lane & 15;
            f32x4 x[C::RPW / 4][NT];
#pragma unroll
            for (int ps = 0; ps < C::RPW / 4; ++ps) {
                int gm = panel * BM + wave * C::RPW + ps * 4 + g;
                gm = gm < M ? gm : M - 1;
                const float* xr = p.X + (long)gm * p.ldx + l * 4;
#pragma unroll
                for (int it = 0; it < NT; ++it) x[ps][it] = *reinterpret_cast<const f32x4*>(xr + it * 64);
            }
#pragma unroll
            for (int ps = 0; ps < C::RPW / 4; ++ps) {
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
                const int row = wave * C::RPW + ps * 4 + g;
                int gm = panel * BM + row;
                gm = gm < M ? gm : M - 1;
#pragma unroll
                for (int it = 0; it < NT; ++it) {
                    const f32x4 gam = *reinterpret_cast<const f32x4*>(p.gamma + it * 64 + l * 4);
                    const f32x4 bet = *reinterpret_cast<const f32x4*>(p.beta + it * 64 + l * 4);
                    float y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = x[ps][it][e] * rstd * gam[e] + bet[e];
                    uint2 hi, lo;
                    split_bf16x2(y[0], y[1], hi.x, lo.x);
                    split_bf16x2(y[2], y[3], hi.y, lo.y);
                    // column c = it*64 + l*4: k-slab c >> 5, 16-byte chunk (c & 31) >> 3, byte (c & 7) * 2
                    char* dst = sA + (it * 2 + (l >> 3)) * C::A_SLAB + off64(row, (l >> 1) & 3) + (l & 1) * 8;
                    *reinterpret_cast<uint2*>(dst) = hi;
                    if (PL == 2) *reinterpret_cast<uint2*>(dst + C::A_PLANE) = lo;
                    if (p.a_out) {
                        bf16_t* ao = p.a_out + (long)gm * KD + it * 64 + l * 4;
                        *reinterpret_cast<uint2*>(ao) = hi;
                        if (PL == 2) *reinterpret_cast<uint2*>(ao + p.a_plane) = lo;
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
        // fill the fragment pipeline: this panel's first two stages were waited for already (kernel start / the last k-steps of
        // the previous panel)
        issue_reads(std::integral_constant<int, 0>{}, 0, 0, cs_slot);
        issue_reads(std::integral_constant<int, 1>{}, 0, 1, cs_slot);
        for (int bn = 0; bn < nbn; ++bn) {
            const bool cols_valid = bn + 1 < nbn || wave_cols_valid_last;      // wave-uniform
            init_acc(bias_cur);
            if (cols_valid) {
#pragma unroll 1
                for (int j = 0; j < NK / 3; ++j) kgroup(std::true_type{}, bn, j);
            } else {
#pragma unroll 1
                for (int j = 0; j < NK / 3; ++j) kgroup(std::false_type{}, bn, j);
            }
            // ---- the tile is complete: hand it to the deferred epilogue
#pragma unroll
            for (int i = 0; i < MI; ++i) accp[i][0] = acc[i][0];
            have_prev = true;
            prev_valid = cols_valid;
#pragma unroll
            for (int i = 0; i < MI; ++i) prev_row_off[i] = row_off[i];
            prev_n0 = bn * BN + wc * 32;
            if (!DEFER) {       // straight away (the copy above is then only a renaming)
                drain_block(std::integral_constant<int, 0>{});
                if (NBLK > 1) drain_block(std::integral_constant<int, 1 % NBLK>{});
                if (NBLK > 2) drain_block(std::integral_constant<int, 2 % NBLK>{});
                if (NBLK > 3) drain_block(std::integral_constant<int, 3 % NBLK>{});
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) bias_cur[j] = bias_nxt[j];
        }
    }

    // ---- the last tile has no successor to hide behind
    if (DEFER) {
        drain_block(std::integral_constant<int, 0>{});
        if (NBLK > 1) drain_block(std::integral_constant<int, 1 % NBLK>{});
        if (NBLK > 2) drain_block(std::integral_constant<int, 2 % NBLK>{});
        if (NBLK > 3) drain_block(std::integral_constant<int, 3 % NBLK>{});
    }
}

template <int EPI, class C>
static int launch_ln_cfg(const LnGemmParams& p, hipStream_t s) {
    static PerDeviceOnce once;
    if (once.first()) {
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln_kernel<EPI, C, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln_kernel<EPI, C, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        once.mark();
    }
    const int ncu = device_cu_count();
    if (ncu <= 0) return -2;
    const int npanels = (p.M + C::BM - 1) / C::BM;
    const int grid = npanels < ncu ? npanels : ncu;
    if (p.dbg) hipLaunchKernelGGL((gemm_ln_kernel<EPI, C, true>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, s, p);
    else hipLaunchKernelGGL((gemm_ln_kernel<EPI, C, false>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

bool gemm_ln_supported(int K, int N, int planes, int epi, int dmodel) {
    if (K != 384 || (planes != 1 && planes != 2) || (epi != EPI_QKV && epi != EPI_GELU) || N % 32 != 0) return false;
    return epi != EPI_QKV || (dmodel % 64 == 0 && N == 3 * dmodel);
}

int launch_gemm_ln(const LnGemmParams& p0, int K, int planes, hipStream_t s) {
    if (p0.M <= 0) return 0;
    LnGemmParams p = p0;
    p.dbg = options().gemm_dbg;
    const long out_elems = p.epi == EPI_QKV ? p.qkv_plane : (long)p.M * p.ldo;      // 32-bit row offsets inside the kernel
    if (!gemm_ln_supported(K, p.N, planes, p.epi, p.dmodel) || p.ldx % 4 != 0 || out_elems >= (1L << 31)) {
        dinoseg_set_error("gemm_ln: unsupported shape K=%d N=%d planes=%d epi=%d ldx=%d", K, p.N, planes, p.epi, p.ldx);
        return -1;
    }
    if (planes == 1) return launch_gemm_ln12(p, s);
    if (p.epi == EPI_QKV) return launch_ln_cfg<EPI_QKV, aln::Cfg384x2>(p, s);
    return launch_ln_cfg<EPI_GELU, aln::Cfg384x2>(p, s);
}

}  // namespace dseg
