// Large-tile persistent MFMA GEMM (bf16 fast path):  C[M,N] = A[M,K] . W[N,K]^T, N % 384 == 0, K % 32 == 0.
//
// Why a second GEMM: the 128x128 kernel (gemm.hip) measured ~520 TFLOP/s on every DINOSeg shape while moving
// ~8 TB/s through the L2 -> LDS-DMA path: 64 FLOP per staged byte is too little for this chip.  This kernel
// stages 2.4x fewer bytes per FLOP and keeps more of them in flight:
//   * 256 x 384 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N), wave tile 128 x 96 =
//     4 x 3 v_mfma_f32_32x32x16_bf16 accumulators = 192 registers) -> 154 FLOP per staged byte;
//     every N of the model (384, 1152, 1536; ViT-B 768, 2304, 3072) is a multiple of 384.
//   * BK = 32 k-slabs ([256][32] + [384][32] bf16 = 40 KiB) in a 3-slot LDS ring (120 KiB),
//     filled by LDS-DMA with a counted `s_waitcnt vmcnt(N)` + raw `s_barrier` per slab, so up to 2 slabs (80 KiB)
//     are in flight while one is being multiplied.
//   * persistent: one workgroup per CU walks its tiles; the ring runs continuously across tile boundaries, so
//     the next tile's first slabs land while the current tile's epilogue drains from registers.
//   * smaller persistent tiles with 2-3 workgroups per CU (128x128, 128x192, and 128x384 with the second workgroup
//     started half a tile late) were measured and are slower on every DINOSeg shape: the epilogue is VALU/store-issue
//     work that does not overlap another workgroup's MFMA phase in practice, and the main loop loses intensity.
//   * XCD-aware tile walk: the column tiles of one A row panel are taken by workgroups of one XCD back-to-back.
//   * epilogue through wave-private 4 KiB LDS patches: every global access is 16 bytes per lane on whole 64/128-byte
//     row segments.  (Two earlier versions -- 4 columns per lane touching 32 rows per instruction, and 4-byte
//     row-segment stores straight from the accumulators -- cost 1-4x the main loop: store-issue bound.)
#include "common.h"
#include "kernels.h"

namespace dseg {

namespace big {
constexpr int BK = 32;
// WM x WN waves, each MI x NI accumulators of 32x32; STAGES ring slots; WGS = persistent workgroups per CU
template <int WM_, int WN_, int MI_, int NI_, int STAGES_, int WGS_, int PL_ = 1, int FMT_ = FMT_BF16>
struct Cfg {
    static constexpr int WM = WM_, WN = WN_, MI = MI_, NI = NI_, STAGES = STAGES_, WGS = WGS_, PL = PL_, FMT = FMT_;
    static constexpr int NWAVES = WM * WN, THREADS = NWAVES * 64;
    static constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    static constexpr int A_PLANE_BYTES = BM * BK * 2, W_PLANE_BYTES = BN * BK * 2;        // one operand plane of one k-slab
    static constexpr int A_BYTES = PL * A_PLANE_BYTES, W_BYTES = PL * W_PLANE_BYTES, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int A_PIECES = A_BYTES / 1024;
    static constexpr int PIECES_PER_WAVE = (STAGE_BYTES / 1024) / NWAVES;
    static_assert((STAGE_BYTES / 1024) % NWAVES == 0, "LDS-DMA pieces must divide evenly over the waves");
    static constexpr int WAVES_PER_SIMD = (NWAVES * WGS + 3) / 4;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES + NWAVES * 4096;   // ring + one 32x32 fp32 patch per wave
};
using Cfg256x384 = Cfg<2, 4, 4, 3, 3, 1>;
using Cfg256x384h = Cfg<2, 4, 4, 3, 3, 1, 1, FMT_FP16>;      // the same tile on fp16 operands (GemmParams::fmt)   // 120 KiB ring + 32 KiB epilogue patches, 1 workgroup / CU, 154 FLOP per staged byte
// bf16x3 (hi + lo planes of both operands, 3 MFMAs per product): 128 x 384 tile, wave tile 64 x 96, 2 slots of 64 KiB -- one
// slab in flight behind the one being multiplied, which takes three times as long as a bf16 slab (144 FLOP per staged byte)
using Cfg128x384x2 = Cfg<2, 4, 2, 3, 2, 1, 2>;
using Cfg128x384x2h = Cfg<2, 4, 2, 3, 2, 1, 2, FMT_FP16>;      // ... on fp16 hi + lo planes
// (Cfg<2,2,2,2,2,3> = 128x128 and Cfg<2,2,2,3,2,3> = 128x192 with 3 workgroups per CU were measured: 15-25 % slower;
//  Cfg<1,4,4,3,2,2> = 128x384 with 2 workgroups per CU: 2-28 % slower -- the two workgroups stay in phase, main loop and
//  epilogue times simply add up as with one)

// 64-byte rows (4 chunks of 16 B): XOR the chunk with (row>>2)&3 -> the 16 rows of a ds_read_b128 lane group
// fall on 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int off64(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }
}  // namespace big

template <int EPI, class C>
__global__ __launch_bounds__(C::THREADS, C::WAVES_PER_SIMD) void gemm_big_kernel(GemmParams p) {
    using namespace big;
    constexpr int BM = C::BM, BN = C::BN, STAGES = C::STAGES, STAGE_BYTES = C::STAGE_BYTES, A_BYTES = C::A_BYTES;
    constexpr int PIECES_PER_WAVE = C::PIECES_PER_WAVE, MI = C::MI, NI = C::NI, PL = C::PL;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / C::WN, wc = wave % C::WN;
    const int lr = lane & 31, lh = lane >> 5;
    // the wave's NI = 3 column blocks inside the tile: blocks 0 and 1 are the adjacent halves of 64-column group wc, block 2 is half of
    // group WN + wc / 2 (the other half: the neighbouring wave's) -- the 16-bit epilogues store whole 64-column groups (see there)
    // (the fp32 epilogues keep 96 adjacent columns per wave: their row segments are 128 bytes either way, and the residual one has no
    //  registers to spare for a second set of column offsets)
    constexpr bool GROUPED = NI == 3 && !(EPI == EPI_PLAIN || EPI == EPI_RESID);
    auto col_of = [&](int j) { return GROUPED ? (j < 2 ? wc * 64 + j * 32 : C::WN * 64 + wc * 32) : wc * NI * 32 + j * 32; };

    const int M = p.M, K = p.K;
    const int nk = K / BK;
    const int nbn = p.N / BN, nbm = (M + BM - 1) / BM;

    // ---- tile walk: XCD x (= blockIdx & 7) owns row panels x, x+8, ...; its workgroups take (panel, bn) pairs in order
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int panels_x = (nbm - xcd + 7) >> 3;                    // row panels owned by this XCD
    const int pairs_x = panels_x * nbn;
    // each workgroup takes a CONTIGUOUS run of its XCD's (panel, bn) pairs: its consecutive tiles are the column tiles of
    // one A row panel, so the panel is re-read from L2 while it is still hot (measured HBM over-fetch of the A operand
    // with the strided assignment: qkv 452 MB vs 353 MB algorithmic, fc1 587 vs 442)
    const int cnt = (pairs_x + per_xcd - 1) / per_xcd;
    const int q_begin = local * cnt;
    const int my_tiles = q_begin < pairs_x ? (pairs_x - q_begin < cnt ? pairs_x - q_begin : cnt) : 0;
    const int total_steps = my_tiles * nk;
    if (total_steps == 0) return;

    auto tile_of = [&](int ti, int& bm, int& bn) {
        const int q = q_begin + ti;
        bm = (q / nbn) * 8 + xcd;
        bn = q % nbn;
    };

    // ---- LDS-DMA issue stream: runs STAGES-1 slabs ahead of the multiply stream, across tile boundaries.
    // Per-lane source offsets are set up once per tile (row clamp, row x stride); a k-step then costs one scalar base
    // update and PIECES_PER_WAVE loads.  (The first version recomputed the tile coordinates, the clamp and a 64-bit row
    // product for every piece of every k-step: ~150 scalar / vector instructions ahead of each step's first MFMA.)
    uint32_t voff[PIECES_PER_WAVE];         // byte offset of this lane's 16 bytes inside A (A pieces) / the W column block
    int is_ti = 0, is_kt = 0, is_slot = 0;
    const char* is_wbase = nullptr;
    auto setup_issue_tile = [&](int ti_i) {
        int bm_i, bn_i;
        tile_of(ti_i, bm_i, bn_i);
        is_wbase = reinterpret_cast<const char*>(p.W + (long)(p.n_off + bn_i * BN) * K);
        const int prow = lane >> 2;                         // row inside a 16-row piece
#pragma unroll
        for (int i = 0; i < PIECES_PER_WAVE; ++i) {
            const int piece = wave * PIECES_PER_WAVE + i;   // first A_PIECES pieces: A rows (plane by plane), then W rows
            const bool is_a = piece < C::A_PIECES;          // wave-uniform
            const int idx = is_a ? piece : piece - C::A_PIECES;
            const int per_plane = (is_a ? C::A_PLANE_BYTES : C::W_PLANE_BYTES) / 1024;
            const int pl = idx / per_plane;
            const int row = (idx - pl * per_plane) * 16 + prow;
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            int gm = bm_i * BM + row;
            gm = gm < M ? gm : M - 1;
            const long e = is_a ? pl * p.a_plane + (long)gm * p.lda : pl * p.w_plane + (long)row * K;
            voff[i] = (uint32_t)((e + c * 8) * 2);      // < 2^32: checked by gemm_big_supported
        }
    };
    auto issue_next = [&]() {
        char* sbase = smem + is_slot * STAGE_BYTES;
        const char* abase = reinterpret_cast<const char*>(p.A) + is_kt * (BK * 2);
        const char* wbase = is_wbase + is_kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < PIECES_PER_WAVE; ++i) {
            const int piece = wave * PIECES_PER_WAVE + i;
            glds16((piece < C::A_PIECES ? abase : wbase) + voff[i], sbase + piece * 1024);
        }
        if (++is_kt == nk) {
            is_kt = 0;
            if (++is_ti < my_tiles) setup_issue_tile(is_ti);
        }
        is_slot = is_slot + 1 == STAGES ? 0 : is_slot + 1;
    };
    setup_issue_tile(0);

    f32x16 acc[MI][NI];
    // The bias is the accumulators' initial value (lane = column: one float per 32-column block), so the epilogue has no
    // bias registers and no bias adds.  The next tile's values are fetched at the top of the epilogue, before its stores.
    float bias_col[NI];
    auto bias_ptr = [&](int ti_c) {
        int bm2, bn2;
        tile_of(ti_c, bm2, bn2);
        return p.bias + p.n_off + bn2 * BN + lr;
    };
    auto init_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = bias_col[j];
    };
    {
        const float* bp = bias_ptr(0);
#pragma unroll
        for (int j = 0; j < NI; ++j) bias_col[j] = bp[col_of(j)];
    }
    init_acc();

    // ---- prologue: fill STAGES-1 ring slots
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < total_steps) issue_next();

    int kt = 0, ti = 0, cs_slot = 0;
    for (int step = 0; step < total_steps; ++step) {
        // slab `step` was issued STAGES-1 steps ago; allow the younger ones to stay in flight
        const int ahead = total_steps - 1 - step;           // slabs issued after this one (capped below)
        if (ahead >= STAGES - 2) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PIECES_PER_WAVE) : "memory");
        } else if (ahead == 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * PIECES_PER_WAVE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();   // everyone's pieces of this slab landed; everyone finished reading slot (step-1)%S
        // the two waves of a SIMD (w and w+4) would issue their DMA pieces at the same moment and leave the matrix pipe idle
        // meanwhile (an LDS-DMA piece holds the wave's issue for 60-185 cycles): waves 0-3 issue before the first half of
        // the slab's MFMAs, waves 4-7 between the halves (the target slot is free from the barrier on either way)
        const bool do_issue = step + STAGES - 1 < total_steps && !(p.dbg & 2);
        const bool issue_late = (p.dbg & 4) ? false : wave >= C::NWAVES / 2;
        if (do_issue && !issue_late) issue_next();

        const char* sa = smem + cs_slot * STAGE_BYTES;
        const char* sw = sa + A_BYTES;
        cs_slot = cs_slot + 1 == STAGES ? 0 : cs_slot + 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // (the backend's MFMA / DS interleaving strategy: parity mode 830 -> 835 frames/s; the bf16 configuration loses 1 % with it)
            if constexpr (PL == 2) __builtin_amdgcn_iglp_opt(0);
            bf16x8 af[PL][MI], wf[PL][NI];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[pl][i] = lds_frag(sa + pl * C::A_PLANE_BYTES + off64(wr * MI * 32 + i * 32 + lr, kk * 2 + lh));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    wf[pl][j] = lds_frag(sw + pl * C::W_PLANE_BYTES + off64(col_of(j) + lr, kk * 2 + lh));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {      // lane: column n, registers: rows m
                    if (PL == 2) {                  // small terms first, as in gemm.hip
                        acc[i][j] = mfma32f<C::FMT>(af[PL - 1][i], wf[0][j], acc[i][j]);
                        acc[i][j] = mfma32f<C::FMT>(af[0][i], wf[PL - 1][j], acc[i][j]);
                    }
                    acc[i][j] = mfma32f<C::FMT>(af[0][i], wf[0][j], acc[i][j]);
                }
            if (kk == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (do_issue && issue_late) issue_next();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        if (++kt == nk) {
            // ================= epilogue from registers =================
            // Each 32x32 accumulator goes through a wave-private 4 KiB LDS patch (no workgroup barrier: LDS requests of
            // one wave complete in order) and leaves as 16-byte-per-lane row segments: 8 rows x 128 B per fp32 store
            // instruction, 16 rows x 64 B per bf16 one.  (Narrower stores were store-issue bound: 4-byte stores straight
            // from the accumulators cost as much as the main loop.)
            // The body is straight-line code on purpose.  gfx9 counts loads AND stores in one in-order vmcnt; with a global
            // load (bias) or a branch (row guard) inside, the compiler's wait insertion falls back to vmcnt(0) before every
            // use / in every block, and each store then waits for the previous one to be acknowledged (24-48 serial round
            // trips per wave and tile: that was most of the "epilogue costs as much as the main loop" of the first version).
            // So: no global loads besides the next tile's bias at the top, rows clamped instead of guarded (rows >= M of the
            // last panel are copies of row M-1 -- the loader clamps the same way -- so their stores rewrite identical bytes).
            int bm, bn;
            tile_of(ti, bm, bn);
            kt = 0;
            ++ti;
            const int m0 = bm * BM + wr * MI * 32, n0 = p.n_off + bn * BN;      // (+ col_of(j): the wave's column blocks)
            const bool skip_epi = (p.dbg & 1) && acc[0][0][0] != 12345.678f;   // ablation (keeps accumulators live)
            float* st = reinterpret_cast<float*>(smem + STAGES * STAGE_BYTES + wave * 4096);
            constexpr bool F32_OUT = (EPI == EPI_PLAIN || EPI == EPI_RESID);
            const float* bnext = bias_ptr(ti < my_tiles ? ti : my_tiles - 1);
            float bias_next[NI];
            if (EPI == EPI_RESID) {
                // loads the compiler does not count (see below); complete before the last counted wait of the segment ring
#pragma unroll
                for (int j = 0; j < NI; ++j) asm volatile("global_load_dword %0, %1, off" : "=v"(bias_next[j]) : "v"(bnext + col_of(j)) : "memory");
            } else {
#pragma unroll
                for (int j = 0; j < NI; ++j) bias_next[j] = bnext[col_of(j)];
            }
            if (!skip_epi) {
            if (EPI == EPI_RESID) {
                // x += acc, in place.  The 48 row segments of the wave's tile run through a 4-deep register ring: the load
                // of segment s+4 is issued before the store of segment s, by instructions the compiler does not count, with
                // hand-counted vmcnt (cdna_hip_programming.md 5.7 form ii).  Addresses are a wave-uniform SGPR base per
                // segment + one per-lane VGPR offset.  (A deeper ring spills: 243 VGPRs as is.)
                const int cg = lane & 7;
                const bool full = m0 + MI * 32 <= M;      // wave-uniform; the ragged last row panel takes the plain path
                if (full) {
                    constexpr int NSEG = MI * NI * 4, LA = 4;
                    const uint32_t voff = (uint32_t)(((lane >> 3) * p.ldo_f32 + cg * 4) * 4);
                    float* tile_base = p.out_f32 + (long)m0 * p.ldo_f32 + n0;
                    auto seg_base = [&](int sg) {
                        const int b = sg >> 2, ps = sg & 3, j = b / MI, i = b % MI;
                        return tile_base + (long)(i * 32 + 8 * ps) * p.ldo_f32 + col_of(j);
                    };
                    f32x4 ring[LA];
#pragma unroll
                    for (int sg = 0; sg < LA; ++sg)
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ring[sg]) : "v"(voff), "s"(seg_base(sg)) : "memory");
#pragma unroll
                    for (int sg = 0; sg < NSEG; ++sg) {
                        const int b = sg >> 2, ps = sg & 3, j = b / MI, i = b % MI;
                        if (ps == 0) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) st[acc_row(r, lh) * 32 + lr] = acc[i][j][r];
                        }
                        // operations younger than this segment's load: the later loads issued so far + the stores since
                        const int younger = ((sg + LA - 1 < NSEG - 1 ? sg + LA - 1 : NSEG - 1) - sg) + (sg < LA ? sg : LA);
                        f32x4& cur = ring[sg % LA];
#define DSEG_WAIT_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(cur)::"memory"); break;
                        switch (younger) {      // compile-time after unrolling
                            DSEG_WAIT_VM(0) DSEG_WAIT_VM(1) DSEG_WAIT_VM(2) DSEG_WAIT_VM(3) DSEG_WAIT_VM(4) DSEG_WAIT_VM(5)
                            DSEG_WAIT_VM(6) DSEG_WAIT_VM(7) DSEG_WAIT_VM(8) DSEG_WAIT_VM(9) DSEG_WAIT_VM(10) DSEG_WAIT_VM(11)
                            DSEG_WAIT_VM(12) DSEG_WAIT_VM(13) DSEG_WAIT_VM(14) DSEG_WAIT_VM(15)
                        }
#undef DSEG_WAIT_VM
                        static_assert(2 * LA - 1 <= 15, "extend the vmcnt ladder");
                        const int row = (lane >> 3) + 8 * ps;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(st + row * 32 + cg * 4) + cur;
                        if (sg + LA < NSEG)
                            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ring[sg % LA]) : "v"(voff), "s"(seg_base(sg + LA)) : "memory");
                        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(seg_base(sg)) : "memory");
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < MI * NI; ++b) {
                        const int j = b / MI, i = b % MI;
                        const int gn = n0 + col_of(j) + cg * 4;
#pragma unroll
                        for (int r = 0; r < 16; ++r) st[acc_row(r, lh) * 32 + lr] = acc[i][j][r];
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const int row = (lane >> 3) + 8 * ps;
                            const int gm = m0 + i * 32 + row;
                            if (gm < M) {
                                float* dst = p.out_f32 + (long)gm * p.ldo_f32 + gn;
                                *reinterpret_cast<f32x4*>(dst) =
                                    *reinterpret_cast<const f32x4*>(dst) + *reinterpret_cast<const f32x4*>(st + row * 32 + cg * 4);
                            }
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the uncounted bias loads above
                }
                asm volatile("" : "+v"(bias_next[0]), "+v"(bias_next[1]), "+v"(bias_next[2]));
                static_assert(NI == 3, "bias anchor names three registers");
            } else if constexpr (F32_OUT) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int cg = lane & 7;
                const int gn = n0 + col_of(j) + cg * 4;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[acc_row(r, lh) * 32 + lr] = acc[i][j][r];
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = (lane >> 3) + 8 * ps;
                        int gm = m0 + i * 32 + row;
                        gm = gm < M ? gm : M - 1;
                        *reinterpret_cast<f32x4*>(p.out_f32 + (long)gm * p.ldo_f32 + gn) =
                            *reinterpret_cast<const f32x4*>(st + row * 32 + cg * 4);
                    }
                }
            }
            } else {
                // 16-bit outputs leave as 128-BYTE row segments: 8 lanes x 16 B, 8 rows per store instruction.  (Measured on the fc1 output of
                // the parity mode -- 708 MB, tools/store_pattern.hip, profiles/r05_gemm_epilogue_stores.md: HBM takes 64-byte row segments at
                // 3.2 TB/s, 128-byte ones at 5.3, whole 768-byte rows at 6.6; the first version of this epilogue stored one 32-column block
                // = 64 bytes per row at a time and its stores were 182 of fc1's 534 us.)  A 64-column group goes through the wave's 4 KiB
                // patch as two halves of [16 rows][64 columns] fp32.  The wave's blocks 0 and 1 are one group; its block 2 shares a group
                // with the neighbouring wave's block 2 (wc ^ 1, same rows): the two write their halves of rows 0..15 into the even wave's
                // patch and of rows 16..31 into the odd wave's, and after a workgroup barrier each stores the 16 rows in its own patch.
                static_assert(NI == 3 && C::WN % 2 == 0, "column groups of the 16-bit epilogue");
                const int cg = lane & 7, prow = lane >> 3;
                // patch element (row, col): the two lane halves of an accumulator register write rows 4 apart -- column ^ 32 for odd row / 4
                auto pidx = [](int row, int col) { return row * 64 + (col ^ (((row >> 2) & 1) << 5)); };
                auto out64 = [&](const float* patch, int gm_first, int gn) __attribute__((always_inline)) {
                    int which = 0, hcol = 0;
                    bf16_t* qkv_base = nullptr;
                    if (EPI == EPI_QKV) {
                        which = gn / p.dmodel;
                        hcol = gn - which * p.dmodel;
                        qkv_base = which == 0 ? p.q : (which == 1 ? p.k : p.v);
                    }
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int row = prow + 8 * ps;
                        int gm = gm_first + row;
                        gm = gm < M ? gm : M - 1;
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(patch + pidx(row, cg * 8));
                        const f32x4 v1 = *reinterpret_cast<const f32x4*>(patch + pidx(row, cg * 8) + 4);
                        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (EPI == EPI_GELU) v[e] = PL == 1 ? gelu_fast(v[e]) : gelu_erf(v[e]);
                            if (EPI == EPI_RELU) v[e] = fmaxf(v[e], 0.f);
                            if (EPI == EPI_QKV && which == 0) v[e] *= p.qscale;
                        }
                        uint4 o[2];
                        if (C::FMT == FMT_FP16 && !(EPI == EPI_QKV && which == 2 && (PL == 1 || p.v_bf16))) {      // (one plane, or asked for: V stays bf16)
                            if (PL == 2) {      // (outputs are unbounded: saturate at the fp16 range)
                                split2<C::FMT, true>(v[0], v[1], o[0].x, o[1].x);
                                split2<C::FMT, true>(v[2], v[3], o[0].y, o[1].y);
                                split2<C::FMT, true>(v[4], v[5], o[0].z, o[1].z);
                                split2<C::FMT, true>(v[6], v[7], o[0].w, o[1].w);
                            } else {
                                o[0].x = pack2_sat<C::FMT>(v[0], v[1]);
                                o[0].y = pack2_sat<C::FMT>(v[2], v[3]);
                                o[0].z = pack2_sat<C::FMT>(v[4], v[5]);
                                o[0].w = pack2_sat<C::FMT>(v[6], v[7]);
                                o[1] = o[0];
                            }
                        } else {
                            split_bf16x2(v[0], v[1], o[0].x, o[1].x);
                            split_bf16x2(v[2], v[3], o[0].y, o[1].y);
                            split_bf16x2(v[4], v[5], o[0].z, o[1].z);
                            split_bf16x2(v[6], v[7], o[0].w, o[1].w);
                        }
                        bf16_t* dst;
                        long pstride;
                        if (EPI == EPI_QKV) {
                            const int bq = gm / p.ntok, tok = gm - bq * p.ntok;
                            dst = qkv_base + ((long)(bq * p.heads + (hcol >> 6)) * p.npad + tok) * 64 + cg * 8;
                            pstride = p.qkv_plane;
                        } else {
                            dst = p.out_bf16 + (long)gm * p.ldo + gn + cg * 8;
                            pstride = p.out_plane;
                        }
                        *reinterpret_cast<uint4*>(dst) = o[0];
                        if (PL == 2) *reinterpret_cast<uint4*>(dst + pstride) = o[1];
                    }
                };
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                            for (int r = 0; r < 8; ++r) st[pidx(acc_row(r, lh), jj * 32 + lr)] = acc[i][jj][8 * h + r];
                        out64(st, m0 + i * 32 + 16 * h, n0 + col_of(0));
                    }
                float* const pair_patch = reinterpret_cast<float*>(smem + STAGES * STAGE_BYTES + (wave & ~1) * 4096);
                const int odd = wave & 1;
                auto wg_barrier = [] {      // (a raw s_barrier is no compiler fence for LDS accesses)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                };
                wg_barrier();       // the neighbour is done with the patch this wave writes into next
#pragma unroll
                for (int i = 0; i < MI; ++i) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 8; ++r) pair_patch[h * 1024 + pidx(acc_row(r, lh), odd * 32 + lr)] = acc[i][2][8 * h + r];
                    wg_barrier();
                    out64(st, m0 + i * 32 + 16 * odd, n0 + C::WN * 64 + (wc & ~1) * 32);
                    wg_barrier();
                }
            }
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) bias_col[j] = bias_next[j];
            init_acc();
        }
    }
}

template <int EPI, class C>
static int launch_big_cfg(const GemmParams& p, hipStream_t s) {
    static PerDeviceOnce once;
    if (once.first()) {
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_big_kernel<EPI, C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        once.mark();
    }
    const int ncu = device_cu_count();
    if (ncu <= 0) return -2;
    const int nbn = p.N / C::BN, nbm = (p.M + C::BM - 1) / C::BM;
    // persistent workgroups: WGS per CU, a multiple of the 8 XCDs; no more per XCD than it has (panel, bn) pairs
    int per_xcd = ncu / 8 * C::WGS;
    const int pairs_per_xcd = ((nbm + 7) / 8) * nbn;
    if (per_xcd > pairs_per_xcd) per_xcd = pairs_per_xcd;
    if (per_xcd < 1) per_xcd = 1;
    const int grid = per_xcd * 8;
    hipLaunchKernelGGL((gemm_big_kernel<EPI, C>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int EPI>
static int launch_big_one(const GemmParams& p, hipStream_t s) {
    if (p.fmt == FMT_FP16) {
        if constexpr (EPI == EPI_RESID || EPI == EPI_GELU || EPI == EPI_QKV || EPI == EPI_RELU)
            return p.planes == 2 ? launch_big_cfg<EPI, big::Cfg128x384x2h>(p, s) : launch_big_cfg<EPI, big::Cfg256x384h>(p, s);
        dinoseg_set_error("gemm_big: the fp16 operand format covers the inference epilogues only (epi=%d)", (int)EPI);
        return -1;
    }
    if (p.planes == 2) return launch_big_cfg<EPI, big::Cfg128x384x2>(p, s);
#ifdef BIG_12WAVES      // experiment: three waves per SIMD, 192 x 384 tile (3 x 4 waves of 64 x 96)
    return launch_big_cfg<EPI, big::Cfg<3, 4, 2, 3, 3, 1>>(p, s);
#else
    return launch_big_cfg<EPI, big::Cfg256x384>(p, s);
#endif
}

bool gemm_big_supported(const GemmParams& p) {
    static_assert(big::Cfg128x384x2::BN == big::Cfg256x384::BN && big::Cfg128x384x2::LDS_BYTES <= 160 * 1024, "one N rule, LDS budget");
    const long a_span = (long)(p.planes - 1) * p.a_plane + (long)p.M * p.lda;                  // elements a lane offset can reach
    const long w_span = (long)(p.planes - 1) * p.w_plane + (long)big::Cfg256x384::BN * p.K;
    return (p.planes == 1 || p.planes == 2) && p.bias != nullptr && p.epi <= EPI_QKV && p.resid == nullptr && p.aux_out == nullptr &&
           p.ksplit <= 1 && p.N % big::Cfg256x384::BN == 0 && p.K % big::BK == 0 && p.lda % 8 == 0 && p.M >= 1 &&
           a_span * 2 < (1L << 32) && w_span * 2 < (1L << 32) &&     // 32-bit lane offsets

           (p.epi != EPI_QKV || (p.dmodel % big::Cfg256x384::BN == 0 && p.N == 3 * p.dmodel));
}

int launch_gemm_big(const GemmParams& p, hipStream_t s) {
    switch (p.epi) {
        case EPI_PLAIN: return launch_big_one<EPI_PLAIN>(p, s);
        case EPI_RESID: return launch_big_one<EPI_RESID>(p, s);
        case EPI_GELU: return launch_big_one<EPI_GELU>(p, s);
        case EPI_RELU: return launch_big_one<EPI_RELU>(p, s);
        case EPI_QKV: return launch_big_one<EPI_QKV>(p, s);
    }
    dinoseg_set_error("gemm_big: bad epilogue %d", p.epi);
    return -1;
}

}  // namespace dseg
