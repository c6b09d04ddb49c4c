// Weight-gradient GEMM on row-major operands ("TN"):  dW[n][k] = sum_m dY[m][n] * X[m][k]   (Linear.weight.grad).
//
// Replaces autograd's `grad_output.t() @ input` (SURVEY.md section 2.1 "bwd" row; reference call site
// pl_torch_modules.py:261-268 via loss.backward()).  The contraction runs over the batch rows m, which is the ROW index
// of both operands, so an NT kernel needs both of them transposed first (the first version did that: 8 transposes per
// transformer block, 16 % of the fine-tune step).  Here both tiles stay row-major in LDS -- [64 m][128 columns] bf16,
// filled by LDS-DMA -- and both MFMA fragments are read column-wise with gfx950's transposing ds_read_b64_tr_b16,
// exactly like the V^T fragments of the attention kernels:  A fragment (rows = n, k = m) from the dY tile, B fragment
// (k = m, columns = k) from the X tile.
//
// Tile 128 (n) x 128 (k) per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = 2 x 2 accumulators), 32 batch rows
// per stage in a four-slot ring: the LDS-DMA of stage c + 3 is issued while stage c is multiplied, one counted `s_waitcnt vmcnt`
// + raw `s_barrier` per stage (inline-asm DMA: behind the builtin hipcc drains the queue in front of every LDS read).  The first
// version -- 64 rows per stage, two slots, `vmcnt(0)` + `__syncthreads()` per stage -- was bound by one HBM latency per stage: 2.9 us
// for 16 MFMAs per wave.  The batch is sliced (split-K); every slice writes its partial tile with plain
// 16-byte stores (GemmParams-style split_stride) and launch_splitk_reduce sums the slices into the gradient.
// Swizzle of the 256-byte rows: 16-byte chunk ^ 2*(row & 3): the 4 rows x 32 bytes a 16-lane group of a transposing read
// touches fall on 4 distinct 32-byte slots of one 128-byte window.
// PLANES = 2 (parity mode): hi + lo planes, three MFMAs per product.
#include "attn_common.h"
#include "kernels.h"

namespace dseg {

namespace tn {
constexpr int BN = 128, BKc = 128, BMr = 32;          // tile: BN rows (n) x BKc columns (k); BMr batch rows per stage
constexpr int TILE = BMr * 256;                       // one [32][128] bf16 tile = 8 KiB
constexpr int RING = 4;                               // stages in LDS: one multiplied, up to three in flight
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row & 3) << 1); }
}  // namespace tn

template <int PLANES>
__global__ __launch_bounds__(256, (PLANES == 1 ? 2 : 1)) void gemm_tn_kernel(TnParams p) {
    using namespace tn;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = PLANES * 2 * TILE;     // per plane: dY tile, X tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    const int tiles_k = p.Kc / BKc;
    // XCD-aware form (1-D grid, ksplit % 8 == 0): every output tile of one batch slice runs on ONE XCD at about the same time, so the
    // slice's rows of dY and X come out of that XCD's L2 for all but the first reader (the 2-D grid dealt the tiles of a slice over the
    // eight XCDs: each of them re-read the rows from the fabric -- 531 MB for fc1's gradient at 8 frames where the operands are 110)
    int tile_id = blockIdx.x, slice = blockIdx.y;
    if (gridDim.y == 1 && p.ksplit > 1) {
        const int tiles = gridDim.x / p.ksplit;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile_id = j % tiles;
        slice = xcd + 8 * (j / tiles);
    }
    const int tn_ = tile_id / tiles_k, tk = tile_id - tn_ * tiles_k;
    const int n0 = tn_ * BN, k0 = tk * BKc;
    const int M = p.M;
    const int nchunks = (M + BMr - 1) / BMr;
    const int per = (nchunks + p.ksplit - 1) / p.ksplit;
    const int c_begin = slice * per;
    const int nc = nchunks - c_begin < per ? nchunks - c_begin : per;
    if (nc <= 0) return;

    // loader: a 1-KiB piece = 4 rows x 256 B; 8 pieces per tile, 2 per wave and tile.  Scalar base + 32-bit lane offset (checked by
    // the launcher), LDS destination through m0.
    constexpr int PP = PLANES * 4;                     // LDS-DMA instructions per wave and stage
    auto dma = [&](const bf16_t* base, uint32_t voff, char* lds) __attribute__((always_inline)) {
        const uint32_t lds_dst = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds);
        const uint64_t src = reinterpret_cast<uint64_t>(base);
        const uint64_t src_u = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)src) |
                               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(src >> 32)) << 32);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(src_u), "s"(lds_dst)
                     : "memory");
    };
    auto stage = [&](int st, int chunk) __attribute__((always_inline)) {
        char* sbase = smem + st * STAGE_BYTES;
        const int prow = lane >> 4, slot = lane & 15;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int piece = wave * 2 + i;
                const int row = piece * 4 + prow;
                const int c = swz(row, slot);              // logical chunk stored in physical slot `slot`
                int gm = chunk * BMr + row;
                gm = gm < M ? gm : M - 1;                  // rows >= M are zeroed at fragment level (ragged last chunk)
                // (a dY narrower than the 128-column tile -- the classifier's 64-column d logits -- repeats its last 8 columns: they
                //  only feed output rows >= N, which are not written)
                const int yc = n0 + c * 8 <= p.ldy - 8 ? n0 + c * 8 : p.ldy - 8;
                dma(p.Y, (uint32_t)((pl * p.y_plane + (long)gm * p.ldy + yc) * 2), sbase + (pl * 2 + 0) * TILE + piece * 1024);
                dma(p.X, (uint32_t)((pl * p.x_plane + (long)gm * p.ldx + k0 + c * 8) * 2), sbase + (pl * 2 + 1) * TILE + piece * 1024);
            }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing reads: within a 16-lane group, lane 4q+p addresses row q, columns 4p..4p+3 of a 4-row x 16-column block and
    // receives column (lane & 15) of the 4 rows.  Group g = lane >> 4: columns 16*(g&1) + 0..15 of a 32-column block, batch
    // rows 8*(g>>1) + 0..3 (second read: +4) of the 16-row step  ->  lane (col = lane & 31, half = lane >> 5) holds rows 8*half+0..7.
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g1 = (lane >> 4) & 1;
    auto frag = [&](const char* tile, int colblock, int ks) {
        const int col = colblock * 32 + tr_g1 * 16 + tr_p * 4;          // first of this lane's 4 columns
        const int row0 = ks * 16 + lh * 8 + tr_q;
        const char* a0 = tile + row0 * 256 + (swz(row0, col >> 3) << 4) + (col & 7) * 2;
        const char* a1 = tile + (row0 + 4) * 256 + (swz(row0 + 4, col >> 3) << 4) + (col & 7) * 2;
        return attn::tr_frag(a0, a1);
    };

    // optional bias gradient (Linear.bias.grad = column sums of dY): the workgroups of the first k-tile multiply their dY
    // fragments with a ones fragment as well -- one extra MFMA per 16 batch rows and wave (wave (wr, wc) sums n-block wc of its
    // 64 rows) instead of a separate pass over dY (transpose_planes_kernel read all of dY for these sums: 37 us per layer)
    const bool do_sum = p.colsum != nullptr && tk == 0;       // workgroup-uniform
    f32x16 bsum;
#pragma unroll
    for (int r = 0; r < 16; ++r) bsum[r] = 0.f;
    const uint4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);

#pragma unroll
    for (int c = 0; c < RING - 1; ++c)
        if (c < nc) stage(c, c_begin + c);

    for (int ci = 0; ci < nc; ++ci) {
        // stage ci landed (this wave's pieces); the up to two younger stages stay in flight
        const int rem = nc - 1 - ci;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PP) : "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // everyone's pieces of stage ci landed; everyone is done reading the slot of stage ci - 1
        if (ci + RING - 1 < nc) stage((ci + RING - 1) & (RING - 1), c_begin + ci + RING - 1);
        const char* sb = smem + (ci & (RING - 1)) * STAGE_BYTES;
        const int m_chunk = (c_begin + ci) * BMr;
        const bool ragged = m_chunk + BMr > M;              // wave-uniform: only the last chunk of the batch
#pragma unroll
        for (int ks = 0; ks < BMr / 16; ++ks) {
            bf16x8 a[PLANES][2], b[PLANES][2];
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[pl][i] = frag(sb + (pl * 2 + 0) * TILE, wr * 2 + i, ks);
                    b[pl][i] = frag(sb + (pl * 2 + 1) * TILE, wc * 2 + i, ks);
                }
            if (ragged) {
                // zero the dY elements of batch rows >= M (their X partners are clamped copies of row M-1: finite)
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        uint4 w = __builtin_bit_cast(uint4, a[pl][i]);
                        uint32_t* ww = reinterpret_cast<uint32_t*>(&w);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int m = m_chunk + ks * 16 + lh * 8 + e;
                            if (m >= M) ww[e >> 1] &= (e & 1) ? 0x0000FFFFu : 0xFFFF0000u;
                        }
                        a[pl][i] = __builtin_bit_cast(bf16x8, w);
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (PLANES == 2) {
                        acc[i][j] = mfma32(a[1][i], b[0][j], acc[i][j]);
                        acc[i][j] = mfma32(a[0][i], b[1][j], acc[i][j]);
                    }
                    acc[i][j] = mfma32(a[0][i], b[0][j], acc[i][j]);
                }
            if (do_sum) {
                if (PLANES == 2) bsum = mfma32(wc ? a[PLANES - 1][1] : a[PLANES - 1][0], ones, bsum);
                bsum = mfma32(wc ? a[0][1] : a[0][0], ones, bsum);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's fragment reads of the stage are complete
    }
    __syncthreads();          // everyone is done with the ring: the partial tile is staged in its place

    if (do_sum && lr == 0) {        // every column of bsum holds the same sums: column 0 adds them
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gn = n0 + wr * 64 + wc * 32 + acc_row(r, lh);
            if (gn < p.N) {
                if (p.det) p.det[(long)slice * p.det_ld + gn] = bsum[r];       // deterministic mode: one partial per batch slice
                else if (bsum[r] != 0.f) atomicAdd(p.colsum + gn, bsum[r]);
            }
        }
    }
    // ---- partial tile: accumulators through LDS C[128][128] fp32, then 16-byte row segments ----
    float* C = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(wr * 64 + i * 32 + acc_row(r, lh)) * 128 + wc * 64 + j * 32 + lr] = acc[i][j][r];
    __syncthreads();
    const int c4 = tid & 31, rb = tid >> 5;
    float* out = p.part + (long)slice * p.split_stride;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int row = it * 8 + rb;
        const int gn = n0 + row;
        if (gn < p.N)
            *reinterpret_cast<f32x4*>(out + (long)gn * p.ld_part + k0 + c4 * 4) = *reinterpret_cast<const f32x4*>(C + row * 128 + c4 * 4);
    }
}

int launch_gemm_tn(const TnParams& p0, hipStream_t s) {
    using namespace tn;
    TnParams p = p0;
    p.det = nullptr;
    p.det_ld = ((p.N + BN - 1) / BN) * BN;
    if (p.colsum && det_scratch().ptr) {
        if ((size_t)p.ksplit * p.det_ld > det_scratch().tn_floats) {      // (never a silent fall-back to atomics in the deterministic mode)
            dinoseg_set_error("gemm_tn: deterministic scratch too small (%d slices x %d columns)", p.ksplit, p.det_ld);
            return -1;
        }
        p.det = det_scratch().tn[p.det_region ? 1 : 0];
    }
    if (p.M < 1 || p.N < 1 || p.Kc % BKc != 0 || p.ksplit < 1 || (p.planes != 1 && p.planes != 2) || p.ldy % 8 != 0 || p.ldx % 8 != 0) {
        dinoseg_set_error("gemm_tn: bad shape M=%d N=%d Kc=%d ksplit=%d planes=%d", p.M, p.N, p.Kc, p.ksplit, p.planes);
        return -1;
    }
    const int tiles = ((p.N + BN - 1) / BN) * (p.Kc / BKc);
    const size_t lds = (size_t)RING * p.planes * 2 * TILE;      // 64 KiB (1 plane) / 128 KiB (2 planes); >= the 64 KiB C tile
    if (((long)p.planes * p.y_plane + (long)p.M * p.ldy) * 2 >= (1L << 32) || ((long)p.planes * p.x_plane + (long)p.M * p.ldx) * 2 >= (1L << 32)) {
        dinoseg_set_error("gemm_tn: operand planes beyond 4 GiB (32-bit lane offsets)");
        return -1;
    }
    const bool xcd_aware = p.ksplit % 8 == 0;
    const dim3 grid = xcd_aware ? dim3(tiles * p.ksplit, 1) : dim3(tiles, p.ksplit);
    if (p.planes == 1) {
        hipLaunchKernelGGL(gemm_tn_kernel<1>, grid, dim3(256), lds, s, p);
    } else {
        static PerDeviceOnce once;
        if (once.first()) {
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<2>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.mark();
    }
        hipLaunchKernelGGL(gemm_tn_kernel<2>, grid, dim3(256), lds, s, p);
    }
    DSEG_CHECK_HIP(hipGetLastError());
    if (p.det) {        // the slices that own batch rows, in slice order
        const int nchunks = (p.M + BMr - 1) / BMr, per = (nchunks + p.ksplit - 1) / p.ksplit, used = (nchunks + per - 1) / per;
        return launch_det_finalize(p.det, used, p.N, p.det_ld, p.colsum, s);
    }
    return 0;
}

}  // namespace dseg
