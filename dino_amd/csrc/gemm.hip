// MFMA GEMM for the DINOSeg linear layers:  C[M,N] = A[M,K] . W[N,K]^T  (+ fused epilogues).
//
// Replaces the reference's aten::addmm dispatches: attn.qkv (vision_transformer.py:75,82),
// attn.proj (:105) + residual (:134), mlp.fc1 + GELU (:60-61), mlp.fc2 (:63) + residual (:135), the patch-embed
// conv as a GEMM (:153,157) and the first two head layers (pl_torch_modules.py:118-121).
//
// Structure (gfx950): 128x128 output tile per 256-thread workgroup, 4 waves as 2x2, each wave 64x64 =
// 2x2 v_mfma_f32_32x32x16_bf16 accumulators.  BK = 64: A and W k-slabs are [128][64] bf16 = 128-byte rows,
// filled by LDS-DMA (global_load_lds 16 B/lane) into a double-buffered, XOR-swizzled image (common.h) and read
// with conflict-free ds_read_b128.  PLANES = 2 is the parity mode: every operand is a bf16 hi+lo pair and each
// product is 3 MFMAs (hi*hi + hi*lo + lo*hi), ~16 mantissa bits with fp32 accumulation.
// The accumulators are staged through LDS once so every epilogue writes whole rows (coalesced).
// GELU uses the A&S 7.1.26 erfc form (|err| <= 1.5e-7): libm erff cost more VALU time than the fc1 main loop.
#include "common.h"
#include "kernels.h"

namespace dseg {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 128;   // one [128][64] bf16 slab

// FMT (inference epilogues only): FMT_FP16 = A, W and the 16-bit outputs are fp16 (one plane: EPI_QKV leaves V bf16; hi+lo: all fp16)
// HALFM (EPI_RESID only): 64 x 128 tiles, wave tile 32 x 64 -- twice the workgroups for a small batch (one frame @480: attn.proj and
// mlp.fc2 are 29 x 3 = 87 tiles of 128 x 128 on 256 CUs).  Every output element sees the same MFMAs in the same order: bit-identical.
template <int PLANES, int EPI, int FMT = FMT_BF16, bool HALFM = false>
__global__ __launch_bounds__(256, (PLANES == 1 ? 2 : 1)) void gemm_nt_kernel(GemmParams p) {
    static_assert(!HALFM || EPI == EPI_RESID, "64-row tiles: the residual epilogue only");
    constexpr int BM = HALFM ? 64 : 128;     // (shadows the file-scope tile height)
    constexpr int MI = HALFM ? 1 : 2;
    static_assert(FMT == FMT_BF16 || EPI == EPI_RESID || EPI == EPI_GELU || EPI == EPI_QKV || EPI == EPI_PATCH || (PLANES == 2 && EPI == EPI_RELU),
                  "fp16 operands: inference epilogues only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = PLANES * 2 * TILE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    // XCD-aware tile order: the nbn column tiles of one row panel run back-to-back on one XCD (A panel L2 reuse).
    const int nbn = p.N / BN, nbm = (p.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bm = (slot / nbn) * 8 + xcd, bn = slot % nbn;
    if (bm >= nbm) return;
    const int m0 = bm * BM, n0 = bn * BN;
    const int M = p.M, K = p.K;

    auto stage = [&](int st, int kt) {
        char* sbase = smem + st * STAGE_BYTES;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int piece = wave * 4 + i;            // 1-KiB piece = 8 rows
                const int row = piece * 8 + (lane >> 3);
                const int c = swz_chunk(row, lane & 7);    // logical chunk that lives in physical slot lane&7
                if (!HALFM || i < 2) {                     // (64-row tiles: 8 A pieces, two per wave)
                    const int apiece = HALFM ? wave * 2 + i : piece;
                    const int arow = apiece * 8 + (lane >> 3);
                    int gm = m0 + arow;
                    gm = gm < M ? gm : M - 1;
                    const bf16_t* srcA = p.A + pl * p.a_plane + (long)gm * p.lda + kt * BK + swz_chunk(arow, lane & 7) * 8;
                    glds16(srcA, sbase + (pl * 2 + 0) * TILE_BYTES + apiece * 1024);
                }
                const bf16_t* srcW = p.W + pl * p.w_plane + (long)(n0 + row) * K + kt * BK + c * 8;
                glds16(srcW, sbase + (pl * 2 + 1) * TILE_BYTES + piece * 1024);
            }
        }
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = K / BK;
    int kt_begin = 0, nk = nk_all;
    if (p.ksplit > 1) {     // grid.y slices of the K loop (weight gradients: K = rows of the batch)
        const int per = (nk_all + p.ksplit - 1) / p.ksplit;
        kt_begin = blockIdx.y * per;
        nk = nk_all - kt_begin < per ? nk_all - kt_begin : per;
        if (nk <= 0) return;
    }
    stage(0, kt_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk && !(p.dbg & 2)) stage(cur ^ 1, kt_begin + kt + 1);
        const char* sb = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[PLANES][MI], b[PLANES][2];
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    a[pl][i] = lds_frag(sb + (pl * 2 + 0) * TILE_BYTES + tile_off_bytes(wr * (MI * 32) + i * 32 + lr, kk * 2 + lh));
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    b[pl][i] = lds_frag(sb + (pl * 2 + 1) * TILE_BYTES + tile_off_bytes(wc * 64 + i * 32 + lr, kk * 2 + lh));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (PLANES == 2) {
                        acc[i][j] = mfma32f<FMT>(a[1][i], b[0][j], acc[i][j]);
                        acc[i][j] = mfma32f<FMT>(a[0][i], b[1][j], acc[i][j]);
                    }
                    acc[i][j] = mfma32f<FMT>(a[0][i], b[0][j], acc[i][j]);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if ((p.dbg & 1) && acc[0][0][0] != 12345.678f) return;   // ablation: no epilogue (keeps the accumulators live)
    // ---- stage accumulators through LDS: C[BM][128] fp32 (64 KiB at 128 rows) ----
    float* C = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(wr * (MI * 32) + i * 32 + acc_row(r, lh)) * 128 + wc * 64 + j * 32 + lr] = acc[i][j][r];
    __syncthreads();

    // ---- bf16-plane outputs: 8 columns per thread -> one 16-byte store per plane (8-byte stores were
    //      store-issue bound: ~2.4 TB/s of output against 4-6 TB/s for the 16-byte fp32 epilogues)
    constexpr bool BF16_OUT = (EPI == EPI_GELU || EPI == EPI_RELU || EPI == EPI_QKV || EPI == EPI_BF16 ||
                               EPI == EPI_DGELU || EPI == EPI_DRELU);
    if (BF16_OUT) {
        const int c8 = tid & 15, rb16 = tid >> 4;
        const int gn = n0 + c8 * 8;
        float b8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[e] = p.bias ? p.bias[gn + e] : 0.f;
        // QKV scatter: a 128-column tile lies inside one of Q / K / V, an 8-column group inside one head
        const int which = (EPI == EPI_QKV) ? n0 / p.dmodel : 0;
        const int hcol = (EPI == EPI_QKV) ? n0 % p.dmodel + c8 * 8 : 0;
        bf16_t* qkv_base = (EPI == EPI_QKV) ? (which == 0 ? p.q : (which == 1 ? p.k : p.v)) : nullptr;
        // backward epilogues: the saved pre-activations of all eight row steps are fetched up front (inside the loop every step waited
        // for its own load: eight HBM latencies per workgroup, 149 us for fc2's input gradient at 8 frames where the GEMM itself is 34 GFLOP)
        uint4 aux_h[8], aux_l[8];
        if (EPI == EPI_DGELU || EPI == EPI_DRELU) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int gmr = m0 + it * 16 + rb16;
                const bf16_t* ax = p.aux_in + (long)(gmr < M ? gmr : M - 1) * p.ldo + gn;
                aux_h[it] = *reinterpret_cast<const uint4*>(ax);
                if (PLANES == 2 && EPI == EPI_DGELU) aux_l[it] = *reinterpret_cast<const uint4*>(ax + p.aux_plane);
            }
        }
        auto row_step = [&](int it, const uint4& ah_in, const uint4& al_in) __attribute__((always_inline)) -> bool {
            const int row = it * 16 + rb16;
            const int gm = m0 + row;
            if (gm >= M) return false;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(C + row * 128 + c8 * 8);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(C + row * 128 + c8 * 8 + 4);
            float v[8] = {v0[0] + b8[0], v0[1] + b8[1], v0[2] + b8[2], v0[3] + b8[3],
                          v1[0] + b8[4], v1[1] + b8[5], v1[2] + b8[6], v1[3] + b8[7]};
            if (EPI == EPI_GELU && p.aux_out) {        // training: keep the pre-activation for gelu'
                bf16_t* ad = p.aux_out + (long)gm * p.ldo + gn;
                uint4 ph, plo;
                split_bf16x2(v[0], v[1], ph.x, plo.x);
                split_bf16x2(v[2], v[3], ph.y, plo.y);
                split_bf16x2(v[4], v[5], ph.z, plo.z);
                split_bf16x2(v[6], v[7], ph.w, plo.w);
                *reinterpret_cast<uint4*>(ad) = ph;
                if (PLANES == 2) *reinterpret_cast<uint4*>(ad + p.aux_plane) = plo;
            }
            if (EPI == EPI_DGELU || EPI == EPI_DRELU) {
                const uint4 ah = ah_in;
                const uint32_t aw[4] = {ah.x, ah.y, ah.z, ah.w};
                float a[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[2 * e] = bf16_lo_to_f32(aw[e]);
                    a[2 * e + 1] = bf16_hi_to_f32(aw[e]);
                }
                if (PLANES == 2 && EPI == EPI_DGELU) {
                    const uint4 al = al_in;
                    const uint32_t lw[4] = {al.x, al.y, al.z, al.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[2 * e] += bf16_lo_to_f32(lw[e]);
                        a[2 * e + 1] += bf16_hi_to_f32(lw[e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= (EPI == EPI_DGELU) ? gelu_erf_grad(a[e]) : (a[e] > 0.f ? 1.f : 0.f);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (EPI == EPI_GELU) v[e] = PLANES == 1 ? gelu_fast(v[e]) : gelu_erf(v[e]);
                if (EPI == EPI_RELU) v[e] = fmaxf(v[e], 0.f);
                if (EPI == EPI_QKV && which == 0) v[e] *= p.qscale;
            }
            uint4 hi, lo;
            if (FMT == FMT_FP16 && !(EPI == EPI_QKV && which == 2 && (PLANES == 1 || p.v_bf16))) {      // (which is workgroup-uniform)
                // (GEMM outputs are unbounded: saturate at the fp16 range instead of producing inf)
                if (PLANES == 2) {
                    split2<FMT, true>(v[0], v[1], hi.x, lo.x);
                    split2<FMT, true>(v[2], v[3], hi.y, lo.y);
                    split2<FMT, true>(v[4], v[5], hi.z, lo.z);
                    split2<FMT, true>(v[6], v[7], hi.w, lo.w);
                } else {
                    hi.x = pack2_sat<FMT>(v[0], v[1]);
                    hi.y = pack2_sat<FMT>(v[2], v[3]);
                    hi.z = pack2_sat<FMT>(v[4], v[5]);
                    hi.w = pack2_sat<FMT>(v[6], v[7]);
                    lo = hi;
                }
            } else {
                split_bf16x2(v[0], v[1], hi.x, lo.x);
                split_bf16x2(v[2], v[3], hi.y, lo.y);
                split_bf16x2(v[4], v[5], hi.z, lo.z);
                split_bf16x2(v[6], v[7], hi.w, lo.w);
            }
            bf16_t* dst;
            long plane_stride;
            if (EPI == EPI_QKV) {
                const int b = gm / p.ntok, tok = gm - b * p.ntok;
                dst = qkv_base + ((long)(b * p.heads + (hcol >> 6)) * p.npad + tok) * 64 + (hcol & 63);
                plane_stride = p.qkv_plane;
            } else {
                dst = p.out_bf16 + (long)gm * p.ldo + gn;
                plane_stride = p.out_plane;
            }
            *reinterpret_cast<uint4*>(dst) = hi;
            if (PLANES == 2) *reinterpret_cast<uint4*>(dst + plane_stride) = lo;
            return true;
        };
        if constexpr (EPI == EPI_DGELU || EPI == EPI_DRELU) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
                if (!row_step(it, aux_h[it], aux_l[it])) break;
        } else {
            const uint4 none = {0u, 0u, 0u, 0u};
#pragma unroll 2
            for (int it = 0; it < 8; ++it)
                if (!row_step(it, none, none)) break;
        }
        return;
    }

    const int c4 = tid & 31, rb = tid >> 5;
    const int gn = n0 + c4 * 4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + gn);
    if (EPI == EPI_RESID) {
        // out = resid + acc + bias (usually in place).  The residual rows are fetched 8 at a time before any store:
        // a load issued after a store to the same buffer waits for the store's acknowledgement (one vmcnt for both).
#pragma unroll
        for (int it0 = 0; it0 < BM / 8; it0 += 8) {
            f32x4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int gm = m0 + (it0 + u) * 8 + rb;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const float* src = (p.resid ? p.resid : p.out_f32) + (long)gm * p.ldo_f32 + gn;
                x[u] = gm < M ? *reinterpret_cast<const f32x4*>(src) : z;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = (it0 + u) * 8 + rb;
                const int gm = m0 + row;
                const f32x4 v = *reinterpret_cast<const f32x4*>(C + row * 128 + c4 * 4) + bias4 + x[u];
                if (gm < M) *reinterpret_cast<f32x4*>(p.out_f32 + (long)gm * p.ldo_f32 + gn) = v;
            }
        }
        return;
    }
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int row = it * 8 + rb;
        const int gm = m0 + row;
        if (gm >= M) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(C + row * 128 + c4 * 4);
        v += bias4;
        if (EPI == EPI_PLAIN) {
            *reinterpret_cast<f32x4*>(p.out_f32 + (long)blockIdx.y * p.split_stride + (long)gm * p.ldo_f32 + gn) = v;
        } else if (EPI == EPI_ATOMIC) {
            float* dst = p.out_f32 + (long)gm * p.ldo_f32 + gn;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (p.n_valid == 0 || gn + e < p.n_valid) atomicAdd(dst + e, v[e]);
        } else if (EPI == EPI_PATCH) {
            const int b = gm / p.n_patches, pi = gm - b * p.n_patches;
            const long xrow = (long)b * (p.n_patches + 1) + 1 + pi;
            f32x4 pe = *reinterpret_cast<const f32x4*>(p.pos + (long)(1 + pi) * p.ldo_f32 + gn);
            *reinterpret_cast<f32x4*>(p.out_f32 + xrow * p.ldo_f32 + gn) = v + pe;
        }
    }
}

template <int PLANES, int EPI, int FMT = FMT_BF16, bool HALFM = false>
static int launch_one(const GemmParams& p, hipStream_t s) {
    constexpr int BMt = HALFM ? 64 : BM;
    const int nbn = p.N / BN, nbm = (p.M + BMt - 1) / BMt;
    const int grid = ((nbm + 7) / 8) * 8 * nbn;
    const size_t lds = (size_t)PLANES * 2 * 2 * TILE_BYTES;
    static PerDeviceOnce once;
    if (once.first()) {
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<PLANES, EPI, FMT, HALFM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.mark();
    }
    hipLaunchKernelGGL((gemm_nt_kernel<PLANES, EPI, FMT, HALFM>), dim3(grid, p.ksplit > 1 ? p.ksplit : 1), dim3(256), lds, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm(const GemmParams& p0, hipStream_t s) {
    if (p0.M <= 0) return 0;
    GemmParams p = p0;
    p.dbg = options().gemm_dbg;
    if (options().gemm_big && gemm_big_supported(p)) {
        // measured on MI355X (tools/bench_ops.py): the 256x384 persistent kernel wins when there are >= 2 tiles per CU
        // or the K loop is long; attn.proj (451 tiles, K = 384) is faster on the 128x128 kernel (tile quantisation)
        // A small batch (single-frame predict: M = 3601 -> 15 row panels) leaves most CUs without a 256x384 tile:
        // below 128 tiles the 128x128 kernel has 3-6x more workgroups to spread (fc2 at B = 1: 60 -> 40 us).
        // (planes == 2: 128-row tiles of three times the work)
        const int bm_big = p.planes == 2 ? 128 : 256;
        const int m_disp = p.dispatch_rows > 0 ? p.dispatch_rows : p.M;
        const long tiles = (long)((m_disp + bm_big - 1) / bm_big) * (p.N / 384);
        // (K >= 768: ViT-B's attn.proj at 8-16 frames -- 226-450 tiles -- 1.66 -> 1.41 ms per step on the persistent kernel)
        if (options().gemm_big > 1 || p.K % BK != 0 || (tiles >= 128 && (tiles >= 512 || p.K >= 768))) return launch_gemm_big(p, s);
    }
    return launch_gemm_small(p, s);
}

int launch_gemm_small(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0) return 0;
    if (p.N % BN != 0 || p.K % BK != 0 || p.lda % 8 != 0) {
        dinoseg_set_error("gemm: unsupported shape M=%d N=%d K=%d lda=%d (need N%%128==0, K%%64==0)", p.M, p.N, p.K, p.lda);
        return -1;
    }
    if (p.epi == EPI_QKV && (p.dmodel % 128 != 0 || p.N != 3 * p.dmodel)) {
        dinoseg_set_error("gemm: QKV epilogue needs dmodel%%128==0 and N==3*dmodel");
        return -1;
    }
    // the residual GEMMs of a small batch (attn.proj, mlp.fc2: N = embed_dim, three column tiles): 64-row tiles while 128-row ones
    // would leave half the CUs without a workgroup (one frame @480: 87 -> 171 workgroups; fc2 24 -> 15 us, proj 12.5 -> 8.5)
    {
        const int ncu = device_cu_count();
        const long wgs128 = (long)((p.M + 127) / 128) * (p.N / BN);
        if (p.epi == EPI_RESID && p.ksplit <= 1 && ncu > 0 && 2 * wgs128 <= ncu && !(options().route_ab & 1)) {
            if (p.fmt == FMT_FP16) {
                if (p.planes == 1) return launch_one<1, EPI_RESID, FMT_FP16, true>(p, s);
                if (p.planes == 2) return launch_one<2, EPI_RESID, FMT_FP16, true>(p, s);
            } else {
                if (p.planes == 1) return launch_one<1, EPI_RESID, FMT_BF16, true>(p, s);
                if (p.planes == 2) return launch_one<2, EPI_RESID, FMT_BF16, true>(p, s);
            }
        }
    }
    if (p.fmt == FMT_FP16) {
        if (p.planes == 1 && p.epi == EPI_RESID) return launch_one<1, EPI_RESID, FMT_FP16>(p, s);
        if (p.planes == 1 && p.epi == EPI_GELU && p.aux_out == nullptr) return launch_one<1, EPI_GELU, FMT_FP16>(p, s);
        if (p.planes == 1 && p.epi == EPI_QKV) return launch_one<1, EPI_QKV, FMT_FP16>(p, s);
        if (p.planes == 1 && p.epi == EPI_PATCH) return launch_one<1, EPI_PATCH, FMT_FP16>(p, s);
        if (p.planes == 2 && p.epi == EPI_RESID) return launch_one<2, EPI_RESID, FMT_FP16>(p, s);
        if (p.planes == 2 && p.epi == EPI_GELU && p.aux_out == nullptr) return launch_one<2, EPI_GELU, FMT_FP16>(p, s);
        if (p.planes == 2 && p.epi == EPI_QKV) return launch_one<2, EPI_QKV, FMT_FP16>(p, s);
        if (p.planes == 2 && p.epi == EPI_PATCH) return launch_one<2, EPI_PATCH, FMT_FP16>(p, s);
        if (p.planes == 2 && p.epi == EPI_RELU) return launch_one<2, EPI_RELU, FMT_FP16>(p, s);
        dinoseg_set_error("gemm: the fp16 operand format covers the inference epilogues only (planes=%d epi=%d)", p.planes, p.epi);
        return -1;
    }
#define DSEG_CASE(PL, E) \
    if (p.planes == PL && p.epi == E) return launch_one<PL, E>(p, s);
    DSEG_CASE(1, EPI_PLAIN) DSEG_CASE(2, EPI_PLAIN)
    DSEG_CASE(1, EPI_RESID) DSEG_CASE(2, EPI_RESID)
    DSEG_CASE(1, EPI_GELU) DSEG_CASE(2, EPI_GELU)
    DSEG_CASE(1, EPI_RELU) DSEG_CASE(2, EPI_RELU)
    DSEG_CASE(1, EPI_QKV) DSEG_CASE(2, EPI_QKV)
    DSEG_CASE(1, EPI_PATCH) DSEG_CASE(2, EPI_PATCH)
    DSEG_CASE(1, EPI_BF16) DSEG_CASE(2, EPI_BF16)
    DSEG_CASE(1, EPI_ATOMIC) DSEG_CASE(2, EPI_ATOMIC)
    DSEG_CASE(1, EPI_DGELU) DSEG_CASE(2, EPI_DGELU)
    DSEG_CASE(1, EPI_DRELU) DSEG_CASE(2, EPI_DRELU)
#undef DSEG_CASE
    dinoseg_set_error("gemm: bad planes/epilogue %d/%d", p.planes, p.epi);
    return -1;
}

}  // namespace dseg
