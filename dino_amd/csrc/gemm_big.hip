// Large-tile persistent MFMA GEMM (bf16 fast path):  C[M,N] = A[M,K] . W[N,K]^T, N % 384 == 0, K % 32 == 0.
//
// Why a second GEMM: the 128x128 kernel (gemm.hip) measured ~520 TFLOP/s on every DINOSeg shape while moving
// ~8 TB/s through the L2 -> LDS-DMA path: 64 FLOP per staged byte is too little for this chip.  This kernel
// stages 2.4x fewer bytes per FLOP and keeps more of them in flight:
//   * 256 x 384 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N), wave tile 128 x 96 =
//     4 x 3 v_mfma_f32_32x32x16_bf16 accumulators = 192 registers) -> 154 FLOP per staged byte;
//     every N of the model (384, 1152, 1536; ViT-B 768, 2304, 3072) is a multiple of 384.
//   * BK = 32 k-slabs ([256][32] + [384][32] bf16 = 40 KiB) in a 4-slot LDS ring = all 160 KiB of the CU,
//     filled by LDS-DMA with a counted `s_waitcnt vmcnt(N)` + raw `s_barrier` per slab, so 2-3 slabs (80-120 KiB)
//     are in flight while one is being multiplied.
//   * persistent: one workgroup per CU walks its tiles; the ring runs continuously across tile boundaries, so
//     the next tile's first slabs land while the current tile's epilogue drains from registers.
//   * XCD-aware tile walk: the column tiles of one A row panel are taken by workgroups of one XCD back-to-back.
//   * epilogue straight from registers (the ring owns all of LDS): in the 32x32 accumulator a register's 32 lanes
//     are 32 consecutive columns of one row, so fp32 stores are whole 128-byte row segments; bf16 outputs pair
//     adjacent lanes with one DPP move and store 64-byte row segments.  (A first version with 4 columns per lane
//     and 16-byte accesses touched 32 rows per instruction and cost 4x the main loop.)
#include "common.h"
#include "kernels.h"

namespace dseg {

namespace big {
constexpr int BM = 256, BN = 384, BK = 32, STAGES = 4;
constexpr int A_BYTES = BM * BK * 2;           // 16 KiB
constexpr int W_BYTES = BN * BK * 2;           // 24 KiB
constexpr int STAGE_BYTES = A_BYTES + W_BYTES; // 40 KiB
constexpr int PIECES_PER_WAVE = (STAGE_BYTES / 1024) / 8;   // 5 LDS-DMA instructions per wave per slab

// 64-byte rows (4 chunks of 16 B): XOR the chunk with (row>>2)&3 -> the 16 rows of a ds_read_b128 lane group
// fall on 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int off64(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }
}  // namespace big

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(GemmParams p) {
    using namespace big;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;         // 2 x 4 waves
    const int lr = lane & 31, lh = lane >> 5;

    const int M = p.M, K = p.K;
    const int nk = K / BK;
    const int nbn = p.N / BN, nbm = (M + BM - 1) / BM;

    // ---- tile walk: XCD x (= blockIdx & 7) owns row panels x, x+8, ...; its workgroups take (panel, bn) pairs in order
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int panels_x = (nbm - xcd + 7) >> 3;                    // row panels owned by this XCD
    const int pairs_x = panels_x * nbn;
    const int my_tiles = pairs_x > local ? (pairs_x - local + per_xcd - 1) / per_xcd : 0;
    const int total_steps = my_tiles * nk;
    if (total_steps == 0) return;

    auto tile_of = [&](int ti, int& bm, int& bn) {
        const int q = ti * per_xcd + local;
        bm = (q / nbn) * 8 + xcd;
        bn = q % nbn;
    };

    auto issue = [&](int step) {
        const int ti = step / nk, kt = step - ti * nk;
        int bm, bn;
        tile_of(ti, bm, bn);
        char* sbase = smem + (step % STAGES) * STAGE_BYTES;
        const int prow = lane >> 2;                         // row inside a 16-row piece
#pragma unroll
        for (int i = 0; i < PIECES_PER_WAVE; ++i) {
            const int piece = wave * PIECES_PER_WAVE + i;   // 0..15: A rows, 16..39: W rows
            const bf16_t* src;
            if (piece < 16) {
                const int row = piece * 16 + prow;
                const int c = (lane & 3) ^ ((row >> 2) & 3);
                int gm = bm * BM + row;
                gm = gm < M ? gm : M - 1;
                src = p.A + (long)gm * p.lda + kt * BK + c * 8;
            } else {
                const int row = (piece - 16) * 16 + prow;
                const int c = (lane & 3) ^ ((row >> 2) & 3);
                src = p.W + (long)(p.n_off + bn * BN + row) * K + kt * BK + c * 8;
            }
            glds16(src, sbase + piece * 1024);
        }
    };

    f32x16 acc[4][3];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    // ---- prologue: fill STAGES-1 ring slots
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < total_steps) issue(s);

    int kt = 0, ti = 0;
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
        if (step + STAGES - 1 < total_steps && !(p.dbg & 2)) issue(step + STAGES - 1);

        const char* sa = smem + (step % STAGES) * STAGE_BYTES;
        const char* sw = sa + A_BYTES;
        int bm, bn;
        tile_of(ti, bm, bn);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], wf[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = lds_frag(sa + off64(wr * 128 + i * 32 + lr, kk * 2 + lh));
#pragma unroll
            for (int j = 0; j < 3; ++j) wf[j] = lds_frag(sw + off64(wc * 96 + j * 32 + lr, kk * 2 + lh));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = mfma32(af[i], wf[j], acc[i][j]);   // lane: column n, registers: rows m
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        if (++kt == nk) {
            // ================= epilogue from registers =================
            kt = 0;
            ++ti;
            const int m0 = bm * BM + wr * 128, n0 = p.n_off + bn * BN + wc * 96;
            const bool skip_epi = (p.dbg & 1) && acc[0][0][0] != 12345.678f;   // ablation (keeps accumulators live)
            if (!skip_epi) {
            // acc[i][j][r] = C[m0 + i*32 + acc_row(r, lh)][n0 + j*32 + lr]: for a fixed register the 32 lanes of a
            // half-wave hold 32 consecutive columns of one row -> every store instruction writes whole 64/128-byte
            // row segments (2 rows x 128 B for fp32; 4 rows x 64 B for bf16 after pairing adjacent lanes by DPP).
            float bv[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) bv[j] = p.bias[n0 + j * 32 + lr];
            const int odd = lane & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int mrow0 = m0 + i * 32 + 4 * lh;            // row of register 0; register r adds (r&3) + 8*(r>>2)
                if (EPI == EPI_PLAIN || EPI == EPI_RESID || EPI == EPI_PATCH) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int gn = n0 + j * 32 + lr;
#pragma unroll
                        for (int half = 0; half < 2; ++half) {       // batches of 8 rows: loads issued ahead of the stores
                            float x[8];
                            if (EPI == EPI_RESID || EPI == EPI_PATCH) {
#pragma unroll
                                for (int rr = 0; rr < 8; ++rr) {
                                    const int r = half * 8 + rr;
                                    int gm = mrow0 + (r & 3) + 8 * (r >> 2);
                                    gm = gm < M ? gm : M - 1;
                                    if (EPI == EPI_RESID) {
                                        x[rr] = p.out_f32[(long)gm * p.ldo_f32 + gn];
                                    } else {
                                        const int tok = gm % p.n_patches;
                                        x[rr] = p.pos[(long)(1 + tok) * p.ldo_f32 + gn];
                                    }
                                }
                            }
#pragma unroll
                            for (int rr = 0; rr < 8; ++rr) {
                                const int r = half * 8 + rr;
                                const int gm = mrow0 + (r & 3) + 8 * (r >> 2);
                                if (gm < M) {
                                    float v = acc[i][j][r] + bv[j];
                                    long orow = gm;
                                    if (EPI == EPI_RESID) v += x[rr];
                                    if (EPI == EPI_PATCH) {
                                        v += x[rr];
                                        orow = gm + gm / p.n_patches + 1;       // b*(n+1) + 1 + tok
                                    }
                                    p.out_f32[orow * p.ldo_f32 + gn] = v;
                                }
                            }
                        }
                    }
                } else {
                    // bf16 outputs: even lane stores (col n, n+1) of row A, odd lane (col n-1, n) of row A+1
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int ra = 2 * u, rb = 2 * u + 1;
                        const int gmA = mrow0 + (ra & 3) + 8 * (ra >> 2);
                        const int gm = gmA + odd;
                        long rowbase = 0;
                        if (EPI == EPI_QKV) {
                            int b = gm / p.ntok;
                            const int tok = gm - b * p.ntok;
                            rowbase = ((long)b * p.heads * p.npad + tok) * 64;
                        } else {
                            rowbase = (long)gm * p.ldo;
                        }
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            float va = acc[i][j][ra] + bv[j], vb = acc[i][j][rb] + bv[j];
                            const int gn = n0 + j * 32 + lr;
                            if (EPI == EPI_GELU) {
                                va = gelu_erf(va);
                                vb = gelu_erf(vb);
                            } else if (EPI == EPI_RELU) {
                                va = fmaxf(va, 0.f);
                                vb = fmaxf(vb, 0.f);
                            } else if (EPI == EPI_QKV) {
                                if (gn < p.dmodel) {       // Q columns (uniform per j: 32 | dmodel)
                                    va *= p.qscale;
                                    vb *= p.qscale;
                                }
                            }
                            const float send = odd ? va : vb;
                            const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                                0, __builtin_bit_cast(int, send), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
                            const uint32_t packed = odd ? pack_bf16x2(recv, vb) : pack_bf16x2(va, recv);
                            const int gc = gn - odd;           // first of the two columns this lane stores
                            if (gm < M) {
                                if (EPI == EPI_QKV) {
                                    const int which = gc / p.dmodel;
                                    const int hcol = gc - which * p.dmodel;
                                    bf16_t* basep = which == 0 ? p.q : (which == 1 ? p.k : p.v);
                                    *reinterpret_cast<uint32_t*>(basep + rowbase + (long)(hcol >> 6) * p.npad * 64 + (hcol & 63)) = packed;
                                } else {
                                    *reinterpret_cast<uint32_t*>(p.out_bf16 + rowbase + gc) = packed;
                                }
                            }
                        }
                    }
                }
            }
            }
            zero_acc();
        }
    }
}

template <int EPI>
static int launch_big_one(const GemmParams& p, hipStream_t s) {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        DSEG_CHECK_HIP(hipGetDevice(&dev));
        DSEG_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_big_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, big::STAGES * big::STAGE_BYTES));
    }
    const int nbn = p.N / big::BN, nbm = (p.M + big::BM - 1) / big::BM;
    // one persistent workgroup per CU, a multiple of the 8 XCDs; no more per XCD than it has (panel, bn) pairs
    int per_xcd = ncu / 8;
    const int pairs_per_xcd = ((nbm + 7) / 8) * nbn;
    if (per_xcd > pairs_per_xcd) per_xcd = pairs_per_xcd;
    if (per_xcd < 1) per_xcd = 1;
    const int grid = per_xcd * 8;
    hipLaunchKernelGGL((gemm_big_kernel<EPI>), dim3(grid), dim3(512), big::STAGES * big::STAGE_BYTES, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

bool gemm_big_supported(const GemmParams& p) {
    return p.planes == 1 && p.bias != nullptr && p.epi <= EPI_QKV && p.resid == nullptr && p.aux_out == nullptr &&
           p.ksplit <= 1 && p.N % big::BN == 0 && p.K % big::BK == 0 && p.lda % 8 == 0 && p.M >= 1 &&
           (p.epi != EPI_QKV || (p.dmodel % big::BN == 0 && p.N == 3 * p.dmodel));
}

int launch_gemm_big(const GemmParams& p, hipStream_t s) {
    switch (p.epi) {
        case EPI_PLAIN: return launch_big_one<EPI_PLAIN>(p, s);
        case EPI_RESID: return launch_big_one<EPI_RESID>(p, s);
        case EPI_GELU: return launch_big_one<EPI_GELU>(p, s);
        case EPI_RELU: return launch_big_one<EPI_RELU>(p, s);
        case EPI_QKV: return launch_big_one<EPI_QKV>(p, s);
        case EPI_PATCH: return launch_big_one<EPI_PATCH>(p, s);
    }
    dinoseg_set_error("gemm_big: bad epilogue %d", p.epi);
    return -1;
}

}  // namespace dseg
