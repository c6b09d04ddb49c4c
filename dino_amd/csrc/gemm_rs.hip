// Row-stationary streaming GEMMs, one operand plane, for the wide model (ViT-B/8: embed_dim 768):   C[M, N] = A[M, K] . W[N, K]^T
//
// gemm_big.hip's 256 x 384 tiles run ViT-B's four linears at ~0.32 of the MFMA peak (profiles/r05_vitb_kernel_stats.csv): its epilogue
// (LDS transposition, GELU, 350-MB stores) has the chip to itself once per tile, and the multiply stops every wave at one barrier per 24 MFMAs.
// These two kernels take the structure that mlp_fused3.hip runs at the power limit: a workgroup is FOUR waves, one per SIMD with the whole
// 512-register file; a wave owns 32 rows for a whole item (128 rows per workgroup) and keeps ITS operand in registers; the weights stream
// through a three-slot LDS ring (48-KiB slots packed in MFMA fragment order: a fragment = one linear 1-KiB LDS-DMA piece = one conflict-free
// ds_read_b128), one barrier per 48 MFMAs, every load / store / epilogue instruction placed in an MFMA gap.
//   gemm_bstat (K <= 768: qkv, fc1)   the A rows are the stationary B operand: xn[K / 16] fragments loaded straight from A[M][K] (16 bytes of a
//                                     row per lane = one fragment).  Step n: Z^T[32 features][32 rows] = bias + W_n . xn^T (K / 16 MFMAs); the
//                                     epilogue of tile n - 1 (GELU / the Q scale, pack, two 16-byte stores) rides in the gaps; the next item's
//                                     fragments are prefetched one load per step.
//   gemm_cstat (N <= 768: proj, fc2)  the output rows are the stationary accumulators: o^T[N][32 rows] = x + W . a^T in N / 32 blocks
//                                     (384 registers at N = 768).  Step kt: the 32-wide k-tile kt of W against two B fragments loaded just in
//                                     time from A (three rotating buffers); the last step stores each finished block (+ bias) and loads
//                                     the next item's residual rows in its place.
// Single-plane modes only (bf16 / fp16); the weights are re-packed once per refresh (launch_pack_rs).
#include "rs_common.h"

namespace dseg {


// W [N][K] fp32 -> kind 0 (bstat): [tile n = N / 32][fragment ks = K / 16][64 lanes][8]: A row = output feature 32 n + sigma23(lane & 31),
//                                  k = 16 ks + 8 (lane >> 5) + e
//               -> kind 1 (cstat, N = 768): [half nh of N][pair k2 of k-tiles = K / 64][fragment 24 kk + 2 db + s2][64 lanes][8]: A row = output
//                                  feature 384 nh + 32 db + sigma23(lane & 31), k = 64 k2 + 32 kk + 16 s2 + 8 (lane >> 5) + e
// gamma (kind 0, optional): the weight of the LayerNorm in front of the linear, folded into its columns -- LayerNorm(x) W^T + b =
// ((x - mean) rstd) (W . diag(gamma))^T + (b + W beta): the kernel's LayerNorm prologue then needs no per-feature constants (fold_ln_bias_kernel)
__global__ __launch_bounds__(256) void pack_rs_kernel(const float* __restrict__ W, int N, int K, int kind, bf16_t* __restrict__ dst, int fmt,
                                                      const float* __restrict__ gamma) {
    const long total = (long)N * K;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long q = idx;
        const int e = (int)(q & 7); q >>= 3;
        const int lane = (int)(q & 63); q >>= 6;
        const int i = attn::sigma23(lane & 31), h = lane >> 5;
        float v;
        if (kind == 0) {
            const int ks = (int)(q % (K / 16)), n = (int)(q / (K / 16));
            v = W[(long)(n * 32 + i) * K + ks * 16 + h * 8 + e];
            if (gamma) v *= gamma[ks * 16 + h * 8 + e];
        } else {      // [half of N][pair of k-tiles][fragment 24 kk + 2 db + s2]
            const int fr = (int)(q % 48); q /= 48;
            const int k2 = (int)(q % (K / 64)), nh = (int)(q / (K / 64));
            const int kk = fr / 24, db = (fr % 24) >> 1, s2 = fr & 1;
            v = W[(long)(nh * (N / 2) + db * 32 + i) * K + (2 * k2 + kk) * 32 + s2 * 16 + h * 8 + e];
        }
        dst[idx] = pack1(v, fmt);
    }
}

int launch_pack_rs(const float* W, int N, int K, int kind, bf16_t* dst, hipStream_t s, int fmt) {
    if (!W || !dst || N % 32 != 0 || K % 64 != 0 || (kind != 0 && kind != 1) || (kind == 1 && N != 768)) {
        dinoseg_set_error("pack_rs: null pointer or unsupported shape N=%d K=%d kind=%d", N, K, kind);
        return -1;
    }
    hipLaunchKernelGGL(pack_rs_kernel, dim3(1024), dim3(256), 0, s, W, N, K, kind, dst, fmt, (const float*)nullptr);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// out[n] = bias[n] + sum_k W[n][k] beta[k]  (fp32; one wave per output feature)
__global__ __launch_bounds__(256) void fold_ln_bias_kernel(const float* __restrict__ W, const float* __restrict__ beta, const float* __restrict__ bias,
                                                           int N, int K, float* __restrict__ out) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(W[(long)n * K + k], beta[k], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[n] = bias[n] + acc;
}

// the kind-0 copy of W with the LayerNorm in front of the linear folded in (gamma into the columns, beta into the bias): what launch_gemm_rs
// reads when GemmParams::ln_x is set
int launch_pack_rs_ln(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K, bf16_t* dst_w, float* dst_bias,
                      hipStream_t s, int fmt) {
    if (!W || !gamma || !beta || !bias || !dst_w || !dst_bias || N % 32 != 0 || K % 64 != 0) {
        dinoseg_set_error("pack_rs_ln: null pointer or unsupported shape N=%d K=%d", N, K);
        return -1;
    }
    hipLaunchKernelGGL(pack_rs_kernel, dim3(1024), dim3(256), 0, s, W, N, K, 0, dst_w, fmt, gamma);
    hipLaunchKernelGGL(fold_ln_bias_kernel, dim3((N + 3) / 4), dim3(256), 0, s, W, beta, bias, N, K, dst_bias);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ================================================================================================ gemm_bstat: qkv / fc1
// EPI_GELU: out16[M][N] = gelu_fast(A W^T + b) (one plane, FMT);  EPI_QKV: Q (pre-scaled) / K / V [B, heads, npad, 64], V bf16 whatever FMT is.
// The epilogue stores 16 bytes per lane straight from the packed accumulator values (a lane holds 8 consecutive features of its row twice).
// (Round 6 also built the output through a wave-private LDS patch -- whole 128-byte row segments -- with the bias as an MFMA fragment loaded by
//  an uncounted asm instruction: the same time, launch for launch (profiles/r06_gemm_rs.md), and two hazards hipcc does not pad around inline asm
//  on the way: a VALU-written SGPR pair read by an asm VMEM instruction needs `s_nop 4` in front, an asm store's data registers two wait states
//  behind.  The plain form below has neither.)
// LN: the stationary rows are (ln_x - mean) rstd computed here -- the fp32 row of a lane (384 values) is loaded once, normalised in registers (two-pass
// statistics, one cross-half shuffle) and packed into the fragments; the LayerNorm's weight and bias are folded into p.W / p.bias (launch_pack_rs_ln):
// no LayerNorm launch, no 16-bit A round trip (vision_transformer.py:122 / :134)
template <int FMT, int EPI, int KS, bool LN>
__global__ __launch_bounds__(rs::THREADS, 1) void gemm_bstat_kernel(GemmParams p) {
    using namespace rs;
    static_assert(KS == 48, "the gap programs below are written for 48-gap steps (K = 768)");
    constexpr int SLOTB = KS * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, ntiles = p.N / 32;
    const int nitems = (M + BM - 1) / BM;
    if ((int)blockIdx.x >= nitems) return;
    float* const sBias = reinterpret_cast<float*>(smem + RING * SLOTB);
    for (int i = tid; i < p.N / 4; i += THREADS) reinterpret_cast<f32x4*>(sBias)[i] = reinterpret_cast<const f32x4*>(p.bias)[i];
    RsStream st;
    st.wp = reinterpret_cast<uint64_t>(p.W);
    st.lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    st.piece0 = (uint32_t)wave * (KS / NW) * 1024;
    st.sn = 0; st.ipos = 0; st.rpos = 0; st.nslots = ntiles;
    rs_prologue<KS>(st, (uint32_t)(tid & 63) * 16);
    __syncthreads();

    bf16x8 xn[KS];
    f32x16 Z0, Z1;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        uint32_t zero = 0;
        asm volatile("" : "+v"(zero));
        const uint32_t lane_i = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
        const uint32_t lane16_i = lane_i * 16, lh_i = lane_i >> 5, lr_i = lane_i & 31;
        const uint32_t frag_rd_i = st.lds_base + lane16_i;
        const int r_ = item * BM + wave * 32 + (int)lr_i;
        const int rc = r_ < M ? r_ : M - 1;      // (clamped: a row past the end is a copy of row M - 1 and rewrites its bytes)
        if constexpr (LN) {
            const float* xr = p.ln_x + (long)rc * (KS * 16) + lh_i * 8;
            f32x4 r[2 * KS];
#pragma unroll
            for (int j = 0; j < 2 * KS; ++j) r[j] = *reinterpret_cast<const f32x4*>(xr + (j >> 1) * 16 + (j & 1) * 4);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * KS; j += 2) sum += ((r[j][0] + r[j][1]) + (r[j][2] + r[j][3])) + ((r[j + 1][0] + r[j + 1][1]) + (r[j + 1][2] + r[j + 1][3]));
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.0f / (KS * 16));
            float qv = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * KS; ++j) {
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dlt = r[j][e] - mean;
                    part = fmaf(dlt, dlt, part);
                }
                qv += part;
            }
            qv += __shfl_xor(qv, 32);
            const float rstd = 1.0f / sqrtf(qv * (1.0f / (KS * 16)) + p.ln_eps);
            // (no per-feature constants: the LayerNorm's weight and bias are folded into W and the bias -- launch_pack_rs_ln)
            const float nmr = -mean * rstd;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = fmaf(r[2 * k][e], rstd, nmr);
                    y[4 + e] = fmaf(r[2 * k + 1][e], rstd, nmr);
                }
                uint4 u;
                u.x = pack2_sat<FMT>(y[0], y[1]);
                u.y = pack2_sat<FMT>(y[2], y[3]);
                u.z = pack2_sat<FMT>(y[4], y[5]);
                u.w = pack2_sat<FMT>(y[6], y[7]);
                xn[k] = __builtin_bit_cast(bf16x8, u);
            }
        } else {
            const bf16_t* ar = p.A + (long)rc * p.lda + lh_i * 8;
            mf_for(std::make_integer_sequence<int, KS>{}, [&](auto k_tag) __attribute__((always_inline)) {
                constexpr int k = decltype(k_tag)::value;
                uint4 u = {0x3c003c00u, 0x3c003c00u, (uint32_t)k, 0u};
                if (!(RS_ABL & 32)) u = *reinterpret_cast<const uint4*>(ar + k * 16);
                xn[k] = __builtin_bit_cast(bf16x8, u);
            });
        }
        long orow;      // destination of this lane's row
        if constexpr (EPI == EPI_QKV) {
            const int fr_ = rc / p.ntok, tok_ = rc - fr_ * p.ntok;
            orow = ((long)fr_ * p.heads * p.npad + tok_) * 64 + lh_i * 8;
        } else {
            orow = (long)rc * p.ldo + lh_i * 8;
        }

        float ez[16], eu[16], eq[16];
        uint32_t zp[8];
        auto gelu_op = [&](auto n_tag, auto i_tag, const f32x16& z) __attribute__((always_inline)) {
            constexpr int N = decltype(n_tag)::value, I = decltype(i_tag)::value;
            if (RS_ABL & 1) {
                if constexpr (I == 0) ez[N] = z[N];
                return;
            }
            if constexpr (I == 0) eu[N] = __builtin_amdgcn_fmed3f(z[N], -8.0f, 8.0f);
            if constexpr (I == 1) eq[N] = eu[N] * eu[N];
            if constexpr (I == 2) ez[N] = fmaf(1.01537542e-3f, eq[N], -1.06782573e-1f);
            if constexpr (I == 3) ez[N] = fmaf(ez[N], eq[N], -2.30111381f);
            if constexpr (I == 4) ez[N] = ez[N] * eu[N];
            if constexpr (I == 5) ez[N] = __builtin_amdgcn_exp2f(ez[N]);
            if constexpr (I == 6) ez[N] = 1.0f + ez[N];
            if constexpr (I == 7) ez[N] = __builtin_amdgcn_rcpf(ez[N]);
            if constexpr (I == 8) ez[N] = z[N] * ez[N];
        };
        // the epilogue of tile tq (values in z), gap G of the step that carries it: element n's instruction i at gap START(n) + i, a pair packed one
        // gap after its odd element, the two 16-byte stores in the last two gaps (behind the step's pieces)
        auto epilogue_gap = [&](auto g_tag, const f32x16& z, int tq) __attribute__((always_inline)) {
            constexpr int G = decltype(g_tag)::value;
            constexpr int NOPS = EPI == EPI_GELU ? 9 : 1;
            mf_for(std::make_integer_sequence<int, 16>{}, [&](auto n_tag) __attribute__((always_inline)) {
                constexpr int N = decltype(n_tag)::value;
                constexpr int START = EPI == EPI_GELU ? (5 * N) / 2 + 1 : 2 * N + 8;
                if constexpr (G >= START && G < START + NOPS) {
                    if constexpr (EPI == EPI_GELU) gelu_op(n_tag, std::integral_constant<int, G - START>{}, z);
                    else ez[N] = z[N] * (tq < p.dmodel / 32 ? p.qscale : 1.0f);
                }
                if constexpr ((N & 1) == 1 && G == START + NOPS) {
                    if constexpr (EPI == EPI_QKV) {
                        // (V stays bf16 in the fp16 mode: the zero-reference attention's P.V product; Q / K saturate like gemm_big.hip)
                        if (FMT == FMT_BF16 || tq >= 2 * (p.dmodel / 32)) zp[N >> 1] = pack_bf16x2(ez[N - 1], ez[N]);
                        else zp[N >> 1] = pack2_sat<FMT>(ez[N - 1], ez[N]);
                    } else {
                        zp[N >> 1] = pack2_sat<FMT>(ez[N - 1], ez[N]);
                    }
                }
            });
            if constexpr (G >= KS - 2) {
                constexpr int GG = G - (KS - 2);
                bf16_t* dst;
                if constexpr (EPI == EPI_QKV) {
                    const int nd = p.dmodel / 32, which = tq / nd, hb = tq - which * nd;
                    dst = (which == 0 ? p.q : (which == 1 ? p.k : p.v)) + orow + (long)(hb >> 1) * p.npad * 64 + (hb & 1) * 32 + GG * 16;
                } else {
                    dst = p.out_bf16 + orow + tq * 32 + GG * 16;
                }
                const uint4 u = {zp[4 * GG], zp[4 * GG + 1], zp[4 * GG + 2], zp[4 * GG + 3]};
                if (!(RS_ABL & 64)) *reinterpret_cast<uint4*>(dst) = u;
                else asm volatile("" ::"v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w), "v"(dst));
            }
        };
        // (the four reads go out in front of the step's barrier, the accumulator is written behind the step's first fragment reads)
        f32x4 zc0, zc1, zc2, zc3;
        auto z_bias_load = [&](int n) __attribute__((always_inline)) {
            const float* bp = sBias + n * 32 + lh_i * 8;
            zc0 = *reinterpret_cast<const f32x4*>(bp); zc1 = *reinterpret_cast<const f32x4*>(bp + 4);
            zc2 = *reinterpret_cast<const f32x4*>(bp + 16); zc3 = *reinterpret_cast<const f32x4*>(bp + 20);
        };
        auto z_bias_set = [&](f32x16& z) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z[e] = zc0[e];
                z[4 + e] = zc1[e];
                z[8 + e] = zc2[e];
                z[12 + e] = zc3[e];
            }
        };
        // step n: z_nxt = bias(n) + W_n . xn^T; the epilogue of tile n - 1 (z_cur) in the gaps
        auto step = [&](f32x16& z_nxt, const f32x16& z_cur, int n, auto epi_tag) __attribute__((always_inline)) {
            constexpr bool EPI_ON = decltype(epi_tag)::value;
            z_bias_load(n);
            // (vmcnt 12: the previous step's pieces may stay in flight; its two stores, younger still, then count among the twelve -- a conservative wait)
            rs_step<KS, KS / NW>(
                st, frag_rd_i, lane16_i, [&]() __attribute__((always_inline)) { z_bias_set(z_nxt); },
                [&](auto j_tag, const bf16x8& fr) __attribute__((always_inline)) {
                    constexpr int J = decltype(j_tag)::value;
                    if (!(RS_ABL & 4)) z_nxt = mfma32f<FMT>(fr, xn[J], z_nxt);
                },
                [&](auto g_tag) __attribute__((always_inline)) {
                    if constexpr (EPI_ON) epilogue_gap(g_tag, z_cur, n - 1);
                });
        };
        // (ntiles is even: qkv 72 / fc1 96 tiles)
        step(Z0, Z0, 0, std::false_type{});
#pragma unroll 1
        for (int n = 1; n + 1 < ntiles; n += 2) {
            step(Z1, Z0, n, std::true_type{});
            step(Z0, Z1, n + 1, std::true_type{});
        }
        step(Z1, Z0, ntiles - 1, std::true_type{});
        mf_for(std::make_integer_sequence<int, KS>{}, [&](auto g_tag) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            epilogue_gap(g_tag, Z1, ntiles - 1);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ================================================================================================ gemm_cstat: proj / fc2 (+ residual)
// X[M][768] += A[M][K] . W^T + bias, in place.  An item = 128 rows x ONE HALF of the 768 output features (12 accumulator blocks = 192
// registers: MFMA accumulators live in the 256 AGPRs); a step = two 32-wide k-tiles of that half of W (48 fragments); items 2 r and 2 r + 1
// are the two halves of row block r and run on neighbouring workgroups at the same time (the second read of the A rows is an L2 hit); the
// grid is even, so a workgroup keeps its half -- and its weight stream -- for the whole launch.
template <int FMT>
__global__ __launch_bounds__(rs::THREADS, 1) void gemm_cstat_kernel(GemmParams p) {
    using namespace rs;
    constexpr int NB = 12, NG = 48, SLOTB = NG * 1024, NH = NB * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, nk2 = p.K / 64;
    const int nitems = 2 * ((M + BM - 1) / BM);
    if ((int)blockIdx.x >= nitems) return;
    const int nh = blockIdx.x & 1;
    float* const sBias = reinterpret_cast<float*>(smem + RING * SLOTB);
    for (int i = tid; i < NH / 4; i += THREADS) reinterpret_cast<f32x4*>(sBias)[i] = reinterpret_cast<const f32x4*>(p.bias + nh * NH)[i];
    RsStream st;
    st.wp = reinterpret_cast<uint64_t>(p.W) + (uint64_t)nh * nk2 * SLOTB;
    st.lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    st.piece0 = (uint32_t)wave * (NG / NW) * 1024;
    st.sn = 0; st.ipos = 0; st.rpos = 0; st.nslots = nk2;
    rs_prologue<NG>(st, (uint32_t)(tid & 63) * 16);
    __syncthreads();

    f32x16 o[NB];
    bf16x8 bq[3][4];      // [buffer][2 (k-tile of the pair) + s2]
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        uint32_t zero = 0;
        asm volatile("" : "+v"(zero));
        const uint32_t lane_i = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
        const uint32_t lane16_i = lane_i * 16, lh_i = lane_i >> 5;
        const uint32_t frag_rd_i = st.lds_base + lane16_i;
        auto lane_row = [&](int it) __attribute__((always_inline)) -> int {      // this lane's row of item `it` (row block it >> 1), clamped
            uint32_t z = 0;
            asm volatile("" : "+v"(z));
            const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
            const int r = (it >> 1) * BM + wave * 32 + (int)(l & 31);
            return r < M ? r : M - 1;
        };
        auto lane_half8 = [&]() __attribute__((always_inline)) -> uint32_t {
            uint32_t z = 0;
            asm volatile("" : "+v"(z));
            return (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z)) >> 5) * 8;
        };
        // residual block db: features 32 db + 16 s2 + 8 lh + (0 .. 7) = registers 8 s2 + (0 .. 7): four 16-byte loads / stores
        auto load_x = [&](const float* xr, auto db_tag) __attribute__((always_inline)) {
            constexpr int DB = decltype(db_tag)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 a = {1.f, 2.f, 3.f, (float)q};
                if (!(RS_ABL & 32)) a = *reinterpret_cast<const f32x4*>(xr + DB * 32 + (q >> 1) * 16 + (q & 1) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[DB][q * 4 + e] = a[e];
            }
        };
        // B fragment i = 2 kk + s2 of k-tile pair k2: 16 bytes of the A row at k = 64 k2 + 32 kk + 16 s2 + 8 lh
        auto load_b = [&](const bf16_t* ar, int k2, auto buf_tag, auto i_tag) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_tag)::value, I = decltype(i_tag)::value;
            uint4 u = {0x3c003c00u, 0x3c003c00u, (uint32_t)k2, 0u};
            if (!(RS_ABL & 32)) u = *reinterpret_cast<const uint4*>(ar + k2 * 64 + I * 16);
            bq[BUF][I] = __builtin_bit_cast(bf16x8, u);
        };
        auto load_b4 = [&](const bf16_t* ar, int k2, auto buf_tag) __attribute__((always_inline)) {
            mf_for(std::make_integer_sequence<int, 4>{}, [&](auto i_tag) __attribute__((always_inline)) { load_b(ar, k2, buf_tag, i_tag); });
        };
        const bool has_next = item + (int)gridDim.x < nitems;
        const bf16_t* const ar = p.A + (long)lane_row(item) * p.lda + lane_half8();
        if (item == (int)blockIdx.x) {
            const float* xr = p.out_f32 + (long)lane_row(item) * p.ldo_f32 + nh * NH + lane_half8();
            mf_for(std::make_integer_sequence<int, NB>{}, [&](auto db_tag) __attribute__((always_inline)) { load_x(xr, db_tag); });
        }
        // (the first two fragment pairs at the item's start, in flight beside the bias pass below: prefetched from the previous item's last steps
        //  they are loop-carried values, which hipcc parks in scratch across the item boundary)
        load_b4(ar, 0, std::integral_constant<int, 0>{});
        load_b4(ar, 1, std::integral_constant<int, 1>{});
        // the bias joins the residual rows (an LDS read inside a step would be waited for with the compiler's lgkmcnt count, which knows nothing of
        // the fragment reads in flight: here nothing is)
#pragma unroll
        for (int db = 0; db < NB; ++db) {
            __builtin_amdgcn_sched_barrier(0);
            const float* bp = sBias + db * 32 + lh_i * 8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(bp + (q >> 1) * 16 + (q & 1) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[db][q * 4 + e] += c[e];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float* xrow = nullptr;
        const float* nxr = nullptr;
        // step k2 (buffer BUF = k2 % 3): o[db] += W(kk, db, s2) . b(kk, s2) for the pair's two k-tiles; the fragments of pair k2 + 2 are
        // loaded in gaps 2 .. 8.  LAST: block db is final after gap 25 + 2 db: stored two gaps later, the next item's rows loaded in its place
        auto step = [&](int k2, auto buf_tag, auto vm_tag, auto last_tag) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;
            if constexpr (LAST) {
                const uint32_t h8 = lane_half8();
                xrow = p.out_f32 + (long)lane_row(item) * p.ldo_f32 + nh * NH + h8;
                nxr = p.out_f32 + (long)lane_row(item + (int)gridDim.x) * p.ldo_f32 + nh * NH + h8;
            }

            auto finish = [&](auto db_tag) __attribute__((always_inline)) {
                constexpr int DB = decltype(db_tag)::value;
                // (no row guard: a lane past the last row works on a copy of row M - 1 and stores the same bits to the same place)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = o[DB][q * 4 + e];
                    if (!(RS_ABL & 64)) *reinterpret_cast<f32x4*>(xrow + DB * 32 + (q >> 1) * 16 + (q & 1) * 4) = a;
                    else asm volatile("" ::"v"(a));
                }
                if (has_next) load_x(nxr, db_tag);
            };
            rs_step<NG, decltype(vm_tag)::value>(
                st, frag_rd_i, lane16_i, []() __attribute__((always_inline)) {},
                [&](auto j_tag, const bf16x8& fr) __attribute__((always_inline)) {
                    constexpr int J = decltype(j_tag)::value, KK = J / 24, JJ = J % 24;
                    if (!(RS_ABL & 4)) o[JJ >> 1] = mfma32f<FMT>(fr, bq[BUF][2 * KK + (JJ & 1)], o[JJ >> 1]);
                },
                [&](auto g_tag) __attribute__((always_inline)) {
                    constexpr int G = decltype(g_tag)::value;
                    if constexpr (!LAST && G >= 2 && G <= 8 && (G & 1) == 0) {
                        if (k2 + 2 < nk2) load_b(ar, k2 + 2, std::integral_constant<int, (BUF + 2) % 3>{}, std::integral_constant<int, (G - 2) / 2>{});
                    }
                    if constexpr (LAST && G >= 27 && (G & 1) == 1) finish(std::integral_constant<int, (G - 27) / 2>{});
                });
            if constexpr (LAST) {
                finish(std::integral_constant<int, NB - 1>{});      // (blocks 0 .. NB - 2: gaps 27 .. 47)
            }
        };
        using VM = std::integral_constant<int, NG / NW>;
        using VMB = std::integral_constant<int, 48>;      // an item's first two steps: >= 48 row loads were issued behind the pieces they wait for
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        using B2 = std::integral_constant<int, 2>;
        step(0, B0{}, VMB{}, std::false_type{});
        step(1, B1{}, VMB{}, std::false_type{});
        step(2, B2{}, VM{}, std::false_type{});
#pragma unroll 1
        for (int k2 = 3; k2 + 3 < nk2; k2 += 3) {
            step(k2, B0{}, VM{}, std::false_type{});
            step(k2 + 1, B1{}, VM{}, std::false_type{});
            step(k2 + 2, B2{}, VM{}, std::false_type{});
        }
        step(nk2 - 3, B0{}, VM{}, std::false_type{});
        step(nk2 - 2, B1{}, VM{}, std::false_type{});
        step(nk2 - 1, B2{}, VM{}, std::true_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------ host
bool gemm_rs_supported(const GemmParams& p) {
    if (p.planes != 1 || p.bias == nullptr || p.M < 1 || (p.ln_x == nullptr && p.lda % 8 != 0) || (p.ln_x != nullptr && p.epi == EPI_RESID) || p.resid != nullptr || p.aux_out != nullptr || p.ksplit > 1) return false;
    if (p.epi == EPI_RESID) return p.N == 768 && p.K % 192 == 0 && p.K >= 576 && p.ldo_f32 == p.N && p.out_f32 != nullptr;
    if (p.epi == EPI_GELU) return p.K == 768 && p.N % 64 == 0 && p.N >= 128 && p.ldo % 8 == 0 && p.N * 4 <= 16 * 1024 && p.out_bf16 != nullptr;
    if (p.epi == EPI_QKV) return p.K == 768 && p.dmodel == 768 && p.N == 3 * p.dmodel && p.heads * 64 == p.dmodel && p.q && p.k && p.v;
    return false;
}

template <int FMT>
static int launch_gemm_rs_fmt(const GemmParams& p, hipStream_t s) {
    using namespace rs;
    constexpr int LDS_B = RING * 48 * 1024 + 16 * 1024, LDS_C = RING * 48 * 1024 + 4 * 1024;
    static PerDeviceOnce once;
    if (once.first()) {
        auto opt_in = [](const void* fn, int bytes) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
        DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&gemm_bstat_kernel<FMT, EPI_GELU, 48, false>), LDS_B));
        DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&gemm_bstat_kernel<FMT, EPI_QKV, 48, false>), LDS_B));
        DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&gemm_bstat_kernel<FMT, EPI_GELU, 48, true>), LDS_B));
        DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&gemm_bstat_kernel<FMT, EPI_QKV, 48, true>), LDS_B));
        DSEG_CHECK_HIP(opt_in(reinterpret_cast<const void*>(&gemm_cstat_kernel<FMT>), LDS_C));
        once.mark();
    }
    const int ncu = device_cu_count();
    if (ncu <= 0) return -2;
    const int nitems = (p.M + BM - 1) / BM;
    const int rounds = (nitems + ncu - 1) / ncu;
    const int grid = (nitems + rounds - 1) / rounds;
    const dim3 b(THREADS);
    if (p.epi == EPI_RESID) {      // two items per row block (the halves of N); an even grid keeps every workgroup on one half
        const int n2 = 2 * nitems, r2 = (n2 + ncu - 1) / ncu;
        int g2 = (n2 + r2 - 1) / r2;
        g2 += g2 & 1;
        if (g2 > ncu) g2 = ncu & ~1;
        hipLaunchKernelGGL((gemm_cstat_kernel<FMT>), dim3(g2), b, LDS_C, s, p);
        DSEG_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const dim3 g(grid);
    if (p.ln_x) {
        if (p.epi == EPI_GELU) hipLaunchKernelGGL((gemm_bstat_kernel<FMT, EPI_GELU, 48, true>), g, b, LDS_B, s, p);
        else hipLaunchKernelGGL((gemm_bstat_kernel<FMT, EPI_QKV, 48, true>), g, b, LDS_B, s, p);
    } else if (p.epi == EPI_GELU) hipLaunchKernelGGL((gemm_bstat_kernel<FMT, EPI_GELU, 48, false>), g, b, LDS_B, s, p);
    else hipLaunchKernelGGL((gemm_bstat_kernel<FMT, EPI_QKV, 48, false>), g, b, LDS_B, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// p.W: the weight re-packed by launch_pack_rs (kind 0 for GELU / QKV, kind 1 for RESID)
int launch_gemm_rs(const GemmParams& p, hipStream_t s) {
    if (!gemm_rs_supported(p)) {
        dinoseg_set_error("gemm_rs: unsupported shape / epilogue (M=%d N=%d K=%d epi=%d planes=%d)", p.M, p.N, p.K, p.epi, p.planes);
        return -1;
    }
    return p.fmt == FMT_FP16 ? launch_gemm_rs_fmt<FMT_FP16>(p, s) : launch_gemm_rs_fmt<FMT_BF16>(p, s);
}

}  // namespace dseg
