// Backward / optimiser helper kernels of the DINOSeg fine-tune step (gfx950).
// Replaces autograd's nll_loss_backward / _log_softmax_backward_data / native_layer_norm_backward /
// upsample_bicubic2d_backward / sum (bias grads) and the Adam(W) step (SURVEY.md §2.1 "bwd" row;
// reference call sites pl_torch_modules.py:258-268).  The GEMM-shaped parts of the backward (dgrad / wgrad) run on
// gemm.hip; weight gradients use operands transposed by transpose_planes_kernel so that the contraction over the
// batch rows is K-contiguous for the same NT MFMA kernel (split-K + fp32 atomics).
#include "common.h"
#include "gemm_ln_common.h"
#include "kernels.h"

#define DSEG_TRY_RC(expr)        \
    do {                         \
        int _rc = (expr);        \
        if (_rc != 0) return _rc; \
    } while (0)

namespace dseg {

// ------------------------------------------------------------------------------------------------
// [M][C] (fp32, or bf16 hi/lo planes) -> transposed planes T[pl][c][m] with c < c_pad rows and m < m_pad columns
// (zero filled beyond C / M), optionally also the row-major planes N[pl][m][c] and the column sums (bias gradients).
// drop_cls: logical row j = b*(ntok-1)+t-1 is read from physical row b*ntok+t (t >= 1) of the source.
__global__ __launch_bounds__(256) void transpose_planes_kernel(const float* __restrict__ src_f32,
                                                               const bf16_t* __restrict__ src_pl, long src_plane, int lds_,
                                                               int M, int C, bf16_t* __restrict__ T, long t_plane, int m_pad,
                                                               bf16_t* __restrict__ Nout, long n_plane, int ldn,
                                                               float* __restrict__ colsum, int planes, int drop_cls, int ntok, float* __restrict__ det) {
    __shared__ uint16_t th[64][66], tl[64][66];
    __shared__ float cs[4][64];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float csum = 0.f;
    const int c = c0 + tx;
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
        const int ml = ty * 16 + rr, m = m0 + ml;
        uint16_t hi = 0, lo = 0;
        if (m < M && c < C) {
            long prow = m;
            if (drop_cls) {
                const int b = m / (ntok - 1), t = m - b * (ntok - 1);
                prow = (long)b * ntok + t + 1;
            }
            float v;
            if (src_f32) {
                v = src_f32[prow * lds_ + c];
                const uint32_t h2 = pack_bf16x2(v, 0.f);
                hi = (uint16_t)(h2 & 0xFFFF);
                lo = (uint16_t)(pack_bf16x2(v - bf16_lo_to_f32(h2), 0.f) & 0xFFFF);
            } else {
                hi = src_pl[prow * lds_ + c];
                v = bf16_to_f32(hi);
                if (planes == 2) {
                    lo = src_pl[src_plane + prow * lds_ + c];
                    v += bf16_to_f32(lo);
                }
            }
            csum += v;
            if (Nout) {
                Nout[(long)m * ldn + c] = hi;
                if (planes == 2) Nout[n_plane + (long)m * ldn + c] = lo;
            }
        }
        th[ml][tx] = hi;
        tl[ml][tx] = lo;
    }
    if (colsum) cs[ty][tx] = csum;
    __syncthreads();
    if (colsum && ty == 0 && c < C) {
        const float s = cs[0][tx] + cs[1][tx] + cs[2][tx] + cs[3][tx];
        if (det) det[(long)blockIdx.x * C + c] = s;       // deterministic mode: one partial per 64-row block, summed in order afterwards
        else if (s != 0.f) atomicAdd(colsum + c, s);
    }
    if (T) {
#pragma unroll 4
        for (int cc = 0; cc < 16; ++cc) {
            const int cl = ty * 16 + cc;
            const long o = (long)(c0 + cl) * m_pad + m0 + tx;
            T[o] = th[tx][cl];
            if (planes == 2) T[t_plane + o] = tl[tx][cl];
        }
    }
}

int launch_transpose_planes(const float* src_f32, const bf16_t* src_pl, long src_plane, int ld_src, int M, int C,
                            bf16_t* T, long t_plane, int c_pad, int m_pad, bf16_t* Nout, long n_plane, int ldn,
                            float* colsum, int planes, int drop_cls, int ntok, hipStream_t s, int det_region) {
    if (m_pad % 64 != 0 || c_pad % 64 != 0 || m_pad < M || c_pad < C) {
        dinoseg_set_error("transpose_planes: bad padding (M=%d C=%d m_pad=%d c_pad=%d)", M, C, m_pad, c_pad);
        return -1;
    }
    float* det = nullptr;
    if (colsum && det_scratch().ptr) {
        // (a launch on the weight-gradient side stream must not share the caller's stream's region: both streams run at once)
        const size_t cap = det_region ? det_scratch().tn_floats : det_scratch().floats;
        if ((size_t)(m_pad / 64) * C > cap) {
            dinoseg_set_error("transpose_planes: deterministic scratch too small (%d x %d partial sums)", m_pad / 64, C);
            return -1;
        }
        det = det_region ? det_scratch().tn[1] : det_scratch().ptr;
    }
    hipLaunchKernelGGL(transpose_planes_kernel, dim3(m_pad / 64, c_pad / 64), dim3(256), 0, s, src_f32, src_pl, src_plane,
                       ld_src, M, C, T, t_plane, m_pad, Nout, n_plane, ldn, colsum, planes, drop_cls, ntok, det);
    DSEG_CHECK_HIP(hipGetLastError());
    if (det) return launch_det_finalize(det, m_pad / 64, C, C, colsum, s);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// F.nll_loss (pl_torch_modules.py:265; default reduction 'mean', ignore_index = -100) and the backward of log_softmax.
//   nll_reduce_kernel   : acc[0] += sum over valid rows of -logp[m][y_m];  acc[1] += number of valid rows;
//                         flags[0] |= 1 when a label is neither in [0, C) nor ignore_index (F.nll_loss raises there;
//                         the row is then treated as ignored and the caller reads the flag: dinoseg_train_status)
//   logsoftmax_bwd_kernel: dz[m][c] = dl[m][c] - exp(logp[m][c]) * sum_c dl[m][c], hi/lo planes [2][M][ldz] (zero padded
//                         columns), where dl is either the caller's d loss / d logp (autograd path, dinoseg_backward) or
//                         nll_loss's own -[c == y_m] / n_valid (fused path; also finalises loss = acc[0] / acc[1]).
//                         Both paths go through the same arithmetic, so F.nll_loss(model(x), y).backward() and
//                         training_step() produce the same d logits bit for bit.
constexpr int IGNORE_INDEX = -100;

__global__ __launch_bounds__(256) void nll_reduce_kernel(const float* __restrict__ logp, const int64_t* __restrict__ labels, int M,
                                                         int C, float* __restrict__ acc, int* __restrict__ flags, float* __restrict__ det) {
    __shared__ float wsum[2][4];
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f, n = 0.f;
    if (m < M) {
        const long y = labels[m];
        if (y >= 0 && y < C) {
            l = -logp[(long)m * C + y];
            n = 1.f;
        } else if (y != IGNORE_INDEX) {
            atomicOr(flags, 1);
        }
    }
    l = wave_sum(l);
    n = wave_sum(n);
    if (det) {          // deterministic mode: the block's four wave sums in wave order, one partial pair per block
        if ((threadIdx.x & 63) == 0) {
            wsum[0][threadIdx.x >> 6] = l;
            wsum[1][threadIdx.x >> 6] = n;
        }
        __syncthreads();
        if (threadIdx.x < 2) det[blockIdx.x * 2 + threadIdx.x] = ((wsum[threadIdx.x][0] + wsum[threadIdx.x][1]) + wsum[threadIdx.x][2]) + wsum[threadIdx.x][3];
        return;
    }
    if ((threadIdx.x & 63) == 0 && n != 0.f) {
        atomicAdd(acc, l);
        atomicAdd(acc + 1, n);
    }
}

// deterministic mode: dst[c] += sum over the partials of part[p * ld + c] in a FIXED order: thread (column c, group g of 8) adds the
// partials g, g + 8, g + 16, ... in ascending order (eight independent chains: the loads pipeline), then the eight group sums are
// added in group order.  The order depends on nparts only -- never on timing.  (First version: one thread per column walking all
// partials -- 1024 dependent strided loads: 0.2 ms per call, the fine-tune step 2.1x slower; this one: ~10 us.)
__global__ __launch_bounds__(256) void det_finalize_kernel(const float* __restrict__ part, int nparts, int width, int ld,
                                                           float* __restrict__ dst) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < width) {
        int q = g;
        for (; q + 24 < nparts; q += 32) {      // four loads in flight per thread
            const float a0 = part[(long)q * ld + c], a1 = part[(long)(q + 8) * ld + c];
            const float a2 = part[(long)(q + 16) * ld + c], a3 = part[(long)(q + 24) * ld + c];
            s = (((s + a0) + a1) + a2) + a3;
        }
        for (; q < nparts; q += 8) s += part[(long)q * ld + c];
    }
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < width) {
        float t = red[0][cl];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][cl];
        dst[c] += t;
    }
}

DetScratch& det_scratch() {
    static DetScratch d = {nullptr, 0, {nullptr, nullptr}, 0};
    return d;
}

int launch_det_finalize(const float* part, int nparts, int width, int ld, float* dst, hipStream_t s) {
    hipLaunchKernelGGL(det_finalize_kernel, dim3((width + 31) / 32), dim3(256), 0, s, part, nparts, width, ld, dst);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void logsoftmax_bwd_kernel(const float* __restrict__ logp, const float* __restrict__ dlogp,
                                                             const int64_t* __restrict__ labels, const float* __restrict__ acc,
                                                             int M, int C, float* __restrict__ loss, bf16_t* __restrict__ dz,
                                                             long dz_plane, int ldz) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (labels && m == 0) *loss = acc[1] > 0.f ? acc[0] / acc[1] : __builtin_nanf("");   // torch: mean over zero rows is nan
    if (m >= M) return;
    float rowsum = 0.f, own = 0.f;
    int y = -1;
    if (labels) {
        const long yy = labels[m];
        if (yy >= 0 && yy < C) {
            y = (int)yy;
            own = -1.0f / acc[1];
            rowsum = own;
        }
    } else {
        for (int c = 0; c < C; ++c) rowsum += dlogp[(long)m * C + c];
    }
    for (int c = 0; c < ldz; ++c) {
        float g = 0.f;
        if (c < C) {
            const float dl = labels ? (c == y ? own : 0.f) : dlogp[(long)m * C + c];
            g = dl - expf(logp[(long)m * C + c]) * rowsum;
        }
        const uint32_t hi = pack_bf16x2(g, 0.f);
        dz[(long)m * ldz + c] = (bf16_t)(hi & 0xFFFF);
        dz[dz_plane + (long)m * ldz + c] = (bf16_t)(pack_bf16x2(g - bf16_lo_to_f32(hi), 0.f) & 0xFFFF);
    }
}

int launch_nll_loss_grad(const float* logp, const int64_t* labels, const float* dlogp, int M, int C, float* acc, int* flags,
                         float* loss, bf16_t* dz, long dz_plane, int ldz, hipStream_t s) {
    if (labels) {
        DSEG_CHECK_HIP(hipMemsetAsync(acc, 0, 2 * sizeof(float), s));
        float* det = det_scratch().ptr;
        const int nb = (M + 255) / 256;
        if (det && (size_t)nb * 2 > det_scratch().floats) {      // (one policy for every launcher: never a silent fall-back to atomics)
            dinoseg_set_error("nll_loss: deterministic scratch too small (%d blocks)", nb);
            return -1;
        }
        hipLaunchKernelGGL(nll_reduce_kernel, dim3(nb), dim3(256), 0, s, logp, labels, M, C, acc, flags, det);
        if (det) DSEG_TRY_RC(launch_det_finalize(det, nb, 2, 2, acc, s));
    }
    hipLaunchKernelGGL(logsoftmax_bwd_kernel, dim3((M + 255) / 256), dim3(256), 0, s, logp, dlogp, labels, acc, M, C, loss, dz, dz_plane,
                       ldz);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward (native_layer_norm_backward).  One wavefront per row; x is the saved LN input.
//   g = dy * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat));  dgamma += dy * xhat;  dbeta += dy
// dx is added to (accumulate = 1) or written to (0) the residual-stream gradient.  drop_cls: dy has no CLS rows.
// By-products for the layer the gradient flows into next (fc2 / attn.proj of the backward walk): the final dx rows as bf16
// hi[/lo] planes dxp[planes][M][D] -- the operand of its two gradient GEMMs -- and their column sums colsum[D] += sum_m dx[m]
// (its bias gradient).  A separate pack + column-sum pass over dx cost 37 us per layer (transpose_planes_kernel).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, float eps, int M,
                                                            float* __restrict__ dx, int accumulate,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int drop_cls, int ntok, bf16_t* __restrict__ dxp, long dxp_plane,
                                                            int planes, float* __restrict__ colsum, float* __restrict__ det) {
    constexpr int D = 128 * NV;
    __shared__ float red[3][4][D];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wid = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
    f32x2 g[NV], dg[NV], db[NV], dn[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g[i] = *reinterpret_cast<const f32x2*>(gamma + i * 128 + lane * 2);
        dg[i] = f32x2{0.f, 0.f};
        db[i] = f32x2{0.f, 0.f};
        dn[i] = f32x2{0.f, 0.f};
    }
    auto emit = [&](int m, int i, f32x2 r) {      // by-products of one finished pair of dx columns
        dn[i] += r;
        if (dxp) {
            uint32_t hi, lo;
            split_bf16x2(r[0], r[1], hi, lo);
            bf16_t* o = dxp + (long)m * D + i * 128 + lane * 2;
            *reinterpret_cast<uint32_t*>(o) = hi;
            if (planes == 2) *reinterpret_cast<uint32_t*>(o + dxp_plane) = lo;
        }
    };
    for (int m = wid; m < M; m += nw) {
        long drow = m;
        bool has_dy = true;
        if (drop_cls) {
            const int b = m / ntok, t = m - b * ntok;
            has_dy = t != 0;
            drow = (long)b * (ntok - 1) + t - 1;
        }
        float* dxr = dx + (long)m * D;
        if (!has_dy) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f32x2* dst = reinterpret_cast<f32x2*>(dxr + i * 128 + lane * 2);
                const f32x2 r = accumulate ? *dst : f32x2{0.f, 0.f};
                if (!accumulate) *dst = r;
                if (dxp || colsum) emit(m, i, r);
            }
            continue;
        }
        const float* xr = x + (long)m * D;
        const float* dyr = dy + drow * D;
        f32x2 xv[NV], dv[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            xv[i] = *reinterpret_cast<const f32x2*>(xr + i * 128 + lane * 2);
            dv[i] = *reinterpret_cast<const f32x2*>(dyr + i * 128 + lane * 2);
            s += xv[i][0] + xv[i][1];
        }
        const float mean = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            xv[i][0] -= mean;
            xv[i][1] -= mean;
            q += xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1];
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float xh = xv[i][e] * rstd;
                const float ge = dv[i][e] * g[i][e];
                dg[i][e] += dv[i][e] * xh;
                db[i][e] += dv[i][e];
                sg += ge;
                sgx += ge * xh;
                xv[i][e] = xh;      // keep xhat
                dv[i][e] = ge;      // keep g
            }
        }
        sg = wave_sum(sg) * (1.0f / D);
        sgx = wave_sum(sgx) * (1.0f / D);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x2 r;
            r[0] = rstd * (dv[i][0] - sg - xv[i][0] * sgx);
            r[1] = rstd * (dv[i][1] - sg - xv[i][1] * sgx);
            f32x2* dst = reinterpret_cast<f32x2*>(dxr + i * 128 + lane * 2);
            if (accumulate) r += *dst;
            *dst = r;
            if (dxp || colsum) emit(m, i, r);
        }
    }
    // block-level reduction of the parameter gradients, then one atomic per column per block
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        red[0][wv][i * 128 + lane * 2] = dg[i][0];
        red[0][wv][i * 128 + lane * 2 + 1] = dg[i][1];
        red[1][wv][i * 128 + lane * 2] = db[i][0];
        red[1][wv][i * 128 + lane * 2 + 1] = db[i][1];
        red[2][wv][i * 128 + lane * 2] = dn[i][0];
        red[2][wv][i * 128 + lane * 2 + 1] = dn[i][1];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const float a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        const float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        if (det) {      // deterministic mode: per-block partials [block][3][D], summed in block order by det_finalize_kernel
            float* dp = det + (long)blockIdx.x * 3 * D;
            dp[c] = a;
            dp[D + c] = b;
            dp[2 * D + c] = red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c];
            continue;
        }
        if (dgamma) atomicAdd(dgamma + c, a);
        if (dbeta) atomicAdd(dbeta + c, b);
        if (colsum) {
            const float n = red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c];
            if (n != 0.f) atomicAdd(colsum + c, n);
        }
    }
}

// The same with sixteen lanes per row (four rows per wavefront): 16 bytes per lane and access instead of 8, the four row statistics
// by DPP adds inside the lane group instead of wave-wide ds_bpermute butterflies.  D = 64 * NC <= 512.  A row without dy (the CLS
// row of the final norm) runs with dy = 0: the same result as skipping it.  (3 blocks, 8 frames @480: 69 -> ~47 us per launch, 1530 -> 1577 frames/s of the fine-tune step.)
template <int NC>
__global__ __launch_bounds__(256) void layernorm_bwd16_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ gamma, float eps, int M,
                                                              float* __restrict__ dx, int accumulate,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int drop_cls, int ntok, bf16_t* __restrict__ dxp, long dxp_plane,
                                                              int planes, float* __restrict__ colsum, float* __restrict__ det) {
    constexpr int D = 64 * NC;
    __shared__ float red[3][4][D];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane & 15, rg = lane >> 4;
    const int wid = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
    f32x4 g[NC], dg[NC], db[NC], dn[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        g[c] = *reinterpret_cast<const f32x4*>(gamma + c * 64 + sub * 4);
        dg[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[c] = dg[c];
        dn[c] = dg[c];
    }
    const float invD = 1.0f / D;
    for (int m0 = wid * 4; m0 < M; m0 += nw * 4) {
        const int m = m0 + rg;
        const bool valid = m < M;
        const int mc = valid ? m : M - 1;
        long drow = mc;
        bool has_dy = valid;
        if (drop_cls) {
            const int b = mc / ntok, t = mc - b * ntok;
            has_dy = valid && t != 0;
            drow = (long)b * (ntok - 1) + (t > 0 ? t - 1 : 0);
        }
        const float* xr = x + (long)mc * D;
        const float* dyr = dy + drow * D;
        float* dxr = dx + (long)mc * D;
        f32x4 xv[NC], dv[NC], old[NC];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            xv[c] = *reinterpret_cast<const f32x4*>(xr + c * 64 + sub * 4);
            dv[c] = *reinterpret_cast<const f32x4*>(dyr + c * 64 + sub * 4);
            if (accumulate) old[c] = *reinterpret_cast<const f32x4*>(dxr + c * 64 + sub * 4);
            if (!has_dy) dv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            s += (xv[c][0] + xv[c][1]) + (xv[c][2] + xv[c][3]);
        }
        const float mean = aln::row16_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[c][e] -= mean;
                q += xv[c][e] * xv[c][e];
            }
        const float rstd = 1.0f / sqrtf(aln::row16_sum(q) * invD + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = xv[c][e] * rstd;
                const float ge = dv[c][e] * g[c][e];
                dg[c][e] += dv[c][e] * xh;
                db[c][e] += dv[c][e];
                sg += ge;
                sgx += ge * xh;
                xv[c][e] = xh;
                dv[c][e] = ge;
            }
        sg = aln::row16_sum(sg) * invD;
        sgx = aln::row16_sum(sgx) * invD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = rstd * (dv[c][e] - sg - xv[c][e] * sgx);
            if (accumulate) r += old[c];
            if (valid) {
                *reinterpret_cast<f32x4*>(dxr + c * 64 + sub * 4) = r;
                dn[c] += r;
                if (dxp) {
                    uint2 hi, lo;
                    split_bf16x2(r[0], r[1], hi.x, lo.x);
                    split_bf16x2(r[2], r[3], hi.y, lo.y);
                    bf16_t* o = dxp + (long)m * D + c * 64 + sub * 4;
                    *reinterpret_cast<uint2*>(o) = hi;
                    if (planes == 2) *reinterpret_cast<uint2*>(o + dxp_plane) = lo;
                }
            }
        }
    }
    // the four row groups of the wave, then the four waves, then one atomic per column and block
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = dg[c][e], b = db[c][e], n = dn[c][e];
            a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
            n += __shfl_xor(n, 16); n += __shfl_xor(n, 32);
            if (rg == 0) {
                red[0][wv][c * 64 + sub * 4 + e] = a;
                red[1][wv][c * 64 + sub * 4 + e] = b;
                red[2][wv][c * 64 + sub * 4 + e] = n;
            }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const float a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        const float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        if (det) {      // deterministic mode: per-block partials [block][3][D], summed in block order by det_finalize_kernel
            float* dp = det + (long)blockIdx.x * 3 * D;
            dp[c] = a;
            dp[D + c] = b;
            dp[2 * D + c] = red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c];
            continue;
        }
        if (dgamma) atomicAdd(dgamma + c, a);
        if (dbeta) atomicAdd(dbeta + c, b);
        if (colsum) {
            const float n = red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c];
            if (n != 0.f) atomicAdd(colsum + c, n);
        }
    }
}

int launch_layernorm_bwd(const float* dy, const float* x, const float* gamma, float eps, int M, int D, float* dx,
                         int accumulate, float* dgamma, float* dbeta, int drop_cls, int ntok, hipStream_t s, bf16_t* dxp,
                         long dxp_plane, int planes, float* colsum) {
    if (D % 128 != 0 || D > 1024) {
        dinoseg_set_error("layernorm_bwd: D=%d must be a multiple of 128 and <= 1024", D);
        return -1;
    }
    float* det = det_scratch().ptr;
    if (det && (size_t)1024 * 3 * D > det_scratch().floats) {
        dinoseg_set_error("layernorm_bwd: deterministic scratch too small (D=%d)", D);
        return -1;
    }
    auto finalize = [&](int blocks) -> int {
        if (!det) return 0;
        if (dgamma) DSEG_TRY_RC(launch_det_finalize(det, blocks, D, 3 * D, dgamma, s));
        if (dbeta) DSEG_TRY_RC(launch_det_finalize(det + D, blocks, D, 3 * D, dbeta, s));
        if (colsum) DSEG_TRY_RC(launch_det_finalize(det + 2 * D, blocks, D, 3 * D, colsum, s));
        return 0;
    };
    if (D % 64 == 0 && D <= 512 && !(options().route_ab & 2)) {      // sixteen lanes per row (route_ab bit 1: the one-wave-per-row kernel, A/B)
        int grid16 = (M + 15) / 16;
        if (grid16 > 1024) grid16 = 1024;
#define DSEG_LNB16(NC)                                                                                                       \
    case NC:                                                                                                                 \
        hipLaunchKernelGGL((layernorm_bwd16_kernel<NC>), dim3(grid16), dim3(256), 0, s, dy, x, gamma, eps, M, dx, accumulate, \
                           dgamma, dbeta, drop_cls, ntok, dxp, dxp_plane, planes, colsum, det);                              \
        break;
        switch (D / 64) { DSEG_LNB16(2) DSEG_LNB16(4) DSEG_LNB16(6) DSEG_LNB16(8) }
#undef DSEG_LNB16
        DSEG_CHECK_HIP(hipGetLastError());
        return finalize(grid16);
    }
    int grid = (M + 3) / 4;
    if (grid > 1024) grid = 1024;
#define DSEG_LNB(NV)                                                                                                  \
    case NV:                                                                                                          \
        hipLaunchKernelGGL((layernorm_bwd_kernel<NV>), dim3(grid), dim3(256), 0, s, dy, x, gamma, eps, M, dx, accumulate, \
                           dgamma, dbeta, drop_cls, ntok, dxp, dxp_plane, planes, colsum, det);                       \
        break;
    switch (D / 128) { DSEG_LNB(1) DSEG_LNB(2) DSEG_LNB(3) DSEG_LNB(4) DSEG_LNB(5) DSEG_LNB(6) DSEG_LNB(7) DSEG_LNB(8) }
#undef DSEG_LNB
    DSEG_CHECK_HIP(hipGetLastError());
    return finalize(grid);
}

// ------------------------------------------------------------------------------------------------
// out[t][d] = sum_b X[b*ntok + t][d]   (gradient of the broadcast pos-embed add; row 0 is also d cls_token)
// Split-K weight gradients: out[r][c] += sum over slices of part[s][r][c] (c < cols).  The slices are written as plain
// 16-byte stores by the GEMM (EPI_PLAIN, GemmParams::split_stride); 21 slices of a 1536 x 384 gradient are 50 MB of
// streaming traffic (~20 us) where the same sum through fp32 atomics took most of the 183 us the GEMM launch cost.
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int ks, long stride, int rows, int ld_part,
                                     float* __restrict__ out, int ldo, int cols) {
    const long total = (long)rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
        const float* src = part + (long)r * ld_part + c;
        float a = 0.f;
        for (int sl = 0; sl < ks; ++sl) a += src[sl * stride];
        out[(long)r * ldo + c] += a;
    }
}
// cols, ld_part, ldo multiples of 4 (every transformer-block weight): 16 bytes per lane and slice, four slices in flight
__global__ void splitk_reduce4_kernel(const float* __restrict__ part, int ks, long stride, int rows, int ld_part,
                                      float* __restrict__ out, int ldo, int cols4) {
    const long total = (long)rows * cols4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols4), c = (int)(i - (long)r * cols4) * 4;
        const float* src = part + (long)r * ld_part + c;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        int sl = 0;
        for (; sl + 4 <= ks; sl += 4) {
            a0 += *reinterpret_cast<const f32x4*>(src + (long)sl * stride);
            a1 += *reinterpret_cast<const f32x4*>(src + (long)(sl + 1) * stride);
            a2 += *reinterpret_cast<const f32x4*>(src + (long)(sl + 2) * stride);
            a3 += *reinterpret_cast<const f32x4*>(src + (long)(sl + 3) * stride);
        }
        for (; sl < ks; ++sl) a0 += *reinterpret_cast<const f32x4*>(src + (long)sl * stride);
        f32x4* dst = reinterpret_cast<f32x4*>(out + (long)r * ldo + c);
        *dst = *dst + ((a0 + a1) + (a2 + a3));
    }
}

int launch_splitk_reduce(const float* part, int ks, long stride, int rows, int ld_part, float* out, int ldo, int cols, hipStream_t s) {
    const long total = (long)rows * cols;
    if (total < 1) return 0;
    if (cols % 4 == 0 && ld_part % 4 == 0 && ldo % 4 == 0 && stride % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        int grid = (int)((total / 4 + 255) / 256);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(grid), dim3(256), 0, s, part, ks, stride, rows, ld_part, out, ldo, cols / 4);
    } else {
        int grid = (int)((total + 255) / 256);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, part, ks, stride, rows, ld_part, out, ldo, cols);
    }
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ void batch_sum_rows_kernel(const float* __restrict__ X, int B, int ntok, int D, float* __restrict__ out) {
    const long total = (long)ntok * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += X[(long)b * total + i];
        out[i] = s;
    }
}

int launch_batch_sum_rows(const float* X, int B, int ntok, int D, float* out, hipStream_t s) {
    const long total = (long)ntok * D;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(batch_sum_rows_kernel, dim3(grid), dim3(256), 0, s, X, B, ntok, D, out);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// Transpose of pos_resample_kernel (upsample_bicubic2d_backward): every resampled-grid gradient goes to its 16 clamped taps of the
// stored [g*g+1, D] pos-embed gradient (accumulates into dpe).
__device__ __forceinline__ void cubic_w_bwd(float t, float w[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// Gather form (deterministic, no atomics): a stored element (yy, xx, d) is the sum of the resampled-grid gradients whose clamped
// 4 x 4 taps include it -- separable: sum_x Wx(x, xx) sum_y Wy(y, yy) dpos[y][x][d], Wy(y, yy) = the sum of the taps of y that clamp
// to yy; two passes through a [g][o][D] scratch (the one-pass gather read 144 values per element: 44 us).  (The scatter form -- 16 fp32 atomics per resampled element onto 785 x 384 addresses -- took 70 us at 28 -> 60.)
__device__ __forceinline__ float tap_weight(int y, int yy, int g, float scale) {
    const float sy = (y + 0.5f) * scale - 0.5f;
    const float fy = floorf(sy);
    float w[4];
    cubic_w_bwd(sy - fy, w);
    const int iy = (int)fy;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        int t = iy - 1 + a;
        t = t < 0 ? 0 : (t > g - 1 ? g - 1 : t);
        if (t == yy) acc += w[a];
    }
    return acc;
}
__device__ __forceinline__ void tap_range(int yy, int g, int o, float scale, int& lo, int& hi) {
    // outputs y whose taps can reach yy: floor(sy) in [yy - 2, yy + 1] (all smaller / larger ones too at the clamped borders)
    lo = (int)floorf(((float)(yy - 2) + 0.5f) / scale - 0.5f) - 1;
    hi = (int)ceilf(((float)(yy + 2) + 0.5f) / scale - 0.5f) + 1;
    if (yy == 0 || lo < 0) lo = 0;
    if (yy == g - 1 || hi > o - 1) hi = o - 1;
}
// separable: pass 1 sums the rows (tmp[yy][x][d] = sum_y Wy(y, yy) dpos[y][x][d]), pass 2 the columns into the stored gradient
__global__ void pos_resample_bwd_rows_kernel(const float* __restrict__ dpos, int g, int D, int o, float scale, float* __restrict__ tmp) {
    const long total = (long)g * o * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const long t = i / D;
        const int x = (int)(t % o), yy = (int)(t / o);
        int ylo, yhi;
        tap_range(yy, g, o, scale, ylo, yhi);
        float acc = 0.f;
        for (int y = ylo; y <= yhi; ++y) acc = fmaf(tap_weight(y, yy, g, scale), dpos[(1 + (long)y * o + x) * D + d], acc);
        tmp[i] = acc;
    }
}
__global__ void pos_resample_bwd_cols_kernel(const float* __restrict__ dpos, const float* __restrict__ tmp, int g, int D, int o,
                                             float scale, float* __restrict__ dpe) {
    const long total = ((long)g * g + 1) * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const long tokn = i / D;
        if (tokn == 0) {
            dpe[i] += dpos[i];      // class pos: same index
            continue;
        }
        const int yy = (int)((tokn - 1) / g), xx = (int)((tokn - 1) % g);
        int xlo, xhi;
        tap_range(xx, g, o, scale, xlo, xhi);
        float acc = 0.f;
        for (int x = xlo; x <= xhi; ++x) acc = fmaf(tap_weight(x, xx, g, scale), tmp[((long)yy * o + x) * D + d], acc);
        dpe[i] += acc;
    }
}
__global__ void pos_identity_bwd_kernel(const float* __restrict__ dpos, long total, float* __restrict__ dpe) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) dpe[i] += dpos[i];
}

// scratch: g * o * D floats (the row pass)
int launch_pos_resample_bwd(const float* dpos, int g, int D, int o, float* dpe, float* scratch, hipStream_t s) {
    const float scale = (float)(1.0 / (((double)o + 0.1) / (double)g));
    if (o == g) {      // identity grid
        const long total = ((long)g * g + 1) * D;
        hipLaunchKernelGGL(pos_identity_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dpos, total, dpe);
        DSEG_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const long t1 = (long)g * o * D, t2 = ((long)g * g + 1) * D;
    int g1 = (int)((t1 + 255) / 256), g2 = (int)((t2 + 255) / 256);
    if (g1 > 8192) g1 = 8192;
    if (g2 > 8192) g2 = 8192;
    hipLaunchKernelGGL(pos_resample_bwd_rows_kernel, dim3(g1), dim3(256), 0, s, dpos, g, D, o, scale, scratch);
    hipLaunchKernelGGL(pos_resample_bwd_cols_kernel, dim3(g2), dim3(256), 0, s, dpos, scratch, g, D, o, scale, dpe);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// delta[pair][q] = sum_d dO[row(q)][head*64+d] * O[row(q)][head*64+d]   (flash-attention backward row term)
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
                                                         long plane, int planes, int B, int heads, int ntok,
                                                         float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long total = (long)B * ntok * heads;      // one wavefront per (row, head): 64 lanes = 64 d
    if (wid >= total) return;
    const long row = wid / heads;
    const int head = (int)(wid - row * heads);
    const long off = row * (heads * 64) + head * 64 + lane;
    float a = bf16_to_f32(dO[off]), b = bf16_to_f32(O[off]);
    if (planes == 2) {
        a += bf16_to_f32(dO[plane + off]);
        b += bf16_to_f32(O[plane + off]);
    }
    const float s = wave_sum(a * b);
    if (lane == 0) {
        const long bidx = row / ntok, q = row - bidx * ntok;
        delta[(bidx * heads + head) * ntok + q] = s;
    }
}

int launch_attn_delta(const bf16_t* dO, const bf16_t* O, long plane, int planes, int B, int heads, int ntok, float* delta,
                      hipStream_t s) {
    const long total = (long)B * ntok * heads;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, dO, O, plane, planes, B, heads,
                       ntok, delta);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Fused Adam / AdamW step on one tensor (torch.optim semantics; pl_torch_modules.py:258-259).
//   decoupled = 1 (AdamW): p *= 1 - lr*wd;   decoupled = 0 (Adam): g += wd*p
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr, float b1, float b2, float eps, float wd, int decoupled, float bc1, float bc2s,
                            float gscale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float pi = p[i], gi = g[i] * gscale;
        if (decoupled) pi *= 1.0f - lr * wd;
        else if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

int launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd,
                int decoupled, int step, float gscale, hipStream_t s) {
    if (n <= 0) return 0;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    const float bc1 = 1.0f - powf(b1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(b2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, decoupled, bc1, bc2s, gscale);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- multi-tensor forms: one launch for up to 64 parameters (48 trainable tensors at L = 3 were 48 optimizer launches
// and 48 memset nodes per step, ~5 % of the fine-tune step in launch gaps) ----
struct MultiAdamTable {
    float* p[MULTI_MAX];
    const float* g[MULTI_MAX];
    float* m[MULTI_MAX];
    float* v[MULTI_MAX];
    int boff[MULTI_MAX + 1];        // first block of tensor t; boff[count] = grid size
    long n[MULTI_MAX];
    int count;
};
struct MultiZeroTable {
    float* p[MULTI_MAX];
    int boff[MULTI_MAX + 1];
    long n[MULTI_MAX];
    int count;
};
constexpr int MULTI_CHUNK = 256 * 16;      // elements per block

__device__ __forceinline__ int multi_find(const int* boff, int count, int b) {
    int t = 0;
    while (t + 1 < count && boff[t + 1] <= b) ++t;      // count <= 64: a wave-uniform scan of kernel arguments
    return t;
}

__global__ __launch_bounds__(256) void multi_adam_kernel(MultiAdamTable T, float lr, float b1, float b2, float eps, float wd,
                                                         int decoupled, float bc1, float bc2s, float gscale) {
    const int t = multi_find(T.boff, T.count, blockIdx.x);
    const long base = (long)(blockIdx.x - T.boff[t]) * MULTI_CHUNK;
    float* p = T.p[t];
    const float* g = T.g[t];
    float* m = T.m[t];
    float* v = T.v[t];
    const long n = T.n[t];
    for (long i = base + threadIdx.x; i < base + MULTI_CHUNK && i < n; i += 256) {
        float pi = p[i], gi = g[i] * gscale;
        if (decoupled) pi *= 1.0f - lr * wd;
        else if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

__global__ __launch_bounds__(256) void multi_zero_kernel(MultiZeroTable T) {
    const int t = multi_find(T.boff, T.count, blockIdx.x);
    const long base = (long)(blockIdx.x - T.boff[t]) * MULTI_CHUNK;
    float* p = T.p[t];
    const long n = T.n[t];
    for (long i = base + threadIdx.x; i < base + MULTI_CHUNK && i < n; i += 256) p[i] = 0.f;
}

int launch_multi_adam(int count, float* const* p, const float* const* g, float* const* m, float* const* v, const long* n, float lr,
                      float b1, float b2, float eps, float wd, int decoupled, int step, float gscale, hipStream_t s) {
    const float bc1 = 1.0f - powf(b1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(b2, (float)step));
    for (int t0 = 0; t0 < count; t0 += MULTI_MAX) {
        MultiAdamTable T;
        T.count = count - t0 < MULTI_MAX ? count - t0 : MULTI_MAX;
        int blocks = 0;
        for (int t = 0; t < T.count; ++t) {
            T.p[t] = p[t0 + t]; T.g[t] = g[t0 + t]; T.m[t] = m[t0 + t]; T.v[t] = v[t0 + t]; T.n[t] = n[t0 + t];
            T.boff[t] = blocks;
            blocks += (int)((n[t0 + t] + MULTI_CHUNK - 1) / MULTI_CHUNK);
        }
        T.boff[T.count] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(multi_adam_kernel, dim3(blocks), dim3(256), 0, s, T, lr, b1, b2, eps, wd, decoupled, bc1, bc2s, gscale);
        DSEG_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

int launch_multi_zero(int count, float* const* p, const long* n, hipStream_t s) {
    for (int t0 = 0; t0 < count; t0 += MULTI_MAX) {
        MultiZeroTable T;
        T.count = count - t0 < MULTI_MAX ? count - t0 : MULTI_MAX;
        int blocks = 0;
        for (int t = 0; t < T.count; ++t) {
            T.p[t] = p[t0 + t]; T.n[t] = n[t0 + t];
            T.boff[t] = blocks;
            blocks += (int)((n[t0 + t] + MULTI_CHUNK - 1) / MULTI_CHUNK);
        }
        T.boff[T.count] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(multi_zero_kernel, dim3(blocks), dim3(256), 0, s, T);
        DSEG_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace dseg
