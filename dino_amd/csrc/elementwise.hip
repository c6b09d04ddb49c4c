// HBM-bound helper kernels of the DINOSeg path (gfx950): operand packing, LayerNorm, patch gather,
// CLS rows, pos-embed bicubic resample, classifier tail.  All are coalesced row kernels; reductions
// use 64-lane wavefront shuffles.
#include "common.h"
#include "kernels.h"

namespace dseg {

// ------------------------------------------------------------------------------------------------
// fp32 [rows, cols] -> bf16 hi(/lo) planes [planes][rows_pad][cols_pad], zero padded.
// Used once per weight refresh (nn.Linear weights are [out, in] = the W[N,K] operand of gemm.hip).
__global__ void pack_planes_kernel(const float* __restrict__ src, int rows, int cols, bf16_t* __restrict__ dst,
                                   long plane, int rows_pad, int cols_pad, int planes, int fmt) {
    const long total = (long)rows_pad * cols_pad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols_pad), c = (int)(i - (long)r * cols_pad);
        const float v = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
        bf16_t hi, lo;
        split1(v, fmt, hi, lo);
        dst[i] = hi;
        if (planes == 2) dst[plane + i] = lo;
    }
}

int launch_pack_planes(const float* src, int rows, int cols, bf16_t* dst, long plane, int rows_pad, int cols_pad,
                       int planes, hipStream_t s, int fmt) {
    const long total = (long)rows_pad * cols_pad;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(pack_planes_kernel, dim3(grid), dim3(256), 0, s, src, rows, cols, dst, plane, rows_pad, cols_pad, planes, fmt);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// transposed variant: dst[pl][c][r] = src[r][c]   (W^T operand planes for the input-gradient GEMMs)
__global__ void pack_planes_t_kernel(const float* __restrict__ src, int rows, int cols, bf16_t* __restrict__ dst,
                                     long plane, int rows_pad, int cols_pad, int planes) {
    const long total = (long)rows_pad * cols_pad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i / rows_pad), r = (int)(i - (long)c * rows_pad);
        const float v = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
        const uint32_t hi = pack_bf16x2(v, 0.f);
        dst[i] = (bf16_t)(hi & 0xFFFF);
        if (planes == 2) dst[plane + i] = (bf16_t)(pack_bf16x2(v - bf16_lo_to_f32(hi), 0.f) & 0xFFFF);
    }
}

int launch_pack_planes_t(const float* src, int rows, int cols, bf16_t* dst, long plane, int rows_pad, int cols_pad,
                         int planes, hipStream_t s) {
    const long total = (long)rows_pad * cols_pad;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(pack_planes_t_kernel, dim3(grid), dim3(256), 0, s, src, rows, cols, dst, plane, rows_pad, cols_pad, planes);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// Many packs in one launch: block b serves job t with boff[t] <= b < boff[t+1] (a wave-uniform scan of kernel arguments).
constexpr int PACK_MAX = 40;
struct MultiPackTable {
    PackJob job[PACK_MAX];
    int boff[PACK_MAX + 1];
    int count;
};
// a block moves one 64 x 64 source tile through LDS: coalesced 256-byte row reads, coalesced 128-byte row writes on both the plain
// and the transposed side (the first version -- one element per thread, strided on one side, a division per element -- took 51 us
// for the 15 weights of three blocks where the bytes take 10)
__global__ __launch_bounds__(256) void multi_pack_kernel(MultiPackTable T) {
    __shared__ float tile[64][65];
    int t = 0;
    while (t + 1 < T.count && T.boff[t + 1] <= (int)blockIdx.x) ++t;
    const PackJob& j = T.job[t];
    const int lb = blockIdx.x - T.boff[t];
    const int tiles_c = (j.cols_pad + 63) >> 6;
    const int r0 = (lb / tiles_c) * 64, c0 = (lb % tiles_c) * 64;
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < j.rows && c0 + c < j.cols) ? j.src[(long)(r0 + r) * j.cols + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 256) {
        int r, c;
        long d;
        if (j.transposed) {
            c = i >> 6; r = i & 63;
            d = (long)(c0 + c) * j.rows_pad + r0 + r;
        } else {
            r = i >> 6; c = i & 63;
            d = (long)(r0 + r) * j.cols_pad + c0 + c;
        }
        if (r0 + r >= j.rows_pad || c0 + c >= j.cols_pad) continue;
        const float v = tile[r][c];
        bf16_t hi, lo;
        split1(v, j.fmt, hi, lo);
        j.dst[d] = hi;
        if (j.planes == 2) j.dst[j.plane + d] = lo;
    }
}

int launch_multi_pack(const PackJob* jobs, int count, hipStream_t s) {
    for (int t0 = 0; t0 < count; t0 += PACK_MAX) {
        MultiPackTable T;
        T.count = count - t0 < PACK_MAX ? count - t0 : PACK_MAX;
        int blocks = 0;
        for (int t = 0; t < T.count; ++t) {
            T.job[t] = jobs[t0 + t];
            T.boff[t] = blocks;
            blocks += ((jobs[t0 + t].rows_pad + 63) / 64) * ((jobs[t0 + t].cols_pad + 63) / 64);
        }
        T.boff[T.count] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(multi_pack_kernel, dim3(blocks), dim3(256), 0, s, T);
        DSEG_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (nn.LayerNorm(D, eps=1e-6): vision_transformer.py:303; uses :114,:118,:183).
// One wavefront per row; the row lives in registers (D/128 float2 per lane); mean, then the biased variance
// of the centred values (same two-pass arithmetic as the oracle), reductions by wavefront shuffles.
template <int NV>   // D = 128 * NV
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int M,
                                                        bf16_t* __restrict__ out, long out_plane, int planes,
                                                        float* __restrict__ out_f32, int drop_cls, int ntok, int fmt) {
    constexpr int D = 128 * NV;
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nw = gridDim.x * 4;
    f32x2 g[NV], bta[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g[i] = *reinterpret_cast<const f32x2*>(gamma + i * 128 + lane * 2);
        bta[i] = *reinterpret_cast<const f32x2*>(beta + i * 128 + lane * 2);
    }
    for (int m = wid; m < M; m += nw) {
        long orow = m;
        if (drop_cls) {
            const int b = m / ntok, t = m - b * ntok;
            if (t == 0) continue;
            orow = (long)b * (ntok - 1) + t - 1;
        }
        const float* xr = x + (long)m * D;
        f32x2 v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = *reinterpret_cast<const f32x2*>(xr + i * 128 + lane * 2);
            s += v[i][0] + v[i][1];
        }
        const float mean = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i][0] -= mean;
            v[i][1] -= mean;
            q += v[i][0] * v[i][0] + v[i][1] * v[i][1];
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float y0 = v[i][0] * rstd * g[i][0] + bta[i][0];
            const float y1 = v[i][1] * rstd * g[i][1] + bta[i][1];
            const long o = orow * D + i * 128 + lane * 2;
            if (out_f32) {
                f32x2 y = {y0, y1};
                *reinterpret_cast<f32x2*>(out_f32 + o) = y;
            }
            if (out) {
                uint32_t hi, lo;
                if (fmt == FMT_FP16) split2<FMT_FP16>(y0, y1, hi, lo);      // (wave-uniform)
                else split_bf16x2(y0, y1, hi, lo);
                *reinterpret_cast<uint32_t*>(out + o) = hi;
                if (planes == 2) *reinterpret_cast<uint32_t*>(out + out_plane + o) = lo;
            }
        }
    }
}

int launch_layernorm(const float* x, const float* gamma, const float* beta, float eps, int M, int D, bf16_t* out,
                     long out_plane, int planes, float* out_f32, int drop_cls, int ntok, hipStream_t s, int fmt) {
    if (M <= 0) return 0;
    if (D % 128 != 0 || D > 1024) {
        dinoseg_set_error("layernorm: D=%d must be a multiple of 128 and <= 1024", D);
        return -1;
    }
    int grid = (M + 3) / 4;
    if (grid > 4096) grid = 4096;
#define DSEG_LN(NV)                                                                                              \
    case NV:                                                                                                     \
        hipLaunchKernelGGL((layernorm_kernel<NV>), dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, M, out,    \
                           out_plane, planes, out_f32, drop_cls, ntok, fmt);                                     \
        break;
    switch (D / 128) {
        DSEG_LN(1) DSEG_LN(2) DSEG_LN(3) DSEG_LN(4) DSEG_LN(5) DSEG_LN(6) DSEG_LN(7) DSEG_LN(8)
    }
#undef DSEG_LN
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Patch gather for Conv2d(3, D, kernel 8, stride 8) as a GEMM (vision_transformer.py:153,157):
// out[b*n + py*o + px][c*64 + ky*8 + kx] = pixel(b, c, py*8+ky, px*8+kx).
// kind 0: uint8 HWC frames, with albumentations' Normalize fused: (u8 - 255*mean[c]) * (1 / (255*std[c]))
//         (pl_torch_modules.py:37) -- frames stay uint8 on the wire (691 KB instead of 2.76 MB @480).
// kind 1: fp32 CHW tensor as handed to DINOSeg.forward (pl_torch_modules.py:239).
__global__ __launch_bounds__(256) void patch_gather_kernel(const void* __restrict__ xin, int kind, int B, int r,
                                                           f32x4 mean255, f32x4 inv255, bf16_t* __restrict__ out,
                                                           long out_plane, int planes, int fmt) {
    const int o = r >> 3;
    const long total = (long)B * o * o * 8;   // one work item = (patch, ky): 8 pixels x 3 channels
    for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long)gridDim.x * blockDim.x) {
        const int ky = (int)(w & 7);
        const long patch = w >> 3;
        const int px = (int)(patch % o);
        const int py = (int)((patch / o) % o);
        const int b = (int)(patch / ((long)o * o));
        float v[3][8];
        if (kind == 0) {
            const uint8_t* src = reinterpret_cast<const uint8_t*>(xin) + (((long)b * r + py * 8 + ky) * r + px * 8) * 3;
            uint32_t raw[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) raw[i] = reinterpret_cast<const uint32_t*>(src)[i];   // 24 B, 8-byte aligned
#pragma unroll
            for (int kx = 0; kx < 8; ++kx)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int byte = kx * 3 + c;
                    const float u = (float)((raw[byte >> 2] >> ((byte & 3) * 8)) & 0xFF);
                    v[c][kx] = (u - mean255[c]) * inv255[c];
                }
        } else {
            const float* src = reinterpret_cast<const float*>(xin);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* s = src + (((long)b * 3 + c) * r + py * 8 + ky) * r + px * 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(s), bq = *reinterpret_cast<const f32x4*>(s + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[c][e] = a[e];
                    v[c][4 + e] = bq[e];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint4 hi, lo;
            if (fmt == FMT_FP16) {      // (uniform; pixels are bounded: no saturation needed)
                split2<FMT_FP16>(v[c][0], v[c][1], hi.x, lo.x);
                split2<FMT_FP16>(v[c][2], v[c][3], hi.y, lo.y);
                split2<FMT_FP16>(v[c][4], v[c][5], hi.z, lo.z);
                split2<FMT_FP16>(v[c][6], v[c][7], hi.w, lo.w);
            } else {
                split_bf16x2(v[c][0], v[c][1], hi.x, lo.x);
                split_bf16x2(v[c][2], v[c][3], hi.y, lo.y);
                split_bf16x2(v[c][4], v[c][5], hi.z, lo.z);
                split_bf16x2(v[c][6], v[c][7], hi.w, lo.w);
            }
            bf16_t* dst = out + patch * 192 + c * 64 + ky * 8;
            *reinterpret_cast<uint4*>(dst) = hi;
            if (planes == 2) *reinterpret_cast<uint4*>(dst + out_plane) = lo;
        }
    }
}

int launch_patch_gather(const void* x, int kind, int B, int r, const float* mean255, const float* inv_std255,
                        bf16_t* out, long out_plane, int planes, hipStream_t s, int fmt) {
    const long total = (long)B * (r / 8) * (r / 8) * 8;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    f32x4 m = {mean255[0], mean255[1], mean255[2], 0.f}, iv = {inv_std255[0], inv_std255[1], inv_std255[2], 0.f};
    hipLaunchKernelGGL(patch_gather_kernel, dim3(grid), dim3(256), 0, s, x, kind, B, r, m, iv, out, out_plane, planes, fmt);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// CLS rows: X[b*ntok, :] = cls_token + pos[0]   (vision_transformer.py:229-233)
__global__ void cls_rows_kernel(float* __restrict__ X, const float* __restrict__ cls, const float* __restrict__ pos,
                                int B, int ntok, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i - b * D;
    X[(long)b * ntok * D + d] = cls[d] + pos[d];
}

int launch_cls_rows(float* X, const float* cls, const float* pos, int B, int ntok, int D, hipStream_t s) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3((B * D + 255) / 256), dim3(256), 0, s, X, cls, pos, B, ntok, D);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Pos-embed resample (vision_transformer.py:202-222): bicubic, A = -0.75, align_corners=False, source
// coordinate (dst + 0.5) * (g / (o + 0.1)) - 0.5 (torch uses the GIVEN scale factor), taps clamped to the
// border.  Parameter-only: run once per resolution and cached by the caller.  out: [o*o + 1, D], row 0 = class pos.
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ void pos_resample_kernel(const float* __restrict__ pe, int g, int D, int o, float scale,
                                    float* __restrict__ out) {
    const long total = ((long)o * o + 1) * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const long tokn = i / D;
        if (tokn == 0) {
            out[i] = pe[d];
            continue;
        }
        const int y = (int)((tokn - 1) / o), x = (int)((tokn - 1) % o);
        const float sy = (y + 0.5f) * scale - 0.5f, sx = (x + 0.5f) * scale - 0.5f;
        const float fy = floorf(sy), fx = floorf(sx);
        float wy[4], wx[4];
        cubic_w(sy - fy, wy);
        cubic_w(sx - fx, wx);
        const int iy = (int)fy, ix = (int)fx;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int yy = iy - 1 + a;
            yy = yy < 0 ? 0 : (yy > g - 1 ? g - 1 : yy);
            float row = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int xx = ix - 1 + c;
                xx = xx < 0 ? 0 : (xx > g - 1 ? g - 1 : xx);
                row += pe[(1 + (long)yy * g + xx) * D + d] * wx[c];
            }
            acc += row * wy[a];
        }
        out[i] = acc;
    }
}

int launch_pos_resample(const float* pos_embed, int g, int D, int o, float* out, hipStream_t s) {
    const long total = ((long)o * o + 1) * D;
    if (o == g) {   // the reference returns the stored pos_embed untouched (vision_transformer.py:205-206)
        DSEG_CHECK_HIP(hipMemcpyAsync(out, pos_embed, (size_t)total * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    // torch: scale_factor = (o + 0.1) / g in double; ATen uses (float)(1.0 / scale_factor) for source coordinates
    const float scale = (float)(1.0 / (((double)o + 0.1) / (double)g));
    hipLaunchKernelGGL(pos_resample_kernel, dim3(grid), dim3(256), 0, s, pos_embed, g, D, o, scale, out);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Classifier tail: last Linear (K -> C) + log_softmax(dim=1) + argmax (pl_torch_modules.py:122-123, :294).
// One thread per patch row; the hi+lo input planes are recombined to fp32; W (<= 32 x K fp32) is read through
// wave-uniform (scalar) loads.  argmax keeps the first maximum, as torch.argmax does.
template <int CMAX>
__global__ __launch_bounds__(256) void head_final_kernel(const bf16_t* __restrict__ in, long in_plane, int ld, int M,
                                                         int K, const float* __restrict__ W,
                                                         const float* __restrict__ bias, int C,
                                                         float* __restrict__ logp, int32_t* __restrict__ amax, int fmt) {
    // Sixteen lanes per row: a lane group reads 256 contiguous bytes of the row per step (one thread per row -- the round-1 form --
    // touched 64 rows per load instruction and read 4.6x the activation's bytes from HBM: profiles/r04_pmc_hbm.csv), every lane
    // accumulates its eight columns against the classifier rows staged in LDS, a butterfly over the group finishes the sums.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ws = reinterpret_cast<float*>(smem);          // [C][Kp], Kp = ld rounded up to 8, zero beyond K
    const int Kp = (ld + 7) & ~7;
    for (int i = threadIdx.x; i < C * Kp; i += 256) {
        const int c = i / Kp, k = i - c * Kp;
        Ws[i] = k < K ? W[c * K + k] : 0.f;
    }
    __syncthreads();
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int nchunk = ld >> 3;
    for (long m = (long)blockIdx.x * 16 + grp; m < M; m += (long)gridDim.x * 16) {
        float z[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) z[c] = 0.f;
        const bf16_t* hi = in + m * ld;
        const bf16_t* lo = hi + in_plane;
        for (int ch = sub; ch < nchunk; ch += 16) {
            const uint4 h = *reinterpret_cast<const uint4*>(hi + ch * 8), l = *reinterpret_cast<const uint4*>(lo + ch * 8);
            const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
            float xv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (fmt == FMT_FP16) {
                    xv[2 * e] = lo_to_f32<FMT_FP16>(hw[e]) + lo_to_f32<FMT_FP16>(lw[e]);
                    xv[2 * e + 1] = hi_to_f32<FMT_FP16>(hw[e]) + hi_to_f32<FMT_FP16>(lw[e]);
                } else {
                    xv[2 * e] = bf16_lo_to_f32(hw[e]) + bf16_lo_to_f32(lw[e]);
                    xv[2 * e + 1] = bf16_hi_to_f32(hw[e]) + bf16_hi_to_f32(lw[e]);
                }
            }
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if (c < C) {
                    const float4 w0 = *reinterpret_cast<const float4*>(Ws + c * Kp + ch * 8);
                    const float4 w1 = *reinterpret_cast<const float4*>(Ws + c * Kp + ch * 8 + 4);
                    float a = z[c];
                    a = fmaf(xv[0], w0.x, a); a = fmaf(xv[1], w0.y, a); a = fmaf(xv[2], w0.z, a); a = fmaf(xv[3], w0.w, a);
                    a = fmaf(xv[4], w1.x, a); a = fmaf(xv[5], w1.y, a); a = fmaf(xv[6], w1.z, a); a = fmaf(xv[7], w1.w, a);
                    z[c] = a;
                }
            }
        }
        // butterfly over the 16 lanes of the group: every lane ends with the full sums (+ bias)
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C) {
                float a = z[c];
                a += __shfl_xor(a, 1, 16); a += __shfl_xor(a, 2, 16); a += __shfl_xor(a, 4, 16); a += __shfl_xor(a, 8, 16);
                z[c] = a + bias[c];
            } else {
                z[c] = -INFINITY;
            }
        }
        float mx = z[0];
        int am = 0;
#pragma unroll
        for (int c = 1; c < CMAX; ++c)
            if (c < C && z[c] > mx) {
                mx = z[c];
                am = c;
            }
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) sum += expf(z[c] - mx);
        const float lse = logf(sum);
        // lane `sub` writes classes sub, sub + 16 (the group's stores are one contiguous run of C floats)
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C && (c & 15) == sub) logp[m * C + c] = (z[c] - mx) - lse;
        if (amax && sub == 0) amax[m] = am;
    }
}

int launch_head_final(const bf16_t* in, long in_plane, int ld, int M, int K, const float* W, const float* b, int C,
                      float* logp, int32_t* argmax, hipStream_t s, int fmt) {
    if (M <= 0) return 0;
    if (C < 1 || C > 32 || ld % 8 != 0 || K > ld || (long)C * ld > 16384) {
        dinoseg_set_error("head_final: need 1 <= C <= 32, K <= ld, ld %% 8 == 0 and C * ld <= 16384 (C=%d K=%d ld=%d)", C, K, ld);
        return -1;
    }
    const int ncu = device_cu_count();
    const int want = (M + 15) / 16;
    const int grid = ncu > 0 && want > 8 * ncu ? 8 * ncu : want;       // (grid-stride: the classifier is staged once per workgroup)
    const size_t lds = (size_t)C * ld * sizeof(float);
    if (C <= 8)
        hipLaunchKernelGGL((head_final_kernel<8>), dim3(grid), dim3(256), lds, s, in, in_plane, ld, M, K, W, b, C, logp, argmax, fmt);
    else
        hipLaunchKernelGGL((head_final_kernel<32>), dim3(grid), dim3(256), lds, s, in, in_plane, ld, M, K, W, b, C, logp, argmax, fmt);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Materialised attention probabilities of one block (VisionTransformer.get_last_selfattention,
// vision_transformer.py:273-280 -> Attention.forward :85,:101; caller visualize_attention.py:46).
// Not on the inference hot path (the fused kernel never writes the N x N matrix); used for visualisation only.
// grid (ceil(ntok/16), B*H): scores of 16 queries against all keys (Q~ is pre-scaled by scale*log2e, so the
// softmax is exp2-based), then the same workgroup normalises its 16 rows in place.
__global__ __launch_bounds__(256) void attn_probs_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, long plane,
                                                         int planes, int ntok, int npad, float* __restrict__ out, int fmt) {
    __shared__ float qs[16][64];
    const int pair = blockIdx.y, q0 = blockIdx.x * 16, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bf16_t* Qg = q + (long)pair * npad * 64;
    const bf16_t* Kg = k + (long)pair * npad * 64;
    for (int i = tid; i < 16 * 64; i += 256) {
        const int r = i >> 6, d = i & 63;
        const int qr = q0 + r < ntok ? q0 + r : ntok - 1;
        float v = unpack1(Qg[(long)qr * 64 + d], fmt);
        if (planes == 2) v += unpack1(Qg[plane + (long)qr * 64 + d], fmt);
        qs[r][d] = v;
    }
    __syncthreads();
    float* orow = out + ((long)pair * ntok + q0) * ntok;
    for (int key = tid; key < ntok; key += 256) {
        float kv[64];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint4 h = *reinterpret_cast<const uint4*>(Kg + (long)key * 64 + c * 8);
            const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kv[c * 8 + 2 * e] = fmt == FMT_FP16 ? lo_to_f32<FMT_FP16>(hw[e]) : bf16_lo_to_f32(hw[e]);
                kv[c * 8 + 2 * e + 1] = fmt == FMT_FP16 ? hi_to_f32<FMT_FP16>(hw[e]) : bf16_hi_to_f32(hw[e]);
            }
            if (planes == 2) {
                const uint4 l = *reinterpret_cast<const uint4*>(Kg + plane + (long)key * 64 + c * 8);
                const uint32_t lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    kv[c * 8 + 2 * e] += fmt == FMT_FP16 ? lo_to_f32<FMT_FP16>(lw[e]) : bf16_lo_to_f32(lw[e]);
                    kv[c * 8 + 2 * e + 1] += fmt == FMT_FP16 ? hi_to_f32<FMT_FP16>(lw[e]) : bf16_hi_to_f32(lw[e]);
                }
            }
        }
        for (int r = 0; r < 16; ++r) {
            if (q0 + r >= ntok) break;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) s = fmaf(qs[r][d], kv[d], s);
            orow[(long)r * ntok + key] = s;
        }
    }
    __syncthreads();
    for (int r = wv; r < 16; r += 4) {      // one wavefront per row: max, sum, normalise
        if (q0 + r >= ntok) break;
        float* row = orow + (long)r * ntok;
        float m = -INFINITY;
        for (int j = lane; j < ntok; j += 64) m = fmaxf(m, row[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
        for (int j = lane; j < ntok; j += 64) {
            const float e = __builtin_amdgcn_exp2f(row[j] - m);
            row[j] = e;
            sum += e;
        }
        const float inv = 1.0f / wave_sum(sum);
        for (int j = lane; j < ntok; j += 64) row[j] *= inv;
    }
}

int launch_attn_probs(const bf16_t* q, const bf16_t* k, long plane, int planes, int B, int heads, int ntok, int npad, float* out,
                      hipStream_t s, int fmt) {
    hipLaunchKernelGGL(attn_probs_kernel, dim3((ntok + 15) / 16, B * heads), dim3(256), 0, s, q, k, plane, planes, ntok, npad, out, fmt);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Masked CLS attention of the last block: Attention.forward(x, cls_mask) (vision_transformer.py:80-107), the engine of
// VisionTransformer.forward_mask / get_last_selfattention(x, cls_mask) (:250-280).  Only the CLS query row is used; its
// logits are MULTIPLIED by each mask (the CLS key gets factor 0, so masked keys keep logit 0 -- the reference's arithmetic,
// not a -inf mask), softmax over all N keys, context = probs @ V.  One workgroup per (mask, head); one frame.
// Q~ is pre-scaled by head_dim^-0.5 * log2e, so everything runs in the log2 domain (a factor on the logit commutes).
// Outputs: ctx planes [planes][n_masks][heads*64] (rows = masks) and, optionally, probs [heads][n_masks][ntok] fp32.
__global__ __launch_bounds__(256) void cls_mask_attn_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                            const bf16_t* __restrict__ v, long plane, int planes, int heads, int ntok,
                                                            int npad, const float* __restrict__ mask, int n_masks,
                                                            bf16_t* __restrict__ ctx, long ctx_plane, float* __restrict__ probs,
                                                            int fmt) {
    extern __shared__ float sc[];           // [ntok] scores -> probabilities, then [4][64] partial contexts
    __shared__ float qs[64];
    __shared__ float red[8];
    const int m = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long po = (long)hd * npad * 64;
    // (fp16 mode: Q, K and the ctx output are fp16, V is bf16 -- as the fused attention kernel has them)
    auto ld = [&](const bf16_t* base, long idx, int f) {
        float x = unpack1(base[idx], f);
        if (planes == 2) x += unpack1(base[plane + idx], f);
        return x;
    };
    const int vfmt = planes == 2 ? fmt : FMT_BF16;
    if (tid < 64) qs[tid] = ld(q, po + tid, fmt);                 // CLS row = token 0
    __syncthreads();
    const float* mrow = mask + (long)m * (ntok - 1);
    float mx = -INFINITY;
    for (int n = tid; n < ntok; n += 256) {
        float s = 0.f;
        for (int d = 0; d < 64; ++d) s = fmaf(qs[d], ld(k, po + (long)n * 64 + d, fmt), s);
        s *= (n == 0) ? 0.f : mrow[n - 1];
        sc[n] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int n = tid; n < ntok; n += 256) {
        const float e = __builtin_amdgcn_exp2f(sc[n] - mx);
        sc[n] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wv] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    if (probs) {
        float* pr = probs + ((long)hd * n_masks + m) * ntok;
        for (int n = tid; n < ntok; n += 256) pr[n] = sc[n] * inv;
    }
    // context: wave wv sums keys n = wv, wv+4, ...; lane = d
    float acc = 0.f;
    for (int n = wv; n < ntok; n += 4) acc = fmaf(sc[n], ld(v, po + (long)n * 64 + lane, vfmt), acc);
    __syncthreads();                         // everyone is done reading sc[] as probabilities
    sc[wv * 64 + lane] = acc;
    __syncthreads();
    if (tid < 64) {
        const float o = (sc[tid] + sc[64 + tid] + sc[128 + tid] + sc[192 + tid]) * inv;
        bf16_t hi, lo;
        split1(o, fmt, hi, lo);
        const long idx = (long)m * heads * 64 + hd * 64 + tid;
        ctx[idx] = hi;
        if (planes == 2) ctx[ctx_plane + idx] = lo;
    }
}

int launch_cls_mask_attn(const bf16_t* q, const bf16_t* k, const bf16_t* v, long plane, int planes, int heads, int ntok, int npad,
                         const float* mask, int n_masks, bf16_t* ctx, long ctx_plane, float* probs, hipStream_t s, int fmt) {
    const size_t lds = (size_t)(ntok > 256 ? ntok : 256) * sizeof(float);
    if (lds > 60 * 1024) {
        dinoseg_set_error("cls_mask_attn: %d tokens exceed the LDS score buffer", ntok);
        return -1;
    }
    hipLaunchKernelGGL(cls_mask_attn_kernel, dim3(n_masks, heads), dim3(256), lds, s, q, k, v, plane, planes, heads, ntok, npad, mask,
                       n_masks, ctx, ctx_plane, probs, fmt);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// rows 1 .. n of X [*, D] = row 0 (the CLS residual repeated once per mask, Block.forward vision_transformer.py:131-135)
__global__ void broadcast_row0_kernel(float* __restrict__ X, int D, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)n * D) X[D + i] = X[i % D];
}

int launch_broadcast_row0(float* X, int D, int n, hipStream_t s) {
    const long total = (long)n * D;
    hipLaunchKernelGGL(broadcast_row0_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, D, n);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Confusion matrix for the validation metrics (validation_epoch_end, pl_torch_modules.py:310-332):
// cm[gt][pred] += 1 (int64), block-private histogram in LDS first.
__global__ __launch_bounds__(256) void confusion_kernel(const int32_t* __restrict__ pred, const int64_t* __restrict__ gt, long n,
                                                        int C, unsigned long long* __restrict__ cm) {
    __shared__ unsigned int h[32 * 32];
    for (int i = threadIdx.x; i < C * C; i += 256) h[i] = 0;
    __syncthreads();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int g = (int)gt[i], pr = pred[i];
        if (g >= 0 && g < C && pr >= 0 && pr < C) atomicAdd(&h[g * C + pr], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256)
        if (h[i]) atomicAdd(cm + i, (unsigned long long)h[i]);
}

int launch_confusion(const int32_t* pred, const int64_t* gt, long n, int C, int64_t* cm, hipStream_t s) {
    if (C < 1 || C > 32) {
        dinoseg_set_error("confusion: need 1 <= C <= 32");
        return -1;
    }
    int grid = (int)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(confusion_kernel, dim3(grid), dim3(256), 0, s, pred, gt, n, C, reinterpret_cast<unsigned long long*>(cm));
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ uint8 resize
// Resize(r, r) of get_transforms (pl_torch_modules.py:36) = cv2.resize(..., INTER_LINEAR) on uint8 through albumentations.
// OpenCV is a third-party dependency that is not in /root/reference (opencv_python==4.5.5.62, requirements.txt:7); this
// restates its published fixed-point algorithm (modules/imgproc/src/resize.cpp): per axis  f = (float)((d + 0.5) * scale - 0.5),
// s = floor(f), f -= s;  x axis: f = 0 when s is clamped to [0, w-1];  y axis: rows clamped, f kept;  coefficients
// round(c * 2048) as int16;  horizontal pass H = S[s] * a0 + S[s+1] * a1 in int32;  vertical pass
// ((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2.  An exact 2x downscale takes OpenCV's INTER_AREA fast
// path instead ((s00 + s01 + s10 + s11 + 2) >> 2).  Parity unpinned (no cv2 here): pinned only against oracle/resize_oracle.py.
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst,
                                                        int dh, int dw, double scale_y, double scale_x, int area2) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= dh * dw) return;
    const int dy = idx / dw, dx = idx - dy * dw;
    uint8_t* o = dst + (long)idx * 3;
    if (area2) {
        const uint8_t* r0 = src + ((long)(2 * dy) * sw + 2 * dx) * 3;
        const uint8_t* r1 = r0 + (long)sw * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (uint8_t)((r0[c] + r0[3 + c] + r1[c] + r1[3 + c] + 2) >> 2);
        return;
    }
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    auto coef = [](float v) {       // saturate_cast<short>(v * INTER_RESIZE_COEF_SCALE): round half to even, then clamp
        const int r = (int)rintf(v * 2048.f);
        return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
    };
    const int a0 = coef(1.f - fx), a1 = coef(fx), b0 = coef(1.f - fy), b1 = coef(fy);
    const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
    const int y0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
    const int y1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
    const uint8_t* r0 = src + (long)y0 * sw * 3;
    const uint8_t* r1 = src + (long)y1 * sw * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
        const int h1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
        o[c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
}

int launch_resize_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, hipStream_t s) {
    if (sh < 1 || sw < 1 || dh < 1 || dw < 1) {
        dinoseg_set_error("resize: bad shape %dx%d -> %dx%d", sh, sw, dh, dw);
        return -1;
    }
    const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);      // as cv::resize computes them
    const int area2 = (sw == 2 * dw && sh == 2 * dh) ? 1 : 0;
    const long n = (long)dh * dw;
    hipLaunchKernelGGL(resize_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, sh, sw, dst, dh, dw, scale_y,
                       scale_x, area2);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace dseg
