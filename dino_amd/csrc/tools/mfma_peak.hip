// Measurement kernel, not part of the path: what dense bf16 MFMA rate does this chip sustain, as a function of the resident
// waves per SIMD and of the operand bits?  (SURVEY.md section 8d asks for a measured peak next to the vendor's 2.5 PFLOP/s;
// DESIGN.md section 4 leans on two facts this kernel shows directly: one wave issues at most one 32x32x16 MFMA per ~64 cycles,
// and the clock drops with switching activity.)  Registers only: no LDS, no memory traffic in the loop.
// Built into dino_amd/lib/libdinoseg_tools.so (make tools), NOT into the product library; tools/mfma_peak.py drives it.
#include <stdarg.h>
#include <stdio.h>

#include "../common.h"

namespace dseg {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// FV / FS: independent VALU (v_add_f32) / SALU (s_add_u32) instructions after every group of four MFMAs -- how much of a wave's
// non-MFMA work do the other resident waves' MFMAs hide?
// FN / FW: s_nop 0 / s_waitcnt lgkmcnt(0) (nothing outstanding) fillers -- sequencer-internal instructions.
template <int NACC, int FV, int FS, int FN = 0, int FW = 0>
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, uint32_t seed, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    // operands: seed == 0 -> all-zero bits (no switching activity); else pseudo-random bf16 in [-2, 2)
    uint4 au, bu;
    uint32_t h = seed * 2654435761u + (uint32_t)(blockIdx.x * 256 + threadIdx.x) * 40503u;
    auto next = [&]() {
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
        const uint32_t lo = 0x3F80u | (h & 0x807Fu), hi = 0x3F80u | ((h >> 16) & 0x807Fu);      // +-[1, 2)
        return seed ? (lo | (hi << 16)) : 0u;
    };
    au = {next(), next(), next(), next()};
    bu = {next(), next(), next(), next()};
    const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float fv = (float)lane;
    uint32_t fs = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = mfma32(a, b, acc[i]);     // 16 MFMAs per trip, NACC independent chains
            if ((u * NACC) % 4 == 4 - NACC || NACC == 4) {                     // after every fourth MFMA
#pragma unroll
                for (int k = 0; k < FV; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(fv));
#pragma unroll
                for (int k = 0; k < FS; ++k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(fs) : : "scc");
#pragma unroll
                for (int k = 0; k < FN; ++k) asm volatile("s_nop 0");
#pragma unroll
                for (int k = 0; k < FW; ++k) asm volatile("s_waitcnt lgkmcnt(0)");
            }
        }
    }
    float s = fv * 1e-30f + (float)(fs & 1);
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][lane & 15];
    if (s == 12345.678f) out[0] = s;      // keeps the loop alive
}

// The same FLOPs per trip (16 x 32x32x16 = 32 x 16x16x32) and the same 64 accumulator registers with the other bf16 shape, and both
// shapes with the A operand re-read from LDS for every MFMA (one ds_read_b128 = 1 KiB per wave-instruction, as in the attention and
// MLP kernels: 1 KiB of LDS per 32x32x16 MFMA, 2 KiB per the two 16x16x32 that replace it).  MI355X_MICROARCH.md, DVFS give-back
// item 7: on random data the chip may hold a higher clock on one shape than on the other at equal cycles per FLOP.
template <int SHAPE16, int LDSFED>
__global__ __launch_bounds__(256) void mfma_shape_kernel(int iters, uint32_t seed, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint4 lds[4 * 8 * 64];      // per wave: 8 fragments of 1 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t h = seed * 2654435761u + (uint32_t)(blockIdx.x * 256 + threadIdx.x) * 40503u;
    auto next = [&]() {
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
        const uint32_t lo = 0x3F80u | (h & 0x807Fu), hi = 0x3F80u | ((h >> 16) & 0x807Fu);
        return seed ? (lo | (hi << 16)) : 0u;
    };
    uint4 bu = {next(), next(), next(), next()};
    const bf16x8 b = __builtin_bit_cast(bf16x8, bu);
    for (int f = 0; f < 8; ++f) lds[(wave * 8 + f) * 64 + lane] = uint4{next(), next(), next(), next()};
    __syncthreads();
    const uint4* my = lds + wave * 8 * 64 + lane;
    bf16x8 areg[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) areg[f] = __builtin_bit_cast(bf16x8, my[f * 64]);
    f32x16 acc32[4];
    f32x4_t acc16[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc16[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (SHAPE16 == 0) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                bf16x8 a = areg[u & 7];
                if (LDSFED) a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const volatile uint4*>(my + (u & 7) * 64));
                acc32[u & 3] = mfma32(a, b, acc32[u & 3]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                bf16x8 a = areg[u & 7];
                if (LDSFED) a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const volatile uint4*>(my + (u & 7) * 64));
                acc16[u & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[u & 15], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc32[i][lane & 15];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc16[i][lane & 3];
    if (s == 12345.678f) out[0] = s;
}


// One wave's attention tile as a hand-placed instruction stream (registers only): 16 "gaps" per trip, each one MFMA followed by
// the vector work of one packed pair of probabilities.  FILL: 0 bare MFMAs; 1: + 2 v_exp_f32; 2: + 2 v_exp + 2 v_add (row sums,
// two chains, consumers one gap behind their producers); 3: + 2 v_exp + 2 v_add + 1 v_cvt_pk_bf16_f32 (the full mix of
// attention_z.hip's tile: 5 fillers, 28.5 issue cycles by MI355X_MICROARCH.md's table); 4: the same multiset per tile, NOT
// interleaved (16 MFMAs, then 32 exp, 32 add, 16 cvt: what a wave does without software pipelining).
template <int FILL>
__global__ __launch_bounds__(256) void attn_gap_kernel(int iters, uint32_t seed, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    uint32_t h = seed * 2654435761u + (uint32_t)(blockIdx.x * 256 + threadIdx.x) * 40503u;
    auto next = [&]() {
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
        const uint32_t lo = 0x3F80u | (h & 0x807Fu), hi = 0x3F80u | ((h >> 16) & 0x807Fu);
        return seed ? (lo | (hi << 16)) : 0u;
    };
    uint4 au = {next(), next(), next(), next()}, bu = {next(), next(), next(), next()};
    const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float sc[32], pr[32];
    uint32_t pk[16];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        sc[i] = seed ? (float)((int)(next() & 0xFFF) - 2048) * (1.0f / 512.0f) : 0.f;      // scores in [-4, 4)
        pr[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) pk[i] = 0;
    float sum0 = 0.f, sum1 = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (FILL == 4) {
#pragma unroll
            for (int g = 0; g < 16; ++g) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int g = 0; g < 32; ++g) asm volatile("v_exp_f32 %0, %1" : "=v"(pr[g]) : "v"(sc[g]));
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum0) : "v"(pr[2 * g]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum1) : "v"(pr[2 * g + 1]));
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[g]) : "v"(pr[2 * g]), "v"(pr[2 * g + 1]));
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int q = (g + 15) & 15;        // the pair produced one gap earlier
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
                if (FILL >= 1) {
                    asm volatile("v_exp_f32 %0, %1" : "=v"(pr[2 * g]) : "v"(sc[2 * g]));
                    if (FILL >= 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum0) : "v"(pr[2 * q]));
                    asm volatile("v_exp_f32 %0, %1" : "=v"(pr[2 * g + 1]) : "v"(sc[2 * g + 1]));
                    if (FILL >= 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum1) : "v"(pr[2 * q + 1]));
                    if (FILL >= 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[q]) : "v"(pr[2 * q]), "v"(pr[2 * q + 1]));
                }
            }
        }
    }
    float s = sum0 + sum1;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][lane & 15];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += (float)pk[i];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += pr[i];
    if (s == 12345.678f) out[0] = s;
}

// grid = CUs x waves_per_simd workgroups of 4 waves (one per SIMD); returns the launch through *flops = MFMA FLOPs issued
// chains: 1, 2, 4 independent accumulator chains; or 100 + n: four chains and n VALU fillers per four MFMAs (n = 8, 16, 32);
// or 200 + n: n SALU fillers; 300 + n (16, 32): s_nop 0; 400 + n (16, 32): s_waitcnt lgkmcnt(0)
// 504 / 604 / 704: the shape kernel above: 16x16x32 from registers / 32x32x16 LDS-fed / 16x16x32 LDS-fed (500 = 32x32x16 from
// registers through the same kernel, the control)
static int launch_mfma_peak(int waves_per_simd, int iters, uint32_t seed, int chains, float* out, double* flops, hipStream_t s) {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -2;
    if (ncu <= 0 || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1) return -1;
    const int grid = ncu * waves_per_simd;
#define DSEG_PK(N, V, S) hipLaunchKernelGGL((mfma_peak_kernel<N, V, S>), dim3(grid), dim3(256), 0, s, iters, seed, out)
    switch (chains) {
        case 1: DSEG_PK(1, 0, 0); break;
        case 2: DSEG_PK(2, 0, 0); break;
        case 4: DSEG_PK(4, 0, 0); break;
        case 108: DSEG_PK(4, 8, 0); break;
        case 116: DSEG_PK(4, 16, 0); break;
        case 132: DSEG_PK(4, 32, 0); break;
        case 208: DSEG_PK(4, 0, 8); break;
        case 216: DSEG_PK(4, 0, 16); break;
        case 232: DSEG_PK(4, 0, 32); break;
        case 316: hipLaunchKernelGGL((mfma_peak_kernel<4, 0, 0, 16, 0>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 332: hipLaunchKernelGGL((mfma_peak_kernel<4, 0, 0, 32, 0>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 416: hipLaunchKernelGGL((mfma_peak_kernel<4, 0, 0, 0, 16>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 432: hipLaunchKernelGGL((mfma_peak_kernel<4, 0, 0, 0, 32>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 500: hipLaunchKernelGGL((mfma_shape_kernel<0, 0>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 504: hipLaunchKernelGGL((mfma_shape_kernel<1, 0>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 604: hipLaunchKernelGGL((mfma_shape_kernel<0, 1>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 704: hipLaunchKernelGGL((mfma_shape_kernel<1, 1>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 800: hipLaunchKernelGGL((attn_gap_kernel<0>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 801: hipLaunchKernelGGL((attn_gap_kernel<1>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 802: hipLaunchKernelGGL((attn_gap_kernel<2>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 803: hipLaunchKernelGGL((attn_gap_kernel<3>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        case 804: hipLaunchKernelGGL((attn_gap_kernel<4>), dim3(grid), dim3(256), 0, s, iters, seed, out); break;
        default: fprintf(stderr, "mfma_peak: chains = %d\n", chains); return -1;
    }
#undef DSEG_PK
    if (hipGetLastError() != hipSuccess) return -2;
    if (flops) *flops = (double)grid * 4 * iters * 16 * (2.0 * 32 * 32 * 16);
    return 0;
}

}  // namespace dseg

// error plumbing of common.h's macros (unused here, but the header declares them)
extern "C" void dinoseg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
int device_cu_count() { return -1; }

/* Measurement only (no reference counterpart): a register-only / LDS-fed bf16 MFMA loop on waves_per_simd resident waves per SIMD of
 * every CU; *flops_out = FLOPs the launch issues.  tools/mfma_peak.py turns it into the measured dense-bf16 peak SURVEY.md 8d asks for. */
extern "C" int dinoseg_tools_mfma_peak(int32_t waves_per_simd, int32_t iters, uint32_t seed, int32_t chains, float* scratch,
                                       double* flops_out, void* stream) {
    return dseg::launch_mfma_peak(waves_per_simd, iters, seed, chains, scratch, flops_out, reinterpret_cast<hipStream_t>(stream));
}
