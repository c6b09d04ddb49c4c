// Shared device helpers for the dinoseg HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dseg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef uint16_t bf16_t;   // storage type of a bf16 element in HBM / LDS

constexpr int WAVE = 64;

// ---- bf16 conversion (round-to-nearest-even; hipcc emits v_cvt_pk_bf16_f32) ----
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed << 16); }
__device__ __forceinline__ float bf16_hi_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed & 0xFFFF0000u); }
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }

// ---- operand format of the single-plane modes: bf16 (8 significand bits, fp32's exponent range) or fp16 (11 bits, |x| < 65504).
// Both are 16-bit storage (bf16_t) and ride in the same registers (bf16x8 = four VGPRs); what differs is the conversion
// (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32), the way back to fp32, and the MFMA opcode (same rate).  Kernels take it as a template
// parameter; the pack / gather kernels that run once per weight refresh take it at run time.
enum OperandFmt { FMT_BF16 = 0, FMT_FP16 = 1 };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
template <int FMT>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    if constexpr (FMT == FMT_FP16) return pack_f16x2(a, b);
    else return pack_bf16x2(a, b);
}
template <int FMT>
__device__ __forceinline__ float lo_to_f32(uint32_t packed) {
    if constexpr (FMT == FMT_FP16) return (float)__builtin_bit_cast(f16x2, packed)[0];
    else return bf16_lo_to_f32(packed);
}
template <int FMT>
__device__ __forceinline__ float hi_to_f32(uint32_t packed) {
    if constexpr (FMT == FMT_FP16) return (float)__builtin_bit_cast(f16x2, packed)[1];
    else return bf16_hi_to_f32(packed);
}
// run-time forms (pack kernels, the small gather kernels of the visualisation paths)
__device__ __forceinline__ bf16_t pack1(float v, int fmt) {
    return (bf16_t)((fmt == FMT_FP16 ? pack_f16x2(v, 0.f) : pack_bf16x2(v, 0.f)) & 0xFFFFu);
}
__device__ __forceinline__ float unpack1(bf16_t v, int fmt) {
    return fmt == FMT_FP16 ? (float)__builtin_bit_cast(f16x2, (uint32_t)v)[0] : bf16_to_f32(v);
}

// hi/lo split of two floats: hi = bf16(x), lo = bf16(x - hi).  hi + lo carries ~16 mantissa bits.
__device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf16x2(a, b);
    lo = pack_bf16x2(a - bf16_lo_to_f32(hi), b - bf16_hi_to_f32(hi));
}

// the same split in either operand format: fp16 hi + fp16 lo carries ~22 significand bits for |x| >= 2^-13 (below that the lo plane is
// subnormal: the absolute error stays <= 2^-25), bf16 hi + lo ~16 bits at any magnitude.  SAT: clamp to the fp16 range first (an
// unbounded value -- a GEMM output -- saturates at +-65504 instead of turning into inf - inf = NaN); bounded values skip it.
template <int FMT, bool SAT = false>
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    if constexpr (FMT == FMT_FP16) {
        if constexpr (SAT) {
            a = __builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f);
            b = __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f);
        }
        hi = pack_f16x2(a, b);
        // a - (float)hi as ONE instruction per element: v_fma_mix_f32 reads the fp16 half in place (fma(hi, -1, a): the product is exact,
        // one rounding -- the same bits as v_cvt_f32_f16 + v_sub_f32, two issue slots less per pair in the hi+lo attention's tile body)
        float ra, rb;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
        lo = pack_f16x2(ra, rb);
    } else {
        split_bf16x2(a, b, hi, lo);
    }
}
// single plane, saturating where the format has a range to leave
template <int FMT>
__device__ __forceinline__ uint32_t pack2_sat(float a, float b) {
    if constexpr (FMT == FMT_FP16) return pack_f16x2(__builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f));
    else return pack_bf16x2(a, b);
}
// run-time form of the split (pack / gather kernels)
__device__ __forceinline__ void split1(float v, int fmt, bf16_t& hi, bf16_t& lo) {
    hi = pack1(v, fmt);
    lo = pack1(v - unpack1(hi, fmt), fmt);
}

// ---- 128-byte-row LDS tile swizzle -------------------------------------------------
// A tile is [rows][64 bf16] = 128-B rows, i.e. 8 chunks of 16 B per row.  ds_read_b128 is
// served in 16-lane groups over a 256-B bank row; reading the same logical chunk of 16
// different rows would hit only 2 of 16 slots.  XOR the chunk index with (row>>1)&7:
// the 16 rows of every lane group then land on 16 distinct slots (conflict-free) for both
// the 32x32x16 fragment pattern (row = lane&31) used here.
// The image is filled by LDS-DMA (global_load_lds, lane-linear destination), so the
// permutation is applied to the per-lane SOURCE address and again on the read.
__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int tile_off_bytes(int row, int chunk) { return row * 128 + (swz_chunk(row, chunk) << 4); }

// async 16-byte global -> LDS copy (LDS-DMA). lds_wave_base must be wave-uniform;
// lane i's 16 bytes land at lds_wave_base + 16*i.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// the same product on operands of either format (the registers carry bit patterns: bf16x8 is just "eight 16-bit elements")
template <int FMT>
__device__ __forceinline__ f32x16 mfma32f(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (FMT == FMT_FP16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 lds_frag(const char* p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
}

// 32x32 accumulator: register r of lane (col = lane&31, h = lane>>5) is row (r&3) + 8*(r>>2) + 4*h.
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Exact-form GELU 0.5*x*(1+erf(x/sqrt2)) (nn.GELU() default, vision_transformer.py:50,55) with erfc by
// Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7) written without cancellation:
//   1 + erf(z) = 2 - t*P(t)*exp(-z^2) for z >= 0,  = t*P(t)*exp(-z^2) for z < 0,  t = 1/(1 + 0.3275911 |z|).
// ~14 VALU ops (one v_rcp, one v_exp) instead of libm erff's ~60.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float pe = pl * t * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
    return 0.5f * x * (x >= 0.f ? 2.0f - pe : pe);
}

// bf16 fast mode: Phi(x) ~ logistic(x (a1 + a3 x^2 + a5 x^4)), coefficients fitted (minimax over |x| <= 9) to the exact
// erf form: |gelu_fast - gelu_erf| <= 2.6e-5 everywhere, below the bf16 rounding of the stored activation for
// |gelu| > 0.013.  9 VALU instructions (44 issue cycles) against 16 (72): the fc1 epilogue is VALU-issue bound.
// The argument of the polynomial is clamped (a5 < 0 turns it around beyond |x| ~ 11); the product uses the raw x.
__device__ __forceinline__ float gelu_fast(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
    const float u = xc * xc;
    // coefficients of -t in log2 units: -(1.59501055, 7.40160400e-2, -7.03804786e-4) * log2(e)
    float q = fmaf(1.01537542e-3f, u, -1.06782573e-1f);
    q = fmaf(q, u, -2.30111381f);
    const float e = __builtin_amdgcn_exp2f(q * xc);            // exp(-t)
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// d/dx of the exact GELU: Phi(x) + x * phi(x), same erfc approximation (shares the exponential).
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);   // exp(-x^2/2)
    const float pe = pl * t * e;
    const float cdf = x >= 0.f ? 1.0f - 0.5f * pe : 0.5f * pe;
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}

// XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch, speed only).
// Gives every XCD a contiguous range of logical ids; bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

}  // namespace dseg

// ---- host-side error plumbing shared by all translation units ----
#include <atomic>
// hipFuncSetAttribute (the > 64 KiB dynamic-LDS opt-in) and the CU count are per device: once-flags keyed by ordinal
struct PerDeviceOnce {
    std::atomic<bool> done[64];
    PerDeviceOnce() { for (auto& d : done) d.store(false); }
    // true until the per-device setup has been marked done (callers do the setup, then call mark()): two host threads may both do
    // it -- hipFuncSetAttribute is idempotent -- but none launches before its own setup call has returned.  Devices beyond the
    // table always set up again.
    bool first() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
        return !done[d].load(std::memory_order_acquire);
    }
    void mark() {
        int d = 0;
        if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) done[d].store(true, std::memory_order_release);
    }
};
extern "C" void dinoseg_set_error(const char* fmt, ...);
// compute units of the current device (cached per ordinal); <= 0 on failure with the error message set
int device_cu_count();
#define DSEG_CHECK_HIP(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            dinoseg_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -2;                                                                        \
        }                                                                                     \
    } while (0)
