// Fused multi-head attention forward (flash-style, head_dim = 64) for gfx950.
//
// Replaces the reference's materialised attention -- q@k^T, *scale, softmax, @v on a [B,H,N,N] fp32 tensor
// (vision_transformer.py:85,101,104; 297 MiB per frame per block @480) -- with a tiled kernel that never
// writes the N x N scores: per 128-query workgroup, K / V^T tiles of 64 keys stream through LDS (LDS-DMA,
// double-buffered, XOR-swizzled 128-byte rows), scores live in MFMA accumulators, softmax is online.
//
// Orientation ("keys on the MFMA rows"): S^T = K . Q^T with v_mfma_f32_32x32x16_bf16, so each lane owns one
// query column and 32 of the tile's 64 scores -> row max / row sum are in-lane plus one cross-half exchange.
// The K fragment rows are read through the bit-2<->bit-3 swap sigma(i): accumulator registers 8s..8s+7 of a
// lane half h then hold keys 16s+8h+0..7 in natural order, which is exactly the B-operand fragment of
// O^T += V^T . P^T -- the probabilities never leave registers (no LDS round trip, no permutes).
// V is consumed as V^T[d][key] (written transposed by the QKV GEMM epilogue), so both LDS images are plain
// row tiles.  Q arrives pre-scaled by head_dim^-0.5 * log2(e); exp is v_exp_f32 (2^x).
//
// PLANES = 2 (parity mode): Q, K, V and P are bf16 hi+lo pairs; each product is 3 MFMAs.
#include "common.h"
#include "kernels.h"

namespace dseg {

constexpr int QW = 32;          // query rows per wave
constexpr int QB = 128;         // query rows per workgroup (4 waves)
constexpr int KB = 64;          // keys per tile
constexpr int KV_TILE = 64 * 128;   // [64][64] bf16 slab = 8 KiB

__device__ __forceinline__ int sigma23(int i) {   // swap bits 2 and 3
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}

template <int PLANES>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = PLANES * 2 * KV_TILE;   // per plane: K slab + V^T slab

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;

    // XCD-aware order: all q-tiles of one (batch, head) run on one XCD back-to-back (K/V stay in that L2).
    const int nq = (p.ntok + QB - 1) / QB;
    const int npairs = p.B * p.heads;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nq) * 8 + xcd, qt = slot % nq;
    if (pair >= npairs) return;

    const int ntok = p.ntok, npad = p.npad;
    const long pair_off = (long)pair * npad * 64;       // same element count for Q, K ([npad][64]) and V^T ([64][npad])
    const bf16_t* Qg = p.q + pair_off;
    const bf16_t* Kg = p.k + pair_off;
    const bf16_t* Vg = p.vt + pair_off;

    // ---- Q fragments (B operand: k = d, col = query) straight from global into registers ----
    int qrow = qt * QB + wave * QW + lr;
    const int qrow_c = qrow < ntok ? qrow : ntok - 1;
    bf16x8 qf[PLANES][4];
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[pl][s] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const uint4*>(Qg + pl * p.qkv_plane + (long)qrow_c * 64 + s * 16 + lh * 8));

    auto stage = [&](int st, int key0) {
        char* sbase = smem + st * STAGE_BYTES;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int piece = wave * 4 + i;               // 0..7: K rows, 8..15: V^T rows
                const int row = (piece & 7) * 8 + (lane >> 3);
                const int c = swz_chunk(row, lane & 7);
                const bf16_t* src = (piece < 8)
                    ? Kg + pl * p.qkv_plane + (long)(key0 + row) * 64 + c * 8
                    : Vg + pl * p.qkv_plane + (long)row * npad + key0 + c * 8;
                glds16(src, sbase + pl * 2 * KV_TILE + piece * 1024);
            }
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (ntok + KB - 1) / KB;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int krow_perm = sigma23(lr);

    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) stage(cur ^ 1, (t + 1) * KB);
        const char* sb = smem + cur * STAGE_BYTES;

        // ---- S^T[key][q] = K . Q^T ----
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int off = tile_off_bytes(kb * 32 + krow_perm, s * 2 + lh);
                const bf16x8 khi = lds_frag(sb + off);
                if (PLANES == 2) {
                    const bf16x8 klo = lds_frag(sb + 2 * KV_TILE + off);
                    sacc[kb] = mfma32(klo, qf[0][s], sacc[kb]);
                    sacc[kb] = mfma32(khi, qf[PLANES - 1][s], sacc[kb]);
                }
                sacc[kb] = mfma32(khi, qf[0][s], sacc[kb]);
            }
        }
        // lane (query lr, half lh): sacc[kb][8*s2 + j] is key  t*64 + kb*32 + s2*16 + lh*8 + j

        if ((t + 1) * KB > ntok) {   // ragged last tile: mask keys >= ntok (wave-uniform branch)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KB + kb * 32 + (r >> 3) * 16 + lh * 8 + (r & 7);
                    if (key >= ntok) sacc[kb][r] = -INFINITY;
                }
        }

        // ---- online softmax (log2 domain) ----
        float mx = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(sacc[kb][r] - m_new);
                sacc[kb][r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

        // ---- P fragments (B operand: k = key, col = query): registers 8*s2..8*s2+7 of sacc[kb] ----
        bf16x8 pf[PLANES][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 hi, lo;
                split_bf16x2(sacc[kb][s2 * 8 + 0], sacc[kb][s2 * 8 + 1], hi.x, lo.x);
                split_bf16x2(sacc[kb][s2 * 8 + 2], sacc[kb][s2 * 8 + 3], hi.y, lo.y);
                split_bf16x2(sacc[kb][s2 * 8 + 4], sacc[kb][s2 * 8 + 5], hi.z, lo.z);
                split_bf16x2(sacc[kb][s2 * 8 + 6], sacc[kb][s2 * 8 + 7], hi.w, lo.w);
                pf[0][kb * 2 + s2] = __builtin_bit_cast(bf16x8, hi);
                if (PLANES == 2) pf[PLANES - 1][kb * 2 + s2] = __builtin_bit_cast(bf16x8, lo);
            }

        // ---- O^T[d][q] += V^T . P^T ----
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = KV_TILE + tile_off_bytes(db * 32 + lr, ks * 2 + lh);
                const bf16x8 vhi = lds_frag(sb + off);
                if (PLANES == 2) {
                    const bf16x8 vlo = lds_frag(sb + 2 * KV_TILE + off);
                    o[db] = mfma32(vlo, pf[0][ks], o[db]);
                    o[db] = mfma32(vhi, pf[PLANES - 1][ks], o[db]);
                }
                o[db] = mfma32(vhi, pf[0][ks], o[db]);
            }
        }

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- normalise and write ctx[b*ntok + q][head*64 + d] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (qrow < ntok) {
        const int b = pair / p.heads, head = pair - b * p.heads;
        const int dm = p.heads * 64;
        bf16_t* dst = p.ctx + ((long)b * ntok + qrow) * dm + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * lh;   // registers 4g..4g+3 are d .. d+3
                uint2 hi, lo;
                split_bf16x2(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv, hi.x, lo.x);
                split_bf16x2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv, hi.y, lo.y);
                *reinterpret_cast<uint2*>(dst + d) = hi;
                if (PLANES == 2) *reinterpret_cast<uint2*>(dst + p.ctx_plane + d) = lo;
            }
        if (p.lse != nullptr && lh == 0) p.lse[(long)pair * ntok + qrow] = m_run + __builtin_amdgcn_logf(l_tot);
    }
}

template <int PLANES>
static int launch_attn(const AttnParams& p, hipStream_t s) {
    const int nq = (p.ntok + QB - 1) / QB;
    const int npairs = p.B * p.heads;
    const int grid = ((npairs + 7) / 8) * 8 * nq;
    const size_t lds = (size_t)2 * PLANES * 2 * KV_TILE;
    static bool attr_done = false;
    if (!attr_done) {
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<PLANES>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<PLANES>), dim3(grid), dim3(256), lds, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_attention(const AttnParams& p, hipStream_t s) {
    if (p.npad % KB != 0 || p.npad < p.ntok) {
        dinoseg_set_error("attention: npad=%d must be a multiple of 64 and >= ntok=%d", p.npad, p.ntok);
        return -1;
    }
    if (p.planes == 1) return launch_attn<1>(p, s);
    if (p.planes == 2) return launch_attn<2>(p, s);
    dinoseg_set_error("attention: planes must be 1 or 2");
    return -1;
}

}  // namespace dseg
