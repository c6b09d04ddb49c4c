// Fused attention forward, 64 query rows per wave (bf16 mode).  Same mathematics and LDS images as attention.hip;
// what changes is the work per K/V byte and per barrier: a 256-query workgroup (4 waves x 2 query blocks of 32)
// sweeps the key tiles, so every LDS-DMA piece, every V^T fragment read and every barrier feeds twice the MFMAs.
// (attention.hip ablations: the K/V loads cost 24 % of the 128-query kernel; an 8-wave 256-query workgroup with 32
// rows per wave did not help because the LDS->register traffic and the barrier group grow with it.)
// The two query blocks go through the score / softmax phase one after the other (32 live score registers), their
// probabilities are kept as bf16 fragments, and the PV phase uses each V^T fragment for both blocks.
#include "attn_common.h"
#include "kernels.h"

namespace dseg {

namespace a64 {
constexpr int KB = 64, KV_TILE = attn::KV_TILE_BYTES, STAGE_BYTES = 2 * KV_TILE;
using attn::sigma23;
using attn::tr_frag;
__device__ __forceinline__ int swz2(int row, int chunk) { return attn::swz(row, chunk); }
__device__ __forceinline__ int tile_off2(int row, int chunk) { return attn::tile_off(row, chunk); }
}  // namespace a64

__global__ __launch_bounds__(256, 2) void attn_fwd64_kernel(AttnParams p) {
    using namespace a64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;

    const int nq = (p.ntok + 255) / 256;
    const int npairs = p.B * p.heads;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nq) * 8 + xcd, qt = slot % nq;
    if (pair >= npairs) return;

    const int ntok = p.ntok, npad = p.npad;
    const long pair_off = (long)pair * npad * 64;
    const bf16_t* Qg = p.q + pair_off;
    const bf16_t* Kg = p.k + pair_off;
    const bf16_t* Vg = p.v + pair_off;

    int qrow[2];
    bf16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qrow[qb] = qt * 256 + wave * 64 + qb * 32 + lr;
        const int qc = qrow[qb] < ntok ? qrow[qb] : ntok - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[qb][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Qg + (long)qc * 64 + s * 16 + lh * 8));
    }

    auto stage = [&](int st, int key0) {
        char* sbase = smem + st * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;               // 0..7: K rows, 8..15: V rows
            const int row = (piece & 7) * 8 + (lane >> 3);
            const int c = swz2(row, lane & 7);
            glds16((piece < 8 ? Kg : Vg) + (long)(key0 + row) * 64 + c * 8, sbase + piece * 1024);
        }
    };

    f32x16 o[2][2], negm[2];
    float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            negm[qb][r] = 0.f;
            o[qb][0][r] = 0.f;
            o[qb][1][r] = 0.f;
        }
    }
    constexpr float RESCALE_THR = 16.f;

    const int ntiles = (ntok + KB - 1) / KB;
    stage(0, 0);
    const int krow_perm = sigma23(lr);
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_gl = (lane >> 4) & 1;

    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < ntiles) stage((t + 1) & 1, (t + 1) * KB);
        const char* sb = smem + (t & 1) * STAGE_BYTES;
        const bool ragged = (t + 1) * KB > ntok;

        bf16x8 pf[2][4];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // ---- S'^T[key][q] = K . Q~^T - m_run  (K fragments re-read per query block: 32 live score registers) ----
            f32x16 sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                sacc[kb] = negm[qb];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    sacc[kb] = mfma32(lds_frag(sb + tile_off2(kb * 32 + krow_perm, s * 2 + lh)), qf[qb][s], sacc[kb]);
            }
            if (ragged) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * KB + kb * 32 + (r >> 3) * 16 + lh * 8 + (r & 7) >= ntok) sacc[kb][r] = -INFINITY;
            }
            float mx = sacc[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (t == 0 || __any(mx > RESCALE_THR)) {
                const float delta = (t == 0) ? mx : fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_run[qb] += delta;
                l_run[qb] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[qb][r] -= delta;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kb][r] -= delta;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][d][r] *= alpha;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float e[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        e[j] = __builtin_amdgcn_exp2f(sacc[kb][s2 * 8 + j]);
                        l_run[qb] += e[j];          // in-lane partial row sum; the lane halves are joined at the end
                    }
                    const uint4 h = {pack_bf16x2(e[0], e[1]), pack_bf16x2(e[2], e[3]), pack_bf16x2(e[4], e[5]), pack_bf16x2(e[6], e[7])};
                    pf[qb][kb * 2 + s2] = __builtin_bit_cast(bf16x8, h);
                }
        }

        // ---- O^T[d][q] += V^T . P^T, each V^T fragment used for both query blocks ----
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int krow0 = ks * 16 + lh * 8 + tr_q;
                const int vch = db * 4 + tr_gl * 2 + (tr_p >> 1);
                const bf16x8 vf = tr_frag(sb + KV_TILE + tile_off2(krow0, vch) + (tr_p & 1) * 8,
                                          sb + KV_TILE + tile_off2(krow0 + 4, vch) + (tr_p & 1) * 8);
                o[0][db] = mfma32(vf, pf[0][ks], o[0][db]);
                o[1][db] = mfma32(vf, pf[1][ks], o[1][db]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    const int b = pair / p.heads, head = pair - b * p.heads;
    const int dm = p.heads * 64;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        if (qrow[qb] < ntok) {
            const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
            const float inv = 1.0f / l_tot;
            bf16_t* dst = p.ctx + ((long)b * ntok + qrow[qb]) * dm + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = db * 32 + 8 * g + 4 * lh;
                    const uint2 hi = {pack_bf16x2(o[qb][db][4 * g] * inv, o[qb][db][4 * g + 1] * inv),
                                      pack_bf16x2(o[qb][db][4 * g + 2] * inv, o[qb][db][4 * g + 3] * inv)};
                    *reinterpret_cast<uint2*>(dst + d) = hi;
                }
            if (p.lse != nullptr && lh == 0) p.lse[(long)pair * ntok + qrow[qb]] = m_run[qb] + __builtin_amdgcn_logf(l_tot);
        }
    }
}

int launch_attention64(const AttnParams& p, hipStream_t s) {
    const int nq = (p.ntok + 255) / 256;
    const int npairs = p.B * p.heads;
    const int grid = ((npairs + 7) / 8) * 8 * nq;
    const size_t lds = 2 * a64::STAGE_BYTES;
    hipLaunchKernelGGL(attn_fwd64_kernel, dim3(grid), dim3(256), lds, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace dseg
