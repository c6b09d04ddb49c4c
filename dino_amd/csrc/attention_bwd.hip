// Fused multi-head attention backward (flash-style recompute, head_dim 64) for the fine-tune step.
//
// Replaces autograd's _softmax_backward_data + 4 bmm per attention (SURVEY.md §2.1 bwd row; forward at
// vision_transformer.py:85-104) without materialising P: P = exp2(Q~.K^T - LSE2) is recomputed per tile from the
// forward's log2-domain log-sum-exp.  With delta[q] = sum_d dO[q][d] O[q][d]:
//     dP = dO.V^T,   dS = P * (dP - delta),   dQ = scale * dS.K,   dK = ln2 * dS^T.Q~,   dV = P^T.dO
// (Q~ = Q * scale * log2e as stored by the forward; gradients are w.r.t. the un-scaled qkv projection output).
//
// Two kernels, no atomics, deterministic:
//   attn_bwd_dq_kernel : one workgroup per 128 queries, sweeps the key tiles (mirror of the forward kernel).
//                        -LSE2[q] and -delta[q] are the initial accumulators of the S^T and dP^T MFMA chains.
//   attn_bwd_dkv_kernel: one workgroup per 128 keys (K, V fragments in registers), sweeps the query tiles;
//                        per-row -LSE2 / -delta enter as accumulator-initial values read from a small LDS strip.
// Row-major LDS tiles are read row-wise (ds_read_b128) for the score-type products and column-wise
// (ds_read_b64_tr_b16) for the products that contract over the tile's rows.
// Outputs go to dQKV planes [planes][B*ntok][3*heads*64] = gradient of the qkv GEMM output (Q | K | V columns).
#include "attn_common.h"
#include "kernels.h"

namespace dseg {

constexpr int BKV_TILE = attn::KV_TILE_BYTES;   // [64][64] bf16 slab
__device__ __forceinline__ int bswz(int row, int chunk) { return attn::swz(row, chunk); }
__device__ __forceinline__ int boff(int row, int chunk) { return attn::tile_off(row, chunk); }
__device__ __forceinline__ int bsigma23(int i) { return attn::sigma23(i); }
__device__ __forceinline__ bf16x8 btr_frag(const char* p0, const char* p1) { return attn::tr_frag(p0, p1); }

// registers 8*s2..8*s2+7 of a 32x32 accumulator -> bf16 hi(/lo) B-operand fragment
template <int PLANES>
__device__ __forceinline__ void acc_to_frag(const f32x16& a, int s2, bf16x8& hi, bf16x8& lo) {
    uint4 h, l;
    split_bf16x2(a[s2 * 8 + 0], a[s2 * 8 + 1], h.x, l.x);
    split_bf16x2(a[s2 * 8 + 2], a[s2 * 8 + 3], h.y, l.y);
    split_bf16x2(a[s2 * 8 + 4], a[s2 * 8 + 5], h.z, l.z);
    split_bf16x2(a[s2 * 8 + 6], a[s2 * 8 + 7], h.w, l.w);
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

// ------------------------------------------------------------------------------------------------ prep
// neg_lse[pair][q] = -LSE2 (q < ntok) / -inf (pad);  neg_delta[pair][q] = -sum_d dO*O (q < ntok) / 0 (pad); q < npad
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
                                                            long plane, int planes, const float* __restrict__ lse, int B,
                                                            int heads, int ntok, int npad, float* __restrict__ neg_lse,
                                                            float* __restrict__ neg_delta) {
    // a wave takes 8 consecutive rows of one (frame, head): lane = 8 * row + 16-byte chunk of the row's 64 head dimensions
    // (the first version gave a whole wave one 128-byte row: 39 us per layer at 8 frames, nearly all of it waiting)
    const int lane = threadIdx.x & 63, rr = lane >> 3, ch = lane & 7;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int groups = npad / 8;                                   // npad % 64 == 0
    const long total = (long)B * heads * groups;
    if (wid >= total) return;
    const long pair = wid / groups;
    const int q = (int)(wid - pair * groups) * 8 + rr;
    const long out = pair * npad + q;
    if (q >= ntok) {
        if (ch == 0) {
            neg_lse[out] = -INFINITY;
            neg_delta[out] = 0.f;
        }
        return;
    }
    const long b = pair / heads;
    const int head = (int)(pair - b * heads);
    const long off = (b * ntok + q) * (long)(heads * 64) + head * 64 + ch * 8;
    float a[8], o[8];
    {
        const uint4 ua = *reinterpret_cast<const uint4*>(dO + off), uo = *reinterpret_cast<const uint4*>(O + off);
        const uint32_t wa[4] = {ua.x, ua.y, ua.z, ua.w}, wo[4] = {uo.x, uo.y, uo.z, uo.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[2 * i] = __uint_as_float(wa[i] << 16);
            a[2 * i + 1] = __uint_as_float(wa[i] & 0xFFFF0000u);
            o[2 * i] = __uint_as_float(wo[i] << 16);
            o[2 * i + 1] = __uint_as_float(wo[i] & 0xFFFF0000u);
        }
    }
    if (planes == 2) {
        const uint4 ua = *reinterpret_cast<const uint4*>(dO + plane + off), uo = *reinterpret_cast<const uint4*>(O + plane + off);
        const uint32_t wa[4] = {ua.x, ua.y, ua.z, ua.w}, wo[4] = {uo.x, uo.y, uo.z, uo.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[2 * i] += __uint_as_float(wa[i] << 16);
            a[2 * i + 1] += __uint_as_float(wa[i] & 0xFFFF0000u);
            o[2 * i] += __uint_as_float(wo[i] << 16);
            o[2 * i + 1] += __uint_as_float(wo[i] & 0xFFFF0000u);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum = fmaf(a[i], o[i], sum);
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    sum += __shfl_xor(sum, 4);
    if (ch == 0) {
        neg_lse[out] = -lse[pair * ntok + q];
        neg_delta[out] = -sum;
    }
}

// ------------------------------------------------------------------------------------------------ dQ
// NW waves (32 queries each) per workgroup share the K/V tiles: 16 / NW LDS-DMA pieces per wave and tile
template <int PLANES, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dq_kernel(AttnBwdParams p) {
    constexpr int QBLK = NW * 32;
    static_assert(16 % NW == 0, "pieces per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = PLANES * 2 * BKV_TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;

    const int nq = (p.ntok + QBLK - 1) / QBLK;
    const int npairs = p.B * p.heads;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nq) * 8 + xcd, qt = slot % nq;
    if (pair >= npairs) return;
    const int ntok = p.ntok, npad = p.npad, dm = p.heads * 64;
    const int b = pair / p.heads, head = pair - b * p.heads;
    const long pair_off = (long)pair * npad * 64;
    const bf16_t* Qg = p.q + pair_off;
    const bf16_t* Kg = p.k + pair_off;
    const bf16_t* Vg = p.v + pair_off;

    const int qrow = qt * QBLK + wave * 32 + lr;
    const int qc = qrow < ntok ? qrow : ntok - 1;
    bf16x8 qf[PLANES][4], df[PLANES][4];
    const bf16_t* dOrow = p.dO + ((long)b * ntok + qc) * dm + head * 64;
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[pl][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Qg + pl * p.qkv_plane + (long)qc * 64 + s * 16 + lh * 8));
            df[pl][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(dOrow + pl * p.dO_plane + s * 16 + lh * 8));
        }
    f32x16 negl, negd;
    {
        const float nl = p.neg_lse[(long)pair * npad + qc], nd = p.neg_delta[(long)pair * npad + qc];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            negl[r] = nl;
            negd[r] = nd;
        }
    }

    auto stage = [&](int st, int key0) {
        char* sbase = smem + st * STAGE_BYTES;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int i = 0; i < 16 / NW; ++i) {
                const int piece = wave * (16 / NW) + i;
                const int row = (piece & 7) * 8 + (lane >> 3);
                const int c = bswz(row, lane & 7);
                const bf16_t* src = (piece < 8 ? Kg : Vg) + pl * p.qkv_plane + (long)(key0 + row) * 64 + c * 8;
                glds16(src, sbase + pl * 2 * BKV_TILE + piece * 1024);
            }
    };

    f32x16 dq[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    const int ntiles = (ntok + 63) / 64;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int rperm = bsigma23(lr);
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_gl = (lane >> 4) & 1;

    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) stage(cur ^ 1, (t + 1) * 64);
        const char* sb = smem + cur * STAGE_BYTES;

        f32x16 sacc[2], pacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            sacc[kb] = negl;     // S'^T = K.Q~^T - LSE2
            pacc[kb] = negd;     // dP'^T = V.dO^T - delta
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int off = boff(kb * 32 + rperm, s * 2 + lh);
                const bf16x8 khi = lds_frag(sb + off), vhi = lds_frag(sb + BKV_TILE + off);
                if (PLANES == 2) {
                    const bf16x8 klo = lds_frag(sb + 2 * BKV_TILE + off), vlo = lds_frag(sb + 3 * BKV_TILE + off);
                    sacc[kb] = mfma32(klo, qf[0][s], sacc[kb]);
                    sacc[kb] = mfma32(khi, qf[PLANES - 1][s], sacc[kb]);
                    pacc[kb] = mfma32(vlo, df[0][s], pacc[kb]);
                    pacc[kb] = mfma32(vhi, df[PLANES - 1][s], pacc[kb]);
                }
                sacc[kb] = mfma32(khi, qf[0][s], sacc[kb]);
                pacc[kb] = mfma32(vhi, df[0][s], pacc[kb]);
            }
        }
        if ((t + 1) * 64 > ntok) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 64 + kb * 32 + (r >> 3) * 16 + lh * 8 + (r & 7) >= ntok) sacc[kb][r] = -INFINITY;
        }
        // dS^T = P * (dP - delta), in place in pacc
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[kb][r] *= __builtin_amdgcn_exp2f(sacc[kb][r]);

        bf16x8 sf[PLANES][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 hi, lo;
                acc_to_frag<PLANES>(pacc[kb], s2, hi, lo);
                sf[0][kb * 2 + s2] = hi;
                if (PLANES == 2) sf[PLANES - 1][kb * 2 + s2] = lo;
            }
        // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]   (K^T fragments by transposing reads of the K tile)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int krow0 = ks * 16 + lh * 8 + tr_q;
                const int ch = db * 4 + tr_gl * 2 + (tr_p >> 1);
                const int o0 = boff(krow0, ch) + (tr_p & 1) * 8, o1 = boff(krow0 + 4, ch) + (tr_p & 1) * 8;
                const bf16x8 khi = btr_frag(sb + o0, sb + o1);
                if (PLANES == 2) {
                    const bf16x8 klo = btr_frag(sb + 2 * BKV_TILE + o0, sb + 2 * BKV_TILE + o1);
                    dq[db] = mfma32(klo, sf[0][ks], dq[db]);
                    dq[db] = mfma32(khi, sf[PLANES - 1][ks], dq[db]);
                }
                dq[db] = mfma32(khi, sf[0][ks], dq[db]);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (qrow < ntok) {
        bf16_t* dst = p.dqkv + ((long)b * ntok + qrow) * (3 * dm) + head * 64;      // Q columns
        const float sc = 0.125f;   // head_dim^-0.5
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * lh;
                uint2 hi, lo;
                split_bf16x2(dq[db][4 * g] * sc, dq[db][4 * g + 1] * sc, hi.x, lo.x);
                split_bf16x2(dq[db][4 * g + 2] * sc, dq[db][4 * g + 3] * sc, hi.y, lo.y);
                *reinterpret_cast<uint2*>(dst + d) = hi;
                if (PLANES == 2) *reinterpret_cast<uint2*>(dst + p.dqkv_plane + d) = lo;
            }
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int PLANES, int NW>
__global__ __launch_bounds__(NW * 64, (PLANES == 1 ? 2 : 1)) void attn_bwd_dkv_kernel(AttnBwdParams p) {
    constexpr int KBLK = NW * 32;
    static_assert(16 % NW == 0 && NW >= 2, "pieces per wave; waves 0 and 1 fetch the row terms");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = PLANES * 2 * BKV_TILE + 512;     // Q~ and dO slabs per plane + 64 x (-LSE2, -delta)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;

    const int nkb = (p.ntok + KBLK - 1) / KBLK;
    const int npairs = p.B * p.heads;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nkb) * 8 + xcd, kt = slot % nkb;
    if (pair >= npairs) return;
    const int ntok = p.ntok, npad = p.npad, dm = p.heads * 64;
    const int b = pair / p.heads, head = pair - b * p.heads;
    const long pair_off = (long)pair * npad * 64;
    const bf16_t* Qg = p.q + pair_off;
    const bf16_t* Kg = p.k + pair_off;
    const bf16_t* Vg = p.v + pair_off;
    const bf16_t* dOg = p.dO + (long)b * ntok * dm + head * 64;     // row stride dm
    const float* nlg = p.neg_lse + (long)pair * npad;
    const float* ndg = p.neg_delta + (long)pair * npad;

    // this wave's 32 keys: K and V fragments as B operands (k = d, column = key)
    const int krow = kt * KBLK + wave * 32 + lr;
    const int kc = krow < npad ? krow : npad - 1;
    bf16x8 kf[PLANES][4], vf[PLANES][4];
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[pl][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Kg + pl * p.qkv_plane + (long)kc * 64 + s * 16 + lh * 8));
            vf[pl][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Vg + pl * p.qkv_plane + (long)kc * 64 + s * 16 + lh * 8));
        }

    auto stage = [&](int st, int q0) {
        char* sbase = smem + st * STAGE_BYTES;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int i = 0; i < 16 / NW; ++i) {
                const int piece = wave * (16 / NW) + i;       // 0..7: Q~ rows, 8..15: dO rows
                const int row = (piece & 7) * 8 + (lane >> 3);
                const int c = bswz(row, lane & 7);
                const bf16_t* src;
                if (piece < 8) {
                    src = Qg + pl * p.qkv_plane + (long)(q0 + row) * 64 + c * 8;          // padded per pair: rows < npad
                } else {
                    int qr = q0 + row;
                    qr = qr < ntok ? qr : ntok - 1;                                       // dO lives in the [M, D] layout
                    src = dOg + pl * p.dO_plane + (long)qr * dm + c * 8;
                }
                glds16(src, sbase + pl * 2 * BKV_TILE + piece * 1024);
            }
        if (wave < 2) {     // 64 floats each: wave 0 -> -LSE2, wave 1 -> -delta (4-byte LDS-DMA, lane-linear)
            const float* src = (wave == 0 ? nlg : ndg) + q0 + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sbase + PLANES * 2 * BKV_TILE + wave * 256),
                                             4, 0, 0);
        }
    };

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dk[d][r] = 0.f;
            dv[d][r] = 0.f;
        }

    const int ntiles = (ntok + 63) / 64;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int rperm = bsigma23(lr);
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_gl = (lane >> 4) & 1;

    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) stage(cur ^ 1, (t + 1) * 64);
        const char* sb = smem + cur * STAGE_BYTES;
        const float* strip = reinterpret_cast<const float*>(sb + PLANES * 2 * BKV_TILE);

        // S'[q][key] = Q~.K^T - LSE2[q];  dP'[q][key] = dO.V^T - delta[q]: accumulator register 8*s2+j of block qb is
        // query  qb*32 + 16*s2 + 8*lh + j  (rows read through the bit-2<->3 swap), so its initial value is a strip read
        f32x16 sacc[2], pacc[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int q8 = qb * 32 + s2 * 16 + lh * 8;
                const f32x4 l0 = *reinterpret_cast<const f32x4*>(strip + q8), l1 = *reinterpret_cast<const f32x4*>(strip + q8 + 4);
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(strip + 64 + q8), d1 = *reinterpret_cast<const f32x4*>(strip + 64 + q8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sacc[qb][s2 * 8 + e] = l0[e];
                    sacc[qb][s2 * 8 + 4 + e] = l1[e];
                    pacc[qb][s2 * 8 + e] = d0[e];
                    pacc[qb][s2 * 8 + 4 + e] = d1[e];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int off = boff(qb * 32 + rperm, s * 2 + lh);
                const bf16x8 qhi = lds_frag(sb + off), dhi = lds_frag(sb + BKV_TILE + off);
                if (PLANES == 2) {
                    const bf16x8 qlo = lds_frag(sb + 2 * BKV_TILE + off), dlo = lds_frag(sb + 3 * BKV_TILE + off);
                    sacc[qb] = mfma32(qlo, kf[0][s], sacc[qb]);
                    sacc[qb] = mfma32(qhi, kf[PLANES - 1][s], sacc[qb]);
                    pacc[qb] = mfma32(dlo, vf[0][s], pacc[qb]);
                    pacc[qb] = mfma32(dhi, vf[PLANES - 1][s], pacc[qb]);
                }
                sacc[qb] = mfma32(qhi, kf[0][s], sacc[qb]);
                pacc[qb] = mfma32(dhi, vf[0][s], pacc[qb]);
            }
        }
        // P (pad queries have -LSE2 = -inf -> P = 0), then dS = P * dP'
        bf16x8 pf[PLANES][4], sf[PLANES][4];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = __builtin_amdgcn_exp2f(sacc[qb][r]);
                sacc[qb][r] = pr;
                pacc[qb][r] *= pr;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 hi, lo;
                acc_to_frag<PLANES>(sacc[qb], s2, hi, lo);
                pf[0][qb * 2 + s2] = hi;
                if (PLANES == 2) pf[PLANES - 1][qb * 2 + s2] = lo;
                acc_to_frag<PLANES>(pacc[qb], s2, hi, lo);
                sf[0][qb * 2 + s2] = hi;
                if (PLANES == 2) sf[PLANES - 1][qb * 2 + s2] = lo;
            }
        }
        // dV^T[d][key] += dO^T[d][q] . P[q][key];   dK^T[d][key] += Q~^T[d][q] . dS[q][key]
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int qrow0 = ks * 16 + lh * 8 + tr_q;
                const int ch = db * 4 + tr_gl * 2 + (tr_p >> 1);
                const int o0 = boff(qrow0, ch) + (tr_p & 1) * 8, o1 = boff(qrow0 + 4, ch) + (tr_p & 1) * 8;
                const bf16x8 qhi = btr_frag(sb + o0, sb + o1), dhi = btr_frag(sb + BKV_TILE + o0, sb + BKV_TILE + o1);
                if (PLANES == 2) {
                    const bf16x8 qlo = btr_frag(sb + 2 * BKV_TILE + o0, sb + 2 * BKV_TILE + o1);
                    const bf16x8 dlo = btr_frag(sb + 3 * BKV_TILE + o0, sb + 3 * BKV_TILE + o1);
                    dv[db] = mfma32(dlo, pf[0][ks], dv[db]);
                    dv[db] = mfma32(dhi, pf[PLANES - 1][ks], dv[db]);
                    dk[db] = mfma32(qlo, sf[0][ks], dk[db]);
                    dk[db] = mfma32(qhi, sf[PLANES - 1][ks], dk[db]);
                }
                dv[db] = mfma32(dhi, pf[0][ks], dv[db]);
                dk[db] = mfma32(qhi, sf[0][ks], dk[db]);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (krow < ntok) {
        bf16_t* dstK = p.dqkv + ((long)b * ntok + krow) * (3 * dm) + dm + head * 64;
        bf16_t* dstV = dstK + dm;
        const float ln2 = 0.69314718055994530942f;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * lh;
                uint2 hi, lo;
                split_bf16x2(dk[db][4 * g] * ln2, dk[db][4 * g + 1] * ln2, hi.x, lo.x);
                split_bf16x2(dk[db][4 * g + 2] * ln2, dk[db][4 * g + 3] * ln2, hi.y, lo.y);
                *reinterpret_cast<uint2*>(dstK + d) = hi;
                if (PLANES == 2) *reinterpret_cast<uint2*>(dstK + p.dqkv_plane + d) = lo;
                split_bf16x2(dv[db][4 * g], dv[db][4 * g + 1], hi.x, lo.x);
                split_bf16x2(dv[db][4 * g + 2], dv[db][4 * g + 3], hi.y, lo.y);
                *reinterpret_cast<uint2*>(dstV + d) = hi;
                if (PLANES == 2) *reinterpret_cast<uint2*>(dstV + p.dqkv_plane + d) = lo;
            }
    }
}

template <int PLANES, int NW, int NWK = NW>      // waves per workgroup of the dQ / the dK,dV kernel
static int launch_bwd(const AttnBwdParams& p, hipStream_t s) {
    const int nq = (p.ntok + NW * 32 - 1) / (NW * 32), nkb = (p.ntok + NWK * 32 - 1) / (NWK * 32);
    const int npairs = p.B * p.heads;
    const int grid = ((npairs + 7) / 8) * 8 * nq, grid_k = ((npairs + 7) / 8) * 8 * nkb;
    const size_t lds_dq = (size_t)2 * PLANES * 2 * BKV_TILE;
    const size_t lds_dkv = (size_t)2 * (PLANES * 2 * BKV_TILE + 512);
    static PerDeviceOnce once;
    if (once.first()) {
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<PLANES, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq));
        DSEG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<PLANES, NWK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dkv));
        once.mark();
    }
    const long row_groups = (long)p.B * p.heads * (p.npad / 8);      // 8 rows per wave, 4 waves per block
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((row_groups + 3) / 4)), dim3(256), 0, s, p.dO, p.O, p.dO_plane, PLANES,
                       p.lse, p.B, p.heads, p.ntok, p.npad, p.neg_lse, p.neg_delta);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<PLANES, NW>), dim3(grid), dim3(NW * 64), lds_dq, s, p);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<PLANES, NWK>), dim3(grid_k), dim3(NWK * 64), lds_dkv, s, p);
    DSEG_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_attention_bwd(const AttnBwdParams& p, hipStream_t s) {
    if (p.npad % 64 != 0 || p.npad < p.ntok) {
        dinoseg_set_error("attention_bwd: npad=%d must be a multiple of 64 and >= ntok=%d", p.npad, p.ntok);
        return -1;
    }
    if (p.planes == 1) {
        // 256-query workgroups for the dQ kernel once the 128-row grid is a round deep (2 workgroups of 4 waves per CU): half the
        // LDS-DMA pieces per wave and tile, as in attention_z.hip; same arithmetic per row
        const int ncu = device_cu_count();
        const long wgs4 = (long)((p.B * p.heads + 7) / 8) * 8 * ((p.ntok + 127) / 128);
        // (dQ kernel only: the dK,dV kernel holds three 4-wave workgroups per CU at its 166 registers, an 8-wave one would be alone;
        //  measured at 8 frames @480: 1509 -> 1515 frames/s with both wide, 1522 with dQ wide only)
        if (ncu > 0 && wgs4 >= 2L * ncu && !(options().attn_variant & 64)) return launch_bwd<1, 8, 4>(p, s);
        return launch_bwd<1, 4>(p, s);
    }
    if (p.planes == 2) return launch_bwd<2, 4>(p, s);
    dinoseg_set_error("attention_bwd: planes must be 1 or 2");
    return -1;
}

}  // namespace dseg
