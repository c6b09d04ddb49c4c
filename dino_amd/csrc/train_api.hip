// Fine-tune step of the DINOSeg hot path: forward with saved activations, backward, gradients
// written into caller-bound fp32 buffers.  Replaces DINOSeg.training_step + autograd.backward
// (pl_torch_modules.py:261-268) for one data-parallel rank; the cross-rank gradient mean is done by the caller
// (torch.distributed all_reduce over RCCL, dino_amd/parallel.py).  Host code only.
//
// Backward of y = act(x W^T + b), given dY (activation derivative already applied):
//   dX = dY . W          -> gemm.hip NT kernel with the transposed packed weight W^T[K][N] as the "W" operand
//   dW = dY^T . X        -> both operands transposed to [*, rows] planes (transpose_planes_kernel), same NT kernel
//                           with the batch rows as the contraction, split over grid.y, fp32 atomics into dW
//   db = column sums of dY (by-product of the transpose kernel)
#include <string.h>

#include <vector>

#include "handle.h"

namespace {

constexpr size_t DET_FLOATS = (size_t)1024 * 3 * 1024;      // option deterministic: scratch for per-block partial sums (12 MiB) ...
constexpr size_t DET_TN_FLOATS = (size_t)768 * 3072;        // ... and gemm_tn's per-slice bias partials, one region per stream (2 x 9 MiB)
constexpr int SPLITK_TILES = 768;      // partial 128x128 fp32 tiles of one weight-gradient GEMM (48 MiB): what the workspace holds
inline int splitk_budget() {
    const int v = dseg::options().splitk_tiles;
    return v < 1 ? 1 : v > SPLITK_TILES ? SPLITK_TILES : v;
}

struct TrainLayout {
    int n, ntok, npad, M, Mp, Mpad, Mppad, Cmax;
    // per block (offsets are for block 0; block l adds l * blk_stride)
    size_t Xin, A1, Q, K, V, LSE, CTX, Xmid, A2, HPRE, HB, blk_stride;
    size_t Xfin, PATCH, FEAT, H1, H2, LOGP, DZ;
    size_t dX, dA, dXp, G, dCTX, T1, T2, NLSE, NDEL, DPOS, SINK, ACC, SPLITK, DET;
    size_t zero_begin, zero_end;      // Q/K/V of every block (pad rows must be zero)
    size_t total;
    long a_plane, qkv_plane, f_plane, feat_plane, h1_plane, h2_plane, dz_plane, patch_plane, g_plane, t_plane;
};

TrainLayout make_train_layout(const dinoseg_handle* h, int B, int r) {
    const dinoseg_config& c = h->cfg;
    const int D = c.embed_dim, F = D * c.mlp_ratio, P = h->planes, HP = head_planes();
    TrainLayout L;
    memset(&L, 0, sizeof(L));
    L.n = (r / 8) * (r / 8);
    L.ntok = L.n + 1;
    L.npad = (L.ntok + 63) / 64 * 64;
    L.M = B * L.ntok;
    L.Mp = B * L.n;
    L.Mpad = (L.M + 63) / 64 * 64;
    L.Mppad = (L.Mp + 63) / 64 * 64;
    L.Cmax = 3 * D > F ? 3 * D : F;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    L.a_plane = (long)L.M * D;
    L.qkv_plane = (long)B * c.num_heads * L.npad * 64;
    L.f_plane = (long)L.M * F;
    // block 0
    const size_t b0 = off;
    L.Xin = take((size_t)L.M * D * 4);
    L.A1 = take((size_t)P * L.a_plane * 2);
    L.Q = take((size_t)P * L.qkv_plane * 2);
    L.K = take((size_t)P * L.qkv_plane * 2);
    L.V = take((size_t)P * L.qkv_plane * 2);
    L.LSE = take((size_t)B * c.num_heads * L.ntok * 4);
    L.CTX = take((size_t)P * L.a_plane * 2);
    L.Xmid = take((size_t)L.M * D * 4);
    L.A2 = take((size_t)P * L.a_plane * 2);
    L.HPRE = take((size_t)P * L.f_plane * 2);
    L.HB = take((size_t)P * L.f_plane * 2);
    L.blk_stride = off - b0;
    off = b0 + L.blk_stride * (c.n_blocks > 0 ? c.n_blocks : 1);
    L.Xfin = take((size_t)L.M * D * 4);
    L.patch_plane = (long)L.Mp * 192;
    L.PATCH = take((size_t)P * L.patch_plane * 2);
    L.feat_plane = (long)L.Mp * D;
    L.FEAT = take((size_t)HP * L.feat_plane * 2);
    L.h1_plane = (long)L.Mp * 256;
    L.H1 = take((size_t)HP * L.h1_plane * 2);
    L.h2_plane = (long)L.Mp * 128;
    L.H2 = take((size_t)HP * L.h2_plane * 2);
    L.LOGP = take((size_t)L.Mp * c.n_classes * 4);
    L.dz_plane = (long)L.Mp * 64;
    L.DZ = take((size_t)HP * L.dz_plane * 2);
    // backward scratch
    L.dX = take((size_t)L.M * D * 4);
    L.dA = take((size_t)L.M * D * 4);
    L.dXp = take((size_t)2 * L.a_plane * 2);
    L.g_plane = (long)L.M * L.Cmax;
    L.G = take((size_t)2 * L.g_plane * 2);
    {   // d ctx planes [M, D]; also hosts the head's d h1 planes [Mp, 256]
        const size_t e = (size_t)L.a_plane > (size_t)L.Mp * 256 ? (size_t)L.a_plane : (size_t)L.Mp * 256;
        L.dCTX = take(2 * e * 2);
    }
    L.t_plane = (long)L.Cmax * L.Mpad;
    L.T1 = take((size_t)2 * L.t_plane * 2);
    L.T2 = take((size_t)2 * L.t_plane * 2);
    L.NLSE = take((size_t)B * c.num_heads * L.npad * 4);
    L.NDEL = take((size_t)B * c.num_heads * L.npad * 4);
    L.DPOS = take((size_t)L.ntok * D * 4);
    L.SINK = take((size_t)4 * 1024 * 4);
    L.ACC = take(256);        // nll_loss accumulators {sum of -logp[y], valid rows} (the sticky bad-label flag lives in the handle)
    L.SPLITK = take((size_t)SPLITK_TILES * 128 * 128 * 4);     // split-K partial tiles of the weight gradients
    L.DET = take((size_t)(DET_FLOATS + 2 * DET_TN_FLOATS) * 4);       // option deterministic: per-block partial sums (the largest user: LayerNorm backward, 1024 blocks x 3 x D)
    L.total = off;
    return L;
}

struct TLin {
    std::string wname;
    int N, K, n_pad, k_pad, planes;     // W is [N][K]; W^T planes are [k_pad][n_pad]
};

std::vector<TLin> transposed_specs(const dinoseg_handle* h) {
    const dinoseg_config& c = h->cfg;
    const int D = c.embed_dim, F = D * c.mlp_ratio, P = h->planes, HP = head_planes(), C = c.n_classes;
    std::vector<TLin> v;
    for (int i = 0; i < c.n_blocks; ++i) {
        const std::string b = "dino.blocks." + std::to_string(i) + ".";
        v.push_back({b + "attn.qkv.weight", 3 * D, D, 3 * D, D, P});
        v.push_back({b + "attn.proj.weight", D, D, D, D, P});
        v.push_back({b + "mlp.fc1.weight", F, D, F, D, P});
        v.push_back({b + "mlp.fc2.weight", D, F, D, F, P});
    }
    if (c.head_kind == DINOSEG_HEAD_MLP) {
        v.push_back({"clf.layer_1.weight", 200, D, 256, D, HP});
        v.push_back({"clf.layer_2.weight", 100, 200, 128, 256, HP});
        v.push_back({"clf.layer_3.weight", C, 100, 64, 128, HP});
    } else {
        v.push_back({"clf.layer_1.weight", C, D, 64, D, HP});
    }
    return v;
}

struct TW {
    bf16_t* w;
    long plane;
};

}  // namespace

int dinoseg_train_release(dinoseg_handle* h) {
    if (h->tws) (void)hipFree(h->tws);
    if (h->twbuf) (void)hipFree(h->twbuf);
    if (h->bad_label_flag) (void)hipFree(h->bad_label_flag);
    h->bad_label_flag = nullptr;
    h->tws = nullptr;
    h->twbuf = nullptr;
    h->tws_bytes = h->twbuf_bytes = 0;
    return 0;
}

extern "C" int dinoseg_bind_grad(dinoseg_handle* h, const char* name, float* dev_ptr) {
    if (!h || !name) {
        dinoseg_set_error("dinoseg_bind_grad: null argument");
        return -1;
    }
    if (!h->expected.count(name)) {
        dinoseg_set_error("dinoseg_bind_grad: unexpected key '%s'", name);
        return -1;
    }
    if (dev_ptr) h->grads[name] = dev_ptr;
    else h->grads.erase(name);
    return 0;
}

extern "C" int dinoseg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, int32_t decoupled, int32_t step, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || step < 1) {
        dinoseg_set_error("dinoseg_adam_step: bad argument");
        return -1;
    }
    return launch_adam(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, decoupled, step, grad_scale,
                       reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_adam_step_multi(int32_t count, float* const* p, const float* const* g, float* const* m, float* const* v,
                                       const int64_t* n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                       int32_t decoupled, int32_t step, float grad_scale, void* stream) {
    if (count < 0 || (count > 0 && (!p || !g || !m || !v || !n)) || step < 1) {
        dinoseg_set_error("dinoseg_adam_step_multi: bad argument");
        return -1;
    }
    std::vector<long> nn(count);
    for (int i = 0; i < count; ++i) {
        if (!p[i] || !g[i] || !m[i] || !v[i] || n[i] < 0) {
            dinoseg_set_error("dinoseg_adam_step_multi: null pointer or negative size at tensor %d", i);
            return -1;
        }
        nn[i] = (long)n[i];
    }
    return launch_multi_adam(count, p, g, m, v, nn.data(), lr, beta1, beta2, eps, weight_decay, decoupled, step, grad_scale,
                             reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_op_attention_bwd(const void* q, const void* k, const void* v, int64_t qkv_plane, const void* dO,
                                        const void* O, int64_t o_plane, const float* lse, float* scratch, void* dqkv,
                                        int64_t dqkv_plane, int32_t B, int32_t heads, int32_t ntok, int32_t npad,
                                        int32_t planes, void* stream) {
    AttnBwdParams a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.qkv_plane = qkv_plane;
    a.dO = (const bf16_t*)dO; a.O = (const bf16_t*)O; a.dO_plane = o_plane; a.lse = lse;
    a.neg_lse = scratch; a.neg_delta = scratch + (size_t)B * heads * npad;
    a.dqkv = (bf16_t*)dqkv; a.dqkv_plane = dqkv_plane;
    a.B = B; a.heads = heads; a.ntok = ntok; a.npad = npad; a.planes = planes;
    return launch_attention_bwd(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_op_layernorm_bwd(const float* dy, const float* x, const float* gamma, float eps, int32_t M, int32_t D,
                                        float* dx, int32_t accumulate, float* dgamma, float* dbeta, int32_t drop_cls,
                                        int32_t ntok, void* stream) {
    return launch_layernorm_bwd(dy, x, gamma, eps, M, D, dx, accumulate, dgamma, dbeta, drop_cls, ntok,
                                reinterpret_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------------ the step
// Forward with saved activations (DINOSeg.forward under autograd, pl_torch_modules.py:239-256).  The saved state stays valid
// until the next call; train_backward_impl consumes it.
static int train_forward_impl(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r, float* logp_out, hipStream_t s) {
    if (!h || !x || B <= 0) {
        dinoseg_set_error("dinoseg_train_forward: bad argument");
        return -1;
    }
    if (r <= 0 || r % 8 != 0) {
        dinoseg_set_error("Resolution should be a multiple of 8.");
        return -1;
    }
    if (x_kind != DINOSEG_INPUT_U8_HWC && x_kind != DINOSEG_INPUT_F32_CHW) {
        dinoseg_set_error("dinoseg_train_forward: bad x_kind %d", x_kind);
        return -1;
    }
    if (!h->weights_ready) {
        dinoseg_set_error("dinoseg_train_forward: weights not packed (call dinoseg_refresh_weights after binding)");
        return -3;
    }
    if (h->fmt != FMT_BF16) {
        dinoseg_set_error("dinoseg_train_forward: precision fp16 is inference-only (fp16 gradients would need loss scaling); use bf16 or bf16x3");
        return -1;
    }
    h->tr_B = -1;       // no valid saved forward until this one has been enqueued completely
    DSEG_TRY(check_stream_device(h, s));
    DSEG_TRY(dinoseg_prepare_resolution(h, r, reinterpret_cast<void*>(s)));
    const dinoseg_config& c = h->cfg;
    const int D = c.embed_dim, F = D * c.mlp_ratio, P = h->planes, HP = head_planes(), H = c.num_heads, C = c.n_classes;
    const int NB = c.n_blocks;
    const bool mlp_head = c.head_kind == DINOSEG_HEAD_MLP;
    const TrainLayout L = make_train_layout(h, B, r);

    // ---- workspace
    if (!h->bad_label_flag) {       // (its own allocation: a change of batch shape re-lays the workspace, the latched flag must survive it)
        DSEG_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&h->bad_label_flag), 256));
        DSEG_CHECK_HIP(hipMemsetAsync(h->bad_label_flag, 0, 256, s));
    }
    if (L.total > h->tws_bytes) {
        if (h->tws) {
            DSEG_CHECK_HIP(hipStreamSynchronize(s));
            DSEG_CHECK_HIP(hipFree(h->tws));
        }
        h->tws = nullptr;
        h->tws_bytes = 0;
        DSEG_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&h->tws), L.total));
        h->tws_bytes = L.total;
        h->tws_B = -1;
    }
    char* ws = h->tws;
    if (h->tws_B != B || h->tws_r != r) {
        for (int l = 0; l < NB; ++l)   // Q/K/V pad rows must be zero (never written afterwards)
            DSEG_CHECK_HIP(hipMemsetAsync(ws + L.Q + l * L.blk_stride, 0, L.LSE - L.Q, s));
        DSEG_CHECK_HIP(hipMemsetAsync(ws + L.ACC, 0, 256, s));
        h->tws_B = B;
        h->tws_r = r;
    }
    auto F32 = [&](size_t o) { return reinterpret_cast<float*>(ws + o); };
    auto B16 = [&](size_t o) { return reinterpret_cast<bf16_t*>(ws + o); };

    // =============================================================== forward (activations kept)
    float mean255[3], inv255[3];
    norm_consts(mean255, inv255);
    bf16_t* PATCH = B16(L.PATCH);
    DSEG_TRY(launch_patch_gather(x, x_kind, B, r, mean255, inv255, PATCH, L.patch_plane, P, s));
    float* X0 = NB > 0 ? F32(L.Xin) : F32(L.Xfin);
    {
        const PackedLinear& pk = h->packed.at("dino.patch_embed.proj.weight");
        GemmParams g = {};
        g.A = PATCH; g.a_plane = L.patch_plane; g.lda = 192;
        g.W = pk.w; g.w_plane = pk.plane;
        g.M = L.Mp; g.N = D; g.K = 192; g.planes = P; g.epi = EPI_PATCH;
        g.bias = W(h, "dino.patch_embed.proj.bias");
        g.out_f32 = X0; g.ldo_f32 = D;
        g.pos = h->pos_cache; g.n_patches = L.n;
        DSEG_TRY(launch_gemm(g, s));
    }
    DSEG_TRY(launch_cls_rows(X0, W(h, "dino.cls_token"), h->pos_cache, B, L.ntok, D, s));
    const float qscale = 0.125f * 1.44269504088896340736f;

    for (int l = 0; l < NB; ++l) {
        const std::string b = "dino.blocks." + std::to_string(l) + ".";
        const size_t o = l * L.blk_stride;
        float* Xin = F32(L.Xin + o);
        float* Xmid = F32(L.Xmid + o);
        float* Xout = l + 1 < NB ? F32(L.Xin + o + L.blk_stride) : F32(L.Xfin);
        bf16_t *A1 = B16(L.A1 + o), *Q = B16(L.Q + o), *Kb = B16(L.K + o), *V = B16(L.V + o), *CTX = B16(L.CTX + o);
        bf16_t *A2 = B16(L.A2 + o), *HPRE = B16(L.HPRE + o), *HB = B16(L.HB + o);
        const bool fuse_ln = options().gemm_ln != 0 && L.qkv_plane < (1L << 31) && L.f_plane < (1L << 31);
        if (fuse_ln && h->packed_slab.count(b + "attn.qkv.weight")) {
            // LN1 + qkv in one launch; the normalised planes the weight gradient needs are a by-product (a_out)
            LnGemmParams g = {};
            g.X = Xin; g.ldx = D; g.gamma = W(h, b + "norm1.weight"); g.beta = W(h, b + "norm1.bias"); g.eps = c.ln_eps;
            g.W = h->packed_slab.at(b + "attn.qkv.weight"); g.bias = W(h, b + "attn.qkv.bias");
            g.M = L.M; g.N = 3 * D; g.epi = EPI_QKV;
            g.q = Q; g.k = Kb; g.v = V; g.qkv_plane = L.qkv_plane;
            g.ntok = L.ntok; g.npad = L.npad; g.heads = H; g.dmodel = D; g.qscale = qscale;
            g.a_out = A1; g.a_plane = L.a_plane;
            DSEG_TRY(launch_gemm_ln(g, D, P, s));
        } else {
        DSEG_TRY(launch_layernorm(Xin, W(h, b + "norm1.weight"), W(h, b + "norm1.bias"), c.ln_eps, L.M, D, A1, L.a_plane, P,
                                  nullptr, 0, L.ntok, s));
        {
            const PackedLinear& pk = h->packed.at(b + "attn.qkv.weight");
            GemmParams g = {};
            g.A = A1; g.a_plane = L.a_plane; g.lda = D; g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.M; g.N = 3 * D; g.K = D; g.planes = P; g.epi = EPI_QKV; g.bias = W(h, b + "attn.qkv.bias");
            g.q = Q; g.k = Kb; g.v = V; g.qkv_plane = L.qkv_plane;
            g.ntok = L.ntok; g.npad = L.npad; g.heads = H; g.dmodel = D; g.qscale = qscale;
            DSEG_TRY(launch_gemm(g, s));
        }
        }
        {
            AttnParams a = {};
            a.q = Q; a.k = Kb; a.v = V; a.qkv_plane = L.qkv_plane; a.ctx = CTX; a.ctx_plane = L.a_plane;
            a.lse = F32(L.LSE + o);
            a.B = B; a.heads = H; a.ntok = L.ntok; a.npad = L.npad; a.planes = P;
            DSEG_TRY(launch_attention(a, s));
        }
        {
            const PackedLinear& pk = h->packed.at(b + "attn.proj.weight");
            GemmParams g = {};
            g.A = CTX; g.a_plane = L.a_plane; g.lda = D; g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.M; g.N = D; g.K = D; g.planes = P; g.epi = EPI_RESID; g.bias = W(h, b + "attn.proj.bias");
            g.resid = Xin; g.out_f32 = Xmid; g.ldo_f32 = D;
            DSEG_TRY(launch_gemm(g, s));
        }
        if (fuse_ln && h->packed_slab.count(b + "mlp.fc1.weight")) {
            LnGemmParams g = {};
            g.X = Xmid; g.ldx = D; g.gamma = W(h, b + "norm2.weight"); g.beta = W(h, b + "norm2.bias"); g.eps = c.ln_eps;
            g.W = h->packed_slab.at(b + "mlp.fc1.weight"); g.bias = W(h, b + "mlp.fc1.bias");
            g.M = L.M; g.N = F; g.epi = EPI_GELU;
            g.out_bf16 = HB; g.out_plane = L.f_plane; g.ldo = F;
            g.a_out = A2; g.a_plane = L.a_plane;
            g.aux_out = HPRE; g.aux_plane = L.f_plane;
            DSEG_TRY(launch_gemm_ln(g, D, P, s));
        } else {
        DSEG_TRY(launch_layernorm(Xmid, W(h, b + "norm2.weight"), W(h, b + "norm2.bias"), c.ln_eps, L.M, D, A2, L.a_plane, P,
                                  nullptr, 0, L.ntok, s));
        {
            const PackedLinear& pk = h->packed.at(b + "mlp.fc1.weight");
            GemmParams g = {};
            g.A = A2; g.a_plane = L.a_plane; g.lda = D; g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.M; g.N = F; g.K = D; g.planes = P; g.epi = EPI_GELU; g.bias = W(h, b + "mlp.fc1.bias");
            g.out_bf16 = HB; g.out_plane = L.f_plane; g.ldo = F; g.aux_out = HPRE; g.aux_plane = L.f_plane;
            DSEG_TRY(launch_gemm(g, s));
        }
        }
        {
            const PackedLinear& pk = h->packed.at(b + "mlp.fc2.weight");
            GemmParams g = {};
            g.A = HB; g.a_plane = L.f_plane; g.lda = F; g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.M; g.N = D; g.K = F; g.planes = P; g.epi = EPI_RESID; g.bias = W(h, b + "mlp.fc2.bias");
            g.resid = Xmid; g.out_f32 = Xout; g.ldo_f32 = D;
            DSEG_TRY(launch_gemm(g, s));
        }
    }
    float* Xfin = F32(L.Xfin);
    bf16_t *FEAT = B16(L.FEAT), *H1 = B16(L.H1), *H2 = B16(L.H2);
    float* LOGP = F32(L.LOGP);
    DSEG_TRY(launch_layernorm(Xfin, W(h, "dino.norm.weight"), W(h, "dino.norm.bias"), c.ln_eps, L.M, D, FEAT, L.feat_plane, HP,
                              nullptr, 1, L.ntok, s));
    if (mlp_head) {
        {
            const PackedLinear& pk = h->packed.at("clf.layer_1.weight");
            GemmParams g = {};
            g.A = FEAT; g.a_plane = L.feat_plane; g.lda = D; g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.Mp; g.N = 256; g.K = D; g.planes = HP; g.epi = EPI_RELU; g.bias = pk.bias_pad;
            g.out_bf16 = H1; g.out_plane = L.h1_plane; g.ldo = 256;
            DSEG_TRY(launch_gemm(g, s));
        }
        {
            const PackedLinear& pk = h->packed.at("clf.layer_2.weight");
            GemmParams g = {};
            g.A = H1; g.a_plane = L.h1_plane; g.lda = 256; g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.Mp; g.N = 128; g.K = 256; g.planes = HP; g.epi = EPI_RELU; g.bias = pk.bias_pad;
            g.out_bf16 = H2; g.out_plane = L.h2_plane; g.ldo = 128;
            DSEG_TRY(launch_gemm(g, s));
        }
        DSEG_TRY(launch_head_final(H2, L.h2_plane, 128, L.Mp, 100, W(h, "clf.layer_3.weight"), W(h, "clf.layer_3.bias"), C, LOGP,
                                   nullptr, s));
    } else {
        DSEG_TRY(launch_head_final(FEAT, L.feat_plane, D, L.Mp, D, W(h, "clf.layer_1.weight"), W(h, "clf.layer_1.bias"), C, LOGP,
                                   nullptr, s));
    }
    if (logp_out) DSEG_CHECK_HIP(hipMemcpyAsync(logp_out, LOGP, (size_t)L.Mp * C * 4, hipMemcpyDeviceToDevice, s));
    h->tr_B = B;
    h->tr_r = r;
    return 0;
}

// Backward of the last train_forward_impl.  Exactly one of (labels, dlogp) is given:
//   labels : loss = F.nll_loss(logp, labels) (mean over the rows whose label is not -100) -> *loss_out, then backward of it
//   dlogp  : fp32 [B*n, C] upstream gradient d L / d logp (torch.autograd path)
// Gradients are written (not accumulated) into the buffers bound with dinoseg_bind_grad.
static int train_backward_impl(dinoseg_handle* h, const int64_t* labels, const float* dlogp, float* loss_out, hipStream_t s) {
    if (!h || (labels == nullptr) == (dlogp == nullptr) || (labels && !loss_out)) {
        dinoseg_set_error("dinoseg_backward: needs exactly one of labels (+ loss_out) and dlogp");
        return -1;
    }
    if (h->tr_B <= 0 || !h->tws) {
        dinoseg_set_error("dinoseg_backward: no saved forward (call dinoseg_train_forward first)");
        return -3;
    }
    if (!h->weights_ready) {
        dinoseg_set_error("dinoseg_backward: weights were re-bound after the forward; run the forward again");
        return -3;
    }
    const int B = h->tr_B, r = h->tr_r;
    const dinoseg_config& c = h->cfg;
    const int D = c.embed_dim, F = D * c.mlp_ratio, P = h->planes, HP = head_planes(), H = c.num_heads, C = c.n_classes;
    const int NB = c.n_blocks;
    const bool mlp_head = c.head_kind == DINOSEG_HEAD_MLP;
    const TrainLayout L = make_train_layout(h, B, r);
    char* ws = h->tws;
    auto F32 = [&](size_t o) { return reinterpret_cast<float*>(ws + o); };
    auto B16 = [&](size_t o) { return reinterpret_cast<bf16_t*>(ws + o); };

    // ---- transposed packed weights for dX = dY . W  (weights change every optimiser step: repack)
    const std::vector<TLin> tspecs = transposed_specs(h);
    std::map<std::string, TW> tw;
    {
        size_t total = 0;
        for (const TLin& t : tspecs) total += align_up((size_t)t.planes * t.k_pad * t.n_pad * 2, 256);
        if (total > h->twbuf_bytes) {
            if (h->twbuf) {
                DSEG_CHECK_HIP(hipStreamSynchronize(s));
                DSEG_CHECK_HIP(hipFree(h->twbuf));
            }
            h->twbuf = nullptr;
            DSEG_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&h->twbuf), total));
            h->twbuf_bytes = total;
        }
        size_t off = 0;
        std::vector<dseg::PackJob> jobs;
        for (const TLin& t : tspecs) {
            TW e;
            e.w = reinterpret_cast<bf16_t*>(h->twbuf + off);
            e.plane = (long)t.k_pad * t.n_pad;
            off += align_up((size_t)t.planes * t.k_pad * t.n_pad * 2, 256);
            // W [N][K] fp32 -> W^T planes [k_pad][n_pad]: "rows" of the source are N, transposed destination rows are K
            jobs.push_back({W(h, t.wname), e.w, e.plane, t.N, t.K, t.n_pad, t.k_pad, t.planes, 1});
            tw[t.wname] = e;
        }
        DSEG_TRY(launch_multi_pack(jobs.data(), (int)jobs.size(), s));
    }

    auto grad = [&](const std::string& name) -> float* {
        auto it = h->grads.find(name);
        return it == h->grads.end() ? nullptr : it->second;
    };
    auto numel = [&](const std::string& name) {
        size_t n = 1;
        for (int64_t d : h->expected.at(name)) n *= (size_t)d;
        return n;
    };
    bool backbone = false;
    {
        std::vector<float*> zp;
        std::vector<long> zn;
        for (auto& kv : h->grads) {
            zp.push_back(kv.second);
            zn.push_back((long)numel(kv.first));
            if (kv.first.rfind("dino.", 0) == 0) backbone = true;
        }
        if (!zp.empty()) DSEG_TRY(launch_multi_zero((int)zp.size(), zp.data(), zn.data(), s));      // one launch instead of ~50 memset nodes
    }
    // option deterministic: the launchers below write per-block partial sums here and add them in a fixed order (train.hip,
    // gemm_tn.hip) instead of fp32 atomics; cleared on every way out
    // The scratch pointer is process-wide state read by the launchers: a second backward entered while it is set (another host thread
    // stepping another handle) would write its partial sums into THIS handle's workspace -- refused instead.
    struct DetGuard {
        bool mine = false;
        ~DetGuard() { if (mine) det_scratch() = DetScratch{nullptr, 0, {nullptr, nullptr}, 0}; }
    } det_guard;
    if (options().deterministic) {
        if (det_scratch().ptr != nullptr) {
            dinoseg_set_error("dinoseg_backward: option deterministic allows one backward at a time per process (another one is being queued)");
            return -1;
        }
        det_scratch() = DetScratch{F32(L.DET), DET_FLOATS, {F32(L.DET) + DET_FLOATS, F32(L.DET) + DET_FLOATS + DET_TN_FLOATS}, DET_TN_FLOATS};
        det_guard.mine = true;
    }
    const hipStream_t main_stream = s;
    float* Xfin = F32(L.Xfin);
    bf16_t *FEAT = B16(L.FEAT), *H1 = B16(L.H1), *H2 = B16(L.H2), *DZ = B16(L.DZ);
    float* LOGP = F32(L.LOGP);
    bf16_t* PATCH = B16(L.PATCH);

    // =============================================================== backward
    // stage events: a side stream can start reducing a gradient bucket while the rest of backward still runs
    h->stage_done = 0;
    auto stage_mark_on = [&](int stage, hipStream_t on) -> int {
        while ((int)h->stage_ev.size() <= stage) {
            hipEvent_t ev;
            DSEG_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            h->stage_ev.push_back(ev);
        }
        DSEG_CHECK_HIP(hipEventRecord(h->stage_ev[stage], on));
        h->stage_done = stage + 1;
        return 0;
    };
    auto stage_mark = [&](int stage) -> int { return stage_mark_on(stage, s); };
    // Side stream for the blocks' weight gradients (option train_streams = 2).  dW = dY^T . X reads what the input-gradient chain
    // has already produced and feeds nothing but the optimiser, so it runs beside that chain on the handle's internal stream:
    // the chain's tail rounds and memory-bound kernels (LayerNorm backward, the attention prep) leave CUs idle that the
    // weight-gradient tiles fill.  side_begin(): the side stream waits for everything queued on s so far; side_end() returns an
    // event the caller's stream waits on (side_wait) before it overwrites an operand the side kernels read, and before every
    // gradient-stage event.  Fork and join are events only: the call stays stream-ordered for the caller and capturable.
    // (deterministic mode: the side stream's only partial sums are gemm_tn's bias sums: they have their own part of the scratch area)
    const bool side = options().train_streams >= 2 && D % 128 == 0 && F % 128 == 0;   // (narrow layers go through T1 / T2: one stream)
    hipStream_t ws_ = s;
    size_t bw_i = 0;
    auto bw_event = [&](hipEvent_t* out) -> int {
        if (bw_i == h->bw_ev.size()) {
            hipEvent_t ev;
            DSEG_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            h->bw_ev.push_back(ev);
        }
        *out = h->bw_ev[bw_i++];
        return 0;
    };
    if (side) {
        DSEG_TRY(ensure_aux_stream(h));
        ws_ = h->aux_stream;
    }
    auto side_begin = [&]() -> int {
        if (!side) return 0;
        hipEvent_t ev;
        DSEG_TRY(bw_event(&ev));
        DSEG_CHECK_HIP(hipEventRecord(ev, s));
        DSEG_CHECK_HIP(hipStreamWaitEvent(ws_, ev, 0));
        return 0;
    };
    auto side_end = [&](hipEvent_t* done) -> int {
        *done = nullptr;
        if (!side) return 0;
        DSEG_TRY(bw_event(done));
        DSEG_CHECK_HIP(hipEventRecord(*done, ws_));
        return 0;
    };
    auto side_wait = [&](hipEvent_t& done) -> int {
        if (done) DSEG_CHECK_HIP(hipStreamWaitEvent(s, done, 0));
        done = nullptr;
        return 0;
    };
    float* sink = F32(L.SINK);
    bf16_t *T1 = B16(L.T1), *T2 = B16(L.T2);

    // dX[M_, Kin] = dY[M_, Ncols] . W   with optional activation-derivative epilogue
    auto dgrad = [&](const bf16_t* dY, long dy_plane, int ld, int M_, int n_contract, const TW& wt, int k_out, int planes, int epi,
                     float* out_f32, bf16_t* out_bf16, long out_plane, const bf16_t* aux, long aux_plane) -> int {
        GemmParams g = {};
        g.A = dY; g.a_plane = dy_plane; g.lda = ld; g.W = wt.w; g.w_plane = wt.plane;
        g.M = M_; g.N = k_out; g.K = n_contract; g.planes = planes; g.epi = epi;
        g.out_f32 = out_f32; g.ldo_f32 = k_out;
        g.out_bf16 = out_bf16; g.out_plane = out_plane; g.ldo = k_out; g.aux_in = aux; g.aux_plane = aux_plane;
        return launch_gemm_small(g, s);
    };
    // dW[n_rows, k_cols] += dY^T . X  from transposed planes T_dy [n_pad][m_pad], T_x [k_pad128][m_pad]
    auto wgrad = [&](const bf16_t* Tdy, const bf16_t* Tx, long tplane, int m_pad, int n_rows, int k_pad128, int k_cols, int planes,
                     float* dW) -> int {
        if (!dW) return 0;
        GemmParams g = {};
        g.A = Tdy; g.a_plane = tplane; g.lda = m_pad; g.W = Tx; g.w_plane = tplane;
        g.M = n_rows; g.N = k_pad128; g.K = m_pad; g.planes = planes;
        const int row_tiles = (n_rows + 127) / 128, tiles = row_tiles * (k_pad128 / 128), nk = m_pad / 64;
        int ks = splitk_budget() / tiles;
        if (ks > nk / 2) ks = nk / 2;
        if (ks < 1) ks = 1;
        if (ks == 1) {
            g.epi = EPI_ATOMIC;
            g.out_f32 = dW; g.ldo_f32 = k_cols; g.n_valid = k_cols;
            g.ksplit = 1;
            return launch_gemm_small(g, s);
        }
        // slices of the batch dimension write partial tiles (plain stores), one pass sums them into dW
        const int per = (nk + ks - 1) / ks, used = (nk + per - 1) / per;       // slices that own k-steps (gemm.hip)
        float* part = F32(L.SPLITK);
        g.epi = EPI_PLAIN;
        g.out_f32 = part; g.ldo_f32 = k_pad128; g.ksplit = ks;
        g.split_stride = (long)row_tiles * 128 * k_pad128;
        DSEG_TRY(launch_gemm_small(g, s));
        return launch_splitk_reduce(part, used, g.split_stride, n_rows, k_pad128, dW, k_cols, k_cols, s);
    };
    auto pad128 = [](int v) { return (v + 127) / 128 * 128; };
    // weight gradient straight from the row-major dY and layer-input planes (gemm_tn.hip): no operand transposes
    // dbias (optional): the layer's bias gradient = column sums of Y, taken inside the weight-gradient kernel; when that kernel
    // does not run (frozen weight, narrow layer) a pack pass over Y produces them
    // k_pad (optional): the stored width of X when that is a multiple of 128 and k_cols is not (the classifier's zero-padded hidden
    // activations): the kernel multiplies all k_pad columns, the reduce writes the first k_cols
    auto wgrad_tn = [&](const bf16_t* Y, long y_plane, int ldy, const bf16_t* X, long x_plane, int ldx, int m_rows, int n_rows,
                        int k_cols, int planes, float* dW, float* dbias, hipStream_t s, int k_pad = 0) -> int {
        if (k_pad > 0 && k_pad % 128 == 0 && k_pad <= ldx) {
            if (!dW && !dbias) return 0;
            TnParams g = {};
            g.Y = Y; g.y_plane = y_plane; g.ldy = ldy; g.X = X; g.x_plane = x_plane; g.ldx = ldx;
            g.M = m_rows; g.N = n_rows; g.Kc = k_pad; g.planes = planes;
            const int row_tiles = (n_rows + 127) / 128, tiles = row_tiles * (k_pad / 128), nchunks = (m_rows + 31) / 32;
            int ks = splitk_budget() / tiles;
            if (ks > nchunks / 4) ks = nchunks / 4;
            if (ks >= 8 && !(dseg::options().route_ab & 4)) ks &= ~7;      // a multiple of 8: the kernel's XCD-aware form (gemm_tn.hip)
            if (ks < 1) ks = 1;
            const int per = (nchunks + ks - 1) / ks, used = (nchunks + per - 1) / per;
            g.part = F32(L.SPLITK); g.ld_part = k_pad; g.split_stride = (long)row_tiles * 128 * k_pad; g.ksplit = ks;
            g.colsum = dbias;
            g.det_region = s != main_stream;
            DSEG_TRY(launch_gemm_tn(g, s));
            if (!dW) return 0;
            return launch_splitk_reduce(g.part, used, g.split_stride, n_rows, k_pad, dW, k_cols, k_cols, s);
        }
        if (dbias && (!dW || k_cols % 128 != 0))
            DSEG_TRY(launch_transpose_planes(nullptr, Y, y_plane, ldy, m_rows, n_rows, nullptr, 0, pad128(n_rows), L.Mpad, nullptr, 0, 0,
                                             dbias, planes, 0, 0, s, s != main_stream));
        if (!dW) return 0;
        if (k_cols % 128 != 0) {        // narrow layers (embed_dim not a multiple of 128): transposed operands + the NT kernel
            bf16_t* T1f = B16(L.T1);
            bf16_t* T2f = B16(L.T2);
            DSEG_TRY(launch_transpose_planes(nullptr, Y, y_plane, ldy, m_rows, n_rows, T1f, L.t_plane, pad128(n_rows), L.Mpad, nullptr, 0, 0,
                                             nullptr, planes, 0, 0, s));
            DSEG_TRY(launch_transpose_planes(nullptr, X, x_plane, ldx, m_rows, k_cols, T2f, L.t_plane, pad128(k_cols), L.Mpad, nullptr, 0, 0,
                                             nullptr, planes, 0, 0, s));
            return wgrad(T1f, T2f, L.t_plane, L.Mpad, n_rows, pad128(k_cols), k_cols, planes, dW);
        }
        TnParams g = {};
        g.Y = Y; g.y_plane = y_plane; g.ldy = ldy; g.X = X; g.x_plane = x_plane; g.ldx = ldx;
        g.M = m_rows; g.N = n_rows; g.Kc = k_cols; g.planes = planes;
        const int row_tiles = (n_rows + 127) / 128, tiles = row_tiles * (k_cols / 128), nchunks = (m_rows + 31) / 32;
        int ks = splitk_budget() / tiles;
        if (ks > nchunks / 4) ks = nchunks / 4;
        if (ks >= 8 && !(dseg::options().route_ab & 4)) ks &= ~7;      // a multiple of 8: the kernel's XCD-aware form (gemm_tn.hip)
        if (ks < 1) ks = 1;
        const int per = (nchunks + ks - 1) / ks, used = (nchunks + per - 1) / per;
        g.part = F32(L.SPLITK); g.ld_part = k_cols; g.split_stride = (long)row_tiles * 128 * k_cols; g.ksplit = ks;
        g.colsum = dbias;
        g.det_region = s != main_stream;
        DSEG_TRY(launch_gemm_tn(g, s));
        return launch_splitk_reduce(g.part, used, g.split_stride, n_rows, k_cols, dW, k_cols, k_cols, s);
    };

    // ---- loss and d logits (pl_torch_modules.py:264-265)
    DSEG_TRY(launch_nll_loss_grad(LOGP, labels, dlogp, L.Mp, C, F32(L.ACC), h->bad_label_flag, loss_out, DZ,
                                  L.dz_plane, 64, s));
    const long tpl = L.t_plane;
    float* dX = F32(L.dX);
    float* dA = F32(L.dA);
    bf16_t* G = B16(L.G);
    if (mlp_head) {
        // layer_3: z = h2 W3^T + b3      (weight and bias gradients straight from the row-major planes: gemm_tn.hip; h2 / h1 are stored
        // 128 / 256 wide, zero beyond their 100 / 200 columns)
        DSEG_TRY(wgrad_tn(DZ, L.dz_plane, 64, H2, L.h2_plane, 128, L.Mp, C, 100, HP, grad("clf.layer_3.weight"), grad("clf.layer_3.bias"), s, 128));
        bf16_t* dH2 = G;                         // [HP][Mp][128]
        const long dh2_plane = (long)L.Mp * 128;
        DSEG_TRY(dgrad(DZ, L.dz_plane, 64, L.Mp, 64, tw.at("clf.layer_3.weight"), 128, HP, EPI_DRELU, nullptr, dH2, dh2_plane, H2, L.h2_plane));
        // layer_2
        DSEG_TRY(wgrad_tn(dH2, dh2_plane, 128, H1, L.h1_plane, 256, L.Mp, 100, 200, HP, grad("clf.layer_2.weight"), grad("clf.layer_2.bias"), s, 256));
        bf16_t* dH1 = B16(L.dCTX);               // [HP][Mp][256] fits: Mp*256 <= M*D
        const long dh1_plane = (long)L.Mp * 256;
        DSEG_TRY(dgrad(dH2, dh2_plane, 128, L.Mp, 128, tw.at("clf.layer_2.weight"), 256, HP, EPI_DRELU, nullptr, dH1, dh1_plane, H1, L.h1_plane));
        // layer_1
        if (D % 128 == 0) {
            DSEG_TRY(wgrad_tn(dH1, dh1_plane, 256, FEAT, L.feat_plane, D, L.Mp, 200, D, HP, grad("clf.layer_1.weight"), grad("clf.layer_1.bias"), s, D));
        } else {
            DSEG_TRY(launch_transpose_planes(nullptr, dH1, dh1_plane, 256, L.Mp, 200, T1, tpl, 256, L.Mppad, nullptr, 0, 0,
                                             grad("clf.layer_1.bias"), HP, 0, 0, s));
            DSEG_TRY(launch_transpose_planes(nullptr, FEAT, L.feat_plane, D, L.Mp, D, T2, tpl, pad128(D), L.Mppad, nullptr, 0, 0, nullptr, HP, 0, 0, s));
            DSEG_TRY(wgrad(T1, T2, tpl, L.Mppad, 200, pad128(D), D, HP, grad("clf.layer_1.weight")));
        }
        if (backbone)
            DSEG_TRY(dgrad(dH1, dh1_plane, 256, L.Mp, 256, tw.at("clf.layer_1.weight"), D, HP, EPI_PLAIN, dA, nullptr, 0, nullptr, 0));
    } else {
        if (D % 128 == 0) {
            DSEG_TRY(wgrad_tn(DZ, L.dz_plane, 64, FEAT, L.feat_plane, D, L.Mp, C, D, HP, grad("clf.layer_1.weight"), grad("clf.layer_1.bias"), s, D));
        } else {
            DSEG_TRY(launch_transpose_planes(nullptr, DZ, L.dz_plane, 64, L.Mp, C, T1, tpl, 128, L.Mppad, nullptr, 0, 0,
                                             grad("clf.layer_1.bias"), HP, 0, 0, s));
            DSEG_TRY(launch_transpose_planes(nullptr, FEAT, L.feat_plane, D, L.Mp, D, T2, tpl, pad128(D), L.Mppad, nullptr, 0, 0, nullptr, HP, 0, 0, s));
            DSEG_TRY(wgrad(T1, T2, tpl, L.Mppad, C, pad128(D), D, HP, grad("clf.layer_1.weight")));
        }
        if (backbone)
            DSEG_TRY(dgrad(DZ, L.dz_plane, 64, L.Mp, 64, tw.at("clf.layer_1.weight"), D, HP, EPI_PLAIN, dA, nullptr, 0, nullptr, 0));
    }
    DSEG_TRY(stage_mark(0));
    if (!backbone) return 0;      // frozen backbone (freeze_bb, pl_torch_modules.py:434-436): only the head trains

    // ---- final norm (CLS rows get no gradient from the head)
    auto gsink = [&](const std::string& name) { float* g = grad(name); return g ? g : sink; };
    // (every LayerNorm backward also leaves its dX rows as bf16 planes dXp and their column sums = the bias gradient of the
    //  layer the walk reaches next: mlp.fc2 of the last block here)
    bf16_t* dXp = B16(L.dXp);
    DSEG_TRY(launch_layernorm_bwd(dA, Xfin, W(h, "dino.norm.weight"), c.ln_eps, L.M, D, dX, 0, gsink("dino.norm.weight"),
                                  gsink("dino.norm.bias"), 1, L.ntok, s, dXp, L.a_plane, P,
                                  NB > 0 ? grad("dino.blocks." + std::to_string(NB - 1) + ".mlp.fc2.bias") : nullptr));

    bf16_t* dCTX = B16(L.dCTX);
    hipEvent_t w_fc2 = nullptr, w_fc1 = nullptr, w_proj = nullptr, w_qkv = nullptr;
    for (int l = NB - 1; l >= 0; --l) {
        const std::string b = "dino.blocks." + std::to_string(l) + ".";
        const size_t o = l * L.blk_stride;
        bf16_t *A1 = B16(L.A1 + o), *Q = B16(L.Q + o), *Kb = B16(L.K + o), *V = B16(L.V + o), *CTX = B16(L.CTX + o);
        bf16_t *A2 = B16(L.A2 + o), *HPRE = B16(L.HPRE + o), *HB = B16(L.HB + o);
        // ---- mlp.fc2 : X_out = X_mid + H W2^T + b
        // (dXp = bf16 planes of dX and the fc2 bias gradient were left by the LayerNorm backward that produced dX; the weight
        //  gradient reads dXp and HB row-major)
        DSEG_TRY(side_begin());
        DSEG_TRY(wgrad_tn(dXp, L.a_plane, D, HB, L.f_plane, F, L.M, D, F, P, grad(b + "mlp.fc2.weight"), nullptr, ws_));
        DSEG_TRY(side_end(&w_fc2));
        // dHpre = (dX . W2) * gelu'(Hpre)      (writes G: the previous block's qkv weight gradient reads it)
        DSEG_TRY(side_wait(w_qkv));
        DSEG_TRY(dgrad(dXp, L.a_plane, D, L.M, D, tw.at(b + "mlp.fc2.weight"), F, P, EPI_DGELU, nullptr, G, (long)L.M * F, HPRE, L.f_plane));
        // ---- mlp.fc1 : Hpre = A2 W1^T + b
        DSEG_TRY(side_begin());
        DSEG_TRY(wgrad_tn(G, (long)L.M * F, F, A2, L.a_plane, D, L.M, F, D, P, grad(b + "mlp.fc1.weight"), grad(b + "mlp.fc1.bias"), ws_));
        DSEG_TRY(side_end(&w_fc1));
        DSEG_TRY(dgrad(G, (long)L.M * F, F, L.M, F, tw.at(b + "mlp.fc1.weight"), D, P, EPI_PLAIN, dA, nullptr, 0, nullptr, 0));
        // ---- norm2 (input X_mid); the residual branch keeps dX      (rewrites dXp: the fc2 weight gradient reads it)
        DSEG_TRY(side_wait(w_fc2));
        DSEG_TRY(launch_layernorm_bwd(dA, F32(L.Xmid + o), W(h, b + "norm2.weight"), c.ln_eps, L.M, D, dX, 1, gsink(b + "norm2.weight"),
                                      gsink(b + "norm2.bias"), 0, L.ntok, s, dXp, L.a_plane, P, grad(b + "attn.proj.bias")));
        // ---- attn.proj : X_mid = X_in + ctx Wp^T + b
        DSEG_TRY(side_begin());
        DSEG_TRY(wgrad_tn(dXp, L.a_plane, D, CTX, L.a_plane, D, L.M, D, D, P, grad(b + "attn.proj.weight"), nullptr, ws_));
        DSEG_TRY(side_end(&w_proj));
        DSEG_TRY(dgrad(dXp, L.a_plane, D, L.M, D, tw.at(b + "attn.proj.weight"), D, P, EPI_BF16, nullptr, dCTX, L.a_plane, nullptr, 0));
        // ---- attention      (writes G: the fc1 weight gradient reads it)
        DSEG_TRY(side_wait(w_fc1));
        {
            AttnBwdParams a = {};
            a.q = Q; a.k = Kb; a.v = V; a.qkv_plane = L.qkv_plane;
            a.dO = dCTX; a.O = CTX; a.dO_plane = L.a_plane; a.lse = F32(L.LSE + o);
            a.neg_lse = F32(L.NLSE); a.neg_delta = F32(L.NDEL);
            a.dqkv = G; a.dqkv_plane = (long)L.M * 3 * D;
            a.B = B; a.heads = H; a.ntok = L.ntok; a.npad = L.npad; a.planes = P;
            DSEG_PROF(DINOSEG_PROF_ATTN_BWD, DSEG_TRY(launch_attention_bwd(a, s)));
        }
        // ---- attn.qkv : qkv = A1 Wqkv^T + b
        DSEG_TRY(side_begin());
        DSEG_TRY(wgrad_tn(G, (long)L.M * 3 * D, 3 * D, A1, L.a_plane, D, L.M, 3 * D, D, P, grad(b + "attn.qkv.weight"),
                          grad(b + "attn.qkv.bias"), ws_));
        DSEG_TRY(side_end(&w_qkv));
        DSEG_TRY(dgrad(G, (long)L.M * 3 * D, 3 * D, L.M, 3 * D, tw.at(b + "attn.qkv.weight"), D, P, EPI_PLAIN, dA, nullptr, 0, nullptr, 0));
        // ---- norm1 (input X_in)      (rewrites dXp: the proj weight gradient reads it)
        // (by-products for mlp.fc2 of block l-1; the embedding step after block 0 packs dX itself: it drops the CLS rows)
        DSEG_TRY(side_wait(w_proj));
        DSEG_TRY(launch_layernorm_bwd(dA, F32(L.Xin + o), W(h, b + "norm1.weight"), c.ln_eps, L.M, D, dX, 1, gsink(b + "norm1.weight"),
                                      gsink(b + "norm1.bias"), 0, L.ntok, s, l > 0 ? dXp : nullptr, L.a_plane, P,
                                      l > 0 ? grad("dino.blocks." + std::to_string(l - 1) + ".mlp.fc2.bias") : nullptr));
        // this block's gradients are complete once the side stream has finished its qkv weight gradient; the stage event is
        // recorded on the side stream (it has waited for everything the block queued on s up to the qkv weight gradient -- the
        // LayerNorm backward above is covered by the extra fork), so the caller's stream does not stall here
        if (side) {
            DSEG_TRY(side_begin());
            DSEG_TRY(stage_mark_on(1 + (NB - 1 - l), ws_));
            continue;
        }
        DSEG_TRY(stage_mark(1 + (NB - 1 - l)));
    }

    // join: the caller's stream continues (and the call returns) behind everything the side stream did; the embedding step
    // below reuses the split-K workspace
    DSEG_TRY(side_wait(w_qkv));
    // ---- embeddings: tokens = [cls ; conv(patches)] + pos   (vision_transformer.py:224-235)
    float* dpos = F32(L.DPOS);
    DSEG_TRY(launch_batch_sum_rows(dX, B, L.ntok, D, dpos, s));
    if (grad("dino.cls_token"))
        DSEG_CHECK_HIP(hipMemcpyAsync(grad("dino.cls_token"), dpos, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
    // (scratch: the T2 transpose buffer, idle until the patch-embed gradient below; [pos_grid][r/8][D] floats fit its >= Mppad x 256 bf16)
    if (grad("dino.pos_embed")) {
        // (T2 is allocated with two planes whatever the precision: make_train_layout)
        if ((size_t)c.pos_grid * (r / 8) * D * sizeof(float) > (size_t)2 * L.t_plane * sizeof(bf16_t)) {
            dinoseg_set_error("dinoseg_backward: pos-embed scratch does not fit (pos_grid %d, grid %d)", c.pos_grid, r / 8);
            return -1;
        }
        DSEG_TRY(launch_pos_resample_bwd(dpos, c.pos_grid, D, r / 8, grad("dino.pos_embed"), reinterpret_cast<float*>(T2), s));
    }
    DSEG_TRY(launch_transpose_planes(dX, nullptr, 0, D, L.Mp, D, T1, tpl, pad128(D), L.Mppad, nullptr, 0, 0,
                                     grad("dino.patch_embed.proj.bias"), P, 1, L.ntok, s));
    if (grad("dino.patch_embed.proj.weight")) {
        DSEG_TRY(launch_transpose_planes(nullptr, PATCH, L.patch_plane, 192, L.Mp, 192, T2, tpl, 256, L.Mppad, nullptr, 0, 0, nullptr, P, 0, 0, s));
        DSEG_TRY(wgrad(T1, T2, tpl, L.Mppad, D, 256, 192, P, grad("dino.patch_embed.proj.weight")));
    }
    return stage_mark(NB + 1);
}

extern "C" int dinoseg_train_forward(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r, float* logp_out,
                                     void* stream) {
    DeviceGuard guard(h);
    return train_forward_impl(h, x, x_kind, B, r, logp_out, reinterpret_cast<hipStream_t>(stream));
}

// The backward forks weight-gradient kernels onto the handle's side stream (option train_streams = 2) and joins them before it
// returns.  An error in between returns early: join here too, so that the caller's stream never runs ahead of side-stream kernels
// that still read / write the gradient buffers, the split-K workspace or the activations (as dinoseg_forward does for its halves).
static int backward_joined(dinoseg_handle* h, const int64_t* labels, const float* dlogp, float* loss_out, hipStream_t s) {
    const int rc = train_backward_impl(h, labels, dlogp, loss_out, s);
    if (rc != 0 && h && h->aux_stream && h->ev_join) {
        (void)hipEventRecord(h->ev_join, h->aux_stream);
        (void)hipStreamWaitEvent(s, h->ev_join, 0);
    }
    return rc;
}

extern "C" int dinoseg_backward(dinoseg_handle* h, const float* dlogp, void* stream) {
    DeviceGuard guard(h);
    return backward_joined(h, nullptr, dlogp, nullptr, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_train_step(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r,
                                  const int64_t* labels, float* loss_out, float* logp_out, void* stream) {
    if (!labels || !loss_out) {
        dinoseg_set_error("dinoseg_train_step: bad argument");
        return -1;
    }
    DeviceGuard guard(h);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DSEG_TRY(train_forward_impl(h, x, x_kind, B, r, logp_out, s));
    return backward_joined(h, labels, nullptr, loss_out, s);
}

extern "C" int dinoseg_grad_stages(const dinoseg_handle* h) { return h ? h->cfg.n_blocks + 2 : -1; }

extern "C" int dinoseg_stream_wait_grad_stage(dinoseg_handle* h, int32_t stage, void* stream) {
    if (!h || stage < 0 || stage >= h->cfg.n_blocks + 2) {
        dinoseg_set_error("dinoseg_stream_wait_grad_stage: stage out of range");
        return -1;
    }
    if (stage >= h->stage_done) {
        dinoseg_set_error("dinoseg_stream_wait_grad_stage: the last backward recorded %d stage(s); stage %d was not reached "
                          "(frozen backbone?)", h->stage_done, stage);
        return -3;
    }
    DeviceGuard guard(h);
    DSEG_CHECK_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), h->stage_ev[stage], 0));
    return 0;
}

extern "C" int dinoseg_train_status(dinoseg_handle* h, int32_t* bad_labels, void* stream) {
    if (!h || !bad_labels) {
        dinoseg_set_error("dinoseg_train_status: null argument");
        return -1;
    }
    *bad_labels = 0;
    if (!h->bad_label_flag) return 0;
    DeviceGuard guard(h);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int flag = 0;
    DSEG_CHECK_HIP(hipMemcpyAsync(&flag, h->bad_label_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    DSEG_CHECK_HIP(hipStreamSynchronize(s));
    if (flag) DSEG_CHECK_HIP(hipMemsetAsync(h->bad_label_flag, 0, sizeof(int), s));
    *bad_labels = flag;
    return 0;
}
