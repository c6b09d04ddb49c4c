// C-ABI of libdinoseg_hip.so (see include/dinoseg.h): handle, weight binding/packing, workspace, and the
// forward orchestration of the DINOSeg hot path on one MI355X.  Host code only; kernels live in
// gemm.hip / attention.hip / elementwise.hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "handle.h"

using namespace dseg;

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";

extern "C" void dinoseg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* dinoseg_last_error(void) { return g_err; }
extern "C" int dinoseg_version(void) { return 300; }      // 3.0: fused MLP kernel, two streams by default; the measurement kernel moved to libdinoseg_tools.so

int device_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        dinoseg_set_error("hipGetDevice failed");
        return -1;
    }
    if (dev >= 0 && dev < 64 && cache[dev].load() > 0) return cache[dev].load();
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        dinoseg_set_error("hipDeviceGetAttribute(MultiprocessorCount) failed on device %d", dev);
        return -1;
    }
    if (dev >= 0 && dev < 64) cache[dev].store(n);
    return n;
}



static void add_expected(dinoseg_handle* h) {
    const dinoseg_config& c = h->cfg;
    const int64_t D = c.embed_dim, F = (int64_t)c.embed_dim * c.mlp_ratio, C = c.n_classes, p = c.patch;
    auto& e = h->expected;
    e["dino.cls_token"] = {1, 1, D};
    e["dino.pos_embed"] = {1, (int64_t)c.pos_grid * c.pos_grid + 1, D};
    e["dino.patch_embed.proj.weight"] = {D, 3, p, p};
    e["dino.patch_embed.proj.bias"] = {D};
    for (int i = 0; i < c.n_blocks; ++i) {
        const std::string b = "dino.blocks." + std::to_string(i) + ".";
        e[b + "norm1.weight"] = {D};
        e[b + "norm1.bias"] = {D};
        e[b + "attn.qkv.weight"] = {3 * D, D};
        e[b + "attn.qkv.bias"] = {3 * D};
        e[b + "attn.proj.weight"] = {D, D};
        e[b + "attn.proj.bias"] = {D};
        e[b + "norm2.weight"] = {D};
        e[b + "norm2.bias"] = {D};
        e[b + "mlp.fc1.weight"] = {F, D};
        e[b + "mlp.fc1.bias"] = {F};
        e[b + "mlp.fc2.weight"] = {D, F};
        e[b + "mlp.fc2.bias"] = {D};
    }
    e["dino.norm.weight"] = {D};
    e["dino.norm.bias"] = {D};
    if (c.head_kind == DINOSEG_HEAD_MLP) {
        e["clf.layer_1.weight"] = {200, D};
        e["clf.layer_1.bias"] = {200};
        e["clf.layer_2.weight"] = {100, 200};
        e["clf.layer_2.bias"] = {100};
        e["clf.layer_3.weight"] = {C, 100};
        e["clf.layer_3.bias"] = {C};
    } else {
        e["clf.layer_1.weight"] = {C, D};
        e["clf.layer_1.bias"] = {C};
    }
}

extern "C" int dinoseg_create(const dinoseg_config* cfg, dinoseg_handle** out) {
    if (!cfg || !out) {
        dinoseg_set_error("dinoseg_create: null argument");
        return -1;
    }
    if (cfg->embed_dim % 128 != 0 || cfg->embed_dim > 1024 || cfg->num_heads * 64 != cfg->embed_dim || cfg->patch != 8 ||
        cfg->n_blocks < 0 || cfg->n_classes < 1 || cfg->n_classes > 32 || cfg->mlp_ratio < 1 || cfg->pos_grid < 1 ||
        (cfg->precision != DINOSEG_BF16 && cfg->precision != DINOSEG_BF16X3 && cfg->precision != DINOSEG_FP16 &&
         cfg->precision != DINOSEG_FP16X3) ||
        (cfg->head_kind != DINOSEG_HEAD_MLP && cfg->head_kind != DINOSEG_HEAD_LINEAR)) {
        dinoseg_set_error("dinoseg_create: unsupported config (embed_dim=%d heads=%d patch=%d blocks=%d classes=%d)",
                          cfg->embed_dim, cfg->num_heads, cfg->patch, cfg->n_blocks, cfg->n_classes);
        return -1;
    }
    dinoseg_handle* h = new dinoseg_handle();
    h->cfg = *cfg;
    h->planes = (cfg->precision == DINOSEG_BF16X3 || cfg->precision == DINOSEG_FP16X3) ? 2 : 1;
    h->fmt = (cfg->precision == DINOSEG_FP16 || cfg->precision == DINOSEG_FP16X3) ? FMT_FP16 : FMT_BF16;
    add_expected(h);
    *out = h;
    return 0;
}

// the inference workspaces and the split-forward stream / events (they live on the handle's device)
static void release_workspaces(dinoseg_handle* h) {
    if (h->ws) (void)hipFree(h->ws);
    if (h->ws2) (void)hipFree(h->ws2);
    if (h->aux_stream) {
        (void)hipStreamSynchronize(h->aux_stream);
        (void)hipStreamDestroy(h->aux_stream);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (auto& e : h->bw_ev) (void)hipEventDestroy(e);
    h->bw_ev.clear();
    h->ws = h->ws2 = nullptr;
    h->ws_bytes = h->ws2_bytes = 0;
    h->ws_B = h->ws_r = h->ws2_B = h->ws2_r = -1;
    h->aux_stream = nullptr;
    h->ev_fork = h->ev_join = nullptr;
}

extern "C" int dinoseg_destroy(dinoseg_handle* h) {
    if (!h) return 0;
    DeviceGuard guard(h);
    if (h->wbuf) (void)hipFree(h->wbuf);
    if (h->pos_cache) (void)hipFree(h->pos_cache);
    release_workspaces(h);
    (void)dinoseg_train_release(h);
    for (auto& r : h->prof_recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto& e : h->prof_pool) (void)hipEventDestroy(e);
    for (auto& e : h->stage_ev) (void)hipEventDestroy(e);
    delete h;
    return 0;
}

extern "C" int dinoseg_bind_weight(dinoseg_handle* h, const char* name, const void* dev_ptr, const int64_t* shape,
                                   int32_t ndim) {
    if (!h || !name || !dev_ptr || !shape) {
        dinoseg_set_error("dinoseg_bind_weight: null argument");
        return -1;
    }
    auto it = h->expected.find(name);
    if (it == h->expected.end()) {
        dinoseg_set_error("dinoseg_bind_weight: unexpected key '%s'", name);
        return -1;
    }
    const std::vector<int64_t>& want = it->second;
    bool ok = (int)want.size() == ndim;
    for (int i = 0; ok && i < ndim; ++i) ok = want[i] == shape[i];
    if (!ok) {
        dinoseg_set_error("dinoseg_bind_weight: shape mismatch for '%s'", name);
        return -1;
    }
    {
        // the device that owns the parameters owns the handle: workspace, packed weights and every launch follow it
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, dev_ptr) != hipSuccess || attr.type != hipMemoryTypeDevice) {
            (void)hipGetLastError();
            dinoseg_set_error("dinoseg_bind_weight: '%s' is not a device pointer", name);
            return -1;
        }
        if (h->device >= 0 && h->device != attr.device && !h->bound.empty() && h->bound.count(name) == 0) {
            dinoseg_set_error("dinoseg_bind_weight: '%s' lives on device %d, earlier tensors on device %d", name, attr.device, h->device);
            return -1;
        }
        if (h->device != attr.device) {
            if (h->device >= 0) {        // model.to(another device): everything the library allocated is on the old one
                DeviceGuard old(h);
                if (h->wbuf) (void)hipFree(h->wbuf);
                if (h->pos_cache) (void)hipFree(h->pos_cache);
                release_workspaces(h);
                (void)dinoseg_train_release(h);
                // events belong to the device they were created on: the gradient-stage and profiler events too
                for (auto& e : h->stage_ev) (void)hipEventDestroy(e);
                h->stage_ev.clear();
                h->stage_done = 0;
                for (auto& r : h->prof_recs) {
                    (void)hipEventDestroy(r.a);
                    (void)hipEventDestroy(r.b);
                }
                h->prof_recs.clear();
                for (auto& e : h->prof_pool) (void)hipEventDestroy(e);
                h->prof_pool.clear();
                h->wbuf = nullptr; h->pos_cache = nullptr; h->pos_r = -1;
                h->wbuf_bytes = h->pos_cap = 0;
                h->tws_B = h->tws_r = h->tr_B = -1;
                h->packed.clear();
                h->packed_slab.clear();
                h->packed_mlp.clear();
                h->packed_proj.clear();
                h->packed_qkvf.clear();
                h->packed_mlp3.clear();
    h->packed_mlp4.clear();
    h->packed_rs_bias.clear();
                h->packed_mlp4.clear();
                h->packed_rs_bias.clear();
                h->packed_rs.clear();
                h->bound.clear();
                h->grads.clear();
            }
            h->device = attr.device;
        }
    }
    BoundTensor t;
    t.ptr = reinterpret_cast<const float*>(dev_ptr);
    t.shape.assign(shape, shape + ndim);
    h->bound[name] = t;
    h->weights_ready = false;
    h->pos_stale = true;        // (dinoseg_refresh_weights resamples the cached resolution again)
    return 0;
}



// names of every nn.Linear-shaped weight that feeds gemm.hip, with its logical [N, K] and padded [n_pad, k_pad]
struct LinSpec {
    std::string wname, bname;
    int N, K, n_pad, k_pad, planes, fmt;
};

static std::vector<LinSpec> linear_specs(const dinoseg_handle* h) {
    const dinoseg_config& c = h->cfg;
    const int D = c.embed_dim, F = c.embed_dim * c.mlp_ratio, P = h->planes, FM = h->fmt;
    std::vector<LinSpec> v;
    // (fp16 mode: the patch embedding runs split like the head -- 0.13 % of the FLOPs, and its operands are raw pixels)
    v.push_back({"dino.patch_embed.proj.weight", "dino.patch_embed.proj.bias", D, 3 * c.patch * c.patch, D, 192, patch_planes(h), patch_fmt(h)});
    for (int i = 0; i < c.n_blocks; ++i) {
        const std::string b = "dino.blocks." + std::to_string(i) + ".";
        v.push_back({b + "attn.qkv.weight", b + "attn.qkv.bias", 3 * D, D, 3 * D, D, P, FM});
        v.push_back({b + "attn.proj.weight", b + "attn.proj.bias", D, D, D, D, P, FM});
        v.push_back({b + "mlp.fc1.weight", b + "mlp.fc1.bias", F, D, F, D, P, FM});
        v.push_back({b + "mlp.fc2.weight", b + "mlp.fc2.bias", D, F, D, F, P, FM});
    }
    if (c.head_kind == DINOSEG_HEAD_MLP) {
        v.push_back({"clf.layer_1.weight", "clf.layer_1.bias", 200, D, 256, D, head_planes(), split_fmt(h)});
        v.push_back({"clf.layer_2.weight", "clf.layer_2.bias", 100, 200, 128, 256, head_planes(), split_fmt(h)});
    }
    return v;
}

static int ensure_mlp_packs(dinoseg_handle* h, hipStream_t s);

extern "C" int dinoseg_refresh_weights(dinoseg_handle* h, void* stream) {
    if (!h) return -1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DeviceGuard guard(h);
    DSEG_TRY(check_stream_device(h, s));
    for (auto& kv : h->expected)
        if (!h->bound.count(kv.first)) {
            dinoseg_set_error("dinoseg_refresh_weights: missing key '%s' (strict load)", kv.first.c_str());
            return -3;
        }
    // the process-wide option is read ONCE per refresh: the packs below and every forward until the next refresh use this value
    // (a later dinoseg_set_option on a live handle takes effect with the next refresh, never between a pack and its GEMM)
    h->fp16_patch_planes_snap = options().fp16_patch_planes;
    const std::vector<LinSpec> specs = linear_specs(h);
    auto ln_fed = [&](const LinSpec& sp) {        // qkv / fc1: also kept slab-major for the LayerNorm-fused kernel
        const bool qkv = sp.wname.find("attn.qkv.weight") != std::string::npos;
        const bool fc1 = sp.wname.find("mlp.fc1.weight") != std::string::npos;
        if (sp.planes == 2 && sp.fmt != FMT_BF16) return false;      // (the hi+lo LayerNorm-fused kernel is bf16 only: gemm_ln.hip)
        return (qkv || fc1) && gemm_ln_supported(sp.K, sp.N, sp.planes, qkv ? EPI_QKV : EPI_GELU, h->cfg.embed_dim);
    };
    size_t total = 0;
    for (const LinSpec& sp : specs) {
        total += align_up((size_t)sp.planes * sp.n_pad * sp.k_pad * sizeof(bf16_t), 256);
        if (sp.n_pad != sp.N) total += align_up((size_t)sp.n_pad * sizeof(float), 256);
        if (ln_fed(sp)) total += align_up((size_t)gemm_ln_slab_elems(sp.N, sp.K, sp.planes) * sizeof(bf16_t), 256);
    }
    const int Dm = h->cfg.embed_dim, Fh = h->cfg.embed_dim * h->cfg.mlp_ratio;
    const bool mlp_fusable = mlp_fused_supported(Dm, Fh, h->planes);
    const bool mlp3_fusable = mlp_fused3_supported(Dm, Fh, h->planes);
    const bool mlp4_fusable = options().mlp_fused4 && mlp_fused4_supported(Dm, Fh, h->planes);      // (read at refresh time: a fine-tune step re-packs what exists)
    // one-plane modes of the wide model: fragment-order copies of the four block linears for the row-stationary GEMMs (gemm_rs.hip)
    auto rs_kind = [&](const LinSpec& sp) -> int {
        if (!options().gemm_rs || h->planes != 1 || Dm != 768 || sp.wname.rfind("dino.blocks.", 0) != 0) return -1;      // (read at refresh time)
        const int bit = sp.wname.find("mlp.fc1.weight") != std::string::npos ? 1 : sp.wname.find("attn.qkv.weight") != std::string::npos ? 2 : 4;
        if (!(options().gemm_rs & bit)) return -1;                                                                              // (bit set: 1 fc1, 2 qkv, 4 proj + fc2)
        if (sp.K == 768 && (sp.wname.find("attn.qkv.weight") != std::string::npos || sp.wname.find("mlp.fc1.weight") != std::string::npos)) return 0;
        if (sp.N == 768 && sp.K % 192 == 0 && (sp.wname.find("attn.proj.weight") != std::string::npos || sp.wname.find("mlp.fc2.weight") != std::string::npos)) return 1;
        return -1;
    };
    // (kind 0 with option gemm_rs_ln, read here: the copy carries the LayerNorm in front of the linear -- norm1 for qkv, norm2 for fc1 -- and a folded bias)
    auto rs_ln = [&](const LinSpec& sp) -> bool { return rs_kind(sp) == 0 && options().gemm_rs_ln; };
    for (const LinSpec& sp : specs)
        if (rs_kind(sp) >= 0) total += align_up((size_t)sp.N * sp.K * sizeof(bf16_t), 256) + (rs_ln(sp) ? align_up((size_t)sp.N * sizeof(float), 256) : 0);
    if (mlp3_fusable) total += (size_t)h->cfg.n_blocks * align_up((size_t)mlp_fused3_pack_elems(Dm, Fh) * sizeof(bf16_t), 256);
    if (mlp4_fusable) total += (size_t)h->cfg.n_blocks * align_up((size_t)mlp_fused4_pack_elems(Dm, Fh) * sizeof(bf16_t), 256);
    if (mlp_fusable)
        total += (size_t)h->cfg.n_blocks * (align_up((size_t)mlp_fused_pack_elems(Dm, Fh) * sizeof(bf16_t), 256) +
                                            align_up((size_t)mlp_fused_proj_pack_elems(Dm) * sizeof(bf16_t), 256) +
                                            align_up((size_t)mlp_fused_qkv_pack_elems(Dm) * sizeof(bf16_t), 256));
    if (total > h->wbuf_bytes) {
        ++h->generation;
        if (h->wbuf) DSEG_CHECK_HIP(hipFree(h->wbuf));
        h->wbuf = nullptr;
        DSEG_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&h->wbuf), total));
        h->wbuf_bytes = total;
    }
    size_t off = 0;
    std::vector<PackJob> jobs;
    for (const LinSpec& sp : specs) {
        PackedLinear pk;
        pk.w = reinterpret_cast<bf16_t*>(h->wbuf + off);
        pk.plane = (long)sp.n_pad * sp.k_pad;
        pk.n_pad = sp.n_pad;
        pk.k_pad = sp.k_pad;
        off += align_up((size_t)sp.planes * sp.n_pad * sp.k_pad * sizeof(bf16_t), 256);
        jobs.push_back({W(h, sp.wname), pk.w, pk.plane, sp.N, sp.K, sp.n_pad, sp.k_pad, sp.planes, 0, sp.fmt});
        if (sp.n_pad != sp.N) {
            pk.bias_pad = reinterpret_cast<float*>(h->wbuf + off);
            off += align_up((size_t)sp.n_pad * sizeof(float), 256);
            DSEG_CHECK_HIP(hipMemsetAsync(pk.bias_pad, 0, (size_t)sp.n_pad * sizeof(float), s));
            DSEG_CHECK_HIP(hipMemcpyAsync(pk.bias_pad, W(h, sp.bname), (size_t)sp.N * sizeof(float),
                                          hipMemcpyDeviceToDevice, s));
        }
        h->packed[sp.wname] = pk;
        if (ln_fed(sp)) {
            bf16_t* slab = reinterpret_cast<bf16_t*>(h->wbuf + off);
            off += align_up((size_t)gemm_ln_slab_elems(sp.N, sp.K, sp.planes) * sizeof(bf16_t), 256);
            DSEG_TRY(launch_pack_slabs(W(h, sp.wname), sp.N, sp.K, sp.planes, slab, s, sp.fmt));
            h->packed_slab[sp.wname] = slab;
        }
    }
    DSEG_TRY(launch_multi_pack(jobs.data(), (int)jobs.size(), s));
    h->packed_mlp.clear();
    h->packed_proj.clear();
    h->packed_qkvf.clear();
    h->packed_mlp3.clear();
    h->packed_rs.clear();
    for (const LinSpec& sp : specs)
        if (rs_kind(sp) >= 0) {
            bf16_t* dst = reinterpret_cast<bf16_t*>(h->wbuf + off);
            off += align_up((size_t)sp.N * sp.K * sizeof(bf16_t), 256);
            if (rs_ln(sp)) {
                float* fb = reinterpret_cast<float*>(h->wbuf + off);
                off += align_up((size_t)sp.N * sizeof(float), 256);
                const std::string blk = sp.wname.substr(0, sp.wname.find(sp.wname.find("attn.qkv") != std::string::npos ? "attn.qkv" : "mlp.fc1"));
                const std::string nrm = blk + (sp.wname.find("attn.qkv") != std::string::npos ? "norm1" : "norm2");
                DSEG_TRY(launch_pack_rs_ln(W(h, sp.wname), W(h, nrm + ".weight"), W(h, nrm + ".bias"), W(h, sp.bname), sp.N, sp.K, dst, fb, s, sp.fmt));
                h->packed_rs_bias[sp.wname] = fb;
            } else {
                DSEG_TRY(launch_pack_rs(W(h, sp.wname), sp.N, sp.K, rs_kind(sp), dst, s, sp.fmt));
            }
            h->packed_rs[sp.wname] = dst;
        }
    if (mlp3_fusable)
        for (int i = 0; i < h->cfg.n_blocks; ++i) {
            h->packed_mlp3["dino.blocks." + std::to_string(i) + "."] = reinterpret_cast<bf16_t*>(h->wbuf + off);
            off += align_up((size_t)mlp_fused3_pack_elems(Dm, Fh) * sizeof(bf16_t), 256);
        }
    if (mlp4_fusable)
        for (int i = 0; i < h->cfg.n_blocks; ++i) {
            h->packed_mlp4["dino.blocks." + std::to_string(i) + "."] = reinterpret_cast<bf16_t*>(h->wbuf + off);
            off += align_up((size_t)mlp_fused4_pack_elems(Dm, Fh) * sizeof(bf16_t), 256);
        }
    if (mlp_fusable)
        for (int i = 0; i < h->cfg.n_blocks; ++i) {
            const std::string b = "dino.blocks." + std::to_string(i) + ".";
            bf16_t* dst = reinterpret_cast<bf16_t*>(h->wbuf + off);
            off += align_up((size_t)mlp_fused_pack_elems(Dm, Fh) * sizeof(bf16_t), 256);
            h->packed_mlp[b] = dst;
            if (mlp_fused_proj_pack_elems(Dm) > 0) {
                h->packed_proj[b] = reinterpret_cast<bf16_t*>(h->wbuf + off);
                off += align_up((size_t)mlp_fused_proj_pack_elems(Dm) * sizeof(bf16_t), 256);
            }
            if (i > 0 && mlp_fused_qkv_pack_elems(Dm) > 0) {      // (block 0's qkv has no fused kernel in front of it)
                h->packed_qkvf[b] = reinterpret_cast<bf16_t*>(h->wbuf + off);
                off += align_up((size_t)mlp_fused_qkv_pack_elems(Dm) * sizeof(bf16_t), 256);
            }
        }
    // Always packed here, in stream order with the other packs: a forward captured in a graph contains no pack kernels, so a
    // deferred pack (round 3 skipped these while gradient buffers were bound) would let a replay after a fine-tune step read stale
    // fused-kernel weights next to fresh ones.  Three small launches per block.
    h->packed_mlp_stale = true;
    DSEG_TRY(ensure_mlp_packs(h, s));
    h->weights_ready = true;
    // pos_embed may have changed in place (load_state_dict into the same storage, an optimizer step on an unfrozen backbone).  A
    // captured forward contains no resample launch and never calls dinoseg_prepare_resolution, so "resample on the next forward"
    // (pos_r = -1 alone) would let a replay read the OLD rows next to freshly packed linears: resample here, in stream order with
    // the packs, into the SAME buffer: like the re-packed linears, the captured pointers stay valid and the replay reads new rows.
    if (h->pos_r > 0 && h->pos_cache != nullptr && h->bound.count("dino.pos_embed")) {
        DSEG_TRY(launch_pos_resample(W(h, "dino.pos_embed"), h->cfg.pos_grid, h->cfg.embed_dim, h->pos_r / 8, h->pos_cache, s));
    } else {
        h->pos_r = -1;      // (nothing cached: the next forward resamples, and dinoseg_prepare_resolution counts a new generation)
    }
    h->pos_stale = false;
    return 0;
}

extern "C" int dinoseg_prepare_resolution(dinoseg_handle* h, int32_t r, void* stream) {
    if (!h) return -1;
    if (r <= 0 || r % 8 != 0) {
        dinoseg_set_error("Resolution should be a multiple of 8.");
        return -1;
    }
    if (!h->bound.count("dino.pos_embed")) {
        dinoseg_set_error("dinoseg_prepare_resolution: dino.pos_embed not bound");
        return -3;
    }
    if (h->pos_r == r && !h->pos_stale) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DeviceGuard guard(h);
    const int o = r / 8, D = h->cfg.embed_dim;
    ++h->generation;        // (the cache holds ONE resolution: a captured forward of another one would read this one's rows)
    const size_t need = ((size_t)o * o + 1) * D * sizeof(float);
    if (need > h->pos_cap) {
        if (h->pos_cache) DSEG_CHECK_HIP(hipFree(h->pos_cache));
        h->pos_cache = nullptr;
        DSEG_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&h->pos_cache), need));
        h->pos_cap = need;
    }
    DSEG_TRY(launch_pos_resample(W(h, "dino.pos_embed"), h->cfg.pos_grid, D, o, h->pos_cache, s));
    h->pos_r = r;
    h->pos_stale = false;
    return 0;
}

// ------------------------------------------------------------------------------------------------ workspace
struct WsLayout {
    size_t X, A, Q, K, V, CTX, HB, FEAT, H1, H2, total;
    long a_plane, qkv_plane, ctx_plane, hb_plane, feat_plane, h1_plane, h2_plane;
    int n, ntok, npad, M, Mp;
};

static WsLayout make_layout(const dinoseg_handle* h, int B, int r) {
    const dinoseg_config& c = h->cfg;
    const int D = c.embed_dim, F = D * c.mlp_ratio, P = h->planes, HP = head_planes();
    WsLayout L;
    L.n = (r / 8) * (r / 8);
    L.ntok = L.n + 1;
    L.npad = (L.ntok + 63) / 64 * 64;
    L.M = B * L.ntok;
    L.Mp = B * L.n;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    L.X = take((size_t)L.M * D * 4);
    L.a_plane = (long)L.M * D;                 // LN output; also hosts the patch-gather matrix: patch_planes x [Mp, 192]
    {
        const size_t ln_bytes = (size_t)P * L.a_plane * 2, pg_bytes = (size_t)patch_planes(h) * L.Mp * 192 * 2;
        L.A = take(ln_bytes > pg_bytes ? ln_bytes : pg_bytes);
    }
    L.qkv_plane = (long)B * c.num_heads * L.npad * 64;
    L.Q = take((size_t)P * L.qkv_plane * 2);
    L.K = take((size_t)P * L.qkv_plane * 2);
    L.V = take((size_t)P * L.qkv_plane * 2);
    L.ctx_plane = (long)L.M * D;
    L.CTX = take((size_t)P * L.ctx_plane * 2);
    L.hb_plane = (long)L.M * F;
    L.HB = take((size_t)P * L.hb_plane * 2);
    L.feat_plane = (long)L.Mp * D;
    L.FEAT = take((size_t)HP * L.feat_plane * 2);
    L.h1_plane = (long)L.Mp * 256;
    L.H1 = take((size_t)HP * L.h1_plane * 2);
    L.h2_plane = (long)L.Mp * 128;
    L.H2 = take((size_t)HP * L.h2_plane * 2);
    L.total = off;
    return L;
}

extern "C" int64_t dinoseg_state_generation(const dinoseg_handle* h) { return h ? h->generation : -1; }

extern "C" int64_t dinoseg_workspace_bytes(const dinoseg_handle* h, int32_t B, int32_t r) {
    if (!h || B <= 0 || r <= 0 || r % 8 != 0) return -1;
    return (int64_t)(make_layout(h, B, r).total + h->wbuf_bytes);
}

// slot 0: the caller's stream; slot 1: the second half-batch of a split forward (its own buffer, the handle's internal stream)
static int ensure_workspace(dinoseg_handle* h, int slot, const WsLayout& L, int B, int r, hipStream_t s) {
    char*& ws = slot ? h->ws2 : h->ws;
    size_t& bytes = slot ? h->ws2_bytes : h->ws_bytes;
    int& wB = slot ? h->ws2_B : h->ws_B;
    int& wr = slot ? h->ws2_r : h->ws_r;
    if (L.total > bytes) {
        ++h->generation;
        if (ws) {
            DSEG_CHECK_HIP(hipStreamSynchronize(s));
            DSEG_CHECK_HIP(hipFree(ws));
        }
        ws = nullptr;
        bytes = 0;
        DSEG_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&ws), L.total));
        bytes = L.total;
        wB = -1;
    }
    if (wB != B || wr != r) {
        // key/value pad rows beyond ntok must be finite: zero Q/K/V once per layout (never written afterwards).  A layout change is a
        // new state generation: a forward captured under the OLD layout holds no memset node, and another layout's launches have since
        // written other things (fp32 residual rows ...) where its pad rows live -- the owner of the graph must capture again.
        DSEG_CHECK_HIP(hipMemsetAsync(ws + L.Q, 0, L.CTX - L.Q, s));
        ++h->generation;
        wB = B;
        wr = r;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ forward
struct MaskRequest {            // forward_mask / get_last_selfattention(x, cls_mask): see dinoseg_forward_mask
    const float* cls_mask;
    int n_masks;
    float* emb_out;
    float* attn_out;
    float* feat_out;            // dinoseg_features: final-norm tokens [B, N, D] after feat_blocks blocks (0 = all), then stop
    int feat_blocks;
};

// the fused MLP kernel runs for this many token rows (options mlp_fused / mlp_fused_min_rows)
static bool mlp_fuse_wanted(const dinoseg_handle* h, long rows) {
    return (!h->packed_mlp.empty() || !h->packed_mlp3.empty()) &&
           (options().mlp_fused == 2 || (options().mlp_fused == 1 && rows >= options().mlp_fused_min_rows));
}
// fragment-order MLP weights (mlp_fused2.hip), packed on first use after a weight refresh
static int ensure_mlp_packs(dinoseg_handle* h, hipStream_t s) {
    if (!h->packed_mlp_stale) return 0;
    const int Dm = h->cfg.embed_dim, Fh = h->cfg.embed_dim * h->cfg.mlp_ratio;
    for (auto& kv : h->packed_mlp)
        DSEG_TRY(launch_pack_mlp(W(h, kv.first + "mlp.fc1.weight"), W(h, kv.first + "mlp.fc2.weight"), Dm, Fh, kv.second, s, h->fmt));
    for (auto& kv : h->packed_proj) DSEG_TRY(launch_pack_proj(W(h, kv.first + "attn.proj.weight"), Dm, kv.second, s, h->fmt));
    for (auto& kv : h->packed_qkvf) DSEG_TRY(launch_pack_qkv(W(h, kv.first + "attn.qkv.weight"), Dm, kv.second, s, h->fmt));
    for (int i = 0; i < h->cfg.n_blocks; ++i) {      // (one-plane copies of the same streams: mlp_fused4.hip)
        const std::string b = "dino.blocks." + std::to_string(i) + ".", nb = "dino.blocks." + std::to_string(i + 1) + ".";
        if (!h->packed_mlp4.count(b)) continue;
        MlpFused3Weights w = {};
        w.Wproj = W(h, b + "attn.proj.weight"); w.W1 = W(h, b + "mlp.fc1.weight"); w.b1 = W(h, b + "mlp.fc1.bias"); w.W2 = W(h, b + "mlp.fc2.weight");
        w.gamma2 = W(h, b + "norm2.weight"); w.beta2 = W(h, b + "norm2.bias");
        if (i + 1 < h->cfg.n_blocks) {
            w.Wqkv_next = W(h, nb + "attn.qkv.weight"); w.bqkv_next = W(h, nb + "attn.qkv.bias");
            w.gamma1_next = W(h, nb + "norm1.weight"); w.beta1_next = W(h, nb + "norm1.bias");
        }
        DSEG_TRY(launch_pack_mlp4(w, Dm, Fh, h->packed_mlp4.at(b), s, h->fmt));
    }
    for (int i = 0; i < h->cfg.n_blocks; ++i) {      // (block i's stream ends with the qkv weight of block i + 1: the tail of its fused launch)
        const std::string b = "dino.blocks." + std::to_string(i) + ".", nb = "dino.blocks." + std::to_string(i + 1) + ".";
        if (!h->packed_mlp3.count(b)) continue;
        MlpFused3Weights w = {};
        w.Wproj = W(h, b + "attn.proj.weight"); w.W1 = W(h, b + "mlp.fc1.weight"); w.b1 = W(h, b + "mlp.fc1.bias"); w.W2 = W(h, b + "mlp.fc2.weight");
        w.gamma2 = W(h, b + "norm2.weight"); w.beta2 = W(h, b + "norm2.bias");
        if (i + 1 < h->cfg.n_blocks) {
            w.Wqkv_next = W(h, nb + "attn.qkv.weight"); w.bqkv_next = W(h, nb + "attn.qkv.bias");
            w.gamma1_next = W(h, nb + "norm1.weight"); w.beta1_next = W(h, nb + "norm1.bias");
        }
        DSEG_TRY(launch_pack_mlp3(w, Dm, Fh, h->packed_mlp3.at(b), s, h->fmt));
    }
    h->packed_mlp_stale = false;
    return 0;
}

static int forward_impl(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r, float* logp_out,
                        int32_t* argmax_out, int32_t tap_block, float* tap_out, float* attn_out, void* stream,
                        const MaskRequest* mreq = nullptr, int slot = 0, int disp_B = 0) {
    if (!h || !x || B <= 0) {
        dinoseg_set_error("dinoseg_forward: bad argument");
        return -1;
    }
    if (r <= 0 || r % 8 != 0) {
        dinoseg_set_error("Resolution should be a multiple of 8.");
        return -1;
    }
    if (x_kind != DINOSEG_INPUT_U8_HWC && x_kind != DINOSEG_INPUT_F32_CHW) {
        dinoseg_set_error("dinoseg_forward: bad x_kind %d", x_kind);
        return -1;
    }
    if (!h->weights_ready) {
        dinoseg_set_error("dinoseg_forward: weights not packed (call dinoseg_refresh_weights after binding)");
        return -3;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DeviceGuard guard(h);
    DSEG_TRY(check_stream_device(h, s));
    DSEG_TRY(dinoseg_prepare_resolution(h, r, stream));

    const dinoseg_config& c = h->cfg;
    const int D = c.embed_dim, F = D * c.mlp_ratio, P = h->planes, HP = head_planes(), H = c.num_heads, FM = h->fmt, SF = split_fmt(h);
    const WsLayout L = make_layout(h, B, r);
    DSEG_TRY(ensure_workspace(h, slot, L, B, r, s));
    // every size-dependent kernel choice below is made for the rows of the WHOLE call: the half-batches of a split forward (disp_B =
    // the call's batch) then take the routes -- and the summation order -- the unsplit batch takes, so the split changes no bit
    const int dB = disp_B > 0 ? disp_B : B;
    const int disp_M = dB * L.ntok, disp_Mp = dB * L.n;
    // a block linear through the row-stationary streaming kernels (gemm_rs.hip) where a fragment-order copy exists and the batch fills the chip
    // ... and with the LayerNorm in front of it in its prologue (qkv / fc1 of the wide model: no LayerNorm launch, no 16-bit A round trip)
    auto rs_takes_ln = [&](const std::string& wname) -> bool {
        return options().gemm_rs && options().gemm_rs_ln && disp_M >= options().gemm_rs_min_rows && h->packed_rs_bias.count(wname) && P == 1 &&
               !(mreq && mreq->cls_mask);
    };
    auto gemm_any = [&](GemmParams& g, const std::string& wname) -> int {
        // (a copy that carries the LayerNorm serves only the launch with the LayerNorm inside)
        if (options().gemm_rs && disp_M >= options().gemm_rs_min_rows && h->packed_rs.count(wname) && (g.ln_x != nullptr) == (h->packed_rs_bias.count(wname) != 0)) {
            GemmParams r = g;
            r.W = h->packed_rs.at(wname);
            if (g.ln_x) r.bias = h->packed_rs_bias.at(wname);
            if (gemm_rs_supported(r)) return launch_gemm_rs(r, s);
        }
        if (g.ln_x) {      // (rs_takes_ln said the LayerNorm runs inside: no normalised A exists for another kernel)
            dinoseg_set_error("internal: %s was routed to gemm_rs with its LayerNorm inside, which does not take it", wname.c_str());
            return -1;
        }
        return launch_gemm(g, s);
    };
    char* ws = slot ? h->ws2 : h->ws;
    float* X = reinterpret_cast<float*>(ws + L.X);
    bf16_t* A = reinterpret_cast<bf16_t*>(ws + L.A);
    bf16_t* Q = reinterpret_cast<bf16_t*>(ws + L.Q);
    bf16_t* Kb = reinterpret_cast<bf16_t*>(ws + L.K);
    bf16_t* V = reinterpret_cast<bf16_t*>(ws + L.V);
    bf16_t* CTX = reinterpret_cast<bf16_t*>(ws + L.CTX);
    bf16_t* HB = reinterpret_cast<bf16_t*>(ws + L.HB);
    bf16_t* FEAT = reinterpret_cast<bf16_t*>(ws + L.FEAT);
    bf16_t* H1 = reinterpret_cast<bf16_t*>(ws + L.H1);
    bf16_t* H2 = reinterpret_cast<bf16_t*>(ws + L.H2);

    // ---- prepare_tokens (vision_transformer.py:224-235) ----
    float mean255[3], inv255[3];
    norm_consts(mean255, inv255);
    const long pg_plane = (long)L.Mp * 192;
    const int PP = patch_planes(h);
    DSEG_PROF(DINOSEG_PROF_PATCH, DSEG_TRY(launch_patch_gather(x, x_kind, B, r, mean255, inv255, A, pg_plane, PP, s, patch_fmt(h))));
    {
        const PackedLinear& pk = h->packed.at("dino.patch_embed.proj.weight");
        GemmParams g = {};
        g.A = A; g.a_plane = pg_plane; g.lda = 192;
        g.W = pk.w; g.w_plane = pk.plane;
        g.M = L.Mp; g.N = D; g.K = 192; g.planes = PP; g.fmt = patch_fmt(h); g.epi = EPI_PATCH; g.dispatch_rows = disp_Mp;
        g.bias = W(h, "dino.patch_embed.proj.bias");
        g.out_f32 = X; g.ldo_f32 = D;
        g.pos = h->pos_cache; g.n_patches = L.n;
        DSEG_PROF(DINOSEG_PROF_PATCH, DSEG_TRY(launch_gemm(g, s)));
    }
    DSEG_PROF(DINOSEG_PROF_PATCH, DSEG_TRY(launch_cls_rows(X, W(h, "dino.cls_token"), h->pos_cache, B, L.ntok, D, s)));
    const size_t xbytes = (size_t)L.M * D * sizeof(float);
    if (tap_block == 0 && tap_out) DSEG_CHECK_HIP(hipMemcpyAsync(tap_out, X, xbytes, hipMemcpyDeviceToDevice, s));

    const float qscale = 0.125f * 1.44269504088896340736f;   // head_dim^-0.5 (vision_transformer.py:73) * log2(e)
    // fp16 hi + lo planes: from two rounds of 256-query workgroups on, the attention is the zero-reference assembly kernel, whose
    // probabilities and V are bf16 hi + lo planes -- the qkv epilogue writes V that way (decided for the batch of the WHOLE call; never
    // on the visualisation paths, whose small kernels read V in the mode's own format)
    const int v_bf16 = (P == 2 && FM == FMT_FP16 && !attn_out && !(mreq && mreq->cls_mask) && attention_x3_za(dB, H, L.ntok)) ? 1 : 0;

    // ---- transformer blocks (vision_transformer.py:122-140) ----
    bool qkv_ready = false;      // Q / K / V of block i were written by block i-1's fused launch (mlp_fused2.hip, QKV tail)
    for (int i = 0; i < c.n_blocks; ++i) {
        const std::string b = "dino.blocks." + std::to_string(i) + ".";
        // gemm_ln: 0 never fused, 2 always, 1 (default) by measurement (round 4, tools/r4_smallbatch.sh, 1..6 frames @480):
        //  * single plane (bf16 / fp16): fused from 80 row panels of 128 on -- below that its persistent 128 x 384 panels leave most
        //    CUs idle (one frame = 29 panels: qkv 31 against 21 us with LayerNorm + the 128x128 kernel, fc1 40 against 23; the
        //    whole single-frame forward 1.72 -> 1.33 ms at 12 blocks; crossover between 2 and 3 frames);
        //  * hi+lo planes: its 64-row panels run one workgroup per CU, so it wins only while they fill about one round of the chip
        //    (9 600 .. 16 384 rows = 3-4 frames: fc1 1.15 against 1.24 ms; 1 frame 1.01 against 0.48, 6 frames 2.13 against 1.33);
        //    from 512 tiles of 128 x 384 on, LayerNorm + the hi+lo configuration of the persistent GEMM (B = 32: fc1 30 + 534 us
        //    against 670 fused).
        const bool big_x3 = P == 2 && options().gemm_big && (long)((disp_M + 127) / 128) * 3 >= 512;
        const bool ln_small = P == 1 ? (disp_M + 127) / 128 < 80 : (disp_M < 9600 || disp_M > 16384);
        const bool fuse_ln = options().gemm_ln == 2 || (options().gemm_ln == 1 && !big_x3 && !ln_small);
        // (the fused kernel keeps 32-bit output row offsets)
        if (qkv_ready) {
            qkv_ready = false;
        } else if (fuse_ln && h->packed_slab.count(b + "attn.qkv.weight") && L.qkv_plane < (1L << 31)) {
            // LN1 + qkv in one launch: X rows are normalised in the GEMM's prologue, no bf16 A round trip (gemm_ln.hip)
            LnGemmParams g = {};
            g.X = X; g.ldx = D; g.gamma = W(h, b + "norm1.weight"); g.beta = W(h, b + "norm1.bias"); g.eps = c.ln_eps;
            g.W = h->packed_slab.at(b + "attn.qkv.weight"); g.bias = W(h, b + "attn.qkv.bias");
            g.M = L.M; g.N = 3 * D; g.epi = EPI_QKV;
            g.q = Q; g.k = Kb; g.v = V; g.qkv_plane = L.qkv_plane;
            g.ntok = L.ntok; g.npad = L.npad; g.heads = H; g.dmodel = D; g.qscale = qscale; g.fmt = FM;
            DSEG_PROF(DINOSEG_PROF_QKV, DSEG_TRY(launch_gemm_ln(g, D, P, s)));
        } else {
        const bool ln_inside = rs_takes_ln(b + "attn.qkv.weight");
        if (!ln_inside)
        DSEG_PROF(DINOSEG_PROF_LN, DSEG_TRY(launch_layernorm(X, W(h, b + "norm1.weight"), W(h, b + "norm1.bias"), c.ln_eps, L.M, D, A, L.a_plane, P,
                                  nullptr, 0, L.ntok, s, FM)));
        {
            const PackedLinear& pk = h->packed.at(b + "attn.qkv.weight");
            GemmParams g = {};
            if (ln_inside) { g.ln_x = X; g.ln_eps = c.ln_eps; }
            g.A = A; g.a_plane = L.a_plane; g.lda = D;
            g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.M; g.N = 3 * D; g.K = D; g.planes = P; g.fmt = FM; g.epi = EPI_QKV; g.dispatch_rows = disp_M;
            g.v_bf16 = v_bf16;
            g.bias = W(h, b + "attn.qkv.bias");
            g.q = Q; g.k = Kb; g.v = V; g.qkv_plane = L.qkv_plane;
            g.ntok = L.ntok; g.npad = L.npad; g.heads = H; g.dmodel = D; g.qscale = qscale;
            DSEG_PROF(DINOSEG_PROF_QKV, DSEG_TRY(gemm_any(g, b + "attn.qkv.weight")));
        }
        }
        if (attn_out && i == c.n_blocks - 1)      // get_last_selfattention: probabilities of the last block, then stop
            return launch_attn_probs(Q, Kb, L.qkv_plane, P, B, H, L.ntok, L.npad, attn_out, s, FM);
        if (mreq && mreq->cls_mask && i == c.n_blocks - 1) {
            // last block with cls_mask (Block.forward, vision_transformer.py:127-140): the CLS token attends through each mask;
            // its residual is repeated once per mask; MLP and the final norm run on those n_masks rows only.  The patch-token
            // rows of X / A / CTX / HB are dead from here on and host the n_masks rows (checked: n_masks < ntok).
            const int Nm = mreq->n_masks;
            DSEG_TRY(launch_cls_mask_attn(Q, Kb, V, L.qkv_plane, P, H, L.ntok, L.npad, mreq->cls_mask, Nm, CTX, L.ctx_plane,
                                          mreq->attn_out, s, FM));
            if (!mreq->emb_out) return 0;
            float* Xm = X + D;                        // rows 1 .. Nm
            DSEG_TRY(launch_broadcast_row0(X, D, Nm, s));
            auto lin = [&](const std::string& name, const bf16_t* Ain, long a_plane, int lda, int N, int K, int epi, bf16_t* ob,
                           long o_plane) -> int {
                const PackedLinear& pk = h->packed.at(name + ".weight");
                GemmParams g = {};
                g.A = Ain; g.a_plane = a_plane; g.lda = lda;
                g.W = pk.w; g.w_plane = pk.plane;
                g.M = Nm; g.N = N; g.K = K; g.planes = P; g.fmt = FM; g.epi = epi;
                g.bias = W(h, name + ".bias");
                g.out_f32 = Xm; g.ldo_f32 = D;
                g.out_bf16 = ob; g.out_plane = o_plane; g.ldo = N;
                return launch_gemm_small(g, s);
            };
            DSEG_TRY(lin(b + "attn.proj", CTX, L.ctx_plane, D, D, D, EPI_RESID, nullptr, 0));
            DSEG_TRY(launch_layernorm(Xm, W(h, b + "norm2.weight"), W(h, b + "norm2.bias"), c.ln_eps, Nm, D, A, L.a_plane, P, nullptr, 0,
                                      L.ntok, s, FM));
            DSEG_TRY(lin(b + "mlp.fc1", A, L.a_plane, D, F, D, EPI_GELU, HB, L.hb_plane));
            DSEG_TRY(lin(b + "mlp.fc2", HB, L.hb_plane, F, D, F, EPI_RESID, nullptr, 0));
            return launch_layernorm(Xm, W(h, "dino.norm.weight"), W(h, "dino.norm.bias"), c.ln_eps, Nm, D, A, L.a_plane, P,
                                    mreq->emb_out, 0, L.ntok, s, FM);
        }
        {
            AttnParams a = {};
            a.q = Q; a.k = Kb; a.v = V; a.qkv_plane = L.qkv_plane;
            a.ctx = CTX; a.ctx_plane = L.ctx_plane; a.lse = nullptr;
            a.B = B; a.heads = H; a.ntok = L.ntok; a.npad = L.npad; a.planes = P; a.fmt = FM;
            a.shared_gpu = h->in_split ? 1 : 0;
            a.dispatch_B = dB;
            a.v_bf16 = v_bf16;
            DSEG_PROF(DINOSEG_PROF_ATTN, DSEG_TRY(launch_attention(a, s)));
        }
        const bool fuse_mlp3 = P == 2 && h->packed_mlp3.count(b) &&      // hi + lo planes: mlp_fused3.hip
                               (options().mlp_fused == 2 || (options().mlp_fused == 1 && disp_M >= options().mlp_fused3_min_rows));
        const bool fuse_mlp = fuse_mlp3 || (h->packed_mlp.count(b) && mlp_fuse_wanted(h, disp_M));
        // (the fused MLP kernels take the attention output projection along: x += proj(ctx) + b, then the MLP, one launch)
        const bool fuse_proj = fuse_mlp && options().proj_fused && (fuse_mlp3 || (P == 1 && h->packed_proj.count(b)));
        if (!fuse_proj) {
            const PackedLinear& pk = h->packed.at(b + "attn.proj.weight");
            GemmParams g = {};
            g.A = CTX; g.a_plane = L.ctx_plane; g.lda = D;
            g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.M; g.N = D; g.K = D; g.planes = P; g.fmt = FM; g.epi = EPI_RESID; g.dispatch_rows = disp_M;
            g.bias = W(h, b + "attn.proj.bias");
            g.out_f32 = X; g.ldo_f32 = D;
            DSEG_PROF(DINOSEG_PROF_PROJ, DSEG_TRY(gemm_any(g, b + "attn.proj.weight")));
        }
        if (fuse_mlp3) {
            DSEG_TRY(ensure_mlp_packs(h, s));
            // projection + LN2 + fc1 + GELU + fc2 + residual on hi + lo planes in one launch (mlp_fused3.hip)
            MlpFused3Params g = {};
            g.X = X; g.eps = c.ln_eps;
            g.Wp = h->packed_mlp3.at(b); g.b2 = W(h, b + "mlp.fc2.bias");
            g.M = L.M; g.fmt = FM;
            if (fuse_proj) {
                g.ctx = CTX; g.ctx_plane = L.ctx_plane; g.bproj = W(h, b + "attn.proj.bias");
                // ... and LayerNorm1 + qkv of the next block (a tap of this block's output still reads X, which is complete)
                if (options().qkv_fused3 && i + 1 < c.n_blocks && L.qkv_plane < (1L << 31)) {
                    g.q = Q; g.k = Kb; g.v = V; g.qkv_plane = L.qkv_plane;
                    g.ntok = L.ntok; g.npad = L.npad; g.heads = H; g.qscale = qscale; g.v_bf16 = v_bf16;
                    qkv_ready = true;
                }
            }
            DSEG_PROF(DINOSEG_PROF_FC1, DSEG_TRY(launch_mlp_fused3(g, s)));
        } else if (fuse_mlp && fuse_proj && options().mlp_fused4 && !options().qkv_fused && h->packed_mlp4.count(b)) {
            DSEG_TRY(ensure_mlp_packs(h, s));
            // the same launch with one wave per SIMD (mlp_fused4.hip)
            MlpFused3Params g = {};
            g.X = X; g.eps = c.ln_eps;
            g.Wp = h->packed_mlp4.at(b); g.b2 = W(h, b + "mlp.fc2.bias");
            g.M = L.M; g.fmt = FM;
            g.ctx = CTX; g.bproj = W(h, b + "attn.proj.bias");
            // ... and LayerNorm1 + qkv of the next block (a tap of this block's output still reads X, which is complete)
            if (options().qkv_fused4 && i + 1 < c.n_blocks) {
                g.q = Q; g.k = Kb; g.v = V; g.ntok = L.ntok; g.npad = L.npad; g.heads = H; g.qscale = qscale;
                qkv_ready = true;
            }
            DSEG_PROF(DINOSEG_PROF_FC1, DSEG_TRY(launch_mlp_fused4(g, s)));
        } else if (fuse_mlp) {
            DSEG_TRY(ensure_mlp_packs(h, s));      // (a split forward has done this before its fork)
            // LN2 + fc1 + GELU + fc2 + residual in one launch: the hidden activation never reaches HBM (mlp_fused2.hip)
            MlpFusedParams g = {};
            g.X = X; g.ldx = D; g.gamma = W(h, b + "norm2.weight"); g.beta = W(h, b + "norm2.bias"); g.eps = c.ln_eps;
            g.Wp = h->packed_mlp.at(b); g.b1 = W(h, b + "mlp.fc1.bias"); g.b2 = W(h, b + "mlp.fc2.bias");
            g.M = L.M; g.fmt = FM;
            if (fuse_proj) {
                g.ctx = CTX; g.Wproj = h->packed_proj.at(b); g.bproj = W(h, b + "attn.proj.bias");
                const std::string nb = "dino.blocks." + std::to_string(i + 1) + ".";
                // ... and LayerNorm1 + qkv of the next block (a tap of this block's output still reads X, which is complete)
                if (options().qkv_fused && i + 1 < c.n_blocks && h->packed_qkvf.count(nb)) {
                    g.Wqkv = h->packed_qkvf.at(nb); g.bqkv = W(h, nb + "attn.qkv.bias");
                    g.gamma1 = W(h, nb + "norm1.weight"); g.beta1 = W(h, nb + "norm1.bias");
                    g.q = Q; g.k = Kb; g.v = V; g.ntok = L.ntok; g.npad = L.npad; g.heads = H; g.qscale = qscale;
                    qkv_ready = true;
                }
            }
            DSEG_PROF(DINOSEG_PROF_FC1, DSEG_TRY(launch_mlp_fused2(g, s)));
        } else {
        if (fuse_ln && h->packed_slab.count(b + "mlp.fc1.weight") && L.hb_plane < (1L << 31)) {
            LnGemmParams g = {};
            g.X = X; g.ldx = D; g.gamma = W(h, b + "norm2.weight"); g.beta = W(h, b + "norm2.bias"); g.eps = c.ln_eps;
            g.W = h->packed_slab.at(b + "mlp.fc1.weight"); g.bias = W(h, b + "mlp.fc1.bias");
            g.M = L.M; g.N = F; g.epi = EPI_GELU; g.fmt = FM;
            g.out_bf16 = HB; g.out_plane = L.hb_plane; g.ldo = F;
            DSEG_PROF(DINOSEG_PROF_FC1, DSEG_TRY(launch_gemm_ln(g, D, P, s)));
        } else {
        const bool ln_inside = rs_takes_ln(b + "mlp.fc1.weight");
        if (!ln_inside)
        DSEG_PROF(DINOSEG_PROF_LN, DSEG_TRY(launch_layernorm(X, W(h, b + "norm2.weight"), W(h, b + "norm2.bias"), c.ln_eps, L.M, D, A, L.a_plane, P,
                                  nullptr, 0, L.ntok, s, FM)));
        {
            const PackedLinear& pk = h->packed.at(b + "mlp.fc1.weight");
            GemmParams g = {};
            if (ln_inside) { g.ln_x = X; g.ln_eps = c.ln_eps; }
            g.A = A; g.a_plane = L.a_plane; g.lda = D;
            g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.M; g.N = F; g.K = D; g.planes = P; g.fmt = FM; g.epi = EPI_GELU; g.dispatch_rows = disp_M;
            g.bias = W(h, b + "mlp.fc1.bias");
            g.out_bf16 = HB; g.out_plane = L.hb_plane; g.ldo = F;
            DSEG_PROF(DINOSEG_PROF_FC1, DSEG_TRY(gemm_any(g, b + "mlp.fc1.weight")));
        }
        }
        {
            const PackedLinear& pk = h->packed.at(b + "mlp.fc2.weight");
            GemmParams g = {};
            g.A = HB; g.a_plane = L.hb_plane; g.lda = F;
            g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.M; g.N = D; g.K = F; g.planes = P; g.fmt = FM; g.epi = EPI_RESID; g.dispatch_rows = disp_M;
            g.bias = W(h, b + "mlp.fc2.bias");
            g.out_f32 = X; g.ldo_f32 = D;
            DSEG_PROF(DINOSEG_PROF_FC2, DSEG_TRY(gemm_any(g, b + "mlp.fc2.weight")));
        }
        }
        if (tap_block == i + 1 && tap_out) DSEG_CHECK_HIP(hipMemcpyAsync(tap_out, X, xbytes, hipMemcpyDeviceToDevice, s));
        if (mreq && mreq->feat_out && mreq->feat_blocks == i + 1 && i + 1 < c.n_blocks)      // forward(x, intermediate=k)
            return launch_layernorm(X, W(h, "dino.norm.weight"), W(h, "dino.norm.bias"), c.ln_eps, L.M, D, nullptr, 0, 1, mreq->feat_out, 0,
                                    L.ntok, s);
    }
    if (mreq && mreq->feat_out)     // VisionTransformer.forward(x, all=True): every token through the final norm, fp32
        return launch_layernorm(X, W(h, "dino.norm.weight"), W(h, "dino.norm.bias"), c.ln_eps, L.M, D, nullptr, 0, 1, mreq->feat_out, 0,
                                L.ntok, s);

    // ---- final norm, drop CLS (vision_transformer.py:243; pl_torch_modules.py:243,253) ----
    DSEG_PROF(DINOSEG_PROF_LN, DSEG_TRY(launch_layernorm(X, W(h, "dino.norm.weight"), W(h, "dino.norm.bias"), c.ln_eps, L.M, D, FEAT, L.feat_plane,
                              HP, nullptr, 1, L.ntok, s, SF)));

    // ---- segmentation head (pl_torch_modules.py:108-138), always in split precision ----
    if (c.head_kind == DINOSEG_HEAD_MLP) {
        {
            const PackedLinear& pk = h->packed.at("clf.layer_1.weight");
            GemmParams g = {};
            g.A = FEAT; g.a_plane = L.feat_plane; g.lda = D;
            g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.Mp; g.N = 256; g.K = D; g.planes = HP; g.fmt = SF; g.epi = EPI_RELU;
            g.bias = pk.bias_pad;
            g.out_bf16 = H1; g.out_plane = L.h1_plane; g.ldo = 256;
            DSEG_PROF(DINOSEG_PROF_HEAD, DSEG_TRY(launch_gemm(g, s)));
        }
        {
            const PackedLinear& pk = h->packed.at("clf.layer_2.weight");
            GemmParams g = {};
            g.A = H1; g.a_plane = L.h1_plane; g.lda = 256;
            g.W = pk.w; g.w_plane = pk.plane;
            g.M = L.Mp; g.N = 128; g.K = 256; g.planes = HP; g.fmt = SF; g.epi = EPI_RELU;
            g.bias = pk.bias_pad;
            g.out_bf16 = H2; g.out_plane = L.h2_plane; g.ldo = 128;
            DSEG_PROF(DINOSEG_PROF_HEAD, DSEG_TRY(launch_gemm(g, s)));
        }
        DSEG_PROF(DINOSEG_PROF_HEAD, DSEG_TRY(launch_head_final(H2, L.h2_plane, 128, L.Mp, 100, W(h, "clf.layer_3.weight"), W(h, "clf.layer_3.bias"),
                                   c.n_classes, logp_out ? logp_out : reinterpret_cast<float*>(ws + L.HB), argmax_out, s, SF)));
    } else {
        DSEG_PROF(DINOSEG_PROF_HEAD, DSEG_TRY(launch_head_final(FEAT, L.feat_plane, D, L.Mp, D, W(h, "clf.layer_1.weight"), W(h, "clf.layer_1.bias"),
                                   c.n_classes, logp_out ? logp_out : reinterpret_cast<float*>(ws + L.HB), argmax_out, s, SF)));
    }
    return 0;
}

int ensure_aux_stream(dinoseg_handle* h) {
    if (!h->aux_stream) {
        DSEG_CHECK_HIP(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
        DSEG_CHECK_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        DSEG_CHECK_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    return 0;
}

// Option "streams" = 2: a batch of >= split_min frames runs as two half-batches, the first on the caller's stream, the second on
// the handle's internal stream (forked from and joined to the caller's stream by events, so the call keeps its stream-ordered
// semantics and stays capturable).  Frames are independent (pl_torch_modules.py:253 flattens them); kernels of different
// layers of the two halves overlap: one half's attention fills the CUs the other half's GEMM tail rounds and memory phases
// leave idle (measured: +4.5 % frames/s at B = 32; four quarter-batches: -5 %).  The two workspaces together are the size of one.
extern "C" int dinoseg_forward(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r, float* logp_out,
                               int32_t* argmax_out, int32_t tap_block, float* tap_out, void* stream) {
    const bool split = h && x && options().streams >= 2 && B >= options().split_min && B >= 2 && tap_block < 0 && !tap_out &&
                       r > 0 && r % 8 == 0 && (x_kind == DINOSEG_INPUT_U8_HWC || x_kind == DINOSEG_INPUT_F32_CHW) && h->weights_ready;
    if (!split) return forward_impl(h, x, x_kind, B, r, logp_out, argmax_out, tap_block, tap_out, nullptr, stream);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DeviceGuard guard(h);
    DSEG_TRY(check_stream_device(h, s));
    DSEG_TRY(ensure_aux_stream(h));
    DSEG_TRY(dinoseg_prepare_resolution(h, r, stream));      // the resampled position embedding: before the fork, both halves read it
    const int B0 = (B + 1) / 2, B1 = B - B0;
    const long n = (long)(r / 8) * (r / 8);
    const size_t frame_bytes = x_kind == DINOSEG_INPUT_U8_HWC ? (size_t)r * r * 3 : (size_t)r * r * 3 * sizeof(float);
    const void* x1 = reinterpret_cast<const char*>(x) + (size_t)B0 * frame_bytes;
    const long ntok_ = n + 1;
    if (mlp_fuse_wanted(h, B * ntok_)) DSEG_TRY(ensure_mlp_packs(h, s));   // before the fork: both halves read them
    DSEG_CHECK_HIP(hipEventRecord(h->ev_fork, s));
    DSEG_CHECK_HIP(hipStreamWaitEvent(h->aux_stream, h->ev_fork, 0));
    h->in_split = true;
    const int rc0 = forward_impl(h, x, x_kind, B0, r, logp_out, argmax_out, -1, nullptr, nullptr, stream, nullptr, 0, B);
    const int rc1 = forward_impl(h, x1, x_kind, B1, r, logp_out ? logp_out + (size_t)B0 * n * h->cfg.n_classes : nullptr,
                                 argmax_out ? argmax_out + (size_t)B0 * n : nullptr, -1, nullptr, nullptr, h->aux_stream, nullptr, 1, B);
    h->in_split = false;
    // join even after an error: the caller's stream must not run ahead of work already queued on the internal one
    DSEG_CHECK_HIP(hipEventRecord(h->ev_join, h->aux_stream));
    DSEG_CHECK_HIP(hipStreamWaitEvent(s, h->ev_join, 0));
    return rc0 ? rc0 : rc1;
}

extern "C" int dinoseg_last_selfattention(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r, float* attn_out,
                                          void* stream) {
    if (!attn_out || !h || h->cfg.n_blocks < 1) {
        dinoseg_set_error("dinoseg_last_selfattention: needs an output buffer and at least one block");
        return -1;
    }
    return forward_impl(h, x, x_kind, B, r, nullptr, nullptr, -1, nullptr, attn_out, stream);
}

extern "C" int dinoseg_forward_mask(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t r, const float* cls_mask,
                                    int32_t n_masks, float* emb_out, float* attn_out, void* stream) {
    if (!h || h->cfg.n_blocks < 1 || !cls_mask || n_masks < 1 || (!emb_out && !attn_out)) {
        dinoseg_set_error("dinoseg_forward_mask: needs at least one block, n_masks >= 1 masks and one output buffer");
        return -1;
    }
    if (r > 0 && r % 8 == 0 && n_masks >= (r / 8) * (r / 8) + 1) {
        dinoseg_set_error("dinoseg_forward_mask: n_masks=%d must be smaller than the token count %d", n_masks, (r / 8) * (r / 8) + 1);
        return -1;
    }
    const MaskRequest mr = {cls_mask, n_masks, emb_out, attn_out, nullptr, 0};
    return forward_impl(h, x, x_kind, 1, r, nullptr, nullptr, -1, nullptr, nullptr, stream, &mr);
}

extern "C" int dinoseg_features(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r, int32_t n_blocks,
                                float* tokens_out, void* stream) {
    if (!h || !tokens_out || n_blocks < 0 || n_blocks > h->cfg.n_blocks) {
        dinoseg_set_error("dinoseg_features: needs an output buffer and 0 <= n_blocks <= %d", h ? h->cfg.n_blocks : 0);
        return -1;
    }
    const MaskRequest mr = {nullptr, 0, nullptr, nullptr, tokens_out, n_blocks};
    return forward_impl(h, x, x_kind, B, r, nullptr, nullptr, -1, nullptr, nullptr, stream, &mr);
}

extern "C" int dinoseg_op_resize_u8(const uint8_t* src, int32_t sh, int32_t sw, uint8_t* dst, int32_t dh, int32_t dw, void* stream) {
    if (!src || !dst) {
        dinoseg_set_error("dinoseg_op_resize_u8: null pointer");
        return -1;
    }
    return launch_resize_u8(src, sh, sw, dst, dh, dw, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_op_confusion(const int32_t* pred, const int64_t* gt, int64_t n, int32_t n_classes, int64_t* cm, void* stream) {
    return launch_confusion(pred, gt, n, n_classes, cm, reinterpret_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------------ options
namespace dseg {
Options& options() {
    static Options o;
    return o;
}
}  // namespace dseg

extern "C" int dinoseg_set_option(const char* key, int32_t value) {
    if (!key) return -1;
    if (strcmp(key, "gemm_ln") == 0) {
        dseg::options().gemm_ln = value;
        return 0;
    }
    if (strcmp(key, "gemm_big") == 0) {
        dseg::options().gemm_big = value;
        return 0;
    }
    if (strcmp(key, "route_ab") == 0) {
        dseg::options().route_ab = value;
        return 0;
    }
    if (strcmp(key, "fp16_patch_planes") == 0) {
        dseg::options().fp16_patch_planes = value == 1 ? 1 : 2;
        return 0;
    }
    if (strcmp(key, "op_v_bf16") == 0) {      // dinoseg_op_attention with fp16 hi + lo planes: V is given as bf16 hi + lo planes (AttnParams::v_bf16)
        dseg::options().op_v_bf16 = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "op_fmt") == 0) {      // operand format of the single-plane stand-alone ops (dinoseg_op_*): 0 bf16, 1 fp16
        if (value != FMT_BF16 && value != FMT_FP16) {
            dinoseg_set_error("dinoseg_set_option: op_fmt must be 0 (bf16) or 1 (fp16)");
            return -1;
        }
        dseg::options().op_fmt = value;
        return 0;
    }
    if (strcmp(key, "streams") == 0) {
        dseg::options().streams = value;
        return 0;
    }
    if (strcmp(key, "mlp_fused") == 0) {
        dseg::options().mlp_fused = value;
        return 0;
    }
    if (strcmp(key, "qkv_fused") == 0) {
        dseg::options().qkv_fused = value;
        return 0;
    }
    if (strcmp(key, "proj_fused") == 0) {
        dseg::options().proj_fused = value;
        return 0;
    }
    if (strcmp(key, "mlp_stagger") == 0) {
        dseg::options().mlp_stagger = value;
        return 0;
    }
    if (strcmp(key, "mlp_grid") == 0) {
        dseg::options().mlp_grid = value;
        return 0;
    }
    if (strcmp(key, "splitk_tiles") == 0) {
        dseg::options().splitk_tiles = value;
        return 0;
    }
    if (strcmp(key, "deterministic") == 0) {       // the fine-tune step's reductions in a fixed order (kernels.h Options::deterministic)
        dseg::options().deterministic = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "train_streams") == 0) {
        dseg::options().train_streams = value;
        return 0;
    }
    if (strcmp(key, "mlp_variant") == 0) return 0;      // (accepted and ignored: the one-wave-per-SIMD build was removed in round 4)
    if (strcmp(key, "gemm_rs") == 0) {
        dseg::options().gemm_rs = value & 7;
        return 0;
    }
    if (strcmp(key, "gemm_rs_ln") == 0) {
        dseg::options().gemm_rs_ln = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "gemm_rs_min_rows") == 0) {
        dseg::options().gemm_rs_min_rows = value;
        return 0;
    }
    if (strcmp(key, "qkv_fused3") == 0) {
        dseg::options().qkv_fused3 = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "qkv_fused4") == 0) {
        dseg::options().qkv_fused4 = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "mlp_fused4") == 0) {
        dseg::options().mlp_fused4 = value ? 1 : 0;
        return 0;
    }
    if (strcmp(key, "mlp_fused3_min_rows") == 0) {
        dseg::options().mlp_fused3_min_rows = value;
        return 0;
    }
    if (strcmp(key, "mlp_fused_min_rows") == 0) {
        dseg::options().mlp_fused_min_rows = value;
        return 0;
    }
    if (strcmp(key, "split_min") == 0) {
        dseg::options().split_min = value < 2 ? 2 : value;
        return 0;
    }
    if (strcmp(key, "gemm_dbg") == 0) {
        dseg::options().gemm_dbg = value;
        return 0;
    }
    if (strcmp(key, "attn_variant") == 0) {
        dseg::options().attn_variant = value;
        return 0;
    }
    if (strcmp(key, "attn_dbg") == 0) {
        dseg::options().attn_dbg = value;
        return 0;
    }
    dinoseg_set_error("dinoseg_set_option: unknown key '%s'", key);
    return -1;
}

// ------------------------------------------------------------------------------------------------ profiling
extern "C" int dinoseg_profile(dinoseg_handle* h, int32_t level) {
    if (!h || level < 0 || level > 2) {
        dinoseg_set_error("dinoseg_profile: level must be 0, 1 or 2");
        return -1;
    }
    for (auto& r : h->prof_recs) {
        h->prof_pool.push_back(r.a);
        h->prof_pool.push_back(r.b);
    }
    h->prof_recs.clear();
    h->prof_level = level;
    return 0;
}

extern "C" int dinoseg_profile_read(dinoseg_handle* h, float* ms_sum, int32_t* counts) {
    if (!h || !ms_sum || !counts) {
        dinoseg_set_error("dinoseg_profile_read: null argument");
        return -1;
    }
    for (int c = 0; c < DINOSEG_PROF_COUNT; ++c) {
        ms_sum[c] = 0.f;
        counts[c] = 0;
    }
    for (auto& r : h->prof_recs) {
        DSEG_CHECK_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        DSEG_CHECK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        ms_sum[r.cat] += ms;
        counts[r.cat] += 1;
        h->prof_pool.push_back(r.a);
        h->prof_pool.push_back(r.b);
    }
    h->prof_recs.clear();
    return 0;
}

// ------------------------------------------------------------------------------------------------ stand-alone ops
extern "C" int dinoseg_op_pack(const float* src, int32_t rows, int32_t cols, void* dst, int64_t plane_stride,
                               int32_t rows_pad, int32_t cols_pad, int32_t planes, void* stream) {
    return launch_pack_planes(src, rows, cols, reinterpret_cast<bf16_t*>(dst), plane_stride, rows_pad, cols_pad, planes,
                              reinterpret_cast<hipStream_t>(stream), options().op_fmt);
}

extern "C" int dinoseg_op_gemm(const void* A, int64_t a_plane, int32_t lda, const void* Wp, int64_t w_plane, int32_t M,
                               int32_t N, int32_t K, int32_t planes, int32_t epi, const float* bias, float* out_f32,
                               void* out_bf16, int64_t out_plane, int32_t ldo, void* stream) {
    if (epi < EPI_PLAIN || epi > EPI_RELU) {
        dinoseg_set_error("dinoseg_op_gemm: epi must be 0..3");
        return -1;
    }
    GemmParams g = {};
    g.A = reinterpret_cast<const bf16_t*>(A); g.a_plane = a_plane; g.lda = lda;
    g.W = reinterpret_cast<const bf16_t*>(Wp); g.w_plane = w_plane;
    g.M = M; g.N = N; g.K = K; g.planes = planes; g.epi = epi; g.bias = bias;
    g.fmt = options().op_fmt;
    g.out_f32 = out_f32; g.ldo_f32 = N;
    g.out_bf16 = reinterpret_cast<bf16_t*>(out_bf16); g.out_plane = out_plane; g.ldo = ldo;
    return launch_gemm(g, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_op_qkv_gemm(const void* A, int64_t a_plane, const void* Wp, int64_t w_plane, const float* bias,
                                   int32_t B, int32_t ntok, int32_t npad, int32_t heads, int32_t planes, float qscale,
                                   void* q, void* k, void* v, int64_t qkv_plane, void* stream) {
    GemmParams g = {};
    const int D = heads * 64;
    g.A = reinterpret_cast<const bf16_t*>(A); g.a_plane = a_plane; g.lda = D;
    g.W = reinterpret_cast<const bf16_t*>(Wp); g.w_plane = w_plane;
    g.M = B * ntok; g.N = 3 * D; g.K = D; g.planes = planes; g.epi = EPI_QKV; g.bias = bias;
    g.fmt = options().op_fmt;
    g.q = reinterpret_cast<bf16_t*>(q); g.k = reinterpret_cast<bf16_t*>(k); g.v = reinterpret_cast<bf16_t*>(v);
    g.qkv_plane = qkv_plane; g.ntok = ntok; g.npad = npad; g.heads = heads; g.dmodel = D; g.qscale = qscale;
    return launch_gemm(g, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int64_t dinoseg_op_mlp_fused_pack_elems(int32_t D, int32_t F) { return mlp_fused_pack_elems(D, F); }

extern "C" int dinoseg_op_pack_mlp(const float* W1, const float* W2, int32_t D, int32_t F, void* dst, void* stream) {
    if (!W1 || !W2 || !dst) {
        dinoseg_set_error("dinoseg_op_pack_mlp: null pointer");
        return -1;
    }
    return launch_pack_mlp(W1, W2, D, F, reinterpret_cast<bf16_t*>(dst), reinterpret_cast<hipStream_t>(stream), options().op_fmt);
}

extern "C" int dinoseg_op_mlp_fused(float* X, const float* gamma, const float* beta, float eps, const void* Wp, const float* b1,
                                    const float* b2, int32_t M, int32_t D, int32_t F, void* stream) {
    if (!X || !gamma || !beta || !Wp || !b1 || !b2 || !mlp_fused_supported(D, F, 1)) {
        dinoseg_set_error("dinoseg_op_mlp_fused: null pointer or unsupported shape D=%d F=%d", D, F);
        return -1;
    }
    MlpFusedParams g = {};
    g.X = X; g.ldx = D; g.gamma = gamma; g.beta = beta; g.eps = eps;
    g.Wp = reinterpret_cast<const bf16_t*>(Wp); g.b1 = b1; g.b2 = b2; g.M = M; g.fmt = options().op_fmt;
    return launch_mlp_fused2(g, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int64_t dinoseg_op_proj_pack_elems(int32_t D) { return mlp_fused_proj_pack_elems(D); }

extern "C" int dinoseg_op_pack_proj(const float* Wsrc, int32_t D, void* dst, void* stream) {
    if (!Wsrc || !dst || mlp_fused_proj_pack_elems(D) <= 0) {
        dinoseg_set_error("dinoseg_op_pack_proj: null pointer or unsupported width D=%d", D);
        return -1;
    }
    return launch_pack_proj(Wsrc, D, reinterpret_cast<bf16_t*>(dst), reinterpret_cast<hipStream_t>(stream), options().op_fmt);
}

extern "C" int dinoseg_op_proj_mlp_fused(float* X, const void* ctx, const void* Wproj, const float* bproj, const float* gamma,
                                         const float* beta, float eps, const void* Wp, const float* b1, const float* b2, int32_t M,
                                         int32_t D, int32_t F, void* stream) {
    if (!X || !ctx || !Wproj || !bproj || !gamma || !beta || !Wp || !b1 || !b2 || !mlp_fused_supported(D, F, 1) ||
        mlp_fused_proj_pack_elems(D) <= 0) {
        dinoseg_set_error("dinoseg_op_proj_mlp_fused: null pointer or unsupported shape D=%d F=%d", D, F);
        return -1;
    }
    MlpFusedParams g = {};
    g.X = X; g.ldx = D; g.gamma = gamma; g.beta = beta; g.eps = eps;
    g.Wp = reinterpret_cast<const bf16_t*>(Wp); g.b1 = b1; g.b2 = b2; g.M = M;
    g.ctx = reinterpret_cast<const bf16_t*>(ctx); g.Wproj = reinterpret_cast<const bf16_t*>(Wproj); g.bproj = bproj;
    g.fmt = options().op_fmt;
    return launch_mlp_fused2(g, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_op_pack_rs(const float* W, int32_t N, int32_t K, int32_t kind, void* dst, void* stream) {
    return launch_pack_rs(W, N, K, kind, reinterpret_cast<bf16_t*>(dst), reinterpret_cast<hipStream_t>(stream), options().op_fmt);
}

extern "C" int dinoseg_op_gemm_rs(const void* A, int32_t lda, const void* Wp, const float* bias, int32_t M, int32_t N, int32_t K, int32_t epi,
                                  float* x_inout, void* out16, int32_t ldo, void* q, void* k, void* v, int32_t ntok, int32_t npad,
                                  int32_t heads, float qscale, void* stream) {
    GemmParams g = {};
    g.A = reinterpret_cast<const bf16_t*>(A); g.lda = lda; g.W = reinterpret_cast<const bf16_t*>(Wp); g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.planes = 1; g.fmt = options().op_fmt; g.epi = epi;
    g.out_f32 = x_inout; g.ldo_f32 = N; g.out_bf16 = reinterpret_cast<bf16_t*>(out16); g.ldo = ldo;
    g.q = reinterpret_cast<bf16_t*>(q); g.k = reinterpret_cast<bf16_t*>(k); g.v = reinterpret_cast<bf16_t*>(v);
    g.ntok = ntok; g.npad = npad; g.heads = heads; g.dmodel = heads * 64; g.qscale = qscale;
    return launch_gemm_rs(g, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_op_pack_rs_ln(const float* W, const float* gamma, const float* beta, const float* bias, int32_t N, int32_t K, void* dst_w,
                                     float* dst_bias, void* stream) {
    return launch_pack_rs_ln(W, gamma, beta, bias, N, K, reinterpret_cast<bf16_t*>(dst_w), dst_bias, reinterpret_cast<hipStream_t>(stream),
                             options().op_fmt);
}

extern "C" int dinoseg_op_ln_gemm_rs(const float* X, float eps, const void* Wp, const float* bias_folded, int32_t M, int32_t N, int32_t K,
                                     int32_t epi, void* out16, int32_t ldo, void* q, void* k, void* v, int32_t ntok, int32_t npad, int32_t heads,
                                     float qscale, void* stream) {
    if (!X) {
        dinoseg_set_error("dinoseg_op_ln_gemm_rs: null rows");
        return -1;
    }
    GemmParams g = {};
    g.ln_x = X; g.ln_eps = eps;
    g.W = reinterpret_cast<const bf16_t*>(Wp); g.bias = bias_folded;
    g.M = M; g.N = N; g.K = K; g.planes = 1; g.fmt = options().op_fmt; g.epi = epi;
    g.out_bf16 = reinterpret_cast<bf16_t*>(out16); g.ldo = ldo;
    g.q = reinterpret_cast<bf16_t*>(q); g.k = reinterpret_cast<bf16_t*>(k); g.v = reinterpret_cast<bf16_t*>(v);
    g.ntok = ntok; g.npad = npad; g.heads = heads; g.dmodel = heads * 64; g.qscale = qscale;
    return launch_gemm_rs(g, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int64_t dinoseg_op_mlp3_pack_elems(int32_t D, int32_t F) { return mlp_fused3_pack_elems(D, F); }

extern "C" int dinoseg_op_pack_mlp3(const float* Wproj, const float* W1, const float* b1, const float* W2, const float* gamma2, const float* beta2,
                                    const float* Wqkv_next, const float* bqkv_next, const float* gamma1_next, const float* beta1_next, int32_t D,
                                    int32_t F, int32_t fmt, void* dst, void* stream) {
    if (fmt != FMT_BF16 && fmt != FMT_FP16) {
        dinoseg_set_error("dinoseg_op_pack_mlp3: bad operand format %d", fmt);
        return -1;
    }
    MlpFused3Weights w = {Wproj, W1, b1, W2, gamma2, beta2, Wqkv_next, bqkv_next, gamma1_next, beta1_next};
    return launch_pack_mlp3(w, D, F, reinterpret_cast<bf16_t*>(dst), reinterpret_cast<hipStream_t>(stream), fmt);
}

extern "C" int dinoseg_op_proj_mlp_fused3(float* X, const void* ctx, int64_t ctx_plane, const float* bproj, float eps, const void* Wp, const float* b2,
                                          int32_t M, int32_t D, int32_t F, int32_t fmt, void* stream) {
    if (!mlp_fused3_supported(D, F, 2) || (fmt != FMT_BF16 && fmt != FMT_FP16)) {
        dinoseg_set_error("dinoseg_op_proj_mlp_fused3: unsupported shape D=%d F=%d or format %d", D, F, fmt);
        return -1;
    }
    MlpFused3Params g = {};
    g.X = X; g.eps = eps;
    g.Wp = reinterpret_cast<const bf16_t*>(Wp); g.b2 = b2; g.M = M;
    g.ctx = reinterpret_cast<const bf16_t*>(ctx); g.ctx_plane = ctx_plane; g.bproj = bproj; g.fmt = fmt;
    return launch_mlp_fused3(g, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_op_block_tail_fused3(float* X, const void* ctx, int64_t ctx_plane, const float* bproj, float eps, const void* Wp,
                                            const float* b2, void* q, void* k, void* v, int64_t qkv_plane, int32_t B, int32_t ntok, int32_t npad,
                                            int32_t heads, float qscale, int32_t v_bf16, int32_t D, int32_t F, int32_t fmt, void* stream) {
    if (!mlp_fused3_supported(D, F, 2) || (fmt != FMT_BF16 && fmt != FMT_FP16) || !q || B <= 0 || npad % 64 != 0) {
        dinoseg_set_error("dinoseg_op_block_tail_fused3: unsupported shape D=%d F=%d, format %d, or null q", D, F, fmt);
        return -1;
    }
    MlpFused3Params g = {};
    g.X = X; g.eps = eps;
    g.Wp = reinterpret_cast<const bf16_t*>(Wp); g.b2 = b2; g.M = B * ntok;
    g.ctx = reinterpret_cast<const bf16_t*>(ctx); g.ctx_plane = ctx_plane; g.bproj = bproj; g.fmt = fmt;
    g.q = reinterpret_cast<bf16_t*>(q); g.k = reinterpret_cast<bf16_t*>(k); g.v = reinterpret_cast<bf16_t*>(v); g.qkv_plane = qkv_plane;
    g.ntok = ntok; g.npad = npad; g.heads = heads; g.qscale = qscale; g.v_bf16 = v_bf16;
    return launch_mlp_fused3(g, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int64_t dinoseg_op_mlp4_pack_elems(int32_t D, int32_t F) { return mlp_fused4_pack_elems(D, F); }

extern "C" int dinoseg_op_pack_mlp4(const float* Wproj, const float* W1, const float* b1, const float* W2, const float* gamma2, const float* beta2,
                                    const float* Wqkv_next, const float* bqkv_next, const float* gamma1_next, const float* beta1_next, int32_t D,
                                    int32_t F, int32_t fmt, void* dst, void* stream) {
    if (fmt != FMT_BF16 && fmt != FMT_FP16) {
        dinoseg_set_error("dinoseg_op_pack_mlp4: bad operand format %d", fmt);
        return -1;
    }
    MlpFused3Weights w = {Wproj, W1, b1, W2, gamma2, beta2, Wqkv_next, bqkv_next, gamma1_next, beta1_next};
    return launch_pack_mlp4(w, D, F, reinterpret_cast<bf16_t*>(dst), reinterpret_cast<hipStream_t>(stream), fmt);
}

extern "C" int dinoseg_op_proj_mlp_fused4(float* X, const void* ctx, const float* bproj, float eps, const void* Wp, const float* b2, int32_t M,
                                          int32_t D, int32_t F, int32_t fmt, void* stream) {
    if (!mlp_fused4_supported(D, F, 1) || (fmt != FMT_BF16 && fmt != FMT_FP16)) {
        dinoseg_set_error("dinoseg_op_proj_mlp_fused4: unsupported shape D=%d F=%d or format %d", D, F, fmt);
        return -1;
    }
    MlpFused3Params g = {};
    g.X = X; g.eps = eps;
    g.Wp = reinterpret_cast<const bf16_t*>(Wp); g.b2 = b2; g.M = M;
    g.ctx = reinterpret_cast<const bf16_t*>(ctx); g.bproj = bproj; g.fmt = fmt;
    return launch_mlp_fused4(g, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_op_block_tail_fused4(float* X, const void* ctx, const float* bproj, float eps, const void* Wp, const float* b2, void* q, void* k,
                                            void* v, int32_t B, int32_t ntok, int32_t npad, int32_t heads, float qscale, int32_t D, int32_t F,
                                            int32_t fmt, void* stream) {
    if (!mlp_fused4_supported(D, F, 1) || (fmt != FMT_BF16 && fmt != FMT_FP16) || !q || B <= 0 || npad % 64 != 0) {
        dinoseg_set_error("dinoseg_op_block_tail_fused4: unsupported shape D=%d F=%d, format %d, or null q", D, F, fmt);
        return -1;
    }
    MlpFused3Params g = {};
    g.X = X; g.eps = eps;
    g.Wp = reinterpret_cast<const bf16_t*>(Wp); g.b2 = b2; g.M = B * ntok;
    g.ctx = reinterpret_cast<const bf16_t*>(ctx); g.bproj = bproj; g.fmt = fmt;
    g.q = reinterpret_cast<bf16_t*>(q); g.k = reinterpret_cast<bf16_t*>(k); g.v = reinterpret_cast<bf16_t*>(v);
    g.ntok = ntok; g.npad = npad; g.heads = heads; g.qscale = qscale;
    return launch_mlp_fused4(g, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int64_t dinoseg_op_qkv_pack_elems(int32_t D) { return mlp_fused_qkv_pack_elems(D); }

extern "C" int dinoseg_op_pack_qkv(const float* Wsrc, int32_t D, void* dst, void* stream) {
    if (!Wsrc || !dst || mlp_fused_qkv_pack_elems(D) <= 0) {
        dinoseg_set_error("dinoseg_op_pack_qkv: null pointer or unsupported width D=%d", D);
        return -1;
    }
    return launch_pack_qkv(Wsrc, D, reinterpret_cast<bf16_t*>(dst), reinterpret_cast<hipStream_t>(stream), options().op_fmt);
}

extern "C" int dinoseg_op_block_tail_fused(float* X, const void* ctx, const void* Wproj, const float* bproj, const float* gamma2,
                                           const float* beta2, float eps, const void* Wp, const float* b1, const float* b2,
                                           const void* Wqkv, const float* bqkv, const float* gamma1, const float* beta1, void* q, void* k,
                                           void* v, int32_t B, int32_t ntok, int32_t npad, int32_t heads, float qscale, int32_t D,
                                           int32_t F, void* stream) {
    if (!X || !ctx || !Wproj || !bproj || !gamma2 || !beta2 || !Wp || !b1 || !b2 || !Wqkv || !bqkv || !gamma1 || !beta1 || !q || !k ||
        !v || B <= 0 || ntok <= 0 || npad < ntok || npad % 64 != 0 || heads * 64 != D || !mlp_fused_supported(D, F, 1) ||
        mlp_fused_proj_pack_elems(D) <= 0 || mlp_fused_qkv_pack_elems(D) <= 0) {
        dinoseg_set_error("dinoseg_op_block_tail_fused: null pointer or unsupported shape D=%d F=%d heads=%d", D, F, heads);
        return -1;
    }
    MlpFusedParams g = {};
    g.X = X; g.ldx = D; g.gamma = gamma2; g.beta = beta2; g.eps = eps;
    g.Wp = reinterpret_cast<const bf16_t*>(Wp); g.b1 = b1; g.b2 = b2; g.M = B * ntok;
    g.ctx = reinterpret_cast<const bf16_t*>(ctx); g.Wproj = reinterpret_cast<const bf16_t*>(Wproj); g.bproj = bproj;
    g.Wqkv = reinterpret_cast<const bf16_t*>(Wqkv); g.bqkv = bqkv; g.gamma1 = gamma1; g.beta1 = beta1;
    g.q = reinterpret_cast<bf16_t*>(q); g.k = reinterpret_cast<bf16_t*>(k); g.v = reinterpret_cast<bf16_t*>(v);
    g.ntok = ntok; g.npad = npad; g.heads = heads; g.qscale = qscale; g.fmt = options().op_fmt;
    return launch_mlp_fused2(g, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int64_t dinoseg_op_ln_gemm_slab_elems(int32_t N, int32_t K, int32_t planes) {
    return gemm_ln_supported(K, N, planes, EPI_GELU, 0) ? gemm_ln_slab_elems(N, K, planes) : -1;
}

extern "C" int dinoseg_op_pack_slabs(const float* Wsrc, int32_t N, int32_t K, int32_t planes, void* dst, void* stream) {
    if (!Wsrc || !dst || !gemm_ln_supported(K, N, planes, EPI_GELU, 0)) {
        dinoseg_set_error("dinoseg_op_pack_slabs: unsupported shape N=%d K=%d planes=%d", N, K, planes);
        return -1;
    }
    return launch_pack_slabs(Wsrc, N, K, planes, reinterpret_cast<bf16_t*>(dst), reinterpret_cast<hipStream_t>(stream),
                             planes == 1 ? options().op_fmt : FMT_BF16);
}

extern "C" int dinoseg_op_ln_gemm(const float* X, const float* gamma, const float* beta, float eps, const void* Wp, int64_t w_plane,
                                  const float* bias, int32_t M, int32_t N, int32_t K, int32_t planes, int32_t epi, void* out_bf16,
                                  int64_t out_plane, void* q, void* k, void* v, int64_t qkv_plane, int32_t ntok, int32_t npad,
                                  int32_t heads, float qscale, void* a_out, void* aux_out, void* stream) {
    LnGemmParams g = {};
    g.X = X; g.ldx = K; g.gamma = gamma; g.beta = beta; g.eps = eps;
    g.W = reinterpret_cast<const bf16_t*>(Wp); g.w_plane = w_plane; g.bias = bias;
    g.M = M; g.N = N; g.epi = epi;
    g.out_bf16 = reinterpret_cast<bf16_t*>(out_bf16); g.out_plane = out_plane; g.ldo = N;
    g.q = reinterpret_cast<bf16_t*>(q); g.k = reinterpret_cast<bf16_t*>(k); g.v = reinterpret_cast<bf16_t*>(v);
    g.qkv_plane = qkv_plane; g.ntok = ntok; g.npad = npad; g.heads = heads; g.dmodel = heads * 64; g.qscale = qscale;
    g.a_out = reinterpret_cast<bf16_t*>(a_out); g.a_plane = (long)M * K;
    g.aux_out = reinterpret_cast<bf16_t*>(aux_out); g.aux_plane = out_plane;
    g.fmt = planes == 1 ? options().op_fmt : FMT_BF16;
    return launch_gemm_ln(g, K, planes, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_op_attention(const void* q, const void* k, const void* v, int64_t qkv_plane, void* ctx,
                                    int64_t ctx_plane, float* lse, int32_t B, int32_t heads, int32_t ntok, int32_t npad,
                                    int32_t planes, void* stream) {
    AttnParams a = {};
    a.q = reinterpret_cast<const bf16_t*>(q); a.k = reinterpret_cast<const bf16_t*>(k);
    a.v = reinterpret_cast<const bf16_t*>(v); a.qkv_plane = qkv_plane;
    a.ctx = reinterpret_cast<bf16_t*>(ctx); a.ctx_plane = ctx_plane; a.lse = lse;
    a.B = B; a.heads = heads; a.ntok = ntok; a.npad = npad; a.planes = planes;
    a.fmt = options().op_fmt;
    a.v_bf16 = options().op_v_bf16;
    return launch_attention(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_op_layernorm(const float* x, const float* gamma, const float* beta, float eps, int32_t M, int32_t D,
                                    void* out_bf16, int64_t out_plane, int32_t planes, float* out_f32, int32_t drop_cls,
                                    int32_t ntok, void* stream) {
    return launch_layernorm(x, gamma, beta, eps, M, D, reinterpret_cast<bf16_t*>(out_bf16), out_plane, planes, out_f32,
                            drop_cls, ntok, reinterpret_cast<hipStream_t>(stream), options().op_fmt);
}

extern "C" int dinoseg_op_pos_resample(const float* pos_embed, int32_t g, int32_t D, int32_t o, float* out, void* stream) {
    return launch_pos_resample(pos_embed, g, D, o, out, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dinoseg_op_patch_gather(const void* x, int32_t x_kind, int32_t B, int32_t r, void* out, int64_t out_plane,
                                       int32_t planes, void* stream) {
    if (r <= 0 || r % 8 != 0) {
        dinoseg_set_error("Resolution should be a multiple of 8.");
        return -1;
    }
    float mean255[3], inv255[3];
    norm_consts(mean255, inv255);
    return launch_patch_gather(x, x_kind, B, r, mean255, inv255, reinterpret_cast<bf16_t*>(out), out_plane, planes,
                               reinterpret_cast<hipStream_t>(stream), planes == 2 ? options().op_fmt : FMT_BF16);
}

extern "C" int dinoseg_op_head_final(const void* in, int64_t in_plane, int32_t ld, int32_t M, int32_t K, const float* Wc,
                                     const float* b, int32_t C, float* logp, int32_t* argmax, void* stream) {
    return launch_head_final(reinterpret_cast<const bf16_t*>(in), in_plane, ld, M, K, Wc, b, C, logp, argmax,
                             reinterpret_cast<hipStream_t>(stream), options().op_fmt);
}
