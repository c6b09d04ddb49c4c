// Internal: the handle behind the C-ABI (shared by api.hip and train_api.hip).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/dinoseg.h"
#include "common.h"
#include "kernels.h"

using namespace dseg;   // internal header: only included by the two API translation units

#define DSEG_TRY(expr)            \
    do {                          \
        int _rc = (expr);         \
        if (_rc != 0) return _rc; \
    } while (0)

// ------------------------------------------------------------------------------------------------ handle
struct BoundTensor {
    const float* ptr = nullptr;
    std::vector<int64_t> shape;
};

struct PackedLinear {       // W[N,K] operand planes of one nn.Linear, padded to the GEMM tile
    bf16_t* w = nullptr;
    long plane = 0;
    int n_pad = 0, k_pad = 0;
    float* bias_pad = nullptr;   // only when N was padded (head layers); else the bound bias is used
};

struct dinoseg_handle {
    dinoseg_config cfg;
    int planes;
    int fmt = 0;                // operand format: FMT_BF16 / FMT_FP16 (DINOSEG_FP16: planes = 1, DINOSEG_FP16X3: planes = 2, both fmt = FMT_FP16)
    int device = -1;            // ordinal of the GPU that owns the bound tensors (set by the first dinoseg_bind_weight)
    std::map<std::string, BoundTensor> bound;
    std::map<std::string, std::vector<int64_t>> expected;
    // packed weights (library-owned)
    char* wbuf = nullptr;
    size_t wbuf_bytes = 0;
    std::map<std::string, PackedLinear> packed;
    std::map<std::string, bf16_t*> packed_slab;     // slab-major copies of the LN-fed weights (gemm_ln.hip), when supported
    std::map<std::string, bf16_t*> packed_mlp;      // per block ("dino.blocks.i."): fc1 + fc2 in MFMA fragment order (mlp_fused2.hip)
    std::map<std::string, bf16_t*> packed_proj;     // per block: attn.proj.weight in the same fragment order (mlp_fused2.hip, PROJ)
    std::map<std::string, bf16_t*> packed_rs;       // per Linear weight name, one-plane modes at embed_dim 768: the fragment-order copy gemm_rs.hip streams
    std::map<std::string, float*> packed_rs_bias;   // ... those of them that carry the LayerNorm in front of the linear (qkv / fc1, option gemm_rs_ln at refresh): the folded bias
    std::map<std::string, bf16_t*> packed_mlp4;     // per block, one-plane modes: attn.proj + fc1 + fc2 as the slot stream of mlp_fused4.hip
    std::map<std::string, bf16_t*> packed_mlp3;     // per block, hi + lo modes: attn.proj + fc1 + fc2 as the slot stream of mlp_fused3.hip
    std::map<std::string, bf16_t*> packed_qkvf;     // per block: attn.qkv.weight in fragment order (mlp_fused2.hip, QKV tail of the block before)
    bool packed_mlp_stale = false;         // the fragment-order packs are made on the first forward that uses them (the fine-tune
                                           // step refreshes the weights every step and never runs the fused MLP kernel)
    bool weights_ready = false;
    int fp16_patch_planes_snap = 1;        // option fp16_patch_planes as of the last dinoseg_refresh_weights (what the packs were made for)
    int64_t generation = 0;     // dinoseg_state_generation: bumped when an address or cached content a captured forward bakes in changes
    // pos-embed cache
    float* pos_cache = nullptr;
    int pos_r = -1;             // resolution the cache holds (-1: nothing)
    bool pos_stale = false;     // dino.pos_embed was (re)bound since the cache was filled
    size_t pos_cap = 0;
    // activation workspace (library-owned)
    char* ws = nullptr;
    size_t ws_bytes = 0;
    int ws_B = -1, ws_r = -1;
    // second half-batch of a split forward (option "streams" = 2): its own workspace, an internal stream, fork / join events
    char* ws2 = nullptr;
    size_t ws2_bytes = 0;
    int ws2_B = -1, ws2_r = -1;
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool in_split = false;                 // a split forward is being queued (both halves' launches share the chip)
    // optional per-kernel-class timing with HIP events on the caller's stream (bench.py roofline leg)
    // fine-tune step state (train_api.hip)
    std::map<std::string, float*> grads;   // bound gradient buffers (absent / null = frozen tensor)
    char* tws = nullptr;                   // training workspace: saved activations + backward scratch
    size_t tws_bytes = 0;
    int tws_B = -1, tws_r = -1;
    int tr_B = -1, tr_r = -1;              // batch / resolution of the saved forward dinoseg_backward will differentiate
    int* bad_label_flag = nullptr;         // sticky "a label outside [0, C) other than -100 was seen" (device int, owned by the handle:
                                           // it must survive the training workspace being re-laid out for another batch shape)
    // gradient-stage events of the last backward (dinoseg_stream_wait_grad_stage): stage 0 = head, 1 + k = final norm and block
    // n_blocks-1-k, n_blocks + 1 = embeddings; stage_done = number of stages the last backward recorded
    std::vector<hipEvent_t> stage_ev;
    int stage_done = 0;
    // fork / join events of the weight-gradient side stream (option train_streams = 2: train_api.hip); aux_stream is shared with
    // the split forward
    std::vector<hipEvent_t> bw_ev;
    char* twbuf = nullptr;                 // transposed packed weights for the input-gradient GEMMs
    size_t twbuf_bytes = 0;
    int prof_level = 0;                    // 0 off, 1 attention only, 2 every class
    struct ProfRec { int cat; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
};

// Every entry point that takes a handle runs on the handle's device, whatever the caller's current device is (the reference's
// model.to('cuda:1') pattern leaves the current device at 0): allocations, hipFuncSetAttribute and launches all follow it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const dinoseg_handle* h) {
        if (!h || h->device < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != h->device) switched = hipSetDevice(h->device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
// a non-default stream must live on the handle's device: a launch on another device's stream with this device's pointers faults
static inline int check_stream_device(const dinoseg_handle* h, hipStream_t s) {
    if (!h || h->device < 0 || s == nullptr) return 0;
    hipDevice_t d = -1;
    if (hipStreamGetDevice(s, &d) != hipSuccess) return 0;      // legacy / per-thread default stream handles
    if ((int)d != h->device) {
        dinoseg_set_error("stream belongs to device %d but the model's tensors live on device %d", (int)d, h->device);
        return -1;
    }
    return 0;
}

// the handle's side stream + its fork / join events (api.hip): created on first use, destroyed with the workspaces
int ensure_aux_stream(dinoseg_handle* h);

static inline int prof_begin(dinoseg_handle* h, int cat, hipStream_t s) {
    if (h->prof_level == 0 || (h->prof_level == 1 && cat != DINOSEG_PROF_ATTN)) return -1;
    hipEvent_t ev[2];
    for (int i = 0; i < 2; ++i) {
        if (!h->prof_pool.empty()) {
            ev[i] = h->prof_pool.back();
            h->prof_pool.pop_back();
        } else if (hipEventCreate(&ev[i]) != hipSuccess) {
            return -1;
        }
    }
    h->prof_recs.push_back({cat, ev[0], ev[1]});
    (void)hipEventRecord(ev[0], s);
    return (int)h->prof_recs.size() - 1;
}
static inline void prof_end(dinoseg_handle* h, int idx, hipStream_t s) {
    if (idx >= 0) (void)hipEventRecord(h->prof_recs[idx].b, s);
}
#define DSEG_PROF(cat, stmt)                  \
    do {                                      \
        const int _pi = prof_begin(h, cat, s); \
        stmt;                                 \
        prof_end(h, _pi, s);                  \
    } while (0)


static inline int head_planes() { return 2; }
// planes of the patch-embedding GEMM: the mode's own, except that the fp16 mode runs it split like the head (raw pixel operands,
// 0.13 % of the FLOPs)
static inline int patch_planes(const dinoseg_handle* h) { return (h->fmt == FMT_FP16 && h->fp16_patch_planes_snap == 2) ? 2 : h->planes; }
// format of the split (two-plane) operands of the head (and of the patch embedding where it runs split): fp16 only in the fp16 hi+lo
// mode -- the single-plane fp16 mode keeps them bf16 hi+lo (their error is far below that mode's)
static inline int split_fmt(const dinoseg_handle* h) { return h->planes == 2 ? h->fmt : (int)FMT_BF16; }
// format of the patch-embedding operands: the mode's own when it runs on the mode's planes, the split format otherwise
static inline int patch_fmt(const dinoseg_handle* h) { return patch_planes(h) == h->planes ? h->fmt : split_fmt(h); }   // the classifier head always runs in split precision (it is tiny)
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline const float* W(const dinoseg_handle* h, const std::string& k) { return h->bound.at(k).ptr; }
static inline void norm_consts(float mean255[3], float inv255[3]) {
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c) {
        mean255[c] = mean[c] * 255.0f;          // albumentations: mean * max_pixel_value (fp32)
        inv255[c] = 1.0f / (sd[c] * 255.0f);    // reciprocal of std * max_pixel_value (fp32)
    }
}
int dinoseg_train_release(dinoseg_handle* h);   // train_api.hip: frees the training workspace
