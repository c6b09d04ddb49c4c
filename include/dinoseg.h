/* dinoseg.h -- C-ABI of libdinoseg_hip.so: the MI355X (gfx950) DINOSeg hot path.
 *
 * Drop-in boundary for the reference's Python model class (reference = sachaMorin/dino, all
 * citations relative to its root).  The reference has no FFI of its own -- its "operator interface"
 * for this path is the DINOSeg module API -- so each entry point names the Python call it replaces.
 * The host-side mirror (dino_amd/dinoseg.py) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions: extern "C", plain pointers and sizes, no torch / C++ types.  Every function returns
 * 0 on success and a negative code on failure (-1 bad argument, -2 HIP runtime error, -3 state error);
 * dinoseg_last_error() returns the message.  All device pointers are caller-owned (PyTorch-ROCm
 * tensors); the library owns only its packed-weight copies and its activation workspace, released by
 * dinoseg_destroy().  Every call is asynchronous on the caller's `stream` (a hipStream_t passed as
 * void*; pass torch.cuda.current_stream().cuda_stream) and performs no host synchronisation, except
 * the lazy workspace (re)allocation on the first call for a larger (B, r).
 * A handle is not re-entrant: one handle per process per GPU.
 */
#ifndef DINOSEG_H
#define DINOSEG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dinoseg_handle dinoseg_handle;

/* Precision of the GEMM / attention operands (accumulation is always fp32):
 *   DINOSEG_BF16   : bf16 operands, 1 MFMA per product  (benchmark mode, BASELINE.json "bf16")
 *   DINOSEG_BF16X3 : bf16 hi+lo operand pairs, 3 MFMAs per product (~16 mantissa bits): the parity mode
 *                    that meets "argmax identical, |dlogp| <= 1e-3" against the fp32 reference.
 *   DINOSEG_FP16   : fp16 operands (11 significand bits), 1 MFMA per product at the bf16 rate: the linears and Q.K^T on
 *                    v_mfma_f32_32x32x16_f16; the probabilities and V (the P.V product) stay bf16 -- 2^S against the fixed
 *                    reference 0 needs bf16's exponent range; the head runs split (bf16 hi+lo).
 *                    ~6x closer to the reference than DINOSEG_BF16 at the same speed; inference only (the fine-tune
 *                    entry points refuse it: fp16 gradients would need loss scaling).
 *   DINOSEG_FP16X3 : fp16 hi+lo operand pairs everywhere (patch embedding and head included), 3 MFMAs per product at the
 *                    bf16x3 rate, ~22 significand bits instead of ~16: the parity mode with margin (2x closer to the reference
 *                    than DINOSEG_BF16X3 on the goldens, ~8x on ill-conditioned weights -- outlier channels, sharp heads);
 *                    values beyond +-65504 saturate; inference only. */
enum { DINOSEG_BF16 = 0, DINOSEG_BF16X3 = 1, DINOSEG_FP16 = 2, DINOSEG_FP16X3 = 3 };
enum { DINOSEG_HEAD_LINEAR = 0, DINOSEG_HEAD_MLP = 1 };
enum { DINOSEG_INPUT_U8_HWC = 0,      /* uint8 [B,r,r,3] frames; ImageNet normalisation fused on device   */
       DINOSEG_INPUT_F32_CHW = 1 };   /* fp32  [B,3,r,r] already-normalised tensor (DINOSeg.forward input) */

/* Architecture = the reference ctor arguments that shape the path.
 * DINOSeg.__init__(head, n_blocks, n_classes, backbone='vit')   dt_segmentation/src/pl_torch_modules.py:144-222
 * vit_small / vit_base(patch_size=8)                             dt_segmentation/src/vision_transformer.py:300-311 */
typedef struct dinoseg_config {
    int32_t embed_dim;    /* 384 (ViT-S) / 768 (ViT-B); multiple of 128                         */
    int32_t num_heads;    /* embed_dim / 64                                                      */
    int32_t n_blocks;     /* transformer blocks kept (dino.blocks[:n_blocks], :177)              */
    int32_t patch;        /* 8                                                                   */
    int32_t mlp_ratio;    /* 4                                                                   */
    int32_t n_classes;    /* <= 32                                                               */
    int32_t head_kind;    /* DINOSEG_HEAD_*  (pl_torch_modules.py:219-222)                       */
    int32_t pos_grid;     /* 28: stored pos_embed is [1, 28*28+1, D]                             */
    float   ln_eps;       /* 1e-6 (vision_transformer.py:303)                                    */
    int32_t precision;    /* DINOSEG_BF16 / DINOSEG_BF16X3 / DINOSEG_FP16 / DINOSEG_FP16X3       */
} dinoseg_config;

const char* dinoseg_last_error(void);
int dinoseg_version(void);

/* Replaces DINOSeg.__init__ (pl_torch_modules.py:144-237) minus the network fetch (dt_utils.py:19-29). */
int dinoseg_create(const dinoseg_config* cfg, dinoseg_handle** out);
int dinoseg_destroy(dinoseg_handle* h);

/* Replaces load_state_dict() inside LightningModule.load_from_checkpoint (call sites README.md:31,
 * visualize.py:23): bind one fp32 device tensor by its state_dict key ("dino.blocks.0.attn.qkv.weight",
 * "clf.layer_1.bias", ...).  The pointer is borrowed and must stay valid; shape is checked. */
int dinoseg_bind_weight(dinoseg_handle* h, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim);

/* (Re)build the library's packed bf16 operand planes from the bound fp32 tensors.  Call after binding
 * and after every optimiser step (the role `.to(device)` / optimizer.step() play for torch, README.md:31). */
int dinoseg_refresh_weights(dinoseg_handle* h, void* stream);

/* Replaces DINOSeg.set_resolution (pl_torch_modules.py:270-274) on the device side: bicubic pos-embed
 * resample for an (r/8)x(r/8) grid (vision_transformer.py:202-222), cached per resolution.  r % 8 != 0 -> -1. */
int dinoseg_prepare_resolution(dinoseg_handle* h, int32_t r, void* stream);

/* Replaces DINOSeg.forward (pl_torch_modules.py:239-256) and the argmax of predict() (:294):
 *   x        : B frames at r x r, layout per x_kind
 *   logp_out : fp32 [B*(r/8)^2, n_classes] log-probabilities (may be NULL)
 *   argmax_out: int32 [B*(r/8)^2] first-maximum class index (may be NULL)
 *   tap_block / tap_out: optional debug tap -- tap_block = 0 copies the token matrix after prepare_tokens,
 *   i > 0 after block i, into tap_out (fp32 [B*((r/8)^2+1), embed_dim]); pass -1 / NULL to disable. */
int dinoseg_forward(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r, float* logp_out,
                    int32_t* argmax_out, int32_t tap_block, float* tap_out, void* stream);

/* Replaces VisionTransformer.get_last_selfattention (vision_transformer.py:273-280; caller visualize_attention.py:46):
 * the materialised softmax(q k^T / 8) of the LAST block, attn_out fp32 [B, heads, N, N] with N = (r/8)^2 + 1.
 * Visualisation path, not the inference hot path (which never writes the N x N matrix). */
int dinoseg_last_selfattention(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r, float* attn_out,
                               void* stream);

/* Replaces VisionTransformer.forward_mask (vision_transformer.py:250-271) and get_last_selfattention(x, cls_mask)
 * (:273-280) for ONE frame x (kinds as dinoseg_forward): every block but the last runs as usual; in the last block the CLS
 * query attends through each of the n_masks masks (its logits MULTIPLIED by the mask, CLS key by 0; Attention.forward
 * :80-107), the CLS residual is repeated once per mask (Block.forward :127-140), then MLP and the final norm.
 * cls_mask: fp32 [n_masks, (r/8)^2] on device, n_masks <= (r/8)^2.  emb_out: fp32 [n_masks, embed_dim] (nullable);
 * attn_out: fp32 [heads, n_masks, (r/8)^2 + 1] masked attention of the last block (nullable; with emb_out NULL the call
 * stops after the attention, like get_last_selfattention). */
int dinoseg_forward_mask(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t r, const float* cls_mask, int32_t n_masks,
                         float* emb_out, float* attn_out, void* stream);

/* Replaces VisionTransformer.forward(x, all=True, intermediate=k) (vision_transformer.py:237-248), i.e. `model.dino(x)`:
 * tokens through n_blocks blocks (0 = all the handle has) and the final LayerNorm, tokens_out fp32 [B, (r/8)^2 + 1, embed_dim]
 * (row 0 of every frame is the CLS token; DINOSeg.forward takes [:, 1:], pl_torch_modules.py:243). */
int dinoseg_features(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r, int32_t n_blocks,
                     float* tokens_out, void* stream);

/* Replaces the Resize(r, r) of get_transforms (pl_torch_modules.py:36-38, applied in predict at :291) for frames that are
 * not already r x r: uint8 HWC [sh, sw, 3] -> [dh, dw, 3] on device, restating cv2.resize(INTER_LINEAR)'s fixed-point
 * arithmetic (albumentations 1.1.0 -> opencv 4.5.5, third-party: parity unpinned) so that predict() keeps
 * the frame on the wire as uint8 and resizes it ahead of the patch-embedding gather. */
int dinoseg_op_resize_u8(const uint8_t* src, int32_t sh, int32_t sw, uint8_t* dst, int32_t dh, int32_t dw, void* stream);

/* Confusion matrix for the validation metrics (validation_epoch_end, pl_torch_modules.py:310-332):
 * cm[gt][pred] += 1 over n patches; cm int64 [n_classes, n_classes] on device (zero it first). */
int dinoseg_op_confusion(const int32_t* pred, const int64_t* gt, int64_t n, int32_t n_classes, int64_t* cm, void* stream);

/* ---- fine-tune step (replaces DINOSeg.training_step + autograd + optimizer.step, pl_torch_modules.py:258-268) ---- */

/* Bind (or, with NULL, unbind) the fp32 gradient buffer of a parameter, same shape as the bound weight.  A parameter
 * without a bound gradient is frozen; with no "dino.*" gradient bound the backward stops at the head
 * (freeze_bb, pl_torch_modules.py:434-436). */
int dinoseg_bind_grad(dinoseg_handle* h, const char* name, float* dev_ptr);

/* One training step on this rank's B frames: forward with saved activations, loss = F.nll_loss(log_probs, labels)
 * (mean over the patches whose label is not -100; labels int64 on device), backward.  Every bound gradient buffer is OVERWRITTEN with
 * d loss / d parameter; *loss_out (device float) receives the loss; logp_out (optional) the log-probabilities.
 * Call dinoseg_refresh_weights() after the optimiser changed the parameters. */
int dinoseg_train_step(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r, const int64_t* labels,
                       float* loss_out, float* logp_out, void* stream);

/* The two halves of the step, for callers that own the loss (torch.autograd: `loss = F.nll_loss(model(x), y); loss.backward()`,
 * pl_torch_modules.py:261-266).  dinoseg_train_forward = DINOSeg.forward with the activations kept (logp_out fp32
 * [B*(r/8)^2, n_classes]); dinoseg_backward takes dlogp = d loss / d log-probabilities (same shape, fp32, device) and
 * OVERWRITES every bound gradient buffer with d loss / d parameter.  The saved forward stays valid until the next
 * dinoseg_train_forward / dinoseg_train_step on this handle.  dinoseg_train_step(labels) == train_forward + nll_loss +
 * backward, through the same kernels (same d logits bit for bit). */
int dinoseg_train_forward(dinoseg_handle* h, const void* x, int32_t x_kind, int32_t B, int32_t r, float* logp_out, void* stream);
int dinoseg_backward(dinoseg_handle* h, const float* dlogp, void* stream);

/* Gradient stages of the backward, for overlapping the data-parallel all-reduce with it (no reference
 * counterpart: the reference trains on one GPU).  dinoseg_backward / dinoseg_train_step record an event on their stream when the
 * gradients of a stage are final: stage 0 = head (clf.*), stage 1 + k = dino.norm.* and dino.blocks[n_blocks-1-k].*,
 * stage n_blocks + 1 = embeddings (cls_token, pos_embed, patch_embed).  dinoseg_grad_stages returns n_blocks + 2;
 * dinoseg_stream_wait_grad_stage makes `stream` (e.g. the communication stream) wait for stage `stage` of the LAST backward
 * enqueued on this handle, without blocking the host. */
int dinoseg_grad_stages(const dinoseg_handle* h);
int dinoseg_stream_wait_grad_stage(dinoseg_handle* h, int32_t stage, void* stream);

/* Label check of the training steps since the last call.  F.nll_loss ignores rows labelled -100 (ignore_index; the mean is
 * over the other rows) and raises for any other label outside [0, n_classes): the kernels treat such a row as ignored
 * and latch a flag; *bad_labels receives it (1 = at least one out-of-range label was seen) and the flag is cleared.
 * Synchronises `stream`. */
int dinoseg_train_status(dinoseg_handle* h, int32_t* bad_labels, void* stream);

/* Fused Adam (decoupled = 0: torch.optim.Adam, weight decay added to the gradient) / AdamW (decoupled = 1) update of
 * one tensor; step counts from 1; grad_scale multiplies the gradient first (1/world_size after a sum all-reduce). */
int dinoseg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int32_t decoupled, int32_t step, float grad_scale, void* stream);

/* The same update for `count` parameters in one launch per 64 tensors (host arrays of device pointers and element counts):
 * what optimizer.step() does for the whole parameter list (pl_torch_modules.py:258-259). */
int dinoseg_adam_step_multi(int32_t count, float* const* p, const float* const* g, float* const* m, float* const* v,
                            const int64_t* n, float lr, float beta1, float beta2, float eps, float weight_decay,
                            int32_t decoupled, int32_t step, float grad_scale, void* stream);

/* Per-kernel-class timing with HIP events recorded on the forward's stream (used by bench.py for the
 * roofline leg).  level 0 = off, 1 = the dominant kernel only (fused attention), 2 = every class.
 * dinoseg_profile_read() waits for the recorded events, writes the summed milliseconds and launch counts
 * per class (arrays of DINOSEG_PROF_COUNT) and clears the records. */
enum { DINOSEG_PROF_PATCH = 0, DINOSEG_PROF_LN = 1, DINOSEG_PROF_QKV = 2, DINOSEG_PROF_ATTN = 3, DINOSEG_PROF_PROJ = 4,
       DINOSEG_PROF_FC1 = 5, DINOSEG_PROF_FC2 = 6, DINOSEG_PROF_HEAD = 7,
       DINOSEG_PROF_ATTN_BWD = 8,      /* the flash-style attention backward of the fine-tune step: prep + dQ + dK,dV kernels */
       DINOSEG_PROF_COUNT = 9 };
int dinoseg_profile(dinoseg_handle* h, int32_t level);
int dinoseg_profile_read(dinoseg_handle* h, float* ms_sum, int32_t* counts);

/* Process-wide switches.  Keys:
 *   "streams"    2 [default] / 1: with 2, dinoseg_forward runs a batch of >= "split_min" (default 8) frames as two half-batches,
 *                the first on the caller's stream, the second on an internal stream forked from / joined to it by events (the
 *                call stays stream-ordered and capturable; outputs bit-identical to the one-stream run in every precision -- each
 *                half takes the kernel routes of the whole batch: tests/test_model_gpu.py::test_two_stream_split_equals_one_stream*;
 *                +6 to +9 % frames/s at B = 32 on MI355X: one half's attention fills the CUs the other half's GEMM tails and
 *                memory phases leave idle); 1 = one stream;
 *   "fp16_patch_planes" 1 [default] / 2: precision fp16 only -- the patch embedding on one fp16 plane, or on bf16 hi+lo planes;
 *   "op_fmt"     0 [default] / 1: operand format of the single-plane stand-alone ops (dinoseg_op_*: tests, tools): bf16 / fp16;
 *   "gemm_big"   1 [default] = the persistent 256x384 (bf16) / 128x384 (bf16x3) GEMM where it applies, 0 = always the 128x128
 *                kernel, 2 = wherever its shape rules allow;
 *   "gemm_ln"    1 [default] = qkv / fc1 through the LayerNorm-fused kernels where measured faster, 0 never, 2 wherever supported;
 *   "mlp_fused"  1 [default] = the MLP half of a block (LayerNorm2, fc1, GELU, fc2, residual) as ONE launch where it applies (bf16
 *                mode, embed_dim 384, batches of >= 4 frames at 480x480), 0 never, 2 wherever the shape allows;
 *   "proj_fused" 1 [default] = where that launch runs, it also carries the block's attention output projection + residual
 *                (x += proj(ctx) + b first: vision_transformer.py:104-105), 0 = the projection stays a GEMM launch of its own;
 *   "qkv_fused"  0 [default] / 1 = ... and LayerNorm1 + qkv of the next block at its end (blocks 1.. have no LN + qkv launch then:
 *                +1 % frames/s on one stream, none on two -- the tail writes Q / K / V in the same HBM burst as the launch it replaces);
 *   "mlp_variant" accepted and ignored (the one-wave-per-SIMD build of the fused MLP kernel was removed in round 4);
 *   "train_streams" 2 [default] = dinoseg_backward / dinoseg_train_step run the blocks' weight-gradient GEMMs on an internal stream
 *                beside the input-gradient chain (forked from / joined to the caller's stream by events: stream-ordered, capturable),
 *                1 = everything on the caller's stream (use it when several processes share one GPU);
 *   "deterministic" 0 [default] / 1: the fine-tune step sums the loss, the bias gradients and the LayerNorm gamma / beta gradients from
 *                per-block partials in a FIXED order instead of fp32 atomics (the side stream's bias sums in their own scratch region):
 *                two runs from the same state are bit-identical, at 4-5 % of the step (SURVEY.md section 8e "fixed reduction tree"; with world_size > 1 the
 *                all-reduce's own order is RCCL's); one training step at a time per process while it is on;
 *   "op_v_bf16"  0 [default] / 1: dinoseg_op_attention with fp16 hi + lo planes takes V as bf16 hi + lo planes (what the forward hands
 *                the zero-reference kernels at large batch: attention_za.hip);
 *   "splitk_tiles" 512 [default]: partial 128x128 tiles of one weight-gradient GEMM (<= 768);
 *   "route_ab"   0 [default]: A/B switches of dispatch routes that do not change results (bit 0: 128-row tiles for the residual GEMMs
 *                of a small batch; bit 1: the one-wave-per-row LayerNorm backward; bit 2: the weight-gradient GEMM's 2-D grid);
 *   "attn_variant", "gemm_dbg", "attn_dbg": kernel A/B and timing-ablation switches (tools/bench_ops.py; attn_variant's bits:
 *                dino_amd/csrc/kernels.h -- default 11 | 1024 | 65536: from four rounds of workgroups on the attention runs its tile loop
 *                as a generated assembly pipeline, attention_za.hip, bit-identical to the compiled kernel). */
int dinoseg_set_option(const char* key, int32_t value);

/* Bytes of library-owned device memory a (B, r) forward needs (activations + packed weights). */
int64_t dinoseg_workspace_bytes(const dinoseg_handle* h, int32_t B, int32_t r);
/* Counts the events that invalidate device addresses or cached contents a CAPTURED dinoseg_forward has baked in: a re-allocation of
 * the activation workspace or of the packed weights, a re-computation of the resampled position embedding (another resolution).
 * A caller that replays a HIP graph of the forward compares it with the value read after the capture and re-captures on a change
 * (DINOSeg.predict does: pl_torch_modules.py:276-300 is a single-frame call, launch-bound when issued kernel by kernel). */
int64_t dinoseg_state_generation(const dinoseg_handle* h);

/* ---- stand-alone operators (same kernels the forward uses; exported for unit parity tests) ------------- */

/* fp32 [rows, cols] -> bf16 planes [planes][rows_pad][cols_pad] (zero padded); plane stride in elements */
int dinoseg_op_pack(const float* src, int32_t rows, int32_t cols, void* dst, int64_t plane_stride, int32_t rows_pad,
                    int32_t cols_pad, int32_t planes, void* stream);

/* C[M,N] = A[M,K] . W[N,K]^T on packed planes.  epi: 0 plain(+bias) -> out_f32; 1 residual: out_f32 += acc+bias;
 * 2 GELU(erf) -> out_bf16 planes; 3 ReLU -> out_bf16 planes.  (aten::addmm of vision_transformer.py:60-63,75,105) */
int dinoseg_op_gemm(const void* A, int64_t a_plane, int32_t lda, const void* W, int64_t w_plane, int32_t M, int32_t N,
                    int32_t K, int32_t planes, int32_t epi, const float* bias, float* out_f32, void* out_bf16,
                    int64_t out_plane, int32_t ldo, void* stream);

/* attn.qkv GEMM with the head-scatter epilogue (vision_transformer.py:82): Q (times qscale), K, V, each
 * [planes][B,H,npad,64]; M = B*ntok rows. */
int dinoseg_op_qkv_gemm(const void* A, int64_t a_plane, const void* W, int64_t w_plane, const float* bias, int32_t B,
                        int32_t ntok, int32_t npad, int32_t heads, int32_t planes, float qscale, void* q, void* k,
                        void* v, int64_t qkv_plane, void* stream);

/* Slab-major bf16 copy of an fp32 weight W [N][K] for dinoseg_op_ln_gemm (dst holds dinoseg_op_ln_gemm_slab_elems(N, K, planes)
 * bf16 elements; -1 = unsupported shape): [column tile][k-step][plane][rows][32 k], pre-swizzled, zero padded. */
int64_t dinoseg_op_ln_gemm_slab_elems(int32_t N, int32_t K, int32_t planes);
int dinoseg_op_pack_slabs(const float* W, int32_t N, int32_t K, int32_t planes, void* dst, void* stream);

/* LayerNorm (eps as given) fused into the GEMM that consumes it: out = epilogue(LN(X) W^T + bias), X fp32 [M, K] (K = 384);
 * Wp = the slab-major copy made by dinoseg_op_pack_slabs (w_plane is ignored).
 * epi 4 (QKV): scatter to q (x qscale), k, v [planes][B*heads][npad][64]; epi 2 (GELU): out_bf16 [planes][M][N].
 * a_out / aux_out (nullable): the normalised planes [planes][M][K] / the pre-GELU planes, kept by training forwards.
 * Replaces nn.LayerNorm + nn.Linear pairs vision_transformer.py:123 -> :75 and :135 -> :60-61. */
int dinoseg_op_ln_gemm(const float* X, const float* gamma, const float* beta, float eps, const void* Wp, int64_t w_plane,
                       const float* bias, int32_t M, int32_t N, int32_t K, int32_t planes, int32_t epi, void* out_bf16,
                       int64_t out_plane, void* q, void* k, void* v, int64_t qkv_plane, int32_t ntok, int32_t npad,
                       int32_t heads, float qscale, void* a_out, void* aux_out, void* stream);

/* The whole MLP half of a block in one launch (bf16 mode, embed_dim 384, hidden 1536):  X += fc2(gelu(fc1(LayerNorm(X)))),
 * X fp32 [M, 384] updated in place.  Replaces Block.forward's `x = x + self.mlp(self.norm2(x))` (vision_transformer.py:135 ->
 * :59-65): LayerNorm2, fc1, exact GELU (fitted form of the bf16 mode), fc2, residual add; the hidden activation stays on chip.
 * Wp = both weights re-packed in MFMA fragment order by dinoseg_op_pack_mlp (dinoseg_op_mlp_fused_pack_elems(D, F) bf16
 * elements; 0 = unsupported shape). */
int64_t dinoseg_op_mlp_fused_pack_elems(int32_t D, int32_t F);
int dinoseg_op_pack_mlp(const float* W1, const float* W2, int32_t D, int32_t F, void* dst, void* stream);
int dinoseg_op_mlp_fused(float* X, const float* gamma, const float* beta, float eps, const void* Wp, const float* b1,
                         const float* b2, int32_t M, int32_t D, int32_t F, void* stream);

/* The same launch with the block's attention output projection in front (role-split kernel only):
 *     X += ctx . Wproj^T + bproj;   X += fc2(gelu(fc1(LayerNorm(X))))
 * = Attention.forward's `x = self.proj(x)` + Block.forward's two residual adds (vision_transformer.py:104-105, :123, :135).
 * ctx: bf16 [M, 384] (the attention output, row stride 384); Wproj: the [384, 384] weight re-packed by dinoseg_op_pack_proj
 * (dinoseg_op_proj_pack_elems(D) bf16 elements; 0 = unsupported width).  Library option "proj_fused" (default 1) makes
 * dinoseg_forward use it wherever the fused MLP runs. */
int64_t dinoseg_op_proj_pack_elems(int32_t D);
int dinoseg_op_pack_proj(const float* W, int32_t D, void* dst, void* stream);
int dinoseg_op_proj_mlp_fused(float* X, const void* ctx, const void* Wproj, const float* bproj, const float* gamma,
                              const float* beta, float eps, const void* Wp, const float* b1, const float* b2, int32_t M,
                              int32_t D, int32_t F, void* stream);

/* Row-stationary streaming GEMMs of the wide model (gemm_rs.hip; one operand plane in the format of option "op_fmt", embed_dim 768): what
 * nn.Linear computes in Attention.qkv / Mlp.fc1 (K = 768: epi 4 = the Q / K / V scatter of dinoseg_op_qkv_gemm, epi 2 = GELU into out16 [M][ldo])
 * and in Attention.proj / Mlp.fc2 with the residual add (N = 768, epi 1: x_inout [M][768] += A W^T + bias) -- vision_transformer.py:75, :60-61, :105,
 * :63 + :123 / :135.  Wp: the fp32 weight [N][K] re-packed by dinoseg_op_pack_rs (N * K 16-bit elements; kind 0 for epi 2 / 4, kind 1 for epi 1).
 * dinoseg_forward uses them for ViT-B/8 batches of >= option "gemm_rs_min_rows" rows; option "gemm_rs" is a bit per linear (1 mlp.fc1,
 * 2 attn.qkv, 4 attn.proj + mlp.fc2; default 3: the two that measure faster than the generic kernel). */
int dinoseg_op_pack_rs(const float* W, int32_t N, int32_t K, int32_t kind, void* dst, void* stream);
/* ... with the LayerNorm in front of the linear inside the launch (epi 2 / 4 only): `self.qkv(self.norm1(x))` / `self.fc1(self.norm2(x))`
 * (vision_transformer.py:122 -> :75, :134 -> :60), X fp32 [M][K] rows, K = 768.  The kernel's prologue computes (x - mean) rstd per row; the LayerNorm's
 * weight and bias ride in the packed copy: dinoseg_op_pack_rs_ln writes W . diag(gamma) in fragment order (N * K 16-bit elements) and the folded bias
 * bias + W beta ([N] fp32) -- LayerNorm(x) W^T + b = ((x - mean) rstd) (W diag(gamma))^T + (b + W beta).  Library option "gemm_rs_ln" (default 1, read by
 * dinoseg_refresh_weights). */
int dinoseg_op_pack_rs_ln(const float* W, const float* gamma, const float* beta, const float* bias, int32_t N, int32_t K, void* dst_w,
                          float* dst_bias, void* stream);
int dinoseg_op_ln_gemm_rs(const float* X, float eps, const void* Wp, const float* bias_folded, int32_t M, int32_t N, int32_t K, int32_t epi,
                          void* out16, int32_t ldo, void* q, void* k, void* v, int32_t ntok, int32_t npad, int32_t heads, float qscale,
                          void* stream);
int dinoseg_op_gemm_rs(const void* A, int32_t lda, const void* Wp, const float* bias, int32_t M, int32_t N, int32_t K, int32_t epi,
                       float* x_inout, void* out16, int32_t ldo, void* q, void* k, void* v, int32_t ntok, int32_t npad, int32_t heads,
                       float qscale, void* stream);

/* The same fusion on hi + lo operand planes (the parity modes; mlp_fused3.hip), one launch for
 *     X += ctx . Wproj^T + bproj;   X += fc2(gelu(fc1(LayerNorm(X))))      (vision_transformer.py:104-105, :123, :135 -> :59-65)
 * ctx: the attention output as two planes [2][M][384] (hi, then lo at + ctx_plane elements), or null = the MLP half only (bproj unused).
 * Wp: dinoseg_op_pack_mlp3's copy (dinoseg_op_mlp3_pack_elems(D, F) 16-bit elements; 0 = unsupported shape): Wproj, W1, W2 -- and optionally the NEXT
 * block's Wqkv [1152, 384] -- as hi + lo fragment pairs in the order the kernel walks them, with the LayerNorms folded in: norm2's weight into the columns
 * of W1 and its bias into b1, the next block's norm1 into Wqkv / bqkv (LayerNorm(x) W^T + b = ((x - mean) rstd) (W diag(gamma))^T + (b + W beta)); the kernel
 * computes (x - mean) rstd only.  fmt: 0 = bf16 planes, 1 = fp16 planes.
 * dinoseg_op_block_tail_fused3: ... and LayerNorm1 + the qkv projection of the NEXT block at the end of the same launch
 * (vision_transformer.py:122 -> :75): afterwards X holds the block's output and q / k / v (each two planes [2][B, heads, npad, 64], lo at
 * + qkv_plane elements; q pre-scaled by qscale; rows >= ntok untouched) hold what LayerNorm + the qkv GEMM would have written from it;
 * v_bf16 (fmt 1 only): V as bf16 planes, what the zero-reference hi + lo attention reads.  M = B * ntok rows.  Library option "qkv_fused3"
 * (default 1) makes dinoseg_forward use it wherever the hi + lo fused launch runs. */
int64_t dinoseg_op_mlp3_pack_elems(int32_t D, int32_t F);
int dinoseg_op_pack_mlp3(const float* Wproj, const float* W1, const float* b1, const float* W2, const float* gamma2, const float* beta2,
                         const float* Wqkv_next, const float* bqkv_next, const float* gamma1_next, const float* beta1_next, int32_t D, int32_t F,
                         int32_t fmt, void* dst, void* stream);
int dinoseg_op_proj_mlp_fused3(float* X, const void* ctx, int64_t ctx_plane, const float* bproj, float eps, const void* Wp, const float* b2,
                               int32_t M, int32_t D, int32_t F, int32_t fmt, void* stream);
int dinoseg_op_block_tail_fused3(float* X, const void* ctx, int64_t ctx_plane, const float* bproj, float eps, const void* Wp, const float* b2,
                                 void* q, void* k, void* v, int64_t qkv_plane, int32_t B, int32_t ntok, int32_t npad, int32_t heads, float qscale,
                                 int32_t v_bf16, int32_t D, int32_t F, int32_t fmt, void* stream);

/* The single-plane fusion with ONE wave per SIMD (mlp_fused4.hip): the same result as dinoseg_op_proj_mlp_fused (fp16 / bf16 operands, the
 * logistic GELU of the benchmark modes), the structure of the hi + lo kernel above -- 128-row items, 32 rows per wave held in registers for
 * the whole item, the weights as one linear stream of 48-KiB slots, LayerNorm2's weight folded into W1's columns and its bias into b1 (the kernel
 * computes (x - mean) rstd).  ctx: [M][384] in the operand format, or null = the MLP half only.  Wp: dinoseg_op_pack_mlp4
 * (dinoseg_op_mlp4_pack_elems(D, F) 16-bit elements; Wproj may be null with ctx == null).  Library option "mlp_fused4" (default 0: it measures
 * equal; set before dinoseg_refresh_weights) makes dinoseg_forward use it instead of dinoseg_op_proj_mlp_fused.
 * vision_transformer.py:104-105, :123, :135 -> :59-65. */
int64_t dinoseg_op_mlp4_pack_elems(int32_t D, int32_t F);
int dinoseg_op_pack_mlp4(const float* Wproj, const float* W1, const float* b1, const float* W2, const float* gamma2, const float* beta2,
                         const float* Wqkv_next, const float* bqkv_next, const float* gamma1_next, const float* beta1_next, int32_t D, int32_t F,
                         int32_t fmt, void* dst, void* stream);
int dinoseg_op_proj_mlp_fused4(float* X, const void* ctx, const float* bproj, float eps, const void* Wp, const float* b2, int32_t M, int32_t D,
                               int32_t F, int32_t fmt, void* stream);
/* ... and LayerNorm1 + the qkv projection of the NEXT block at the end of the same launch (Wp packed with Wqkv_next; as
 * dinoseg_op_block_tail_fused3 on one plane: q / k / v [B, heads, npad, 64] in the operand format, V as bf16, q pre-scaled, rows >= ntok untouched).
 * Library option "qkv_fused4" (default 1). */
int dinoseg_op_block_tail_fused4(float* X, const void* ctx, const float* bproj, float eps, const void* Wp, const float* b2, void* q, void* k, void* v,
                                 int32_t B, int32_t ntok, int32_t npad, int32_t heads, float qscale, int32_t D, int32_t F, int32_t fmt,
                                 void* stream);

/* ... and with LayerNorm1 + the qkv projection of the NEXT block at its end (Block.forward of block i from `x = x + attn` on, then
 * block i+1 up to `qkv = self.qkv(self.norm1(x))`: vision_transformer.py:123, :135, :122 -> :75): after the launch X holds block i's
 * output and q / k / v ([B, heads, npad, 64] bf16 each, q pre-scaled by qscale = 64^-0.5 * log2(e), rows >= ntok untouched) hold what
 * dinoseg_op_ln_gemm(EPI_QKV) would have written from it.  M = B * ntok rows.  Wqkv: the [1152, 384] weight re-packed by
 * dinoseg_op_pack_qkv (dinoseg_op_qkv_pack_elems(D) bf16 elements).  Library option "qkv_fused" (default 0: see there). */
int64_t dinoseg_op_qkv_pack_elems(int32_t D);
int dinoseg_op_pack_qkv(const float* W, int32_t D, void* dst, void* stream);
int dinoseg_op_block_tail_fused(float* X, const void* ctx, const void* Wproj, const float* bproj, const float* gamma2,
                                const float* beta2, float eps, const void* Wp, const float* b1, const float* b2, const void* Wqkv,
                                const float* bqkv, const float* gamma1, const float* beta1, void* q, void* k, void* v, int32_t B,
                                int32_t ntok, int32_t npad, int32_t heads, float qscale, int32_t D, int32_t F, void* stream);

/* fused softmax(q k^T) v (vision_transformer.py:85,101,104); q must be pre-scaled by 64^-0.5 * log2(e).
 * q, k, v: [planes][B,heads,npad,64] (rows >= ntok zero); ctx: bf16 planes [planes][B*ntok][heads*64];
 * lse (optional): fp32 [B,heads,ntok], log2 domain. */
int dinoseg_op_attention(const void* q, const void* k, const void* v, int64_t qkv_plane, void* ctx, int64_t ctx_plane,
                         float* lse, int32_t B, int32_t heads, int32_t ntok, int32_t npad, int32_t planes, void* stream);

/* nn.LayerNorm over the last dim (vision_transformer.py:303).  out_bf16 / out_f32 may each be NULL. */
int dinoseg_op_layernorm(const float* x, const float* gamma, const float* beta, float eps, int32_t M, int32_t D,
                         void* out_bf16, int64_t out_plane, int32_t planes, float* out_f32, int32_t drop_cls,
                         int32_t ntok, void* stream);

/* interpolate_pos_encoding (vision_transformer.py:202-222): pos_embed fp32 [g*g+1, D] -> out fp32 [o*o+1, D] */
int dinoseg_op_pos_resample(const float* pos_embed, int32_t g, int32_t D, int32_t o, float* out, void* stream);

/* patch gather (+ fused Normalize for uint8 input) -> bf16 planes [planes][B*(r/8)^2][192] */
int dinoseg_op_patch_gather(const void* x, int32_t x_kind, int32_t B, int32_t r, void* out, int64_t out_plane,
                            int32_t planes, void* stream);

/* last Linear + log_softmax + argmax (pl_torch_modules.py:122-123,:294); in: hi/lo planes [2][M][ld] */
int dinoseg_op_head_final(const void* in, int64_t in_plane, int32_t ld, int32_t M, int32_t K, const float* W,
                          const float* b, int32_t C, float* logp, int32_t* argmax, void* stream);

/* flash-attention backward: q,k,v as the forward; dO, O: ctx-layout planes [planes][B*ntok][heads*64]; lse from the
 * forward; scratch: 2*B*heads*npad floats; dqkv out: planes [planes][B*ntok][3*heads*64] (gradient of the qkv
 * projection output, Q|K|V columns). */
int dinoseg_op_attention_bwd(const void* q, const void* k, const void* v, int64_t qkv_plane, const void* dO, const void* O,
                             int64_t o_plane, const float* lse, float* scratch, void* dqkv, int64_t dqkv_plane, int32_t B,
                             int32_t heads, int32_t ntok, int32_t npad, int32_t planes, void* stream);

/* native_layer_norm_backward: dx (+)= ..., dgamma += ..., dbeta += ... (atomics; zero them first) */
int dinoseg_op_layernorm_bwd(const float* dy, const float* x, const float* gamma, float eps, int32_t M, int32_t D, float* dx,
                             int32_t accumulate, float* dgamma, float* dbeta, int32_t drop_cls, int32_t ntok, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DINOSEG_H */
