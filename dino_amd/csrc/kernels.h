// Host-side launch interface of the dinoseg HIP kernels (internal; the public C-ABI is include/dinoseg.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dseg {

typedef uint16_t bf16_t;

enum GemmEpilogue {
    EPI_PLAIN = 0,   // out_f32[M,N] = acc (+ bias)
    EPI_RESID = 1,   // out_f32[M,N] += acc + bias            (attn.proj / mlp.fc2 + residual add)
    EPI_GELU = 2,    // out_bf16 planes = gelu_erf(acc + bias) (mlp.fc1)
    EPI_RELU = 3,    // out_bf16 planes = relu(acc + bias)     (head layer_1 / layer_2)
    EPI_QKV = 4,     // scatter to Q (pre-scaled), K, V : each [B,H,Npad,64]
    EPI_PATCH = 5,   // token rows: X[b*(n+1)+1+p, :] = acc + bias + pos[1+p, :]
    // --- backward / training (gemm.hip only) ---
    EPI_BF16 = 6,    // out_bf16 planes = acc (+ bias)
    EPI_ATOMIC = 7,  // out_f32 += acc by fp32 atomics (split-K weight gradients; out must be zeroed)
    EPI_DGELU = 8,   // out_bf16 planes = acc * gelu'(aux_in)   (aux_in = saved fc1 pre-activation planes)
    EPI_DRELU = 9,   // out_bf16 planes = acc * (aux_in > 0)    (aux_in = saved post-ReLU activation planes)
};

// C[M,N] = A[M,K] . W[N,K]^T  (both operands K-contiguous, bf16 hi(/lo) planes, fp32 accumulate)
struct GemmParams {
    const bf16_t* A; long a_plane; int lda;     // [planes][M][lda]
    const bf16_t* W; long w_plane;              // [planes][N][K]
    int M, N, K;
    int n_off;                                  // (gemm_big only) first output column / W row of this launch
    int n_valid;                                // ATOMIC: only columns < n_valid are written (0 = all N)
    int dbg;                                    // timing-only ablations: bit0 skip epilogue stores, bit1 skip steady-state loads
    int planes;                                 // 1: bf16 ; 2: bf16 hi+lo split (3 MFMAs per product)
    int fmt;                                    // operand format: 0 = bf16, 1 = fp16 (FMT_FP16: A, W and the 16-bit outputs of GELU / RELU / QKV).
                                                // One fp16 plane: EPI_QKV still writes V as bf16 (the fast attention's P.V product stays bf16);
                                                // fp16 hi+lo planes (planes == 2): every plane is fp16.  Inference epilogues only
    int epi;
    const float* bias;                          // [N] or null
    float* out_f32; int ldo_f32;                // PLAIN / RESID / PATCH
    bf16_t* out_bf16; long out_plane; int ldo;  // GELU / RELU : [planes][M][ldo]
    bf16_t* q; bf16_t* k; bf16_t* v; long qkv_plane;    // QKV
    int ntok, npad, heads, dmodel; float qscale;
    const float* pos; int n_patches;            // PATCH
    const float* resid;                         // RESID: out = resid + acc + bias (null: in place on out_f32)
    bf16_t* aux_out;                            // GELU: also save the pre-activation planes here (training), ld = ldo
    const bf16_t* aux_in; long aux_plane;       // DGELU / DRELU operand planes [planes][M][ldo]; also plane stride of aux_out
    int v_bf16;                                 // EPI_QKV with fp16 hi + lo planes: write V as bf16 hi + lo planes (AttnParams::v_bf16)
    int dispatch_rows;                          // launch_gemm's kernel choice is made for this many rows instead of M when > 0: the two
                                                // half-batches of a split forward take the route of the whole batch (same summation order)
    int ksplit;                                 // >1: split the K loop over grid.y (EPI_ATOMIC, or EPI_PLAIN partial tiles)
    long split_stride;                          // EPI_PLAIN with ksplit > 1: slice y writes out_f32 + y * split_stride (floats)
    // launch_gemm_rs, EPI_GELU / EPI_QKV only: A = (ln_x - mean) rstd per row, computed in the kernel's prologue (ln_x fp32 [M][K] rows, row stride K;
    // A unused); W / bias = launch_pack_rs_ln's copies, which carry the LayerNorm's weight and bias
    const float* ln_x; float ln_eps;
};
int launch_gemm(const GemmParams& p, hipStream_t s);       // dispatches between the two kernels below
int launch_gemm_small(const GemmParams& p, hipStream_t s); // 128x128 tile, bf16 or bf16x3 (gemm.hip)
bool gemm_big_supported(const GemmParams& p);
int launch_gemm_big(const GemmParams& p, hipStream_t s);   // 256x384 persistent tile, bf16 only (gemm_big.hip)

// Row-stationary streaming GEMMs for the wide model (gemm_rs.hip; one plane, embed_dim 768): p.W = the weight re-packed by launch_pack_rs
// (kind 0 for EPI_GELU / EPI_QKV: K = 768; kind 1 for EPI_RESID: N = 768, in place on out_f32)
bool gemm_rs_supported(const GemmParams& p);
int launch_pack_rs(const float* W, int N, int K, int kind, bf16_t* dst, hipStream_t s, int fmt);
int launch_pack_rs_ln(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K, bf16_t* dst_w, float* dst_bias,
                      hipStream_t s, int fmt);
int launch_gemm_rs(const GemmParams& p, hipStream_t s);

// LayerNorm-fused A-stationary GEMM for qkv / fc1 (gemm_ln.hip): X fp32 rows are normalised in the prologue, W streams
struct LnGemmParams {
    const float* X; int ldx;                    // residual stream [M, K] fp32
    const float* gamma; const float* beta; float eps;
    const bf16_t* W; long w_plane;              // slab-major packed copy (launch_pack_slabs); w_plane unused
    const float* bias;                          // [N]
    int M, N;
    int epi;                                    // EPI_QKV or EPI_GELU
    bf16_t* out_bf16; long out_plane; int ldo;  // GELU: [planes][M][ldo]
    bf16_t* q; bf16_t* k; bf16_t* v; long qkv_plane; int ntok, npad, heads, dmodel; float qscale;   // QKV
    bf16_t* a_out; long a_plane;                // optional: the normalised planes [planes][M][K] (training: weight gradients)
    bf16_t* aux_out; long aux_plane;            // optional (GELU): pre-activation planes [planes][M][ldo] (training: gelu')
    int dbg;                                    // timing-only ablations (wrong results): 1 skip epilogue, 2 skip W DMA, 4 skip LN prologue
    int fmt;                                    // operand format (FMT_BF16 / FMT_FP16), as GemmParams::fmt
};
bool gemm_ln_supported(int K, int N, int planes, int epi, int dmodel);
long gemm_ln_slab_elems(int N, int K, int planes);      // bf16 elements of the slab-major copy of W [N][K]
int launch_pack_slabs(const float* src, int N, int K, int planes, bf16_t* dst, hipStream_t s, int fmt = 0);
int launch_gemm_ln(const LnGemmParams& p, int K, int planes, hipStream_t s);

// Fused MLP of one transformer block, single-plane modes, D = 384 (mlp_fused2.hip): X += fc2(gelu(fc1(LayerNorm(X))))   in place
struct MlpFusedParams {
    float* X; int ldx;                          // residual stream [M, 384] fp32
    const float* gamma; const float* beta; float eps;
    const bf16_t* Wp;                           // both weights in MFMA fragment order (launch_pack_mlp)
    const float* b1; const float* b2;           // [1536], [384]
    int M;
    int* queue;                                 // (diagnostic builds: MF2_STAMP buffer)
    int n_long, sleep_max;                      // set by launch_mlp_fused2: start-time spread of the workgroups with one item fewer
    // optional (mlp_fused2 only): the block's attention output projection in the same launch, X += ctx . Wproj^T + bproj first
    const bf16_t* ctx;                          // [M, 384] bf16 attention output (row stride 384), null = MLP only
    const bf16_t* Wproj;                        // launch_pack_proj
    const float* bproj;
    // optional (with ctx): LayerNorm1 + the qkv projection of the NEXT block at the end of the same launch (Q pre-scaled by qscale)
    const bf16_t* Wqkv;                         // launch_pack_qkv: [1152, 384] in fragment order; null = no qkv tail
    const float* bqkv; const float* gamma1; const float* beta1;
    bf16_t* q; bf16_t* k; bf16_t* v;            // each [B, heads, npad, 64] bf16 (rows >= ntok are never written)
    int ntok, npad, heads; float qscale;
    int fmt;                                    // mlp_fused2 only: operand format of ctx, the packed weights and everything in between
                                                // (FMT_BF16 / FMT_FP16); the qkv tail writes V as bf16 either way
};
long mlp_fused_qkv_pack_elems(int D);           // bf16 elements of the packed qkv weight (0: unsupported width)
int launch_pack_qkv(const float* W, int D, bf16_t* dst, hipStream_t s, int fmt = 0);
long mlp_fused_proj_pack_elems(int D);          // bf16 elements of the packed projection weight (0: unsupported width)
int launch_pack_proj(const float* W, int D, bf16_t* dst, hipStream_t s, int fmt = 0);
bool mlp_fused_supported(int D, int F, int planes);
long mlp_fused_pack_elems(int D, int F);        // bf16 elements of the packed copy (0: unsupported shape)
int launch_pack_mlp(const float* W1, const float* W2, int D, int F, bf16_t* dst, hipStream_t s, int fmt = 0);
int launch_mlp_fused2(const MlpFusedParams& p, hipStream_t s);     // role-split wave pairs, two waves per SIMD (mlp_fused2.hip)

// Attention output projection + fused MLP of one block on hi + lo operand planes (mlp_fused3.hip): X += ctx . Wproj^T + bproj (when ctx is
// given), then X += fc2(gelu(fc1(LayerNorm(X)))), in place, one launch, D = 384
struct MlpFused3Params {
    float* X;                                   // residual stream [M, 384] fp32
    float eps;                                  // LayerNorm epsilon (norm2; norm1 of the next block in the qkv tail)
    const bf16_t* Wp;                           // launch_pack_mlp3: proj + fc1 + fc2 as 108 slots of (lo, hi) fragment pairs in stream order (+ the 36
                                                // slots of the next block's Wqkv), then the folded biases -- the LayerNorms' weights and biases ride in it
    const float* b2;                            // [384]
    int M;
    const bf16_t* ctx; long ctx_plane;          // attention output planes [2][M][384] (hi, then lo at + ctx_plane elements); null = MLP only
    const float* bproj;
    int fmt;                                    // operand format of ctx, the packed weights and everything in between (FMT_BF16 / FMT_FP16)
    // optional (with ctx; q != null): LayerNorm1 + the qkv projection of the NEXT block at the end of the same launch (Wp packed with Wqkv_next);
    // Q pre-scaled by qscale
    bf16_t* q; bf16_t* k; bf16_t* v; long qkv_plane;      // each [2][B, heads, npad, 64] (lo plane at + qkv_plane elements; rows >= ntok never written)
    int ntok, npad, heads; float qscale;
    int v_bf16;                                 // fmt == FMT_FP16: V as bf16 hi + lo planes (GemmParams::v_bf16)
};
// what launch_pack_mlp3 reads: the fp32 parameters of a block's second half (and of the next block's first linear)
struct MlpFused3Weights {
    const float* Wproj; const float* W1; const float* b1; const float* W2;
    const float* gamma2; const float* beta2;                                         // norm2 of this block: folded into W1 / b1
    const float* Wqkv_next; const float* bqkv_next; const float* gamma1_next; const float* beta1_next;      // nullable together: the qkv tail
};
bool mlp_fused3_supported(int D, int F, int planes);
long mlp_fused3_pack_elems(int D, int F);       // 16-bit elements of the packed copy, qkv tail slots and folded biases included (0: unsupported shape)
int launch_pack_mlp3(const MlpFused3Weights& w, int D, int F, bf16_t* dst, hipStream_t s, int fmt);
int launch_mlp_fused3(const MlpFused3Params& p, hipStream_t s);
// The same launch on ONE operand plane, one wave per SIMD (mlp_fused4.hip; MlpFused3Params without the plane strides; the qkv tail writes one plane
// each, V as bf16): Wp = launch_pack_mlp4's 54 (+ 18: the next block's Wqkv) slots of 48 fragments (Wproj may be null for an MLP-only copy: its six
// slots are zero and never read), the LayerNorms folded in as in launch_pack_mlp3, then the folded biases
bool mlp_fused4_supported(int D, int F, int planes);
long mlp_fused4_pack_elems(int D, int F);       // 16-bit elements of the packed copy, tail slots and folded biases included (0: unsupported shape)
int launch_pack_mlp4(const MlpFused3Weights& w, int D, int F, bf16_t* dst, hipStream_t s, int fmt);
int launch_mlp_fused4(const MlpFused3Params& p, hipStream_t s);

// tuning knobs (dinoseg_set_option): see api.hip
struct Options {
    int gemm_ln = 1;         // qkv / fc1 through the LayerNorm-fused kernel (gemm_ln.hip): 0 never, 2 wherever it applies, 1 = by measurement (api.hip)
    int gemm_big = 1;        // use gemm_big.hip where it applies
    int gemm_dbg = 0;        // ablation bits copied into GemmParams::dbg (wrong results; timing only)
    int attn_dbg = 0;        // same for AttnParams::dbg (bits 0-2: attention.hip ablations, wrong results); bit 3: attention_za's one-block body for the
                             // ragged last q-tile OFF (an A/B switch, results unchanged)
    int mlp_fused = 1;       // the MLP half of a block as one launch (mlp_fused2.hip): 0 never, 1 for >= mlp_fused_min_rows rows, 2 wherever supported
    int mlp_fused_min_rows = 12000;      // (4 frames @480: +3 %; 6 frames: +18 % with the projection inside; 2 frames: even)
    int mlp_fused3_min_rows = 10000;     // hi + lo planes (mlp_fused3.hip): 3 frames @480 +10 %, 4: +16 %, 8: +19 %; 2 frames -17 %, 1: -36 % (tools/m3_thresh.sh)
    int mlp_stagger = 0;     // experiment: > 0 = one workgroup per CU, those with one item fewer start up to this many x 3.9 us late
    int mlp_grid = 0;        // workgroups of the fused MLP launch (0 = the fewest that need no extra round: mlp_fused2.hip)
    int qkv_fused = 0;       // 1: ... and LayerNorm1 + qkv of the NEXT block at its end (blocks 1.. then have no LN+qkv launch); measured
                             // +1 % on one stream, +-0 on two: the tail is bound by the same HBM write burst as the launch it replaces
    int mlp_fused4 = 0;      // one-plane modes: 1 = the fused projection + MLP launch with ONE wave per SIMD (mlp_fused4.hip) instead of mlp_fused2.hip's two
                             // (no qkv tail: only while qkv_fused is 0).  The launch 354-360 against 359-371 us, the headline +0.3-0.4 % (profiles/r06_mlp_fused4.md):
                             // inside the boxes' spread, so the older kernel stays the default.
                             // Read when the weights are packed (dinoseg_refresh_weights) and at every forward
    int qkv_fused4 = 1;      // ... with LayerNorm1 + qkv of the NEXT block at its end (as qkv_fused3 for the hi + lo launch)
    int qkv_fused3 = 1;      // hi + lo planes (mlp_fused3.hip): 1 = LayerNorm1 + qkv of the NEXT block at the end of the fused projection + MLP launch
    int gemm_rs = 3;         // the row-stationary streaming GEMMs (gemm_rs.hip; embed_dim 768, one plane, >= gemm_rs_min_rows rows), a bit per linear:
                             // 1 = mlp.fc1 (its GELU epilogue rides in the MFMA gaps: 329 against gemm_big's 357 us at 57 616 rows), 2 = attn.qkv
                             // (-3 % on the launch), 4 = attn.proj and mlp.fc2 (slower); 0 = never.  configs.vitb on one box: 850 / 865 / 873
                             // frames/s with 0 / 1 / 3 (profiles/r06_gemm_rs.md).  Read when the weights are packed and at every forward
    int gemm_rs_min_rows = 24000;
    int gemm_rs_ln = 1;      // ... with the LayerNorm in front of qkv / fc1 computed in the kernel's prologue (no LayerNorm launch, no 16-bit A round trip)
    int proj_fused = 1;      // 1: the block's attention output projection runs inside the fused MLP launch
    int streams = 2;         // 2: dinoseg_forward runs a batch of >= split_min frames as two half-batches on two streams (api.hip)
    int split_min = 8;       // (8 frames @480: +6 %, 12: +16 %, 16: +12 %; 6 frames and fewer: slower split)
    int route_ab = 0;        // A/B switches of dispatch routes that do not change results: bit 0 = 128-row tiles for the residual GEMMs of a small
                             // batch (gemm.hip HALFM off), bit 1 = one wave per row in the LayerNorm backward (train.hip), bit 2 = the weight-
                             // gradient GEMM's split count not rounded to a multiple of 8 (its XCD-aware grid off: train_api.hip, gemm_tn.hip)
    int fp16_patch_planes = 1;      // precision fp16: 1 = the patch embedding on one fp16 plane like the rest of the mode (2466 -> 2486 frames/s,
                                    // 0.0263 / 8 flips -> 0.0218 / 5 on the G3 fixture), 2 = on bf16 hi+lo planes (round 4's first build).
                                    // Read when the weights are packed: set it before the first forward
    int op_v_bf16 = 0;       // dinoseg_op_attention: AttnParams::v_bf16 (tests)
    int op_fmt = 0;          // operand format (FMT_BF16 / FMT_FP16) of the single-plane stand-alone ops (dinoseg_op_*: tests, tools); a
                             // handle's forward follows its own precision instead
    int deterministic = 0;   // 1: the fine-tune step sums the loss, the bias and the LayerNorm gamma / beta gradients in a FIXED order (per-block
                             // partials into a scratch area + one ordered pass; the side stream's gemm_tn launches have their own region)
                             // instead of fp32 atomics: two runs from the same state are bit-identical (tests/test_train_gpu.py)
    int train_streams = 2;   // 2: backward runs the weight-gradient GEMMs of the blocks on the handle's side stream (train_api.hip)
    int splitk_tiles = 512;  // weight-gradient GEMMs: partial 128x128 tiles per launch (<= 768, the workspace holds that many)
    int attn_variant = 11 | 1024 | 65536;
                             // bit 0: overflow check on the row sums instead of a per-tile row maximum; bit 1: idle waves skip the
                             // tile work; bit 2: unused (round 1-3's software-pipelined kernel, removed); bit 3 (bf16 mode):
                             // the zero-reference kernel, four waves per SIMD (attention_z.hip); bit 9: never split the keys of a
                             // q-tile over wave groups (attn_fwd_zs_kernel, the small-grid form: A/B and tests);
                             // bit 10 (round 5, default): grids of four rounds and more run the tile loop as a hand-scheduled assembly
                             // pipeline (attention_za.hip: bit-identical outputs), bit 16: with 64 queries per wave at two waves per
                             // SIMD; bit 11: that kernel at every grid size, bit 12: attention_z.hip's 256-query workgroups at every
                             // grid size (tests); bit 6: never the 256-query workgroups (nor attention_za.hip)
};
Options& options();

// Flash-style fused multi-head attention, head_dim 64.
struct AttnParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; long qkv_plane;   // as written by EPI_QKV: [planes][B,H,npad,64]
    bf16_t* ctx; long ctx_plane;       // [planes][B*ntok][heads*64]
    float* lse;                        // optional [B,H,ntok] log2-domain log-sum-exp (for backward), may be null
    int B, heads, ntok, npad, planes;
    int dbg;                           // timing-only ablations: bit0 skip max/exp, bit1 skip steady-state loads, bit2 skip PV
    int shared_gpu;                    // hint: another stream's kernels run beside this launch (the split forward): prefer wide workgroups
    int dispatch_B;                    // > 0: choices that change the arithmetic (the key split of small grids) are made for this batch: the
                                       // half-batches of a split forward run what the whole call would (as GemmParams::dispatch_rows)
    int v_bf16;                        // hi + lo planes with fmt == FMT_FP16: V arrives as bf16 hi + lo planes (GemmParams::v_bf16) -- the
                                       // zero-reference kernels (attention_z.hip / attention_za.hip: probabilities and V bf16, Q / K / ctx
                                       // fp16); 0: every plane fp16, the reference-based kernel of attention.hip
    int fmt;                           // FMT_FP16, one plane (attention_z.hip): Q, K and ctx are fp16, V and the probabilities stay bf16 (2^S against
                                       // the fixed reference 0 needs bf16's exponent range); hi+lo planes (attention.hip): everything fp16, the
                                       // probabilities bounded by the running reference (<= 2^15)
};
int launch_attention(const AttnParams& p, hipStream_t s);
int launch_attention_z(const AttnParams& p, hipStream_t s);    // zero-reference softmax, <= 128 registers: 4 waves per SIMD (attention_z.hip)
int launch_attention_za(const AttnParams& p, hipStream_t s);
// hi + lo planes: does a forward of `batch` frames (the WHOLE call's batch) run the zero-reference assembly kernel (attention_za.hip, X3)?
// The caller of the qkv GEMM asks too: in the fp16 hi + lo mode that kernel wants V as bf16 planes (GemmParams::v_bf16).
bool attention_x3_za(int batch, int heads, int ntok);   // the same arithmetic, tile loop as a hand-scheduled assembly pipeline (attention_za.hip)

// fp32 [rows, cols] -> bf16 planes [planes][rows_pad][cols_pad], zero padded
int launch_pack_planes(const float* src, int rows, int cols, bf16_t* dst, long plane, int rows_pad, int cols_pad,
                       int planes, hipStream_t s, int fmt = 0);

// LayerNorm over the last dim (D % 128 == 0, D <= 1024). rows of x: [M, D] fp32.
// drop_cls != 0: input row m = b*ntok + t is skipped for t == 0 and written to output row b*(ntok-1) + t-1.
int launch_layernorm(const float* x, const float* gamma, const float* beta, float eps, int M, int D,
                     bf16_t* out, long out_plane, int planes, float* out_f32, int drop_cls, int ntok, hipStream_t s, int fmt = 0);

// Patch gather ("im2col") for the 8x8/stride-8 patch embedding. k index = c*64 + ky*8 + kx.
// kind 0: uint8 HWC frames [B,r,r,3] with the ImageNet normalisation fused; kind 1: fp32 CHW [B,3,r,r] as is.
int launch_patch_gather(const void* x, int kind, int B, int r, const float* mean255, const float* inv_std255,
                        bf16_t* out, long out_plane, int planes, hipStream_t s, int fmt = 0);

// X[b*ntok + 0, :] = cls[:] + pos[0, :]
int launch_cls_rows(float* X, const float* cls, const float* pos, int B, int ntok, int D, hipStream_t s);

// Bicubic (A=-0.75, align_corners=False, scale_factor rule) resample of the patch pos-embed grid.
int launch_pos_resample(const float* pos_embed, int g, int D, int o, float* out, hipStream_t s);

// Final classifier layer + log_softmax + argmax.  in: bf16 hi/lo planes [2][M][ld]; W fp32 [C][K]; C <= 32.
int launch_head_final(const bf16_t* in, long in_plane, int ld, int M, int K, const float* W, const float* b, int C,
                      float* logp, int32_t* argmax, hipStream_t s, int fmt = 0);

// materialised softmax(q k^T) of one block, fp32 [B,H,ntok,ntok] (get_last_selfattention; visualisation only)
int launch_attn_probs(const bf16_t* q, const bf16_t* k, long plane, int planes, int B, int heads, int ntok, int npad, float* out,
                      hipStream_t s, int fmt = 0);
// cm[gt][pred] += 1 over n patches (int64 [C,C], accumulates)
int launch_cls_mask_attn(const bf16_t* q, const bf16_t* k, const bf16_t* v, long plane, int planes, int heads, int ntok, int npad,
                         const float* mask, int n_masks, bf16_t* ctx, long ctx_plane, float* probs, hipStream_t s, int fmt = 0);
int launch_broadcast_row0(float* X, int D, int n, hipStream_t s);
int launch_resize_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, hipStream_t s);
int launch_confusion(const int32_t* pred, const int64_t* gt, long n, int C, int64_t* cm, hipStream_t s);

// ---- fine-tune step (train.hip, attention_bwd.hip) ----
struct AttnBwdParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; long qkv_plane;   // forward operands [planes][B,H,npad,64]
    const bf16_t* dO; const bf16_t* O; long dO_plane;                      // d ctx and ctx, [planes][B*ntok][H*64]
    const float* lse;                                                      // forward log2-domain LSE [B,H,ntok]
    float* neg_lse; float* neg_delta;                                      // scratch [B,H,npad] each
    bf16_t* dqkv; long dqkv_plane;                                         // out: [planes][B*ntok][3*H*64]
    int B, heads, ntok, npad, planes;
};
int launch_attention_bwd(const AttnBwdParams& p, hipStream_t s);

int launch_transpose_planes(const float* src_f32, const bf16_t* src_pl, long src_plane, int ld_src, int M, int C,
                            bf16_t* T, long t_plane, int c_pad, int m_pad, bf16_t* Nout, long n_plane, int ldn,
                            float* colsum, int planes, int drop_cls, int ntok, hipStream_t s, int det_region = 0);
                            // det_region (deterministic mode, colsum given): 0 = the caller's stream, 1 = the weight-gradient side stream,
                            // whose partial sums go to its own part of the scratch (as TnParams::det_region)
// labels != null: F.nll_loss (mean over rows whose label is not -100) + d logits; labels == null: d logits from the caller's dlogp
int launch_nll_loss_grad(const float* logp, const int64_t* labels, const float* dlogp, int M, int C, float* acc, int* flags,
                         float* loss, bf16_t* dz, long dz_plane, int ldz, hipStream_t s);
// dxp / colsum (optional): the final dx rows as bf16 planes [planes][M][D] and their column sums (train.hip)
int launch_layernorm_bwd(const float* dy, const float* x, const float* gamma, float eps, int M, int D, float* dx,
                         int accumulate, float* dgamma, float* dbeta, int drop_cls, int ntok, hipStream_t s,
                         bf16_t* dxp = nullptr, long dxp_plane = 0, int planes = 1, float* colsum = nullptr);
// Weight gradient on row-major operands (gemm_tn.hip): part[slice][n][k] = sum over the slice's batch rows of Y[m][n] X[m][k]
struct TnParams {
    const bf16_t* Y; long y_plane; int ldy;     // dY planes [planes][M][ldy], columns n < N (N % 128 == 0 not required: rows guarded)
    const bf16_t* X; long x_plane; int ldx;     // layer input planes [planes][M][ldx], columns k < Kc (Kc % 128 == 0)
    int M, N, Kc, planes;
    float* part; int ld_part; long split_stride; int ksplit;
    float* colsum;                              // optional [N]: += column sums of dY over the M rows (the layer's bias gradient)
    float* det; int det_ld;                     // set by launch_gemm_tn in deterministic mode: per-slice partial sums [ksplit][det_ld]
    int det_region;                             // deterministic mode: 0 = the caller's stream, 1 = the weight-gradient side stream (its own part of the scratch)
};
int launch_gemm_tn(const TnParams& p, hipStream_t s);
// Deterministic mode (Options::deterministic): the scratch area the partial sums of the current backward go to (set by
// dinoseg_backward around its launches; nullptr = atomics) and the ordered pass dst[c] += sum_p part[p * ld + c], p ascending.
struct DetScratch { float* ptr; size_t floats; float* tn[2]; size_t tn_floats; };      // tn[r]: gemm_tn's partials, per stream region
DetScratch& det_scratch();
int launch_det_finalize(const float* part, int nparts, int width, int ld, float* dst, hipStream_t s);
int launch_splitk_reduce(const float* part, int ks, long stride, int rows, int ld_part, float* out, int ldo, int cols, hipStream_t s);
int launch_batch_sum_rows(const float* X, int B, int ntok, int D, float* out, hipStream_t s);
int launch_pos_resample_bwd(const float* dpos, int g, int D, int o, float* dpe, float* scratch /* g*o*D floats */, hipStream_t s);
constexpr int MULTI_MAX = 64;       // tensors per multi-tensor launch (table passed as a kernel argument)
int launch_multi_adam(int count, float* const* p, const float* const* g, float* const* m, float* const* v, const long* n, float lr,
                      float b1, float b2, float eps, float wd, int decoupled, int step, float gscale, hipStream_t s);
int launch_multi_zero(int count, float* const* p, const long* n, hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd,
                int decoupled, int step, float gscale, hipStream_t s);
// several pack_planes / pack_planes_t jobs in ONE launch (a weight refresh is ~30 of them, each a few microseconds of work
// behind a launch: the fine-tune step repacks every weight after every optimiser step)
struct PackJob {
    const float* src; bf16_t* dst; long plane;
    int rows, cols, rows_pad, cols_pad, planes, transposed;
    int fmt = 0;                    // FMT_BF16 / FMT_FP16
};
int launch_multi_pack(const PackJob* jobs, int count, hipStream_t s);
// fp32 [rows, cols] -> TRANSPOSED bf16 planes [planes][cols_pad][rows_pad] (zero padded): W^T operands for dgrad
int launch_pack_planes_t(const float* src, int rows, int cols, bf16_t* dst, long plane, int rows_pad, int cols_pad,
                         int planes, hipStream_t s);

}  // namespace dseg
