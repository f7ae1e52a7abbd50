/*
 * u2tok.h -- C ABI of libu2tok_hip.so, the MI355X (gfx950) implementation of the u2Tokenizer forward path.
 *
 * The reference (Siyou-Li/u2Tokenizer) has no FFI / operator registry: its hot path
 * `u2MetaForCausalLM.prepare_inputs_for_multimodal` (src/model/u2_arch.py:96-117) is a chain of Python
 * nn.Module calls into stock torch ops.  The seam a maintainer would bind is therefore the three module
 * forwards + the embedding splice; each entry point below names the reference code it replaces.
 * The Python host side (u2tokenizer_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions for EVERY function:
 *   - all pointers are DEVICE pointers (HBM) unless the name says `host`; tensors are dense row-major;
 *   - "bf16" = raw bfloat16 bits (uint16_t); parameters are expected in bf16 (model.to(torch.bfloat16));
 *   - work is enqueued on `stream` (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream);
 *     nothing allocates device memory, nothing synchronises, the caller owns every buffer including the workspace;
 *   - work is launched on the CURRENT HIP device: make the device of the buffers current first (hipSetDevice);
 *   - return value: 0 = success, negative = U2TOK_ERR_* (the Python shim raises RuntimeError).
 */
#ifndef U2TOK_H_
#define U2TOK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define U2TOK_OK 0
#define U2TOK_ERR_ARG (-1)        /* bad dimension / null pointer / unsupported combination */
#define U2TOK_ERR_LAUNCH (-2)     /* a kernel launch failed */
#define U2TOK_ERR_WORKSPACE (-3)  /* workspace too small: call the matching *_workspace_bytes */
#define U2TOK_ERR_DEVICE (-4)     /* current device is not gfx950 */

typedef void* u2tok_stream_t; /* hipStream_t */

/* ---- library identity / gating -------------------------------------------------------------- */
int u2tok_version(void);               /* MAJOR*10000 + MINOR*100 + PATCH */
const char* u2tok_arch(void);          /* "gfx950" */
/* The 16-bit ELEMENT TYPE of this build: "bf16" (libu2tok_hip.so) or "f16" (libu2tok_hip_f16.so -- the same sources compiled
 * with -DU2_ELEM_F16 for checkpoints loaded in float16, evalscipt/ourmodel_amos.py:33,70).  Wherever this header says "bf16"
 * (parameter / activation buffers, function names such as u2tok_gemm_bf16) read "the element type of the build": both
 * libraries export exactly the same symbols, accumulate in fp32 and differ only in the storage format and the MFMA opcode. */
const char* u2tok_elem(void);
int u2tok_device_check(void);          /* 0 if the current HIP device is gfx950, else U2TOK_ERR_DEVICE */
/* ---- execution contexts --------------------------------------------------------------------------------------------
 * Options, the tokenizer's side streams / events (one set per caller stream), split-K scratch registrations and
 * profiling records belong to a CONTEXT.  A host thread works on its current context: the one bound with
 * u2tok_ctx_set_current (thread-local, like hipSetDevice), or the process default context when none is bound.  Give
 * every model -- or every host thread -- its own context and they share no mutable state; a context used from several
 * threads at once is safe for launches as long as its options are not changed concurrently.  A context's side streams
 * are created on the HIP device that is current at their first use: keep one context per device. */
typedef struct u2tok_ctx* u2tok_ctx_t;
int u2tok_ctx_create(u2tok_ctx_t* out);
int u2tok_ctx_destroy(u2tok_ctx_t ctx);      /* no work of this context may still be being enqueued */
int u2tok_ctx_set_current(u2tok_ctx_t ctx);  /* NULL: back to the process default context */
u2tok_ctx_t u2tok_ctx_get_current(void);     /* NULL when the thread uses the default context */

/* Tuning / diagnostics switches of the calling thread's current context; U2TOK_ERR_ARG if unknown / out of range:
    "gemm_tile" {0 heuristic, 64, 128: tile of the small-tile kernel}, "gemm_splitk" {-1 never, 0 heuristic, 2..16 force
    that many K slices where scratch allows}, "gemm_big" {-1 never, 0 heuristic; force a form of the big-tile kernel: 20 / 21 =
    256x256 / 256x192 tiles with two LDS stages, 22 = 256x128 ring (three stages), 23 / 24 = 256x192 with three stages for A /
    for B, 25 / 26 = the same at 256x256}, "gemm_big_grid" {persistent workgroups}, "gemm_big_gelu" {1: GELU products may take
    the big-tile kernel, 0: never}, "gemm_big_ring" / "gemm_big_deep" {1: the heuristic may pick the ring / launches the deep
    forms, 0: two-stage forms only}, "gemm_big_splitk" {K slices of a FORCED big-tile launch}, "gemm_big_skinny" {1: partial-round
    products may take the big-tile kernel with K slices}, "kmajor_b" {1: P V and the DiffTS aggregation read V / X in place as
    K-major operands, 0: through transposed copies}, "gemm_tail_fused" {1: <= 16 rows behind a multiple of 256 rows (the ViT's cls
    rows) are computed inside the big-tile launch by the few-rows kernel's arithmetic, 0: a few-rows launch of their own},
    "flash_mode" {0 pick, 1 plain 128-row units, 7 double pipeline (generated asm KV loop)}, "flash_q_prescaled" {1: the q handed to
    u2tok_flash_attention_d64 already carries scale * log2 e}, "vit_flash" {0 unfused attention, 1}, "tta_overlap" {0, 1: side
    stream for the TTA k|v projections},
    "vit_vt_epilogue" {1: the ViT's q|k|v product leaves V^T (the flash kernel's operand) from its own V tiles, 0: a transpose launch},
    "tok_flash" {1: the tokenizer's attention cores run the fused kernel of u2tok_tok_attention, 0: GEMM -> softmax -> GEMM},
    "tok_wide" {1: head dims 256 / 512 of that kernel run its 8-wave form (a wave pair per 16-query block, two waves per SIMD),
    0: the 4-wave form},
    "profile" {0, 1} */
int u2tok_set_option(const char* name, int value);
/* Scratch for split-K partial sums of u2tok_gemm_bf16 calls on `stream` (fp32, slices x M x N), registered on the
 * current context: skinny products (few output tiles, long K) are cut along K when a scratch is registered; NULL / 0
 * removes it.  The module forwards below carve their own from their workspace and do not need this. */
int u2tok_set_gemm_scratch(void* device_ptr, size_t bytes, void* stream);
/* Diagnostics only, process-wide (not for concurrent use): device buffer (>= grid*4*8 uint64, zeroed by the caller)
 * for the flash attention kernel; while attached the kernel runs its s_memtime-instrumented build and ADDS per-phase
 * cycle sums per (workgroup, wave). */
int u2tok_flash_debug_buffer(void* device_ptr);
/* Same for u2tok_tok_attention: >= grid*4*8 uint64 (grid*8*16 for the 8-wave form of head dims 256 / 512, which also leaves s_memrealtime stamps in slots 8..15); slots = cycles in
 * {DMA wait, barrier, K DMA issue (8-wave form: all DMA issue), Q K^T, softmax, V DMA issue (8-wave form: unused), P V},
 * [7] = tiles. */
int u2tok_tok_attention_debug_buffer(void* device_ptr);

/* With option "profile" = 1 every launch is bracketed by hipEvents on its stream.  Collect (HOST arrays of ncat <= 6
 * entries; synchronises on the recorded events, then resets): summed milliseconds, algorithmic FLOPs and launch
 * counts per kernel class 0 = MFMA GEMM, 1 = ViT flash attention, 2 = temporal attention, 3 = row ops
 * (LayerNorm / softmax / RoPE / scores / top-k / pooling), 4 = data movement (im2col, transposes, gathers, splice),
 * 5 = fused tokenizer attention (u2tok_tok_attention). */
int u2tok_profile_collect(double* ms_host, double* flops_host, int64_t* count_host, int32_t ncat);
/* Same, plus the summed ALGORITHMIC bytes (every operand read once + every result written once) per class. */
int u2tok_profile_collect2(double* ms_host, double* flops_host, double* bytes_host, int64_t* count_host, int32_t ncat);

/* ---- configuration records -------------------------------------------------------------------- */

/* ViT3DTower (src/model/multimodal_encoder/vit.py:132-164) built on MONAI 1.3.0 PatchEmbeddingBlock
 * (perceptron) + TransformerBlock x depth; hyper-parameters hard-coded at vit.py:33-38,139-146. */
typedef struct {
  int32_t nchunk;          /* B*C: 32-slice chunks in this call (u2_arch.py:105-106) */
  int32_t img[3];          /* config.image_size, e.g. {32,256,256} */
  int32_t patch[3];        /* config.patch_size, e.g. {4,16,16} */
  int32_t hidden;          /* 768 */
  int32_t mlp_dim;         /* 3072 */
  int32_t depth;           /* 12 */
  int32_t heads;           /* 12 (head dim must be 64) */
  int32_t vol_dtype;       /* 0 = fp16, 1 = bf16, 2 = fp32 voxels */
  int32_t keep_cls;        /* 0: select_feature == "patch" (drop cls, vit.py:157-158); 1: "cls_patch" */
  float ln_eps;            /* 1e-5 */
} u2tok_vit_config;

/* ViT weight table: array of (4 + 11*depth + 2) device pointers, all bf16, MONAI state-dict names:
 *   [0] patch_embedding.position_embeddings (1,ntok,hidden)   [1] patch_embedding.patch_embeddings.1.weight
 *   [2] patch_embedding.patch_embeddings.1.bias               [3] cls_token
 *   per block i (base 4 + 11*i): norm1.weight, norm1.bias, attn.qkv.weight, attn.out_proj.weight,
 *       attn.out_proj.bias, norm2.weight, norm2.bias, mlp.linear1.weight, mlp.linear1.bias,
 *       mlp.linear2.weight, mlp.linear2.bias
 *   [4+11*depth] norm.weight   [5+11*depth] norm.bias */
#define U2TOK_VIT_NW(depth) (4 + 11 * (depth) + 2)

/* SpatialPoolingProjector (src/model/multimodal_projector/spatial_pooling_projector.py:7-59). */
typedef struct {
  int32_t nchunk;
  int32_t grid[3];         /* num_patches_pre = image_size / patch_size */
  int32_t pooling_size;    /* 2 */
  int32_t pooling_type;    /* 0 = "spatial" (avg_pool3d), 1 = "sequence" (avg_pool1d over pooling_size^3) */
  int32_t in_dim;          /* 768 */
  int32_t out_dim;         /* E = LLM hidden size */
  int32_t layer_type;      /* 0 = "mlp" (GELU between), 1 = "linear" */
  int32_t layer_num;       /* 2 */
} u2tok_spp_config;
/* SPP weight table: 2*layer_num pointers: projector.{0,2,..}.weight, .bias in order. */

/* u2Tokenizer (src/model/u2tokenizer/u2Tokenizer.py:6-47; builder.py:3-14). */
typedef struct {
  int32_t B;               /* volumes */
  int32_t T;               /* chunks per volume ("frames") */
  int32_t N;               /* projector tokens per chunk */
  int32_t E;               /* embed_size == hidden_size */
  int32_t Lt;              /* text tokens (padded question length) */
  int32_t num_heads;       /* u2t_num_heads */
  int32_t num_layers;      /* u2t_num_layers */
  int32_t top_k;           /* u2t_top_k */
  int32_t num_query;       /* num_3d_query_token */
  int32_t use_multi_scale; /* bool */
  int32_t attn_type;       /* 0 = "rma" (RelativeMultiheadAttention), 1 = "rope", 2 = nn.MultiheadAttention read
                              sequence-first (every other attn_type string, svr.py:16-18): attends across batch entries */
  int32_t enable_diffts;   /* bool: DifferentiableTokenSelection vs TokenSelection */
  int32_t enable_dmtp;     /* bool: DynamicMultiScalePooling */
  int32_t max_seq_len;     /* 512: rma.py:6 / rope.py:19 */
  float diffts_tau;        /* 1.0 (svr.py:94) */
  float ln_eps;            /* 1e-5 */
} u2tok_tokenizer_config;

/* Tokenizer weight table (all bf16), in this order; an attention record "ATT" is
 *   wq.weight, wq.bias, wk.weight, wk.bias, wv.weight, wv.bias, dense.weight, dense.bias, relative_bias
 * (relative_bias = null for MultiHeadCrossAttention and for attn_type rope):
 *   [0] query_tokens
 *   per SVR layer l (base 1 + 18*l):  ATT spatial_attention, ATT temporal_attention
 *   then token_selection.score_net.weight, .bias
 *   then dynamic_pool.gate_fc.weight, .bias   (null when !enable_dmtp)
 *   per TTA layer l (33 pointers): ATT self_attention, ATT visual_cross_attention, ATT text_cross_attention,
 *       norm_self.weight, .bias, norm_cross_v.weight, .bias, norm_cross_t.weight, .bias
 *   then ATT layer_linagg.linear_aggregator (wv/dense present in the state dict but never read: tta.py:47-48,62-65) */
#define U2TOK_TOK_NW(layers) (1 + 18 * (layers) + 4 + 33 * (layers) + 9)

/* ---- the hot path ------------------------------------------------------------------------------ */

/* Replaces ViT3DTower.forward (vit.py:148-164): volume (nchunk,1,D,H,W) -> features
 * (nchunk, ntok[+1], hidden) bf16. */
size_t u2tok_vit_workspace_bytes(const u2tok_vit_config* cfg);
int u2tok_vit_forward(const u2tok_vit_config* cfg, const void* const* weights, const void* volume, void* out,
                      void* workspace, size_t workspace_bytes, u2tok_stream_t stream);

/* Replaces SpatialPoolingProjector.forward (spatial_pooling_projector.py:34-52):
 * (nchunk, g1*g2*g3, in_dim) bf16 -> (nchunk, proj_out_num, out_dim) bf16. */
size_t u2tok_spp_workspace_bytes(const u2tok_spp_config* cfg);
int u2tok_spp_forward(const u2tok_spp_config* cfg, const void* const* weights, const void* x, void* out,
                      void* workspace, size_t workspace_bytes, u2tok_stream_t stream);

/* Replaces u2Tokenizer.forward (u2Tokenizer.py:40-47): v_token (B,T,N,E), t_token (B,Lt,E) bf16 ->
 * aligned tokens (B,num_query,E) bf16.  topk_idx_out (optional, may be null): (B,top_k) int64 indices chosen by
 * TokenSelection (only written when !enable_diffts) -- the path's integer output.  svr_out (optional, may be null):
 * (B,T*N,E) bf16 copy of the refined tokens the selection stage scores (output of svr.py:166-170), so that a checker
 * can replay the selection on identical inputs. */
size_t u2tok_tokenizer_workspace_bytes(const u2tok_tokenizer_config* cfg);
int u2tok_tokenizer_forward(const u2tok_tokenizer_config* cfg, const void* const* weights, const void* v_token,
                            const void* t_token, void* out, int64_t* topk_idx_out, void* svr_out, void* workspace,
                            size_t workspace_bytes, u2tok_stream_t stream);

/* Same forward with PARITY TAPS (tests): any layer's input can be replaced by a caller buffer ("teacher forcing": each
 * layer of the residual-free SVR stack / of the TTA is then compared on the reference's own fp32 input for that layer) and
 * any layer's output copied out.  Every pointer, and every array entry, may be NULL.  Shapes: svr (B,T,N,E), visual
 * (B,Lv,E) with Lv = top_k (+ top_k/2 + top_k/4 with multi-scale pooling), tta (B,num_query,E); all bf16. */
typedef struct {
  const void* const* svr_in;  /* [num_layers]: replaces the input of SVR layer l (svr.py:23-40) */
  void* const* svr_out;       /* [num_layers]: receives the output of SVR layer l */
  const void* visual_in;      /* replaces the visual tokens the TTA attends to (output of svr.py:171-184) */
  void* visual_out;           /* receives them (before a replacement) */
  const void* const* tta_in;  /* [num_layers]: replaces the query input of TTA layer l (tta.py:93-107) */
  void* const* tta_out;       /* [num_layers]: receives the output of TTA layer l */
} u2tok_tokenizer_taps;
int u2tok_tokenizer_forward_taps(const u2tok_tokenizer_config* cfg, const void* const* weights, const void* v_token,
                                 const void* t_token, void* out, int64_t* topk_idx_out, const u2tok_tokenizer_taps* taps,
                                 void* workspace, size_t workspace_bytes, u2tok_stream_t stream);

/* Replaces embed_tokens(ids) + the splice of u2_arch.py:109,113-116:
 * out[b][s] = (1 <= s <= nfeat) ? feats[b][s-1] : table[ids[b][s]].  nfeat = 0 / feats = null: plain lookup. */
int u2tok_embed_splice(const void* table, const int64_t* ids, const void* feats, void* out, int32_t B, int32_t S,
                       int32_t E, int32_t nfeat, int64_t vocab, u2tok_stream_t stream);

/* ---- producer of the path's input ("next" row of the scope table) ------------------------------------------- */
/* u2Transform.adaptive_resize (src/utils/u2Transform.py:62-122 with the validation transforms of :46-54) on the GPU:
 * ScaleIntensityRangePercentiles(lower_pct, upper_pct, 0, 1, clip) -> CropForeground -> anti-aliased trilinear
 * resize to (int(H r), int(W r), min(D, depth_pad)), r = min(target/H, target/W) -> zero pad to
 * depth_pad x target x target.  vol: fp32 [D][H][W] (the tensor of u2Transform.py:68-69 without the channel axis);
 * out: [depth_pad][target][target] == (depth_pad/32, 32, target, target) of out_dtype (0 fp16, 1 bf16, 2 fp32);
 * info (optional, 12 x int32, device): status (0 ok, 1 no foreground, 2 filter too wide), crop lo[3], hi[3] (d,h,w),
 * resized size[3], then the two percentiles as float. */
size_t u2tok_preprocess_workspace_bytes(int32_t D, int32_t H, int32_t W);
int u2tok_preprocess_volume(const float* vol, void* out, int32_t* info, int32_t D, int32_t H, int32_t W, int32_t target,
                            int32_t depth_pad, float lower_pct, float upper_pct, int32_t out_dtype, void* workspace,
                            size_t workspace_bytes, u2tok_stream_t stream);

/* Same with the training-time branch of u2Transform (u2Transform.py:32-44): between CropForeground and the resize the
 * volume is rotated by rot90_k quarter turns in the (H, W) plane (RandRotate90(spatial_axes=(1,2)), torch.rot90
 * convention), flipped along D / H / W (RandFlip x3), multiplied by (1 + scale_factor) (RandScaleIntensity) and shifted by
 * shift_offset (RandShiftIntensity).  The random draws are the CALLER's (host struct): the library is deterministic. */
typedef struct {
  int32_t rot90_k;       /* 0..3 */
  int32_t flip[3];       /* bool per spatial axis (d, h, w) */
  float scale_factor;    /* RandScaleIntensity factor in [-0.1, 0.1] in the reference; 0 = off */
  float shift_offset;    /* RandShiftIntensity offset in [-0.1, 0.1] in the reference; 0 = off */
} u2tok_augment;
int u2tok_preprocess_volume_aug(const float* vol, void* out, int32_t* info, int32_t D, int32_t H, int32_t W, int32_t target,
                                int32_t depth_pad, float lower_pct, float upper_pct, int32_t out_dtype,
                                const u2tok_augment* aug, void* workspace, size_t workspace_bytes, u2tok_stream_t stream);

/* ---- building blocks (exported for the parity tests; same kernels the pipelines launch) -------- */

/* C[z] = epi(alpha * A[z] B[z]^T): A (M,K) lda, B (N,K) ldb, C (M,N) ldc; z = zb*nbh + zh with element strides.
 * flags: 1 bias[n], 2 bias[m], 4 GELU(erf), 8 + R[m][n], 16 C is fp32 (else bf16), 64 B is K-tile-major [K/64][N][64],
 * 256 B is stored K-major, (K,N) with ldb >= N (C = A B: the input-gradient product dX = dY W without a transposed W),
 * 128 | 256 A is stored K-major too, (K,M) with lda >= M (C = A^T B: the weight-gradient product dW = dY^T X without
 * transposed activations); the K-major dimension (M resp. N) must be a multiple of 8.
 * 512: gate | up pair product of a gated MLP (LlamaMLP / Qwen3MLP: act_fn(gate_proj(x)) * up_proj(x)): B = the gate weight's
 * I rows followed by the up weight's I rows (N = 2 I), C (M, I) bf16 = bf16(silu(bf16(x gate^T))) * bf16(x up^T) -- the values
 * u2tok_gemm_bf16 + u2tok_swiglu_bf16 produce, bit for bit, without the (M, 2 I) intermediate; alone (no other flag, nz = 1),
 * K % 64 == 0, I % 16 == 0, 16-byte aligned operands; U2TOK_ERR_ARG otherwise. */
int u2tok_gemm_bf16(const void* A, const void* B, void* C, const void* bias, const void* R, int32_t M, int32_t N,
                    int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int32_t nz, int32_t nbh,
                    int64_t sAb, int64_t sAh, int64_t sBb, int64_t sBh, int64_t sCb, int64_t sCh, int64_t sRb,
                    int64_t sRh, float alpha, int32_t flags, u2tok_stream_t stream);
int u2tok_layernorm_bf16(const void* x, const void* res, const void* w, const void* b, void* y, int32_t rows,
                         int32_t C, float eps, u2tok_stream_t stream);
int u2tok_softmax_rows(const float* S, void* P, int32_t nz, int32_t rows, int32_t n, int64_t lds, int64_t ldp,
                       float scale, const void* rel_bias, int32_t H, int32_t max_len, u2tok_stream_t stream);
/* out[z][c][r] = in[z][r][c], pad columns zeroed.  perm16 != 0 (needs ld_out % 16 == 0): inside each group of 16
 * output columns the order is [0-3, 8-11, 4-7, 12-15] -- the V^T layout u2tok_flash_attention_d64 consumes. */
int u2tok_transpose_bf16(const void* in, void* out, int32_t nz, int32_t R, int32_t C, int64_t ld_in, int64_t ld_out,
                         int64_t in_zs, int64_t out_zs, int32_t perm16, u2tok_stream_t stream);
int u2tok_im2col_patches(const void* vol, int32_t vol_dtype, void* out, int32_t nchunk, int32_t D, int32_t H,
                         int32_t W, int32_t p1, int32_t p2, int32_t p3, u2tok_stream_t stream);
int u2tok_avgpool3d_tokens(const void* x, void* y, int32_t nb, int32_t g1, int32_t g2, int32_t g3, int32_t w1,
                           int32_t w2, int32_t w3, int32_t C, u2tok_stream_t stream);
int u2tok_score_gemv(const void* x, const void* w, const void* bias, float* scores, int32_t rows, int32_t E,
                     u2tok_stream_t stream);
int u2tok_topk_sorted(const float* scores, int64_t* idx, int32_t B, int32_t n, int32_t k, u2tok_stream_t stream);
int u2tok_gather_rows(const void* x, const int64_t* idx, void* out, int32_t B, int32_t n, int32_t k, int32_t E,
                      u2tok_stream_t stream);
/* ws: B*3*16*ceil(E/256) floats (only read/written when gate_w != null) */
int u2tok_multiscale_pool(const void* x, void* out, int32_t B, int32_t k, int32_t E, const void* gate_w,
                          const void* gate_b, float* ws, u2tok_stream_t stream);
int u2tok_temporal_attention(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t T,
                             int32_t N, int32_t H, int32_t d, int64_t ld_qkv, int64_t ld_out, float scale,
                             const void* rel_bias, int32_t max_len, u2tok_stream_t stream);
/* softmax(q k^T * scale) v for head_dim 64 (MONAI SABlock core, vit.py:100-105) over S main rows per batch plus
 * n_extra (0 / 1) extra row per batch that lives elsewhere (the cls token, kept after all patch rows): row r of batch b
 * at q + b*q_bs + r*ld_qk, head h at column 64 h; vt = V^T of the main rows, [nb][H][64][S_pad] in perm16 order, zero
 * padded; extra row of batch b at qx / kx / vx + b*x_bs, its output at outx + b*ox_bs. */
int u2tok_flash_attention_d64(const void* q, const void* k, const void* vt, void* out, int32_t nb, int32_t S,
                              int32_t H, int64_t ld_qk, int64_t q_bs, int64_t ld_out, int64_t out_bs, int32_t S_pad,
                              float scale, const void* qx, const void* kx, const void* vx, void* outx, int64_t x_bs,
                              int64_t ox_bs, int32_t n_extra, u2tok_stream_t stream);
/* Same, and lse[(b * H + h) * lse_ld + row] = log2 sum_k exp2(q_row . k scale log2 e) for every query row (the extra row at
 * index S; lse_ld >= S + n_extra): the row statistics u2tok_flash_attention_d64_bwd takes instead of rebuilding them. */
int u2tok_flash_attention_d64_lse(const void* q, const void* k, const void* vt, void* out, int32_t nb, int32_t S,
                              int32_t H, int64_t ld_qk, int64_t q_bs, int64_t ld_out, int64_t out_bs, int32_t S_pad,
                              float scale, const void* qx, const void* kx, const void* vx, void* outx, int64_t x_bs,
                              int64_t ox_bs, int32_t n_extra, float* lse, int64_t lse_ld,
                                  u2tok_stream_t stream);
/* Fused attention core of the tokenizer's attention modules -- RelativeMultiheadAttention (rma.py:60-75: + relative_bias
 * [j - i + max_len - 1][h], bf16 (2 max_len - 1, H)), RotaryMultiheadAttention (rope.py:82-86), MultiHeadCrossAttention /
 * LinearAggregation (tta.py:55-61; rel_bias NULL):  out = softmax(q k^T scale + bias) v  per (batch, head), scores and
 * probabilities never in HBM.  Row r of batch b at ptr + b*?_bs + r*ld?, head h at column h*d, d in {64, 128, 256, 512};
 * q / k / v 16-byte aligned with strides % 8 == 0, out 8-byte aligned with strides % 4 == 0; with rel_bias: Sq, Skv <=
 * max_len.  splits: 0 = heuristic, n > 0 = cut the keys into n ranges (fp32 partial sums in `workspace`:
 * u2tok_tok_attention_workspace_bytes, 16-byte aligned; NULL = unsplit).  U2TOK_ERR_ARG for shapes it does not take. */
size_t u2tok_tok_attention_workspace_bytes(int32_t nb, int32_t H, int32_t Sq, int32_t Skv, int32_t d);
int u2tok_tok_attention(const void* q, const void* k, const void* v, void* out, int32_t nb, int32_t Sq, int32_t Skv, int32_t H,
                        int32_t d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                        int64_t o_bs, float scale, const void* rel_bias, int32_t max_len, int32_t splits, void* workspace,
                        size_t workspace_bytes, u2tok_stream_t stream);

/* ---- the consumer of the path's output ("next" row f3): prefill of the stock HF decoder on the spliced embeddings ---------
 * (LlamaForCausalLM / Qwen3ForCausalLM.forward with inputs_embeds, src/model/language_model/u2llama.py:76-87,123-126).  The
 * decoder keeps its HuggingFace module tree, parameters, KV cache and generate loop; u2tokenizer_amd/prefill.py runs each
 * layer of the PREFILL as RMSNorm -> packed q|k|v GEMM -> head norm + rotary -> causal grouped-query attention -> out
 * projection (+ residual) -> RMSNorm -> packed gate|up GEMM -> SiLU(gate) * up -> down projection (+ residual). */
/* softmax(q k^T scale [causal: key j <= query i + Skv - Sq]) v with grouped-query heads: query head h (column h*d of q / out)
 * reads key / value head h / (Hq / Hkv) (column of k / v); d in {64, 128, 256, 512}; layout conventions of
 * u2tok_tok_attention. */
int u2tok_attention_gqa(const void* q, const void* k, const void* v, void* out, int32_t nb, int32_t Sq, int32_t Skv, int32_t Hq,
                        int32_t Hkv, int32_t d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs,
                        int64_t v_bs, int64_t o_bs, float scale, int32_t causal, u2tok_stream_t stream);
/* The same without the causal mask and WITH key splits (flash-decoding: partial results per key range in `workspace`, merged in a
 * fixed order): the decode step's attention of a few query rows over a long KV cache -- nb = batch x kv heads entries of
 * (Skv, d) keys, Hq = query heads per kv head, Hkv = 1 (Qwen3 / Llama decode, language_model/u2llama.py:123-126: generate()).
 * workspace: u2tok_tok_attention_workspace_bytes(nb, Hq, Sq, Skv, d) bytes, 16-byte aligned; NULL = unsplit. */
int u2tok_attention_gqa_split(const void* q, const void* k, const void* v, void* out, int32_t nb, int32_t Sq, int32_t Skv,
                              int32_t Hq, int32_t Hkv, int32_t d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs,
                              int64_t k_bs, int64_t v_bs, int64_t o_bs, float scale, void* workspace, size_t workspace_bytes,
                              u2tok_stream_t stream);
/* y[r] = bf16(x[r] * rsqrt(mean(x[r]^2) + eps)) * w   (LlamaRMSNorm / Qwen3RMSNorm); C % 8 == 0, C <= 8192 */
int u2tok_rmsnorm_bf16(const void* x, const void* w, void* y, int64_t rows, int32_t C, int64_t ldx, int64_t ldy, float eps,
                       u2tok_stream_t stream);
/* In place on the q and k heads of a packed projection output qkv[rows][(Hq + 2 Hkv) D] (D = 64 / 128): per-head RMSNorm
 * with weights wq / wk (Qwen3Attention.q_norm / k_norm; both NULL: none, Llama) and x cos + rotate_half(x) sin
 * (apply_rotary_pos_emb) with cos / sin [rows][cs_ld >= D], fp32 (cos_sin_f32 != 0) or bf16. */
int u2tok_qk_norm_rope(void* qkv, const void* wq, const void* wk, const void* cos, const void* sin, int32_t cos_sin_f32,
                       int64_t rows, int32_t Hq, int32_t Hkv, int32_t D, int64_t ld, int64_t cs_ld, float eps,
                       u2tok_stream_t stream);
/* The same, and the finished k heads and the v heads also written in the KV cache's layout: k_cache / v_cache
 * [rows / S][Hkv][capacity][D] bf16 (HF DynamicLayer: (batch, kv heads, seq, head_dim); row = batch * S + position) at positions
 * s_off .. s_off + S - 1, kv_stride = capacity * D elements between (batch, kv head) entries (0: dense, capacity = S) -- the
 * prefill hands the cache its tensors without a transposing copy, a decode step appends in place
 * (language_model/u2llama.py:123-126: generate()). */
int u2tok_qk_norm_rope_kv(void* qkv, const void* wq, const void* wk, const void* cos, const void* sin, int32_t cos_sin_f32,
                          int64_t rows, int32_t Hq, int32_t Hkv, int32_t D, int64_t ld, int64_t cs_ld, float eps,
                          void* k_cache, void* v_cache, int32_t S, int64_t kv_stride, int32_t s_off, u2tok_stream_t stream);
/* out[r][i] = bf16(silu(gate_up[r][i])) * gate_up[r][I + i]   (LlamaMLP / Qwen3MLP with gate | up packed); I % 8 == 0 */
int u2tok_swiglu_bf16(const void* gate_up, void* out, int64_t rows, int32_t I, int64_t ld_in, int64_t ld_out,
                      u2tok_stream_t stream);

/* ---- one decode step of a decoder layer (the step HF generate() repeats per new token: language_model/u2llama.py:123-126,
 * eval/mrg.py:74-77 asks for up to 768) in two calls -- between them the host appends the new k / v to its KV cache.
 *   pre : input RMSNorm -> packed q|k|v projection (few-rows GEMM) -> per-head q / k RMSNorm (NULL weights: none) + rotary;
 *         qkv (B, (Hq + 2 Hkv) D) keeps the finished queries; the step's keys / values go to position s_off of k_cache / v_cache
 *         (B, Hkv, capacity, D; kv_stride = capacity * D, 0: a dense (B, Hkv, 1, D) pair with s_off = 0)
 *   post: attention of the B query rows over the first T positions of K / V (same layout; keys split over workgroups, merged
 *         in a fixed order) -> out projection + residual x -> RMSNorm -> packed gate|up -> SiLU(gate) * up -> down + residual
 * B <= 16, D in {64, 128}, E % 32 == 0, I % 32 == 0; biases may be NULL; same rounding points as the HF modules in bf16.
 * One workspace for both calls: u2tok_decoder_decode_workspace_bytes(cfg, T) bytes. */
typedef struct u2tok_decode_config {
  int32_t B, E, Hq, Hkv, D, I; /* new tokens (= batch), hidden size, query / key-value heads, head dim, MLP width */
  float eps, qk_eps, scale;    /* RMSNorm eps, q / k norm eps, softmax scale */
} u2tok_decode_config;
size_t u2tok_decoder_decode_workspace_bytes(const u2tok_decode_config* cfg, int32_t T);
int u2tok_decoder_decode_pre(const u2tok_decode_config* cfg, const void* x, const void* w_in_norm, const void* Wqkv,
                             const void* bqkv, const void* wq_norm, const void* wk_norm, const void* cos, const void* sin,
                             int32_t cos_sin_f32, int64_t cs_ld, void* qkv, void* k_cache, void* v_cache, int64_t kv_stride,
                             int32_t s_off, void* workspace, size_t workspace_bytes, u2tok_stream_t stream);
int u2tok_decoder_decode_post(const u2tok_decode_config* cfg, const void* x, const void* qkv, const void* K, const void* V,
                              int32_t T, int64_t kv_stride, const void* Wo, const void* bo, const void* w_post_norm, const void* Wgu,
                              const void* bgu, const void* Wdown, const void* bdown, void* out, void* workspace,
                              size_t workspace_bytes, u2tok_stream_t stream);

/* in-place rotate-half RoPE (rope.py:6-13,77-80): rows indexed (outer, s, inner), position = s; inverse != 0 rotates
 * the other way (the backward of the rotation) */
int u2tok_rope_apply(void* x, int64_t n_outer, int32_t S, int32_t n_inner, int32_t H, int32_t d, int64_t ld,
                     int32_t max_len, int32_t inverse, u2tok_stream_t stream);

/* ---- backward-pass building blocks ("next" row f1: training through the path) ---------------------------------------
 * The GEMM-shaped parts of the backward (dX = dY W, dW = dY^T X, dQ / dK / dV / dP of the attention cores) are
 * u2tok_gemm_bf16 calls on (transposed) operands; these are the rest.  The host side (u2tokenizer_amd/autograd.py)
 * sequences them behind torch.autograd.Function so that the drop-in modules train (train_stage1.py:244-251). */
int u2tok_gelu_fwd(const void* z, void* y, int64_t n, u2tok_stream_t stream);                  /* y = gelu(z) (erf) */
int u2tok_gelu_bwd(const void* z, const void* dy, void* dz, int64_t n, u2tok_stream_t stream); /* dz = dy gelu'(z) */
/* out[c] (+)= sum_r x[r][c] (* y[r][c] when y != NULL) in fp32, fixed summation order (bit-repeatable); out (fp32)
 * and / or out_bf16 receive the result; workspace: u2tok_colsum_workspace_bytes. */
size_t u2tok_colsum_workspace_bytes(int32_t rows, int32_t C);
int u2tok_colsum_bf16(const void* x, const void* y, float* out, void* out_bf16, int32_t rows, int32_t C, int64_t ldx,
                      int64_t ldy, void* workspace, int32_t accumulate, u2tok_stream_t stream);
/* Backward of y = LayerNorm(x (+ res)) * w + b (rows x C, dense): dv = gradient w.r.t. x (== w.r.t. res), dw / db:
 * fp32 [C] (overwritten, or added to when accumulate != 0). */
size_t u2tok_layernorm_bwd_workspace_bytes(int32_t rows, int32_t C);
int u2tok_layernorm_bwd(const void* x, const void* res, const void* w, const void* dy, void* dv, float* dw, float* db,
                        int32_t rows, int32_t C, float eps, void* workspace, int32_t accumulate, u2tok_stream_t stream);
/* dS = P * (dP - rowsum(P * dP)) per row; P, dS bf16 [nrows][ldp] (columns >= n of dS zeroed), dP fp32 [nrows][lddp] */
int u2tok_softmax_bwd(const void* P, const float* dP, void* dS, int64_t nrows, int32_t n, int64_t ldp, int64_t lddp,
                      u2tok_stream_t stream);
/* gradient of the relative-bias table (rma.py:64-70): dtable[d + max_len - 1][h] += sum over z % H == h and the
 * diagonal j - i = d of dS[z][i][j];  dS: bf16 [nz][S][ldp];  dtable: fp32 [2 max_len - 1][H] */
int u2tok_relbias_grad(const void* dS, float* dtable, int32_t nz, int32_t S, int32_t H, int64_t ldp, int32_t max_len,
                       u2tok_stream_t stream);
int u2tok_rowdot_bf16(const void* a, const void* b, float* out, int64_t rows, int32_t C, int64_t lda, int64_t ldb,
                      u2tok_stream_t stream);

/* One rank's shard of a ZeRO-1 AdamW step (config/ds_config.json:27-41; u2tokenizer_amd/dp.py): fp32 master weights and
 * moments (n elements each) updated in place from the bf16 gradient piece scaled by grad_scale (1 / world size) and, when
 * grad_coef != NULL, by the device scalar *grad_coef (the clipping coefficient); out_bf16 receives the rounded new master
 * (the piece the all-gather sends).  torch.optim.AdamW arithmetic; step = 1-based step count (bias corrections).  group:
 * optional per-element parameter-group index (uint8) into the host tables lr[ngroups] / weight_decay[ngroups], ngroups <= 8. */
int u2tok_adamw_step(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, const uint8_t* group, void* out_bf16,
                     int64_t n, const float* lr, const float* weight_decay, int32_t ngroups, float beta1, float beta2, float eps,
                     int32_t step, float grad_scale, const float* grad_coef, u2tok_stream_t stream);

/* Fused backward of the ViT attention core (MONAI SABlock, vit.py:100-105; head dim 64, no bias):
 *   out = softmax(q k^T * scale) v   ->   dq, dk, dv   from q, k, v, out and d_out,
 * replacing what torch.autograd does for the reference (softmax / matmul backward through the (S x S) probabilities, 1.6 GB
 * fp32 per layer at S = 2049) by two flash-style kernels that rebuild the probabilities tile by tile.  All S rows of a batch
 * in one row-major bf16 view: q, k, v row r of batch b at + b*bs_qkv + r*ld_qkv, head h at column h*64 (16-byte aligned
 * bases, strides % 8 == 0); out / d_out with ld_o / bs_o; dq, dk, dv with ld_d / bs_d (may be slices of one packed
 * buffer).  lse: the row statistics of u2tok_flash_attention_d64_lse ((nb * H, lse_ld) floats), or NULL -- the backward then
 * rebuilds them in an extra sweep over the keys.  workspace: u2tok_flash_attention_d64_bwd_workspace_bytes(), 256-byte
 * aligned.  No atomics: bit-repeatable. */
size_t u2tok_flash_attention_d64_bwd_workspace_bytes(int32_t nb, int32_t S, int32_t H);
int32_t u2tok_flash_attention_d64_bwd(const void* q, const void* k, const void* v, int64_t ld_qkv, int64_t bs_qkv,
                                      const void* out, const void* d_out, int64_t ld_o, int64_t bs_o, void* dq, void* dk,
                                      void* dv, int64_t ld_d, int64_t bs_d, int32_t nb, int32_t S, int32_t H, float scale,
                                      const float* lse, int64_t lse_ld, void* workspace, size_t workspace_bytes,
                                      u2tok_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* U2TOK_H_ */
