// Internal C++ launcher API of the u2tok HIP library.  Every function enqueues work on `stream`,
// never allocates, never synchronises, and returns U2_OK or a negative U2_ERR_* code.
#pragma once
#include <algorithm>
#include "common.h"
#include "ctx.h"

namespace u2 {

// ------------------------------------------------------------------ profiling (ctx.hip)
enum : int { PROF_GEMM = 0, PROF_FLASH = 1, PROF_TEMPORAL = 2, PROF_ROWOP = 3, PROF_MOVE = 4, PROF_TOKATTN = 5, PROF_NCAT = 6 };
void prof_enable(bool on);
bool prof_enabled();
int prof_collect(double* ms, double* flops, double* bytes, int64_t* count, int ncat);  // bytes may be null
struct ProfScope {  // brackets one launch with hipEvents on `st` when the current context profiles; free otherwise
  // flops / bytes: ALGORITHMIC work of the launch (2MNK; operands + results once), for the roofline lines of bench.py
  ProfScope(int cat, double flops, hipStream_t st, double bytes = 0.0);
  ~ProfScope();
  int idx_;
  hipStream_t st_;
  Context* c_;
};

// ------------------------------------------------------------------ GEMM (gemm.hip)
enum : int {
  GEMM_BIAS_N = 1,     // + bias[n]   (nn.Linear bias)
  GEMM_BIAS_M = 2,     // + bias[m]   (operand-swapped products, e.g. DiffTS scores^T)
  GEMM_GELU = 4,       // exact erf GELU after bias
  GEMM_RESIDUAL = 8,   // + R[m][n]   (residual stream / position embedding), after GELU
  GEMM_OUT_F32 = 16,   // C is float32 instead of bf16
  GEMM_VEC_OK = 32,    // internal: vector epilogue legal (set by the launcher)
  GEMM_B_KTILE = 64,   // C-ABI only: B is K-tile-major [K/64][N][64] (pack_ktile_major); K % 64 == 0, nz == 1
  GEMM_A_KMAJOR = 128, // A is stored [K][M] (leading dim lda >= M, M % 8 == 0): C = A^T B^T-form products without a transpose
  GEMM_SWIGLU = 512,   // B = [gate rows | up rows] (N = 2 I): C[m][j] = bf16(silu(gate_j)) * up_j, C is [M][I] bf16; no other epilogue
                       // flag, K % 64 == 0, I % 16 == 0 (the 256 x 192-tile kernel stages gate / up rows pairwise, gemm_bt.hip)
  GEMM_B_KMAJOR = 256, // B is stored [K][N] (leading dim ldb >= N, N % 8 == 0); A K-major requires B K-major too
};

struct GemmDesc {
  const bf16_t* A = nullptr;  // [M][K] row-major, leading dim lda ([K][M] with GEMM_A_KMAJOR)
  const bf16_t* B = nullptr;  // [N][K] row-major, leading dim ldb ([K][N] with GEMM_B_KMAJOR)
  void* C = nullptr;          // [M][N] bf16 or f32, leading dim ldc
  const bf16_t* bias = nullptr;
  const bf16_t* R = nullptr;  // [M][N] bf16, leading dim ldr
  int M = 0, N = 0, K = 0;
  int64_t lda = 0, ldb = 0, ldc = 0, ldr = 0;
  int64_t ldbk = 0;  // 0: B rows are K-contiguous; else B is K-tile-major [K/64][N][64] (ldb = 64, ldbk = N * 64)
  // batch z in [0, nz): zb = z / nbh, zh = z % nbh; element offsets zb*s?b + zh*s?h
  int nz = 1, nbh = 1;
  int64_t sAb = 0, sAh = 0, sBb = 0, sBh = 0, sCb = 0, sCh = 0, sRb = 0, sRh = 0;
  float alpha = 1.f;
  // columns n < nsplit (a multiple of 16) take alpha_lo instead of alpha: the ViT's q|k|v product leaves its q columns
  // multiplied by softmax scale * log2 e, from the fp32 accumulator (what flash_attention_d64(..., q_prescaled = 1) reads)
  int nsplit = 0;
  float alpha_lo = 1.f;
  int flags = 0;
  int tiles_m = 0, tiles_n = 0;  // filled by the launcher
  // split-K (filled by the launcher): K tiles [s * kt_per, (s + 1) * kt_per) go to grid.z slice s, which leaves raw
  // fp32 sums in partial[s][z][m][n] (dense, ld = N); gemm_splitk_reduce_kernel applies the epilogue
  int ksplit = 1, kt_per = 0;
  float* partial = nullptr;
  // transposed side output (gemm_vt_supported() first): columns n >= vt_n0 are NOT written to C but, transposed, to
  // vt[m / vt_rows][n - vt_n0][m % vt_rows] (leading dim vt_ld keys, chunk stride vt_bs) in the flash kernel's key order
  // (perm16 of transpose_bf16) -- the ViT's q|k|v product writes V^T itself.  Rows of a few-rows tail still go to C.
  bf16_t* vt = nullptr;
  int vt_n0 = 0, vt_rows = 0;
  int64_t vt_ld = 0, vt_bs = 0;
  // in-launch tail (filled by the big-tile launcher): rows [M, M + tail_rows) of A / C / R (<= 16: the ViT's cls rows behind the
  // patch rows) are computed inside the same launch by the few-rows arithmetic (rows16.h) instead of a launch of their own
  int tail_rows = 0;
  // classic tile kernel (filled by its launcher): 1 = the LDS-DMA pieces leave as buffer_load ... lds (both operands of a batch entry
  // span < 2 GB), 0 = FLAT-encoded global_load_lds (option gemm_mubuf 0, or larger operands)
  int mubuf = 0;
  // big-tile kernels (filled by their launchers): row tiles per group of the tile walk (pp_tile: ids walk `group_m` row tiles column by column)
  int group_m = 8;
};

// Options (tile, split-K, big-tile selection) and the split-K scratch registration of the launch stream come from the
// calling thread's current Context (ctx.h).  Without a registered scratch, products run unsplit.
int gemm_bf16(GemmDesc d, hipStream_t stream);
// internal: the kernels behind gemm_bf16 (descriptor already validated there)
int gemm_classic(GemmDesc d, hipStream_t stream);        // gemm.hip: 128^2 / 64^2 tiles, 2+ workgroups per CU
int gemm_big_try(const GemmDesc& d, hipStream_t stream);  // gemm_bt.hip: 1 launched, 0 not applicable, < 0 error
int gemm_skinny_try(const GemmDesc& d, hipStream_t stream);  // gemm_skinny.hip (M <= 256 against a cold weight): same convention
bool gemm_vt_supported(const GemmDesc& d, int vt_n0, int vt_rows);  // gemm_bt.hip: may d.vt be set for this product?
int gemm_splitk_reduce(const GemmDesc& d, hipStream_t stream);  // gemm.hip: epilogue over d.partial[ksplit][nz][M][N]

// ------------------------------------------------------------------ row ops (rowops.hip)
// y[b][r][:] = LayerNorm(x[b][r][:] (+ res[b][r][:])) * w + bias   (bf16 in/out, fp32 math)
int layernorm_bf16(const bf16_t* x, const bf16_t* res, const bf16_t* w, const bf16_t* bias, bf16_t* y,
                   int nb, int rows, int C, int64_t x_bs, int64_t x_ld, int64_t res_bs, int64_t res_ld,
                   int64_t y_bs, int64_t y_ld, float eps, hipStream_t stream);

// P[z][r][c] = softmax_c( S[z][r][c] * scale + rel_bias[(c - r) + max_len - 1][z % H] ), bf16 out,
// columns [n, ldp) of P are written as zero.  rel_bias may be null.
int softmax_rows(const float* S, bf16_t* P, int nz, int rows, int n, int64_t lds, int64_t ldp,
                 int64_t s_zs, int64_t p_zs, float scale, const bf16_t* rel_bias, int H, int max_len,
                 hipStream_t stream);

// out[z][c][r] = in[z][r][c]; columns [R, ld_out) of out are zero-filled.  perm16 != 0: inside every group of
// 16 output columns the order is [0-3, 8-11, 4-7, 12-15] (layout the flash-attention kernel expects for V^T).
int transpose_bf16(const bf16_t* in, bf16_t* out, int nz, int R, int C, int64_t ld_in, int64_t ld_out,
                   int64_t in_zs, int64_t out_zs, int perm16, hipStream_t stream);

// ------------------------------------------------------------------ vision (vision.hip)
enum : int { VOL_F16 = 0, VOL_BF16 = 1, VOL_F32 = 2 };
// im2col of reference PatchEmbeddingBlock(perceptron): "b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)"
// with c == 1.  vol: [nchunk][D][H][W] of vol_dtype; out: [nchunk][nh*nw*nd][p1*p2*p3] bf16.
int im2col_patches(const void* vol, int vol_dtype, bf16_t* out, int nchunk, int D, int H, int W,
                   int p1, int p2, int p3, hipStream_t stream);

// SpatialPoolingProjector pooling: tokens (g1,g2,g3) grid -> avg over ps^3 neighbourhoods.
int avgpool3d_tokens(const bf16_t* x, bf16_t* y, int nb, int g1, int g2, int g3, int w1, int w2, int w3, int C,
                     hipStream_t stream);
// dst[b*dst_bs + e] = src[e] for e < n  (cls token / query token broadcast)
int fill_rows(const bf16_t* src, bf16_t* dst, int nb, int64_t n, int64_t dst_bs, hipStream_t stream);
// in-place rotate-half RoPE on rows indexed (outer, s, inner), position = s, heads = d-wide column slices
int rope_apply(bf16_t* x, int64_t n_outer, int S, int n_inner, int H, int d, int64_t ld, int max_len, int inverse,
               hipStream_t stream);

// out[b][s] = (s == 0 || s > nfeat) ? table[ids[b][s]] : feats[b][s-1]   (embedding lookup + splice)
// feats may be null with nfeat == 0 (plain lookup).
int embed_splice(const bf16_t* table, const int64_t* ids, const bf16_t* feats, bf16_t* out, int B, int S,
                 int E, int nfeat, int64_t vocab, hipStream_t stream);

// ------------------------------------------------------------------ volume preprocessing (preprocess.hip)
// u2Transform.adaptive_resize on the GPU: vol [D][H][W] fp32 (the tensor of u2Transform.py:68-69 without its channel
// axis) -> out [depth_pad][target][target] (== (depth_pad/32, 32, target, target)) in out_dtype (VOL_*).
// info (optional, device, 12 x int32): status, crop box lo[3], hi[3], resized size[3], then a_min, a_max as float.
// aug (may be null): the training-time augmentations of u2Transform.py:37-42 with their random draws made by the caller
struct PreAugment {
  int rot_k = 0;              // RandRotate90(spatial_axes=(1, 2)): quarter turns of the (h, w) plane, torch.rot90 convention
  int flip[3] = {0, 0, 0};    // RandFlip(spatial_axis = 0 / 1 / 2) on the rotated volume
  float mul = 1.f;            // RandScaleIntensity: v * (1 + factor)
  float add = 0.f;            // RandShiftIntensity: v + offset
};
size_t preprocess_workspace_bytes(int D, int H, int W);
int preprocess_volume(const float* vol, void* out, int32_t* info, int D, int H, int W, int target, int depth_pad,
                      float lower_pct, float upper_pct, int out_dtype, const PreAugment* aug, void* ws, size_t ws_bytes,
                      hipStream_t stream);

// ------------------------------------------------------------------ selection / pooling (select.hip)
// scores[b][i] = fp32( sum_e x[b][i][e] * w[e] + bias ), accumulated in fp64.
int score_gemv(const bf16_t* x, const bf16_t* w, const bf16_t* bias, float* scores, int rows, int E,
               hipStream_t stream);
// idx[b][0..k) = indices of the k largest scores[b][0..n), descending, ties -> lower index first.
int topk_sorted(const float* scores, int64_t* idx, int B, int n, int k, hipStream_t stream);
// out[b][i][:] = x[b][idx[b][i]][:]
int gather_rows(const bf16_t* x, const int64_t* idx, bf16_t* out, int B, int n, int k, int E,
                hipStream_t stream);
// Multi-scale pooling {1,2,4} along the token axis (+ optional DMTP gates).  x: [B][k][E];
// out: [B][k + k/2 + k/4][E].  gate_w/gate_b null -> fixed pooling.  ws: >= B*3*16*ceil(E/256) floats.
int multiscale_pool(const bf16_t* x, bf16_t* out, int B, int k, int E, const bf16_t* gate_w,
                    const bf16_t* gate_b, float* ws, hipStream_t stream);

// ------------------------------------------------------------------ backward-pass kernels (backward.hip)
int gelu_fwd(const bf16_t* z, bf16_t* y, int64_t n, hipStream_t stream);                      // y = gelu(z), n % 8 == 0
int gelu_bwd(const bf16_t* z, const bf16_t* dy, bf16_t* dz, int64_t n, hipStream_t stream);  // dz = dy gelu'(z)
// out[c] (+)= sum_r x[r][c] (* y[r][c] when y != null), fp32, fixed summation order; ws: colsum_workspace_bytes
size_t colsum_workspace_bytes(int rows, int C);
int colsum_bf16(const bf16_t* x, const bf16_t* y, float* out, bf16_t* out_bf16, int rows, int C, int64_t ldx, int64_t ldy,
                float* ws, int accumulate, hipStream_t stream);
// LayerNorm backward of y = LN(x (+ res)) w + b: dv = gradient w.r.t. x (and res), dw / db fp32 [C]
size_t layernorm_bwd_workspace_bytes(int rows, int C);
int layernorm_bwd(const bf16_t* x, const bf16_t* res, const bf16_t* w, const bf16_t* dy, bf16_t* dv, float* dw, float* db,
                  int rows, int C, float eps, float* ws, int accumulate, hipStream_t stream);
// dS = P (dP - rowsum(P dP)); pad columns [n, ldp) zeroed
int softmax_bwd(const bf16_t* P, const float* dP, bf16_t* dS, int64_t nrows, int n, int64_t ldp, int64_t lddp,
                hipStream_t stream);
// dtable[d + max_len - 1][h] += sum over z % H == h and the diagonal j - i = d of dS[z][i][j]   (dS: [nz][S][ldp])
int relbias_grad(const bf16_t* dS, float* dtable, int nz, int S, int H, int64_t ldp, int max_len, hipStream_t stream);
int rowdot_bf16(const bf16_t* a, const bf16_t* b, float* out, int64_t rows, int C, int64_t lda, int64_t ldb,
                hipStream_t stream);
// sharded AdamW step (dp.py): see backward.hip
struct AdamWArgs {
  float lr[8], wd[8];  // per parameter group
  float b1, b2, eps, inv_c1, inv_sqrt_c2, gscale;
  const float* gcoef;  // optional device scalar multiplied into the gradient (clipping coefficient)
};
int adamw_step(float* master, float* m, float* v, const bf16_t* grad, const uint8_t* group, bf16_t* out, int64_t n,
               const AdamWArgs& a, hipStream_t stream);

// ------------------------------------------------------------------ attention (attn.hip)
// Temporal attention of SpatioTemporalAttentionLayer: sequences of length T <= 16 that run ACROSS
// chunks.  q/k/v/out rows are laid out [b][t][n] (leading dim ld), heads are column slices of width d.
int temporal_attention(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int B, int T, int N,
                       int H, int d, int64_t ld_qkv, int64_t ld_out, float scale, const bf16_t* rel_bias,
                       int max_len, hipStream_t stream);

// Flash attention for the ViT blocks (head_dim 64) over S main rows per batch plus n_extra (0 / 1) extra row per
// batch stored elsewhere (the cls token).  q,k: row r of batch b at q + b*q_bs + r*ld_qk, head h at column h*64;
// vt: V^T of the main rows, [nb][H][64][S_pad] (S_pad % 64 == 0, zero padded, perm16 column order); out like q with
// ld_out / out_bs.  Extra row of batch b: qx / kx / vx + b*x_bs, output outx + b*ox_bs (head h at column h*64).
int flash_attention_d64(const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* out, int nb, int S, int H,
                        int64_t ld_qk, int64_t q_bs, int64_t ld_out, int64_t out_bs, int S_pad, float scale,
                        const bf16_t* qx, const bf16_t* kx, const bf16_t* vx, bf16_t* outx, int64_t x_bs, int64_t ox_bs,
                        int n_extra, float* lse, int64_t lse_ld, hipStream_t stream,  // lse: optional row statistics out
                        int q_prescaled = 0);  // 1: q and qx already carry scale * log2 e (S >= 512: the double pipeline)
// Backward of the same attention (attn_bwd.hip): dq / dk / dv of out = softmax(q k^T scale) v for head dim 64, all S rows
// of a batch in ONE row-major view (q, k, v: row r of batch b at + b*bs_qkv + r*ld_qkv, head h at column h*64; o / dout with
// ld_o / bs_o; dq / dk / dv with ld_d / bs_d).  lse: optional row statistics of the forward kernel.  Workspace: the row statistics (lse, D).
size_t flash_attention_d64_bwd_workspace_bytes(int nb, int S, int H);
int flash_attention_d64_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld_qkv, int64_t bs_qkv, const bf16_t* o,
                            const bf16_t* dout, int64_t ld_o, int64_t bs_o, bf16_t* dq, bf16_t* dk, bf16_t* dv, int64_t ld_d,
                            int64_t bs_d, int nb, int S, int H, float scale, const float* lse, int64_t lse_ld,
                            void* workspace, size_t workspace_bytes, hipStream_t stream);
// Fused attention core of the tokenizer's attention modules (tokattn.hip; rma.py:60-75, tta.py:55-61, rope.py:82-86):
// out = softmax(q k^T scale + rel_bias[j - i + max_len - 1][h]) v per (batch, head); row r of batch b at + b*?_bs + r*ld?, head
// h at column h*d; d in {64, 128, 256, 512}.  Few (batch, head, 64-query) units -> the key range is cut into splits whose
// fp32 partial results go through `ws` (tok_attention_workspace_bytes; may be null: unsplit).  force_splits > 0 overrides
// the heuristic (tests).  tok_attention_supported: shapes / strides the kernel takes (callers fall back to the GEMM chain).
size_t tok_attention_workspace_bytes(int nb, int H, int Sq, int Skv, int d);
bool tok_attention_supported(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* out, int Sq, int Skv, int d,
                             int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                             int64_t o_bs, const bf16_t* rel_bias, int max_len);
int tok_attention(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int nb, int Sq, int Skv, int H, int d,
                  int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                  float scale, const bf16_t* rel_bias, int max_len, int force_splits, void* ws, size_t ws_bytes,
                  hipStream_t stream);
// The same kernel for the decoder prefill: grouped-query heads (query head h reads key / value head h / (H / Hkv)) and a causal
// mask (key j visible to query i iff j <= i + Skv - Sq); no relative bias and no key splits with causal.
int attention_ex(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int nb, int Sq, int Skv, int H, int Hkv, int d,
                 int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                 float scale, const bf16_t* rel_bias, int max_len, int causal, int force_splits, void* ws, size_t ws_bytes,
                 hipStream_t stream);
int tok_attention_set_debug_buffer(void* p);
// ------------------------------------------------------------------ decoder prefill row kernels (decoder.hip)
int rmsnorm_bf16(const bf16_t* x, const bf16_t* w, bf16_t* y, int64_t rows, int C, int64_t ldx, int64_t ldy, float eps,
                 hipStream_t stream);
int qk_norm_rope(bf16_t* qkv, const bf16_t* wq, const bf16_t* wk, const void* cosp, const void* sinp, int cs_is_f32,
                 int64_t rows, int Hq, int Hkv, int D, int64_t ld, int64_t cs_ld, float eps, bf16_t* kc, bf16_t* vc, int S,
                 int64_t kv_stride, int s_off, hipStream_t stream);
int swiglu_bf16(const bf16_t* gu, bf16_t* out, int64_t rows, int I, int64_t ld_in, int64_t ld_out, hipStream_t stream);
struct DecodeCfg {  // one decode step of a decoder layer (decoder.hip)
  int B, E, Hq, Hkv, D, I;
  float eps, qk_eps, scale;
};
size_t decoder_decode_workspace_bytes(const DecodeCfg& c, int T);
int decoder_decode_pre(const DecodeCfg& c, const bf16_t* x, const bf16_t* w_in_norm, const bf16_t* Wqkv, const bf16_t* bqkv,
                       const bf16_t* wq_norm, const bf16_t* wk_norm, const void* cosp, const void* sinp, int cs_is_f32,
                       int64_t cs_ld, bf16_t* qkv, bf16_t* kc, bf16_t* vc, int64_t kv_stride, int s_off, void* ws, size_t ws_bytes,
                       hipStream_t st);
int decoder_decode_post(const DecodeCfg& c, const bf16_t* x, const bf16_t* qkv, const bf16_t* K, const bf16_t* V, int T,
                        int64_t kv_stride, const bf16_t* Wo, const bf16_t* bo, const bf16_t* w_post_norm, const bf16_t* Wgu, const bf16_t* bgu,
                        const bf16_t* Wdown, const bf16_t* bdown, bf16_t* out, void* ws, size_t ws_bytes, hipStream_t st);  // diagnostics: >= grid * 4 * 8 uint64, zeroed; null detaches (instrumented build)
// Diagnostics only (process-wide, not for concurrent use): s_memtime phase sums per (workgroup, wave) of the double
// pipeline kernel; while a buffer is attached the kernel runs its instrumented build.  See attn.hip.
int flash_set_debug_buffer(void* p);

}  // namespace u2
