// extern "C" surface of libu2tok_hip.so (declared in include/u2tok.h).  Thin: argument marshalling only.
#include <math.h>
#include <string.h>
#include <new>
#include "pipeline.h"

using namespace u2;

static_assert(U2TOK_OK == U2_OK && U2TOK_ERR_ARG == U2_ERR_ARG && U2TOK_ERR_LAUNCH == U2_ERR_LAUNCH &&
                  U2TOK_ERR_WORKSPACE == U2_ERR_WORKSPACE && U2TOK_ERR_DEVICE == U2_ERR_DEVICE,
              "public and internal status codes must agree");

#define BF(p) reinterpret_cast<const bf16_t*>(p)
#define BFW(p) reinterpret_cast<bf16_t*>(p)
#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int u2tok_version(void) { return 100; /* 0.1.0 */ }
const char* u2tok_arch(void) { return "gfx950"; }
const char* u2tok_elem(void) { return U2_ELEM_ASM; }

int u2tok_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return U2_ERR_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return U2_ERR_DEVICE;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? U2_OK : U2_ERR_DEVICE;
}

// ---- contexts (ctx.h)
int u2tok_ctx_create(u2tok_ctx_t* out) {
  if (!out) return U2_ERR_ARG;
  *out = reinterpret_cast<u2tok_ctx_t>(new (std::nothrow) Context());
  return *out ? U2_OK : U2_ERR_ARG;
}
int u2tok_ctx_destroy(u2tok_ctx_t c) {
  if (!c) return U2_ERR_ARG;
  Context* cx = reinterpret_cast<Context*>(c);
  if (ctx_bound() == cx) ctx_bind(nullptr);
  delete cx;
  return U2_OK;
}
int u2tok_ctx_set_current(u2tok_ctx_t c) {
  ctx_bind(reinterpret_cast<Context*>(c));
  return U2_OK;
}
u2tok_ctx_t u2tok_ctx_get_current(void) { return reinterpret_cast<u2tok_ctx_t>(ctx_bound()); }

int u2tok_set_option(const char* name, int value) {
  if (!name) return U2_ERR_ARG;
  Options& o = ctx().opt;
  struct Opt { const char* name; int Options::*field; int lo, hi; };
  static const Opt table[] = {
      {"gemm_splitk", &Options::gemm_splitk, -1, 16},   {"gemm_mubuf", &Options::gemm_mubuf, 0, 1}, {"ln_wide", &Options::ln_wide, 0, 1},   {"gemm_big", &Options::gemm_big, -1, 27},
      {"gemm_big_grid", &Options::gemm_big_grid, 1, 4096}, {"gemm_big_group_m", &Options::gemm_big_group_m, 0, 64}, {"gemm_big_gelu", &Options::gemm_big_gelu, 0, 1},
      {"gemm_big_splitk", &Options::gemm_big_splitk, 0, 16}, {"gemm_big_skinny", &Options::gemm_big_skinny, 0, 1},
      {"gemm_big_ring", &Options::gemm_big_ring, 0, 1},   {"gemm_big_deep", &Options::gemm_big_deep, 0, 1},
      {"gemm_big_drain", &Options::gemm_big_drain, 0, 2}, {"gemm_skinny", &Options::gemm_skinny, 0, 2},
       {"kmajor_b", &Options::kmajor_b, 0, 1},            {"gemm_tail_fused", &Options::gemm_tail_fused, 0, 1},
      {"flash_mode", &Options::flash_mode, 0, 7},        {"flash_q_prescaled", &Options::flash_q_prescaled, 0, 1},
      {"vit_flash", &Options::vit_flash, 0, 1},         {"tta_overlap", &Options::tta_overlap, 0, 1},
      {"vit_vt_epilogue", &Options::vit_vt_epilogue, 0, 1},
      {"tok_flash", &Options::tok_flash, 0, 1},
      {"tok_wide", &Options::tok_wide, 0, 2},
  };
  if (!strcmp(name, "gemm_tile")) {
    if (value != 0 && value != 64 && value != 128) return U2_ERR_ARG;
    o.gemm_tile = value;
    return U2_OK;
  }
  if (!strcmp(name, "profile")) { prof_enable(value != 0); return U2_OK; }
  for (const Opt& t : table)
    if (!strcmp(name, t.name)) {
      if (value < t.lo || value > t.hi) return U2_ERR_ARG;
      if (t.field == &Options::gemm_big && value > 0 && !(value >= 20 && value <= 27)) return U2_ERR_ARG;
      if (t.field == &Options::flash_mode && value != 0 && value != 1 && value != 7) return U2_ERR_ARG;
      o.*(t.field) = value;
      return U2_OK;
    }
  return U2_ERR_ARG;
}

int u2tok_set_gemm_scratch(void* device_ptr, size_t bytes, void* stream) {
  ctx().set_scratch(reinterpret_cast<hipStream_t>(stream), device_ptr, device_ptr ? bytes : 0);
  return U2_OK;
}
int u2tok_flash_debug_buffer(void* device_ptr) { return flash_set_debug_buffer(device_ptr); }
int u2tok_tok_attention_debug_buffer(void* device_ptr) { return tok_attention_set_debug_buffer(device_ptr); }

int u2tok_profile_collect(double* ms, double* flops, int64_t* count, int32_t ncat) {
  if (!ms || !flops || !count || ncat <= 0 || ncat > PROF_NCAT) return U2_ERR_ARG;
  return prof_collect(ms, flops, nullptr, count, ncat);
}
int u2tok_profile_collect2(double* ms, double* flops, double* bytes, int64_t* count, int32_t ncat) {
  if (!ms || !flops || !bytes || !count || ncat <= 0 || ncat > PROF_NCAT) return U2_ERR_ARG;
  return prof_collect(ms, flops, bytes, count, ncat);
}

size_t u2tok_vit_workspace_bytes(const u2tok_vit_config* cfg) {
  if (!cfg) return 0;
  size_t peak = 0;
  if (vit_forward(*cfg, nullptr, nullptr, nullptr, nullptr, 0, true, &peak, nullptr) != U2_OK) return 0;
  return peak + 256;
}
int u2tok_vit_forward(const u2tok_vit_config* cfg, const void* const* weights, const void* volume, void* out,
                      void* workspace, size_t workspace_bytes, u2tok_stream_t stream) {
  if (!cfg || !workspace || ((uintptr_t)workspace & 255)) return U2_ERR_ARG;
  return vit_forward(*cfg, weights, volume, BFW(out), workspace, workspace_bytes, false, nullptr, ST(stream));
}

size_t u2tok_spp_workspace_bytes(const u2tok_spp_config* cfg) {
  if (!cfg) return 0;
  size_t peak = 0;
  if (spp_forward(*cfg, nullptr, nullptr, nullptr, nullptr, 0, true, &peak, nullptr) != U2_OK) return 0;
  return peak + 256;
}
int u2tok_spp_forward(const u2tok_spp_config* cfg, const void* const* weights, const void* x, void* out,
                      void* workspace, size_t workspace_bytes, u2tok_stream_t stream) {
  if (!cfg || !workspace || ((uintptr_t)workspace & 255)) return U2_ERR_ARG;
  return spp_forward(*cfg, weights, BF(x), BFW(out), workspace, workspace_bytes, false, nullptr, ST(stream));
}

size_t u2tok_tokenizer_workspace_bytes(const u2tok_tokenizer_config* cfg) {
  if (!cfg) return 0;
  size_t peak = 0;
  if (tokenizer_forward(*cfg, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, true, &peak,
                        nullptr) != U2_OK)
    return 0;
  return peak + 256;
}
int u2tok_tokenizer_forward(const u2tok_tokenizer_config* cfg, const void* const* weights, const void* v_token,
                            const void* t_token, void* out, int64_t* topk_idx_out, void* svr_out, void* workspace,
                            size_t workspace_bytes, u2tok_stream_t stream) {
  if (!cfg || !workspace || ((uintptr_t)workspace & 255)) return U2_ERR_ARG;
  return tokenizer_forward(*cfg, weights, BF(v_token), BF(t_token), BFW(out), topk_idx_out, BFW(svr_out), nullptr, workspace,
                           workspace_bytes, false, nullptr, ST(stream));
}
int u2tok_tokenizer_forward_taps(const u2tok_tokenizer_config* cfg, const void* const* weights, const void* v_token,
                                 const void* t_token, void* out, int64_t* topk_idx_out, const u2tok_tokenizer_taps* taps,
                                 void* workspace, size_t workspace_bytes, u2tok_stream_t stream) {
  if (!cfg || !taps || !workspace || ((uintptr_t)workspace & 255)) return U2_ERR_ARG;
  return tokenizer_forward(*cfg, weights, BF(v_token), BF(t_token), BFW(out), topk_idx_out, nullptr, taps, workspace,
                           workspace_bytes, false, nullptr, ST(stream));
}

size_t u2tok_preprocess_workspace_bytes(int32_t D, int32_t H, int32_t W) { return preprocess_workspace_bytes(D, H, W); }
int u2tok_preprocess_volume(const float* vol, void* out, int32_t* info, int32_t D, int32_t H, int32_t W, int32_t target,
                            int32_t depth_pad, float lower_pct, float upper_pct, int32_t out_dtype, void* workspace,
                            size_t workspace_bytes, u2tok_stream_t stream) {
  return preprocess_volume(vol, out, info, D, H, W, target, depth_pad, lower_pct, upper_pct, out_dtype, nullptr, workspace,
                           workspace_bytes, ST(stream));
}
int u2tok_preprocess_volume_aug(const float* vol, void* out, int32_t* info, int32_t D, int32_t H, int32_t W, int32_t target,
                                int32_t depth_pad, float lower_pct, float upper_pct, int32_t out_dtype,
                                const u2tok_augment* aug, void* workspace, size_t workspace_bytes, u2tok_stream_t stream) {
  if (!aug) return U2_ERR_ARG;
  PreAugment a;
  a.rot_k = aug->rot90_k;
  for (int i = 0; i < 3; ++i) a.flip[i] = aug->flip[i];
  a.mul = 1.0f + aug->scale_factor;
  a.add = aug->shift_offset;
  return preprocess_volume(vol, out, info, D, H, W, target, depth_pad, lower_pct, upper_pct, out_dtype, &a, workspace,
                           workspace_bytes, ST(stream));
}

int u2tok_embed_splice(const void* table, const int64_t* ids, const void* feats, void* out, int32_t B, int32_t S,
                       int32_t E, int32_t nfeat, int64_t vocab, u2tok_stream_t stream) {
  return embed_splice(BF(table), ids, BF(feats), BFW(out), B, S, E, nfeat, vocab, ST(stream));
}

int u2tok_gemm_bf16(const void* A, const void* B, void* C, const void* bias, const void* R, int32_t M, int32_t N,
                    int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int32_t nz, int32_t nbh,
                    int64_t sAb, int64_t sAh, int64_t sBb, int64_t sBh, int64_t sCb, int64_t sCh, int64_t sRb,
                    int64_t sRh, float alpha, int32_t flags, u2tok_stream_t stream) {
  GemmDesc g;
  g.A = BF(A); g.B = BF(B); g.C = C; g.bias = BF(bias); g.R = BF(R);
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
  g.nz = nz; g.nbh = nbh;
  g.sAb = sAb; g.sAh = sAh; g.sBb = sBb; g.sBh = sBh; g.sCb = sCb; g.sCh = sCh; g.sRb = sRb; g.sRh = sRh;
  g.alpha = alpha;
  g.flags = flags & (GEMM_BIAS_N | GEMM_BIAS_M | GEMM_GELU | GEMM_RESIDUAL | GEMM_OUT_F32 | GEMM_A_KMAJOR | GEMM_B_KMAJOR |
                     GEMM_SWIGLU);
  if (flags & GEMM_B_KTILE) {
    if ((K & 63) || nz != 1) return U2_ERR_ARG;
    g.ldb = 64;
    g.ldbk = (int64_t)N * 64;
  }
  return gemm_bf16(g, ST(stream));
}

int u2tok_layernorm_bf16(const void* x, const void* res, const void* w, const void* b, void* y, int32_t rows,
                         int32_t C, float eps, u2tok_stream_t stream) {
  return layernorm_bf16(BF(x), BF(res), BF(w), BF(b), BFW(y), 1, rows, C, 0, C, 0, C, 0, C, eps, ST(stream));
}

int u2tok_softmax_rows(const float* S, void* P, int32_t nz, int32_t rows, int32_t n, int64_t lds, int64_t ldp,
                       float scale, const void* rel_bias, int32_t H, int32_t max_len, u2tok_stream_t stream) {
  return softmax_rows(S, BFW(P), nz, rows, n, lds, ldp, (int64_t)rows * lds, (int64_t)rows * ldp, scale, BF(rel_bias), H,
                      max_len, ST(stream));
}

int u2tok_transpose_bf16(const void* in, void* out, int32_t nz, int32_t R, int32_t C, int64_t ld_in, int64_t ld_out,
                         int64_t in_zs, int64_t out_zs, int32_t perm16, u2tok_stream_t stream) {
  return transpose_bf16(BF(in), BFW(out), nz, R, C, ld_in, ld_out, in_zs, out_zs, perm16, ST(stream));
}

int u2tok_im2col_patches(const void* vol, int32_t vol_dtype, void* out, int32_t nchunk, int32_t D, int32_t H,
                         int32_t W, int32_t p1, int32_t p2, int32_t p3, u2tok_stream_t stream) {
  return im2col_patches(vol, vol_dtype, BFW(out), nchunk, D, H, W, p1, p2, p3, ST(stream));
}

int u2tok_avgpool3d_tokens(const void* x, void* y, int32_t nb, int32_t g1, int32_t g2, int32_t g3, int32_t w1,
                           int32_t w2, int32_t w3, int32_t C, u2tok_stream_t stream) {
  return avgpool3d_tokens(BF(x), BFW(y), nb, g1, g2, g3, w1, w2, w3, C, ST(stream));
}

int u2tok_score_gemv(const void* x, const void* w, const void* bias, float* scores, int32_t rows, int32_t E,
                     u2tok_stream_t stream) {
  return score_gemv(BF(x), BF(w), BF(bias), scores, rows, E, ST(stream));
}

int u2tok_topk_sorted(const float* scores, int64_t* idx, int32_t B, int32_t n, int32_t k, u2tok_stream_t stream) {
  return topk_sorted(scores, idx, B, n, k, ST(stream));
}

int u2tok_gather_rows(const void* x, const int64_t* idx, void* out, int32_t B, int32_t n, int32_t k, int32_t E,
                      u2tok_stream_t stream) {
  return gather_rows(BF(x), idx, BFW(out), B, n, k, E, ST(stream));
}

int u2tok_multiscale_pool(const void* x, void* out, int32_t B, int32_t k, int32_t E, const void* gate_w,
                          const void* gate_b, float* ws, u2tok_stream_t stream) {
  return multiscale_pool(BF(x), BFW(out), B, k, E, BF(gate_w), BF(gate_b), ws, ST(stream));
}

int u2tok_temporal_attention(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t T,
                             int32_t N, int32_t H, int32_t d, int64_t ld_qkv, int64_t ld_out, float scale,
                             const void* rel_bias, int32_t max_len, u2tok_stream_t stream) {
  return temporal_attention(BF(q), BF(k), BF(v), BFW(out), B, T, N, H, d, ld_qkv, ld_out, scale, BF(rel_bias), max_len,
                            ST(stream));
}

int u2tok_flash_attention_d64(const void* q, const void* k, const void* vt, void* out, int32_t nb, int32_t S,
                              int32_t H, int64_t ld_qk, int64_t q_bs, int64_t ld_out, int64_t out_bs, int32_t S_pad,
                              float scale, const void* qx, const void* kx, const void* vx, void* outx, int64_t x_bs,
                              int64_t ox_bs, int32_t n_extra, u2tok_stream_t stream) {
  return flash_attention_d64(BF(q), BF(k), BF(vt), BFW(out), nb, S, H, ld_qk, q_bs, ld_out, out_bs, S_pad, scale,
                             BF(qx), BF(kx), BF(vx), BFW(outx), x_bs, ox_bs, n_extra, nullptr, 0, ST(stream));
}
int u2tok_flash_attention_d64_lse(const void* q, const void* k, const void* vt, void* out, int32_t nb, int32_t S,
                                  int32_t H, int64_t ld_qk, int64_t q_bs, int64_t ld_out, int64_t out_bs, int32_t S_pad,
                                  float scale, const void* qx, const void* kx, const void* vx, void* outx, int64_t x_bs,
                                  int64_t ox_bs, int32_t n_extra, float* lse, int64_t lse_ld, u2tok_stream_t stream) {
  if (!lse) return U2_ERR_ARG;
  return flash_attention_d64(BF(q), BF(k), BF(vt), BFW(out), nb, S, H, ld_qk, q_bs, ld_out, out_bs, S_pad, scale,
                             BF(qx), BF(kx), BF(vx), BFW(outx), x_bs, ox_bs, n_extra, lse, lse_ld, ST(stream));
}

size_t u2tok_tok_attention_workspace_bytes(int32_t nb, int32_t H, int32_t Sq, int32_t Skv, int32_t d) {
  return tok_attention_workspace_bytes(nb, H, Sq, Skv, d);
}
int u2tok_tok_attention(const void* q, const void* k, const void* v, void* out, int32_t nb, int32_t Sq, int32_t Skv, int32_t H,
                        int32_t d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                        int64_t o_bs, float scale, const void* rel_bias, int32_t max_len, int32_t splits, void* workspace,
                        size_t workspace_bytes, u2tok_stream_t stream) {
  return tok_attention(BF(q), BF(k), BF(v), BFW(out), nb, Sq, Skv, H, d, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, scale,
                       BF(rel_bias), max_len, splits, workspace, workspace_bytes, ST(stream));
}

int u2tok_attention_gqa(const void* q, const void* k, const void* v, void* out, int32_t nb, int32_t Sq, int32_t Skv, int32_t Hq,
                        int32_t Hkv, int32_t d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs,
                        int64_t v_bs, int64_t o_bs, float scale, int32_t causal, u2tok_stream_t stream) {
  return attention_ex(BF(q), BF(k), BF(v), BFW(out), nb, Sq, Skv, Hq, Hkv, d, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, scale,
                      nullptr, 0, causal, 1, nullptr, 0, ST(stream));
}
int u2tok_attention_gqa_split(const void* q, const void* k, const void* v, void* out, int32_t nb, int32_t Sq, int32_t Skv,
                              int32_t Hq, int32_t Hkv, int32_t d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs,
                              int64_t k_bs, int64_t v_bs, int64_t o_bs, float scale, void* workspace, size_t workspace_bytes,
                              u2tok_stream_t stream) {
  return attention_ex(BF(q), BF(k), BF(v), BFW(out), nb, Sq, Skv, Hq, Hkv, d, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, scale,
                      nullptr, 0, 0, 0, workspace, workspace_bytes, ST(stream));
}
int u2tok_rmsnorm_bf16(const void* x, const void* w, void* y, int64_t rows, int32_t C, int64_t ldx, int64_t ldy, float eps,
                       u2tok_stream_t stream) {
  return rmsnorm_bf16(BF(x), BF(w), BFW(y), rows, C, ldx, ldy, eps, ST(stream));
}
int u2tok_qk_norm_rope(void* qkv, const void* wq, const void* wk, const void* cos, const void* sin, int32_t cos_sin_f32,
                       int64_t rows, int32_t Hq, int32_t Hkv, int32_t D, int64_t ld, int64_t cs_ld, float eps,
                       u2tok_stream_t stream) {
  return qk_norm_rope(BFW(qkv), BF(wq), BF(wk), cos, sin, cos_sin_f32, rows, Hq, Hkv, D, ld, cs_ld, eps, nullptr, nullptr, 0, 0, 0,
                      ST(stream));
}
int u2tok_qk_norm_rope_kv(void* qkv, const void* wq, const void* wk, const void* cos, const void* sin, int32_t cos_sin_f32,
                          int64_t rows, int32_t Hq, int32_t Hkv, int32_t D, int64_t ld, int64_t cs_ld, float eps,
                          void* k_cache, void* v_cache, int32_t S, int64_t kv_stride, int32_t s_off, u2tok_stream_t stream) {
  if (!k_cache || !v_cache) return U2_ERR_ARG;
  return qk_norm_rope(BFW(qkv), BF(wq), BF(wk), cos, sin, cos_sin_f32, rows, Hq, Hkv, D, ld, cs_ld, eps, BFW(k_cache),
                      BFW(v_cache), S, kv_stride, s_off, ST(stream));
}
static DecodeCfg dec_cfg(const u2tok_decode_config* c) {
  DecodeCfg d;
  d.B = c->B; d.E = c->E; d.Hq = c->Hq; d.Hkv = c->Hkv; d.D = c->D; d.I = c->I;
  d.eps = c->eps; d.qk_eps = c->qk_eps; d.scale = c->scale;
  return d;
}
static bool dec_ok(const u2tok_decode_config* c) {
  return c && c->B > 0 && c->B <= 16 && c->E > 0 && !(c->E & 31) && c->Hq > 0 && c->Hkv > 0 && c->Hq % c->Hkv == 0 &&
         (c->D == 64 || c->D == 128) && c->I > 0 && !(c->I & 31);
}
size_t u2tok_decoder_decode_workspace_bytes(const u2tok_decode_config* c, int32_t T) {
  return dec_ok(c) && T > 0 ? decoder_decode_workspace_bytes(dec_cfg(c), T) : 0;
}
int u2tok_decoder_decode_pre(const u2tok_decode_config* c, const void* x, const void* w_in_norm, const void* Wqkv, const void* bqkv,
                             const void* wq_norm, const void* wk_norm, const void* cos, const void* sin, int32_t cos_sin_f32,
                             int64_t cs_ld, void* qkv, void* k_cache, void* v_cache, int64_t kv_stride, int32_t s_off,
                             void* workspace, size_t workspace_bytes, u2tok_stream_t stream) {
  if (!dec_ok(c)) return U2_ERR_ARG;
  return decoder_decode_pre(dec_cfg(c), BF(x), BF(w_in_norm), BF(Wqkv), BF(bqkv), BF(wq_norm), BF(wk_norm), cos, sin, cos_sin_f32,
                            cs_ld, BFW(qkv), BFW(k_cache), BFW(v_cache), kv_stride, s_off, workspace, workspace_bytes, ST(stream));
}
int u2tok_decoder_decode_post(const u2tok_decode_config* c, const void* x, const void* qkv, const void* K, const void* V, int32_t T,
                              int64_t kv_stride, const void* Wo, const void* bo, const void* w_post_norm, const void* Wgu, const void* bgu,
                              const void* Wdown, const void* bdown, void* out, void* workspace, size_t workspace_bytes,
                              u2tok_stream_t stream) {
  if (!dec_ok(c)) return U2_ERR_ARG;
  return decoder_decode_post(dec_cfg(c), BF(x), BF(qkv), BF(K), BF(V), T, kv_stride, BF(Wo), BF(bo), BF(w_post_norm), BF(Wgu), BF(bgu),
                             BF(Wdown), BF(bdown), BFW(out), workspace, workspace_bytes, ST(stream));
}
int u2tok_swiglu_bf16(const void* gate_up, void* out, int64_t rows, int32_t I, int64_t ld_in, int64_t ld_out,
                      u2tok_stream_t stream) {
  return swiglu_bf16(BF(gate_up), BFW(out), rows, I, ld_in, ld_out, ST(stream));
}

int u2tok_rope_apply(void* x, int64_t n_outer, int32_t S, int32_t n_inner, int32_t H, int32_t d, int64_t ld,
                     int32_t max_len, int32_t inverse, u2tok_stream_t stream) {
  return rope_apply(BFW(x), n_outer, S, n_inner, H, d, ld, max_len, inverse, ST(stream));
}

// ---- backward-pass building blocks (backward.hip)
int u2tok_gelu_fwd(const void* z, void* y, int64_t n, u2tok_stream_t stream) { return gelu_fwd(BF(z), BFW(y), n, ST(stream)); }
int u2tok_gelu_bwd(const void* z, const void* dy, void* dz, int64_t n, u2tok_stream_t stream) {
  return gelu_bwd(BF(z), BF(dy), BFW(dz), n, ST(stream));
}
size_t u2tok_colsum_workspace_bytes(int32_t rows, int32_t C) { return colsum_workspace_bytes(rows, C); }
int u2tok_colsum_bf16(const void* x, const void* y, float* out, void* out_bf16, int32_t rows, int32_t C, int64_t ldx,
                      int64_t ldy, void* workspace, int32_t accumulate, u2tok_stream_t stream) {
  return colsum_bf16(BF(x), BF(y), out, BFW(out_bf16), rows, C, ldx, ldy, reinterpret_cast<float*>(workspace), accumulate,
                     ST(stream));
}
size_t u2tok_layernorm_bwd_workspace_bytes(int32_t rows, int32_t C) { return layernorm_bwd_workspace_bytes(rows, C); }
int u2tok_layernorm_bwd(const void* x, const void* res, const void* w, const void* dy, void* dv, float* dw, float* db,
                        int32_t rows, int32_t C, float eps, void* workspace, int32_t accumulate, u2tok_stream_t stream) {
  return layernorm_bwd(BF(x), BF(res), BF(w), BF(dy), BFW(dv), dw, db, rows, C, eps, reinterpret_cast<float*>(workspace),
                       accumulate, ST(stream));
}
int u2tok_softmax_bwd(const void* P, const float* dP, void* dS, int64_t nrows, int32_t n, int64_t ldp, int64_t lddp,
                      u2tok_stream_t stream) {
  return softmax_bwd(BF(P), dP, BFW(dS), nrows, n, ldp, lddp, ST(stream));
}
int u2tok_relbias_grad(const void* dS, float* dtable, int32_t nz, int32_t S, int32_t H, int64_t ldp, int32_t max_len,
                       u2tok_stream_t stream) {
  return relbias_grad(BF(dS), dtable, nz, S, H, ldp, max_len, ST(stream));
}
int u2tok_rowdot_bf16(const void* a, const void* b, float* out, int64_t rows, int32_t C, int64_t lda, int64_t ldb,
                      u2tok_stream_t stream) {
  return rowdot_bf16(BF(a), BF(b), out, rows, C, lda, ldb, ST(stream));
}
int u2tok_adamw_step(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, const uint8_t* group, void* out_bf16,
                     int64_t n, const float* lr, const float* weight_decay, int32_t ngroups, float beta1, float beta2, float eps,
                     int32_t step, float grad_scale, const float* grad_coef, u2tok_stream_t stream) {
  if (!lr || !weight_decay || ngroups <= 0 || ngroups > 8 || step <= 0 || (ngroups > 1 && !group)) return U2_ERR_ARG;
  AdamWArgs a;
  for (int i = 0; i < 8; ++i) { a.lr[i] = lr[i < ngroups ? i : 0]; a.wd[i] = weight_decay[i < ngroups ? i : 0]; }
  a.b1 = beta1; a.b2 = beta2; a.eps = eps;
  a.inv_c1 = (float)(1.0 / (1.0 - pow((double)beta1, step)));
  a.inv_sqrt_c2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, step)));
  a.gscale = grad_scale; a.gcoef = grad_coef;
  return adamw_step(master, exp_avg, exp_avg_sq, BF(grad), group, BFW(out_bf16), n, a, ST(stream));
}
size_t u2tok_flash_attention_d64_bwd_workspace_bytes(int32_t nb, int32_t S, int32_t H) {
  return flash_attention_d64_bwd_workspace_bytes(nb, S, H);
}
int32_t u2tok_flash_attention_d64_bwd(const void* q, const void* k, const void* v, int64_t ld_qkv, int64_t bs_qkv,
                                      const void* out, const void* d_out, int64_t ld_o, int64_t bs_o, void* dq, void* dk,
                                      void* dv, int64_t ld_d, int64_t bs_d, int32_t nb, int32_t S, int32_t H, float scale,
                                      const float* lse, int64_t lse_ld, void* workspace, size_t workspace_bytes,
                                      u2tok_stream_t stream) {
  return flash_attention_d64_bwd(BF(q), BF(k), BF(v), ld_qkv, bs_qkv, BF(out), BF(d_out), ld_o, bs_o, BFW(dq), BFW(dk), BFW(dv),
                                 ld_d, bs_d, nb, S, H, scale, lse, lse_ld, workspace, workspace_bytes, ST(stream));
}

}  // extern "C"
