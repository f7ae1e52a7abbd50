// Launch sequences of the three module forwards of the hot path (reference: u2_arch.py:91-117):
//   ViT3DTower (vit.py:114-126,148-164) -> SpatialPoolingProjector (spatial_pooling_projector.py:34-52)
//   -> u2Tokenizer (u2Tokenizer.py:40-47 = svr.py:166-188 + tta.py:126-140).
// Everything is enqueued from C++ on one HIP stream (no per-op Python/ctypes round trip); intermediate
// tensors live in a caller-provided workspace carved by a bump arena.  Each forward is written once and
// can run "dry" (no launches) to size that workspace.
#include "pipeline.h"

namespace u2 {

namespace {

struct Arena {
  char* base;
  size_t cap;
  size_t off = 0, peak = 0;
  bool dry;
  Arena(void* b, size_t c, bool d) : base((char*)b), cap(c), dry(d) {}
  template <typename T>
  T* get(size_t count) {
    const size_t a = (off + 255) & ~(size_t)255;
    off = a + count * sizeof(T);
    if (off > peak) peak = off;
    if (dry) return reinterpret_cast<T*>((uintptr_t)0x1000 + a);  // never dereferenced
    return (off <= cap) ? reinterpret_cast<T*>(base + a) : nullptr;
  }
  bool ok() const { return dry || peak <= cap; }
};

// The split-K scratch registration of `st` for the duration of one module forward, the previous one restored on exit:
// what a forward computes must not depend on a registration some other caller left behind (a direct u2tok_gemm_bf16
// user, the training path) -- the split changes the summation order.  The ViT and the projector run without one.
struct ScratchScope {
  Context& cx;
  hipStream_t st;
  Scratch prev;
  bool on;
  ScratchScope(Context& c, hipStream_t s, void* p, size_t bytes, bool enable) : cx(c), st(s), prev{s, nullptr, 0}, on(enable) {
    if (!on) return;
    prev = cx.scratch_of(st);
    cx.set_scratch(st, p, bytes);
  }
  ~ScratchScope() { if (on) cx.set_scratch(st, prev.p, prev.bytes); }
};

#define U2_RUN(expr)                  \
  do {                                \
    if (!dry) {                       \
      const int e_ = (expr);          \
      if (e_ != U2_OK) return e_;     \
    }                                 \
  } while (0)
#define U2_CHECK_WS(ar) \
  do {                  \
    if (!(ar).ok()) return U2_ERR_WORKSPACE; \
  } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// y[rows][ldy] = x[rows][ldx] @ W[out][in]^T + b  (+GELU) (+R)
static GemmDesc linear_desc(const bf16_t* x, int64_t ldx, const bf16_t* w, const bf16_t* b, bf16_t* y, int64_t ldy, int64_t rows,
                            int in, int out, int extra_flags, const bf16_t* R, int64_t ldr, int nsplit, float alpha_lo) {
  GemmDesc g;
  g.A = x; g.B = w; g.C = y; g.bias = b; g.R = R;
  g.M = (int)rows; g.N = out; g.K = in;
  g.lda = ldx; g.ldb = in; g.ldc = ldy; g.ldr = ldr;
  g.flags = (b ? GEMM_BIAS_N : 0) | extra_flags | (R ? GEMM_RESIDUAL : 0);
  g.nsplit = nsplit; g.alpha_lo = alpha_lo;  // columns [0, nsplit) scaled by alpha_lo (before the bias)
  return g;
}

int linear(const bf16_t* x, int64_t ldx, const bf16_t* w, const bf16_t* b, bf16_t* y, int64_t ldy, int64_t rows,
           int in, int out, int extra_flags, const bf16_t* R, int64_t ldr, hipStream_t st, int nsplit = 0,
           float alpha_lo = 1.f) {
  return gemm_bf16(linear_desc(x, ldx, w, b, y, ldy, rows, in, out, extra_flags, R, ldr, nsplit, alpha_lo), st);
}

struct AttnCore {
  const bf16_t *q, *k, *v;
  int64_t ldq, ldk, ldv, q_bs, k_bs, v_bs;
  bf16_t* out;
  int64_t ldo, o_bs;
  int nb, Sq, Skv, H, d;
  float scale;
  const bf16_t* rel_bias;
  int max_len;
  bool unfused = false;  // force the GEMM chain (the ViT's debug path)
};

// softmax(Q K^T * scale + bias) V.  Reference: rma.py:60-75 / tta.py:55-61 / rope.py:82-86 (which also materialise the
// probabilities).  Default: the fused kernel of tokattn.hip (scores and probabilities never leave the registers; few
// (batch, head, 64-query) units -> key splits whose fp32 partial sums live in the arena).  Option "tok_flash" = 0, or a shape
// that kernel does not take: two batched MFMA GEMMs around fp32 scores and bf16 probabilities in HBM.
int attention_core(Arena& ar, const AttnCore& a, bool dry, hipStream_t st) {
  if (!a.unfused && (dry || opts().tok_flash)) {
    const size_t mark = ar.off;
    const size_t wsb = tok_attention_workspace_bytes(a.nb, a.H, a.Sq, a.Skv, a.d);
    char* tws = wsb ? ar.get<char>(wsb) : nullptr;
    U2_CHECK_WS(ar);
    // (a dry run sizes the arena for BOTH forms: which one a real call takes depends on pointers it does not have)
    if (!dry && tok_attention_supported(a.q, a.k, a.v, a.out, a.Sq, a.Skv, a.d, a.ldq, a.ldk, a.ldv, a.ldo, a.q_bs, a.k_bs,
                                        a.v_bs, a.o_bs, a.rel_bias, a.max_len)) {
      const int e = tok_attention(a.q, a.k, a.v, a.out, a.nb, a.Sq, a.Skv, a.H, a.d, a.ldq, a.ldk, a.ldv, a.ldo, a.q_bs,
                                  a.k_bs, a.v_bs, a.o_bs, a.scale, a.rel_bias, a.max_len, 0, tws, wsb, st);
      ar.off = mark;
      return e;
    }
    ar.off = mark;
  }
  const size_t mark = ar.off;
  const int64_t ldS = round_up(a.Skv, 8), ldp = ldS;
  const int64_t nz = (int64_t)a.nb * a.H;
  if (nz > 65535) return U2_ERR_ARG;
  float* S = ar.get<float>((size_t)nz * a.Sq * ldS);
  bf16_t* P = ar.get<bf16_t>((size_t)nz * a.Sq * ldp);
  // P V: V is read in place as a K-major B operand (rows = keys; GEMM_B_KMAJOR, LDS transpose reads) when the key count
  // keeps the 16-byte chunks of P whole; otherwise through a transposed copy
  const bool v_in_place = opts().kmajor_b && !(a.Skv & 7) && !(a.d & 7) && !(a.ldv & 7) && !(a.v_bs & 7) && !((uintptr_t)a.v & 15);
  bf16_t* Vt = v_in_place ? nullptr : ar.get<bf16_t>((size_t)a.nb * a.H * a.d * ldp);
  U2_CHECK_WS(ar);
  {
    GemmDesc g;
    g.A = a.q; g.B = a.k; g.C = S;
    g.M = a.Sq; g.N = a.Skv; g.K = a.d;
    g.lda = a.ldq; g.ldb = a.ldk; g.ldc = ldS;
    g.nz = (int)nz; g.nbh = a.H;
    g.sAb = a.q_bs; g.sAh = a.d; g.sBb = a.k_bs; g.sBh = a.d;
    g.sCb = (int64_t)a.H * a.Sq * ldS; g.sCh = (int64_t)a.Sq * ldS;
    g.flags = GEMM_OUT_F32;
    U2_RUN(gemm_bf16(g, st));
  }
  U2_RUN(softmax_rows(S, P, (int)nz, a.Sq, a.Skv, ldS, ldp, (int64_t)a.Sq * ldS, (int64_t)a.Sq * ldp, a.scale,
                      a.rel_bias, a.H, a.max_len, st));
  {
    GemmDesc g;
    g.A = P; g.C = a.out;
    g.M = a.Sq; g.N = a.d;
    g.lda = ldp; g.ldc = a.ldo;
    g.nz = (int)nz; g.nbh = a.H;
    g.sAb = (int64_t)a.H * a.Sq * ldp; g.sAh = (int64_t)a.Sq * ldp;
    g.sCb = a.o_bs; g.sCh = a.d;
    if (v_in_place) {
      g.B = a.v; g.K = a.Skv; g.ldb = a.ldv;
      g.sBb = a.v_bs; g.sBh = a.d;
      g.flags = GEMM_B_KMAJOR;
    } else {
      U2_RUN(transpose_bf16(a.v, Vt, a.nb, a.Skv, a.H * a.d, a.ldv, ldp, a.v_bs, (int64_t)a.H * a.d * ldp, 0, st));
      g.B = Vt; g.K = (int)ldp; g.ldb = ldp;
      g.sBb = (int64_t)a.H * a.d * ldp; g.sBh = (int64_t)a.d * ldp;
    }
    U2_RUN(gemm_bf16(g, st));
  }
  ar.off = mark;
  return U2_OK;
}

// attention over the CHUNK axis of rows laid out (b, t, n) [leading dim 3E for q | k | v, E for the output]: Bx * Nx * H
// independent sequences of Tx rows that are Nx rows apart (svr.py:31-36 without its two permute().contiguous() copies).
// Tx <= 16 and a head dim the wave kernel is built for: one launch; otherwise the batched-GEMM core, one batch entry
// at a time (sequence rows are Nx * 3E elements apart, the Nx positions and H heads are the GEMM batch).
int chunk_axis_attention(Arena& ar, const bf16_t* qkv, bf16_t* out, int Bx, int Tx, int Nx, int H, int d, float scale,
                         const bf16_t* rel_bias, int max_len, bool dry, hipStream_t st) {
  const int E = H * d;
  if (Tx <= 16 && (d == 64 || d == 128 || d == 256 || d == 512)) {
    U2_RUN(temporal_attention(qkv, qkv + E, qkv + 2 * E, out, Bx, Tx, Nx, H, d, 3 * E, E, scale, rel_bias, max_len, st));
    return U2_OK;
  }
  for (int b = 0; b < Bx; ++b) {
    const bf16_t* base = qkv + (int64_t)b * Tx * Nx * 3 * E;
    AttnCore a{base, base + E, base + 2 * E, (int64_t)Nx * 3 * E, (int64_t)Nx * 3 * E, (int64_t)Nx * 3 * E, 3 * E, 3 * E,
               3 * E, out + (int64_t)b * Tx * Nx * E, (int64_t)Nx * E, E, Nx, Tx, Tx, H, d, scale, rel_bias, max_len};
    const int e = attention_core(ar, a, dry, st);
    if (e != U2_OK) return e;
  }
  return U2_OK;
}

}  // namespace

// =========================================================================== ViT3DTower
// Row layout of the residual stream: the nc * ntok PATCH rows first (chunk-major), then the nc cls rows.  The
// reference keeps [cls | patches] per chunk (vit.py:116-118); a transformer is indifferent to the order rows are
// stored in, and this one makes every GEMM see M = nc*ntok (+ nc), the patch rows of a chunk tile exactly into
// 128 / 256-row attention units, and the final "drop the cls token" (vit.py:157-160) a no-op.
int vit_forward(const VitConfig& c, const void* const* W, const void* volume, bf16_t* out, void* ws, size_t ws_bytes,
                bool dry, size_t* peak, hipStream_t st) {
  if (c.nchunk <= 0 || c.depth < 0 || c.heads <= 0 || c.hidden != c.heads * 64) return U2_ERR_ARG;
  for (int i = 0; i < 3; ++i)
    if (c.img[i] <= 0 || c.patch[i] <= 0 || c.img[i] % c.patch[i]) return U2_ERR_ARG;
  if (!dry && (!W || !volume || !out)) return U2_ERR_ARG;
  const int nh = c.img[0] / c.patch[0], nw = c.img[1] / c.patch[1], nd = c.img[2] / c.patch[2];
  const int ntok = nh * nw * nd, S = ntok + 1, Hd = c.hidden, Kp = c.patch[0] * c.patch[1] * c.patch[2];
  const int nc = c.nchunk;
  const int64_t prow = (int64_t)nc * ntok;  // patch rows; cls row of chunk b is row prow + b
  const int64_t rows = prow + nc;
  const int S_pad = (int)round_up(ntok, 64);
  auto w = [&](int i) { return dry ? nullptr : reinterpret_cast<const bf16_t*>(W[i]); };

  ScratchScope scratch_scope(ctx(), st, nullptr, 0, !dry);
  Arena ar(ws, ws_bytes, dry);
  bf16_t* x = ar.get<bf16_t>((size_t)rows * Hd);
  bf16_t* xn = ar.get<bf16_t>((size_t)rows * Hd);
  bf16_t* qkv = ar.get<bf16_t>((size_t)rows * 3 * Hd);
  bf16_t* att = ar.get<bf16_t>((size_t)rows * Hd);
  bf16_t* vt = ar.get<bf16_t>((size_t)nc * Hd * S_pad);
  // patches (only needed for the embedding) and the MLP hidden share one region
  const size_t big = std::max((size_t)nc * ntok * Kp, (size_t)rows * c.mlp_dim);
  bf16_t* h1 = ar.get<bf16_t>(big);
  bf16_t* patches = h1;
  U2_CHECK_WS(ar);

  // ---- patch embedding: im2col + Linear + position embedding; cls token rows appended (vit.py:115-118)
  U2_RUN(im2col_patches(volume, c.vol_dtype, patches, nc, c.img[0], c.img[1], c.img[2], c.patch[0], c.patch[1],
                        c.patch[2], st));
  U2_RUN(fill_rows(w(3), x + prow * Hd, nc, Hd, Hd, st));
  {
    GemmDesc g;
    g.A = patches; g.B = w(1); g.C = x; g.bias = w(2); g.R = w(0);
    g.M = ntok; g.N = Hd; g.K = Kp;
    g.lda = Kp; g.ldb = Kp; g.ldc = Hd; g.ldr = Hd;
    g.nz = nc; g.nbh = 1;
    g.sAb = (int64_t)ntok * Kp; g.sCb = (int64_t)ntok * Hd;
    g.flags = GEMM_BIAS_N | GEMM_RESIDUAL;
    U2_RUN(gemm_bf16(g, st));
  }
  const float scale = 1.0f / sqrtf(64.0f);
  for (int l = 0; l < c.depth; ++l) {
    const int b0 = 4 + 11 * l;
    // x = x + out_proj(attn(qkv(norm1(x))))   (MONAI TransformerBlock / SABlock)
    U2_RUN(layernorm_bf16(x, nullptr, w(b0 + 0), w(b0 + 1), xn, 1, (int)rows, Hd, 0, Hd, 0, 0, 0, Hd, c.ln_eps, st));
    // the double pipeline reads q already multiplied by softmax scale * log2 e: the product scales its q columns from the
    // fp32 accumulator (one rounding, as for the unscaled q of vit.py:100-105; the SABlock qkv projection has no bias)
    const bool q_pre = opts().vit_flash && ntok >= 512 && (opts().flash_mode == 0 || opts().flash_mode == 7);
    // ... and, where the 256 x 192 deep form runs the product (round 5), the V tiles leave V^T for the flash kernel themselves:
    // their K loop runs with the MFMA operands exchanged and the transposed accumulators are stored in that operand's key order
    // (gemm_bt.hip: vt_epilogue) -- no row-major V of the patch rows, no transpose launch (12 per volume).  Same values, bit for bit.
    GemmDesc gq = linear_desc(xn, Hd, w(b0 + 2), nullptr, qkv, 3 * Hd, rows, Hd, 3 * Hd, 0, nullptr, 0, q_pre ? Hd : 0,
                              scale * 1.44269504088896340736f);
    const bool vt_fused = opts().vit_flash && opts().vit_vt_epilogue && S_pad == ntok && gemm_vt_supported(gq, 2 * Hd, ntok);
    if (vt_fused) {
      gq.vt = vt; gq.vt_n0 = 2 * Hd; gq.vt_rows = ntok; gq.vt_ld = S_pad; gq.vt_bs = (int64_t)Hd * S_pad;
    }
    U2_RUN(gemm_bf16(gq, st));
    if (opts().vit_flash) {
      if (!vt_fused)
        U2_RUN(transpose_bf16(qkv + 2 * Hd, vt, nc, ntok, Hd, 3 * Hd, S_pad, (int64_t)ntok * 3 * Hd, (int64_t)Hd * S_pad, 1, st));
      const bf16_t* xq = qkv + prow * 3 * Hd;  // q | k | v of the cls rows
      U2_RUN(flash_attention_d64(qkv, qkv + Hd, vt, att, nc, ntok, c.heads, 3 * Hd, (int64_t)ntok * 3 * Hd, Hd,
                                 (int64_t)ntok * Hd, S_pad, scale, xq, xq + Hd, xq + 2 * Hd, att + prow * Hd, 3 * Hd, Hd,
                                 1, nullptr, 0, st, q_pre ? 1 : 0));
    } else {
      // unfused reference path (debug option "vit_flash" = 0): gather [cls | patches] per chunk, two GEMMs + softmax
      const size_t mark = ar.off;
      bf16_t* qc = ar.get<bf16_t>((size_t)rows * 3 * Hd);
      bf16_t* ac = ar.get<bf16_t>((size_t)rows * Hd);
      U2_CHECK_WS(ar);
      const size_t rb = (size_t)3 * Hd * sizeof(bf16_t), ob = (size_t)Hd * sizeof(bf16_t);
      if (!dry) {
        const hipError_t e1 = hipMemcpy2DAsync(qc + 3 * Hd, (size_t)S * rb, qkv, (size_t)ntok * rb, (size_t)ntok * rb, nc,
                                               hipMemcpyDeviceToDevice, st);
        const hipError_t e2 = hipMemcpy2DAsync(qc, (size_t)S * rb, qkv + prow * 3 * Hd, rb, rb, nc,
                                               hipMemcpyDeviceToDevice, st);
        if (e1 != hipSuccess || e2 != hipSuccess) return U2_ERR_LAUNCH;
      }
      {
        AttnCore a{qc, qc + Hd, qc + 2 * Hd, 3 * Hd, 3 * Hd, 3 * Hd, (int64_t)S * 3 * Hd, (int64_t)S * 3 * Hd,
                   (int64_t)S * 3 * Hd, ac, Hd, (int64_t)S * Hd, nc, S, S, c.heads, 64, scale, nullptr, 0, true};
        const int e = attention_core(ar, a, dry, st);
        if (e != U2_OK) return e;
      }
      if (!dry) {
        const hipError_t e1 = hipMemcpy2DAsync(att, (size_t)ntok * ob, ac + Hd, (size_t)S * ob, (size_t)ntok * ob, nc,
                                               hipMemcpyDeviceToDevice, st);
        const hipError_t e2 = hipMemcpy2DAsync(att + prow * Hd, ob, ac, (size_t)S * ob, ob, nc, hipMemcpyDeviceToDevice, st);
        if (e1 != hipSuccess || e2 != hipSuccess) return U2_ERR_LAUNCH;
      }
      ar.off = mark;
    }
    U2_RUN(linear(att, Hd, w(b0 + 3), w(b0 + 4), x, Hd, rows, Hd, Hd, 0, x, Hd, st));
    // x = x + linear2(gelu(linear1(norm2(x))))
    U2_RUN(layernorm_bf16(x, nullptr, w(b0 + 5), w(b0 + 6), xn, 1, (int)rows, Hd, 0, Hd, 0, 0, 0, Hd, c.ln_eps, st));
    U2_RUN(linear(xn, Hd, w(b0 + 7), w(b0 + 8), h1, c.mlp_dim, rows, Hd, c.mlp_dim, GEMM_GELU, nullptr, 0, st));
    U2_RUN(linear(h1, c.mlp_dim, w(b0 + 9), w(b0 + 10), x, Hd, rows, c.mlp_dim, Hd, 0, x, Hd, st));
  }
  // final norm; the cls rows are dropped unless select_feature == "cls_patch" (vit.py:124,157-160), in which case
  // the output goes back to the reference's [cls | patches] order per chunk
  const int nwt = 4 + 11 * c.depth;
  if (c.keep_cls) {
    U2_RUN(layernorm_bf16(x, nullptr, w(nwt), w(nwt + 1), out + Hd, nc, ntok, Hd, (int64_t)ntok * Hd, Hd, 0, 0,
                          (int64_t)S * Hd, Hd, c.ln_eps, st));
    U2_RUN(layernorm_bf16(x + prow * Hd, nullptr, w(nwt), w(nwt + 1), out, nc, 1, Hd, Hd, Hd, 0, 0, (int64_t)S * Hd, Hd,
                          c.ln_eps, st));
  } else {
    U2_RUN(layernorm_bf16(x, nullptr, w(nwt), w(nwt + 1), out, 1, (int)prow, Hd, 0, Hd, 0, 0, 0, Hd, c.ln_eps, st));
  }
  if (peak) *peak = ar.peak;
  U2_CHECK_WS(ar);
  return U2_OK;
}

// =========================================================================== SpatialPoolingProjector
int spp_forward(const SppConfig& c, const void* const* W, const bf16_t* x, bf16_t* out, void* ws, size_t ws_bytes,
                bool dry, size_t* peak, hipStream_t st) {
  if (c.nchunk <= 0 || c.pooling_size <= 0 || c.layer_num <= 0 || c.in_dim <= 0 || c.out_dim <= 0) return U2_ERR_ARG;
  if (!dry && (!W || !x || !out)) return U2_ERR_ARG;
  const int ps = c.pooling_size;
  int g1 = c.grid[0], g2 = c.grid[1], g3 = c.grid[2], w1 = ps, w2 = ps, w3 = ps;
  if (c.pooling_type == 1) { g3 = g1 * g2 * g3; g1 = g2 = 1; w1 = w2 = 1; w3 = ps * ps * ps; }
  else if (c.pooling_type != 0) return U2_ERR_ARG;
  if (g1 < w1 || g2 < w2 || g3 < w3) return U2_ERR_ARG;
  const int np = (g1 / w1) * (g2 / w2) * (g3 / w3);
  const int64_t rows = (int64_t)c.nchunk * np;
  auto w = [&](int i) { return dry ? nullptr : reinterpret_cast<const bf16_t*>(W[i]); };
  ScratchScope scratch_scope(ctx(), st, nullptr, 0, !dry);
  Arena ar(ws, ws_bytes, dry);
  bf16_t* pooled = ar.get<bf16_t>((size_t)rows * c.in_dim);
  bf16_t* ha = ar.get<bf16_t>((size_t)rows * c.out_dim);
  bf16_t* hb = ar.get<bf16_t>((size_t)rows * c.out_dim);
  U2_CHECK_WS(ar);
  U2_RUN(avgpool3d_tokens(x, pooled, c.nchunk, g1, g2, g3, w1, w2, w3, c.in_dim, st));
  const bf16_t* cur = pooled;
  int cur_dim = c.in_dim;
  for (int l = 0; l < c.layer_num; ++l) {
    const bool last = (l == c.layer_num - 1);
    bf16_t* dst = last ? out : ((l & 1) ? hb : ha);
    const int fl = (!last && c.layer_type == 0) ? GEMM_GELU : 0;
    U2_RUN(linear(cur, cur_dim, w(2 * l), w(2 * l + 1), dst, c.out_dim, rows, cur_dim, c.out_dim, fl, nullptr, 0, st));
    cur = dst;
    cur_dim = c.out_dim;
  }
  if (peak) *peak = ar.peak;
  return U2_OK;
}

// =========================================================================== u2Tokenizer
namespace {
struct Att {
  const bf16_t *wq, *bq, *wk, *bk, *wv, *bv, *wd, *bd, *rb;
};
// The host module packs wq | wk | wv (and their biases) back to back in HBM (tokenizer.py: pack_weights); when it
// did, the three projections of one input are ONE GEMM with N = 3E (2E for the k | v pair of a cross attention):
// the activation panel is read once and the launch fills the machine 3x better at small M.
inline bool kv_packed(const Att& a, int E) {
  return a.wk && a.bk && a.wv == a.wk + (size_t)E * E && a.bv == a.bk + E;
}
inline bool qkv_packed(const Att& a, int E) {
  return a.wq && a.bq && a.wk == a.wq + (size_t)E * E && a.bk == a.bq + E && kv_packed(a, E);
}
// q | k | v = x W^T + b into out[rows][3E]
int qkv_proj(const bf16_t* x, const Att& a, bf16_t* out, int64_t rows, int E, bool dry, hipStream_t st) {
  if (!dry && qkv_packed(a, E)) {
    U2_RUN(linear(x, E, a.wq, a.bq, out, 3 * E, rows, E, 3 * E, 0, nullptr, 0, st));
    return U2_OK;
  }
  U2_RUN(linear(x, E, a.wq, a.bq, out, 3 * E, rows, E, E, 0, nullptr, 0, st));
  U2_RUN(linear(x, E, a.wk, a.bk, out + E, 3 * E, rows, E, E, 0, nullptr, 0, st));
  U2_RUN(linear(x, E, a.wv, a.bv, out + 2 * E, 3 * E, rows, E, E, 0, nullptr, 0, st));
  return U2_OK;
}
}  // namespace

int tokenizer_forward(const TokConfig& c, const void* const* W, const bf16_t* v_token, const bf16_t* t_token,
                      bf16_t* out, int64_t* topk_idx_out, bf16_t* svr_out, const TokTaps* taps, void* ws, size_t ws_bytes,
                      bool dry, size_t* peak, hipStream_t st) {
  if (c.B <= 0 || c.T <= 0 || c.N <= 0 || c.E <= 0 || c.Lt <= 0 || c.num_heads <= 0 || c.num_layers < 0) return U2_ERR_ARG;
  if (c.E % c.num_heads || (c.E / c.num_heads) % 8 || c.top_k <= 0 || c.num_query <= 0) return U2_ERR_ARG;
  if (c.attn_type < 0 || c.attn_type > 2) return U2_ERR_ARG;
  // attn_type 2 = nn.MultiheadAttention read sequence-first (svr.py:16-18,28-35): attention runs across whatever sits
  // in dim 0, INCLUDING the batch entries: "spatial" = across the B*T (batch, chunk) pairs at a token position,
  // "temporal" = across the B*N (batch, token) pairs of a chunk index, TTA self-attention = across the B batch entries
  // of a query index (a sequence of one key for B = 1).
  if (!dry && (!W || !v_token || !t_token || !out)) return U2_ERR_ARG;
  const int B = c.B, T = c.T, N = c.N, E = c.E, H = c.num_heads, d = E / H, L = c.num_layers, Q = c.num_query;
  const int TN = T * N, k = c.top_k;
  if (!c.enable_diffts && k > TN) return U2_ERR_ARG;                       // torch.topk would raise
  if (N > c.max_seq_len || T > c.max_seq_len || Q > c.max_seq_len) return U2_ERR_ARG;  // rma.py:64-68 index range
  const float scale = 1.0f / sqrtf((float)d);
  auto wp = [&](int i) { return dry ? nullptr : reinterpret_cast<const bf16_t*>(W[i]); };
  auto att_at = [&](int i) {
    Att a{wp(i), wp(i + 1), wp(i + 2), wp(i + 3), wp(i + 4), wp(i + 5), wp(i + 6), wp(i + 7), wp(i + 8)};
    if (c.attn_type != 0) a.rb = nullptr;
    return a;
  };
  const int i_sel = 1 + 18 * L, i_gate = i_sel + 2, i_tta = i_gate + 2, i_lin = i_tta + 33 * L;

  Arena ar(ws, ws_bytes, dry);
  // split-K scratch for the skinny linear layers of this forward (M = 256 queries against E x E weights); registered
  // for this stream only while the launches below are being enqueued
  // (24 MB covers the E x E products; the TTA self-attention's packed q|k|v product takes the big-tile kernel in 4 K slices:
  //  4 x rows x 3E fp32)
  const size_t kSplitK = std::max<size_t>(24u << 20, (size_t)16 * B * c.num_query * 3 * E);
  char* skw = ar.get<char>(kSplitK);
  U2_CHECK_WS(ar);
  Context& cx = ctx();
  ScratchScope scratch_scope(cx, st, skw, kSplitK, !dry);
  const int64_t rows = (int64_t)B * TN;
  bf16_t* xa = ar.get<bf16_t>((size_t)rows * E);
  bf16_t* xb = ar.get<bf16_t>((size_t)rows * E);
  bf16_t* qkv = ar.get<bf16_t>((size_t)rows * 3 * E);
  bf16_t* ctx = ar.get<bf16_t>((size_t)rows * E);
  U2_CHECK_WS(ar);

  // ---------------- SVR: SpatioTemporalSignificanceScoring (svr.py:50-62) -- x = attn(x), no residual/norm
  const bf16_t* x = v_token;
  for (int l = 0; l < L; ++l) {
    const Att sp = att_at(1 + 18 * l), tp = att_at(1 + 18 * l + 9);
    if (taps && !dry && taps->svr_in && taps->svr_in[l]) x = reinterpret_cast<const bf16_t*>(taps->svr_in[l]);
    bf16_t* y = (x == xa) ? xb : xa;
    // spatial: sequences of N tokens inside each chunk (svr.py:27-30)
    { const int e = qkv_proj(x, sp, qkv, rows, E, dry, st); if (e != U2_OK) return e; }
    if (c.attn_type == 1) {
      U2_RUN(rope_apply(qkv, (int64_t)B * T, N, 1, H, d, 3 * E, c.max_seq_len, 0, st));
      U2_RUN(rope_apply(qkv + E, (int64_t)B * T, N, 1, H, d, 3 * E, c.max_seq_len, 0, st));
    }
    if (c.attn_type == 2) {  // sequence = the B*T (batch, chunk) pairs, batch = token position
      const int e = chunk_axis_attention(ar, qkv, ctx, 1, B * T, N, H, d, scale, nullptr, c.max_seq_len, dry, st);
      if (e != U2_OK) return e;
    } else {
      AttnCore a{qkv, qkv + E, qkv + 2 * E, 3 * E, 3 * E, 3 * E, (int64_t)N * 3 * E, (int64_t)N * 3 * E,
                 (int64_t)N * 3 * E, ctx, E, (int64_t)N * E, B * T, N, N, H, d, scale, sp.rb, c.max_seq_len};
      const int e = attention_core(ar, a, dry, st);
      if (e != U2_OK) return e;
    }
    U2_RUN(linear(ctx, E, sp.wd, sp.bd, y, E, rows, E, E, 0, nullptr, 0, st));
    // temporal: sequences of T chunks at each token position (svr.py:32-36); rows stay in (b t n) order
    bf16_t* y2 = (y == xa) ? xb : xa;
    { const int e = qkv_proj(y, tp, qkv, rows, E, dry, st); if (e != U2_OK) return e; }
    if (c.attn_type == 1) {
      U2_RUN(rope_apply(qkv, B, T, N, H, d, 3 * E, c.max_seq_len, 0, st));
      U2_RUN(rope_apply(qkv + E, B, T, N, H, d, 3 * E, c.max_seq_len, 0, st));
    }
    if (c.attn_type == 2 && B == 1) {  // sequence = the N tokens of a chunk, batch = chunk
      AttnCore a{qkv, qkv + E, qkv + 2 * E, 3 * E, 3 * E, 3 * E, (int64_t)N * 3 * E, (int64_t)N * 3 * E,
                 (int64_t)N * 3 * E, ctx, E, (int64_t)N * E, T, N, N, H, d, scale, nullptr, 0};
      const int e = attention_core(ar, a, dry, st);
      if (e != U2_OK) return e;
    } else if (c.attn_type == 2) {
      // sequence = the B*N (batch, token) pairs of one chunk index: gather rows (b, t, n) -> (t, b, n), attend, scatter
      const size_t mark = ar.off;
      bf16_t* g = ar.get<bf16_t>((size_t)rows * 3 * E);
      bf16_t* gc = ar.get<bf16_t>((size_t)rows * E);
      U2_CHECK_WS(ar);
      if (!dry)
        for (int b = 0; b < B; ++b)
          if (hipMemcpy2DAsync(g + (size_t)b * N * 3 * E, (size_t)B * N * 3 * E * 2, qkv + (size_t)b * T * N * 3 * E,
                               (size_t)N * 3 * E * 2, (size_t)N * 3 * E * 2, T, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return U2_ERR_LAUNCH;
      const int64_t S2 = (int64_t)B * N;
      AttnCore a{g, g + E, g + 2 * E, 3 * E, 3 * E, 3 * E, S2 * 3 * E, S2 * 3 * E, S2 * 3 * E, gc, E, S2 * E,
                 T, (int)S2, (int)S2, H, d, scale, nullptr, 0};
      const int e = attention_core(ar, a, dry, st);
      if (e != U2_OK) return e;
      if (!dry)
        for (int b = 0; b < B; ++b)
          if (hipMemcpy2DAsync(ctx + (size_t)b * T * N * E, (size_t)N * E * 2, gc + (size_t)b * N * E, (size_t)B * N * E * 2,
                               (size_t)N * E * 2, T, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return U2_ERR_LAUNCH;
      ar.off = mark;
    } else {
      const int e = chunk_axis_attention(ar, qkv, ctx, B, T, N, H, d, scale, tp.rb, c.max_seq_len, dry, st);
      if (e != U2_OK) return e;
    }
    U2_RUN(linear(ctx, E, tp.wd, tp.bd, y2, E, rows, E, E, 0, nullptr, 0, st));
    x = y2;
    if (taps && !dry && taps->svr_out && taps->svr_out[l] &&
        hipMemcpyAsync(taps->svr_out[l], x, (size_t)rows * E * sizeof(bf16_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
      return U2_ERR_LAUNCH;
  }

  if (svr_out && !dry) {  // optional tap: the refined tokens the selection stage sees (parity tests)
    if (hipMemcpyAsync(svr_out, x, (size_t)rows * E * sizeof(bf16_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
      return U2_ERR_LAUNCH;
  }

  // ---------------- token selection (svr.py:171)
  bf16_t* sel = ar.get<bf16_t>((size_t)B * k * E);
  U2_CHECK_WS(ar);
  if (c.enable_diffts) {
    // DifferentiableTokenSelection (svr.py:101-117): weights = softmax_tokens(score_net(x)/tau); out = W^T X.
    // Computed operand-swapped so the scores land transposed: scT[r][tok] = score_w[r] . x[tok] + b[r].
    const size_t mark = ar.off;
    const int64_t ldS = round_up(TN, 8);
    float* scT = ar.get<float>((size_t)B * k * ldS);
    bf16_t* P = ar.get<bf16_t>((size_t)B * k * ldS);
    const bool x_in_place = opts().kmajor_b && !(TN & 7);  // the aggregation reads X as a K-major B operand (rows = tokens)
    bf16_t* Xt = x_in_place ? nullptr : ar.get<bf16_t>((size_t)B * E * ldS);
    U2_CHECK_WS(ar);
    {
      GemmDesc g;
      g.A = wp(i_sel); g.B = x; g.C = scT; g.bias = wp(i_sel + 1);
      g.M = k; g.N = TN; g.K = E; g.lda = E; g.ldb = E; g.ldc = ldS;
      g.nz = B; g.sBb = (int64_t)TN * E; g.sCb = (int64_t)k * ldS;
      g.flags = GEMM_BIAS_M | GEMM_OUT_F32;
      U2_RUN(gemm_bf16(g, st));
    }
    U2_RUN(softmax_rows(scT, P, B, k, TN, ldS, ldS, (int64_t)k * ldS, (int64_t)k * ldS, 1.0f / c.diffts_tau, nullptr, 1,
                        0, st));
    {
      GemmDesc g;
      g.A = P; g.C = sel;
      g.M = k; g.N = E; g.lda = ldS; g.ldc = E;
      g.nz = B; g.sAb = (int64_t)k * ldS; g.sCb = (int64_t)k * E;
      if (x_in_place) {
        g.B = x; g.K = TN; g.ldb = E; g.sBb = (int64_t)TN * E;
        g.flags = GEMM_B_KMAJOR;
      } else {
        U2_RUN(transpose_bf16(x, Xt, B, TN, E, E, ldS, (int64_t)TN * E, (int64_t)E * ldS, 0, st));
        g.B = Xt; g.K = (int)ldS; g.ldb = ldS; g.sBb = (int64_t)E * ldS;
      }
      U2_RUN(gemm_bf16(g, st));
    }
    ar.off = mark;
  } else {
    // TokenSelection (svr.py:75-91): hard top-k over the flattened (t n) axis, sorted by score
    const size_t mark = ar.off;
    float* scores = ar.get<float>((size_t)rows);
    int64_t* idx = topk_idx_out ? topk_idx_out : ar.get<int64_t>((size_t)B * k);
    U2_CHECK_WS(ar);
    U2_RUN(score_gemv(x, wp(i_sel), wp(i_sel + 1), scores, (int)rows, E, st));
    U2_RUN(topk_sorted(scores, idx, B, TN, k, st));
    U2_RUN(gather_rows(x, idx, sel, B, TN, k, E, st));
    ar.off = mark;
  }

  // ---------------- multi-scale pooling (svr.py:173-184)
  const bf16_t* V = sel;
  int Lv = k;
  if (c.use_multi_scale) {
    Lv = k + k / 2 + k / 4;
    bf16_t* pooled = ar.get<bf16_t>((size_t)B * Lv * E);
    float* gws = ar.get<float>((size_t)B * 3 * 16 * cdiv(E, 256));
    U2_CHECK_WS(ar);
    U2_RUN(multiscale_pool(sel, pooled, B, k, E, c.enable_dmtp ? wp(i_gate) : nullptr,
                           c.enable_dmtp ? wp(i_gate + 1) : nullptr, gws, st));
    V = pooled;
  }
  if (taps && !dry) {  // parity taps: the visual tokens the aggregation stage attends to
    if (taps->visual_out &&
        hipMemcpyAsync(taps->visual_out, V, (size_t)B * Lv * E * sizeof(bf16_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
      return U2_ERR_LAUNCH;
    if (taps->visual_in) V = reinterpret_cast<const bf16_t*>(taps->visual_in);
  }

  // ---------------- TTA: TextConditionTokenAggregatorModel (tta.py:126-140)
  const int64_t qrows = (int64_t)B * Q;
  bf16_t* qa = ar.get<bf16_t>((size_t)qrows * E);
  bf16_t* qb = ar.get<bf16_t>((size_t)qrows * E);
  bf16_t* qc = ar.get<bf16_t>((size_t)qrows * E);
  bf16_t* qproj = ar.get<bf16_t>((size_t)qrows * 3 * E);
  bf16_t* qctx = ar.get<bf16_t>((size_t)qrows * E);
  bf16_t* qo = ar.get<bf16_t>((size_t)qrows * E);
  // k | v of the two cross attentions of every layer: one buffer each, filled on the side stream (or in line)
  SideStream* side = (!dry && opts().tta_overlap && L > 0 && L <= 7) ? cx.side_for(st) : nullptr;
  const bool overlap = dry ? (L > 0 && L <= 7) : side != nullptr;  // dry: size for the larger (overlapped) layout
  const int Lmax = Lv > c.Lt ? Lv : c.Lt;
  bf16_t* kv_inline = overlap ? nullptr : ar.get<bf16_t>((size_t)B * Lmax * 2 * E);
  bf16_t* kv_v[8] = {};
  bf16_t* kv_t[8] = {};
  if (overlap)
    for (int l = 0; l < L; ++l) {
      kv_v[l] = ar.get<bf16_t>((size_t)B * Lv * 2 * E);
      kv_t[l] = ar.get<bf16_t>((size_t)B * c.Lt * 2 * E);
    }
  bf16_t* k_lin = ar.get<bf16_t>((size_t)B * Lv * E);  // key projection of the final linear aggregation
  U2_CHECK_WS(ar);
  auto kv_proj = [&](const Att& a, const bf16_t* src, int Ls, bf16_t* kv, hipStream_t s_) -> int {
    if (!dry && kv_packed(a, E)) {
      U2_RUN(linear(src, E, a.wk, a.bk, kv, 2 * E, (int64_t)B * Ls, E, 2 * E, 0, nullptr, 0, s_));
    } else {
      U2_RUN(linear(src, E, a.wk, a.bk, kv, 2 * E, (int64_t)B * Ls, E, E, 0, nullptr, 0, s_));
      U2_RUN(linear(src, E, a.wv, a.bv, kv + E, 2 * E, (int64_t)B * Ls, E, E, 0, nullptr, 0, s_));
    }
    return U2_OK;
  };
  if (overlap && !dry) {
    // fork: everything `st` has enqueued so far (V and t_token are final) precedes the side stream's work
    if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess)
      return U2_ERR_LAUNCH;
    for (int l = 0; l < L; ++l) {
      const int base = i_tta + 33 * l;
      Att va = att_at(base + 9), ta = att_at(base + 18);
      { const int e = kv_proj(va, V, Lv, kv_v[l], side->s); if (e != U2_OK) return e; }
      if (hipEventRecord(side->done[2 * l], side->s) != hipSuccess) return U2_ERR_LAUNCH;
      { const int e = kv_proj(ta, t_token, c.Lt, kv_t[l], side->s); if (e != U2_OK) return e; }
      if (hipEventRecord(side->done[2 * l + 1], side->s) != hipSuccess) return U2_ERR_LAUNCH;
    }
    const Att la = att_at(i_lin);
    U2_RUN(linear(V, E, la.wk, la.bk, k_lin, E, (int64_t)B * Lv, E, E, 0, nullptr, 0, side->s));
    if (hipEventRecord(side->done[2 * L], side->s) != hipSuccess) return U2_ERR_LAUNCH;
  }
  U2_RUN(fill_rows(wp(0), qa, B, (int64_t)Q * E, (int64_t)Q * E, st));  // query_tokens.expand(B,-1,-1)
  bf16_t* qcur = qa;
  // kv_ready: index of the side-stream event that publishes `kv` (-1: compute it here)
  auto cross = [&](const Att& a, const bf16_t* src, int Ls, const bf16_t* qin, bf16_t* dst, bf16_t* kv,
                   int kv_ready) -> int {
    // MultiHeadCrossAttention.forward (tta.py:42-69), is_compress = False
    U2_RUN(linear(qin, E, a.wq, a.bq, qproj, E, qrows, E, E, 0, nullptr, 0, st));
    if (kv_ready < 0) {
      const int e = kv_proj(a, src, Ls, kv, st);
      if (e != U2_OK) return e;
    } else if (!dry) {
      if (hipStreamWaitEvent(st, side->done[kv_ready], 0) != hipSuccess) return U2_ERR_LAUNCH;
    }
    AttnCore ac{qproj, kv, kv + E, E, 2 * E, 2 * E, (int64_t)Q * E, (int64_t)Ls * 2 * E, (int64_t)Ls * 2 * E,
                qctx, E, (int64_t)Q * E, B, Q, Ls, H, d, scale, nullptr, 0};
    const int e = attention_core(ar, ac, dry, st);
    if (e != U2_OK) return e;
    U2_RUN(linear(qctx, E, a.wd, a.bd, dst, E, qrows, E, E, 0, nullptr, 0, st));
    return U2_OK;
  };
  for (int l = 0; l < L; ++l) {
    const int base = i_tta + 33 * l;
    const Att sa = att_at(base);
    Att va = att_at(base + 9), ta = att_at(base + 18);
    va.rb = ta.rb = nullptr;
    const bf16_t *ns_w = wp(base + 27), *ns_b = wp(base + 28), *nv_w = wp(base + 29), *nv_b = wp(base + 30),
                 *nt_w = wp(base + 31), *nt_b = wp(base + 32);
    if (taps && !dry && taps->tta_in && taps->tta_in[l]) qcur = (bf16_t*)taps->tta_in[l];  // (only ever read)
    bf16_t* s1 = (qcur == qa) ? qb : qa;
    bf16_t* s2 = qc;
    // self attention on the query tokens + post-LN residual (tta.py:94-96)
    if (c.attn_type == 2 && B == 1) {
      // (B = 1, Q, E) read sequence-first: every query token is a batch entry with a sequence of ONE key, whose
      // softmax weight is exactly 1 -> the context is the value projection itself (tta.py:84,94)
      U2_RUN(linear(qcur, E, sa.wv, sa.bv, qctx, E, qrows, E, E, 0, nullptr, 0, st));
    } else if (c.attn_type == 2) {
      // (B, Q, E) read sequence-first: the sequence is the B batch entries of one query index
      { const int e = qkv_proj(qcur, sa, qproj, qrows, E, dry, st); if (e != U2_OK) return e; }
      const int e = chunk_axis_attention(ar, qproj, qctx, 1, B, Q, H, d, scale, nullptr, c.max_seq_len, dry, st);
      if (e != U2_OK) return e;
    } else {
      { const int e = qkv_proj(qcur, sa, qproj, qrows, E, dry, st); if (e != U2_OK) return e; }
      if (c.attn_type == 1) {
        U2_RUN(rope_apply(qproj, B, Q, 1, H, d, 3 * E, c.max_seq_len, 0, st));
        U2_RUN(rope_apply(qproj + E, B, Q, 1, H, d, 3 * E, c.max_seq_len, 0, st));
      }
      AttnCore ac{qproj, qproj + E, qproj + 2 * E, 3 * E, 3 * E, 3 * E, (int64_t)Q * 3 * E, (int64_t)Q * 3 * E,
                  (int64_t)Q * 3 * E, qctx, E, (int64_t)Q * E, B, Q, Q, H, d, scale, sa.rb, c.max_seq_len};
      const int e = attention_core(ar, ac, dry, st);
      if (e != U2_OK) return e;
    }
    U2_RUN(linear(qctx, E, sa.wd, sa.bd, qo, E, qrows, E, E, 0, nullptr, 0, st));
    U2_RUN(layernorm_bf16(qcur, qo, ns_w, ns_b, s1, 1, (int)qrows, E, 0, E, 0, E, 0, E, c.ln_eps, st));
    // visual cross attention (tta.py:97-100)
    { const int e = cross(va, V, Lv, s1, qo, overlap ? kv_v[l] : kv_inline, overlap ? 2 * l : -1); if (e != U2_OK) return e; }
    U2_RUN(layernorm_bf16(s1, qo, nv_w, nv_b, s2, 1, (int)qrows, E, 0, E, 0, E, 0, E, c.ln_eps, st));
    // text cross attention (tta.py:101-103) -- no padding mask in the reference
    { const int e = cross(ta, t_token, c.Lt, s2, qo, overlap ? kv_t[l] : kv_inline, overlap ? 2 * l + 1 : -1); if (e != U2_OK) return e; }
    U2_RUN(layernorm_bf16(s2, qo, nt_w, nt_b, s1, 1, (int)qrows, E, 0, E, 0, E, 0, E, c.ln_eps, st));
    qcur = s1;
    if (taps && !dry && taps->tta_out && taps->tta_out[l] &&
        hipMemcpyAsync(taps->tta_out[l], qcur, (size_t)qrows * E * sizeof(bf16_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
      return U2_ERR_LAUNCH;
  }
  // ---------------- LinearAggregation (tta.py:109-116): is_compress=True -> V un-projected, no out-proj
  {
    const Att la = att_at(i_lin);
    U2_RUN(linear(qcur, E, la.wq, la.bq, qproj, E, qrows, E, E, 0, nullptr, 0, st));
    if (!overlap) {
      U2_RUN(linear(V, E, la.wk, la.bk, k_lin, E, (int64_t)B * Lv, E, E, 0, nullptr, 0, st));
    } else if (!dry) {
      if (hipStreamWaitEvent(st, side->done[2 * L], 0) != hipSuccess) return U2_ERR_LAUNCH;
    }
    AttnCore ac{qproj, k_lin, V, E, E, E, (int64_t)Q * E, (int64_t)Lv * E, (int64_t)Lv * E,
                out, E, (int64_t)Q * E, B, Q, Lv, H, d, scale, nullptr, 0};
    const int e = attention_core(ar, ac, dry, st);
    if (e != U2_OK) return e;
  }
  if (peak) *peak = ar.peak;
  U2_CHECK_WS(ar);
  return U2_OK;
}

}  // namespace u2
