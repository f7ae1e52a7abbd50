// Row kernels of the decoder prefill that consumes the path's output (the step AFTER the path --
// LlamaForCausalLM / Qwen3ForCausalLM.forward on the spliced inputs_embeds, src/model/language_model/u2llama.py:76-87,123-126).
// The decoder stays the stock HuggingFace module tree (its parameters, its KV cache, its generate loop); for the PREFILL of
// the S = 1024 spliced embeddings the host side (u2tokenizer_amd/prefill.py) runs each layer as
//   RMSNorm -> packed q|k|v GEMM -> per-head RMSNorm (Qwen3) + rotary embedding -> causal grouped-query attention
//   (tokattn.hip) -> out-projection GEMM (+ residual) -> RMSNorm -> packed gate|up GEMM -> SiLU(gate) * up -> down GEMM (+ residual)
// and these are the HBM-bound pieces between the GEMMs.  All bf16 in / out, fp32 arithmetic, 16-byte accesses.
#include "kernels.h"

namespace u2 {

// ------------------------------------------------------------------------------------------------ RMSNorm
// y[r][:] = bf16(x[r][:] * rsqrt(mean(x[r][:]^2) + eps)) * w   (LlamaRMSNorm / Qwen3RMSNorm: the normalised value is rounded
// to the input dtype before the product with w -- the same two rounding points here)
template <int NC>  // 16-byte chunks per lane held in registers: C <= NC * 512
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ y, int64_t rows, int C, int64_t ldx, int64_t ldy,
                                                      float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xp = x + row * ldx;
  bf16_t* yp = y + row * ldy;
  const int nchunk = C >> 3;
  uint4 v[NC];
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = i * 64 + lane;
    v[i] = uint4{0, 0, 0, 0};
    if (c < nchunk) {
      v[i] = *reinterpret_cast<const uint4*>(xp + c * 8);
      const float a0 = bf16lo(v[i].x), a1 = bf16hi(v[i].x), a2 = bf16lo(v[i].y), a3 = bf16hi(v[i].y);
      const float a4 = bf16lo(v[i].z), a5 = bf16hi(v[i].z), a6 = bf16lo(v[i].w), a7 = bf16hi(v[i].w);
      sq += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3 + a4 * a4 + a5 * a5 + a6 * a6 + a7 * a7;
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = i * 64 + lane;
    if (c < nchunk) {
      const uint4 uw = *reinterpret_cast<const uint4*>(w + c * 8);
      const uint4 n = uint4{pack2_bf16(bf16lo(v[i].x) * rstd, bf16hi(v[i].x) * rstd), pack2_bf16(bf16lo(v[i].y) * rstd, bf16hi(v[i].y) * rstd),
                            pack2_bf16(bf16lo(v[i].z) * rstd, bf16hi(v[i].z) * rstd), pack2_bf16(bf16lo(v[i].w) * rstd, bf16hi(v[i].w) * rstd)};
      *reinterpret_cast<uint4*>(yp + c * 8) =
          uint4{pack2_bf16(bf16lo(n.x) * bf16lo(uw.x), bf16hi(n.x) * bf16hi(uw.x)), pack2_bf16(bf16lo(n.y) * bf16lo(uw.y), bf16hi(n.y) * bf16hi(uw.y)),
                pack2_bf16(bf16lo(n.z) * bf16lo(uw.z), bf16hi(n.z) * bf16hi(uw.z)), pack2_bf16(bf16lo(n.w) * bf16lo(uw.w), bf16hi(n.w) * bf16hi(uw.w))};
    }
  }
}

int rmsnorm_bf16(const bf16_t* x, const bf16_t* w, bf16_t* y, int64_t rows, int C, int64_t ldx, int64_t ldy, float eps,
                 hipStream_t stream) {
  if (!x || !w || !y || rows <= 0 || C <= 0 || (C & 7) || C > 8192 || (ldx & 7) || (ldy & 7)) return U2_ERR_ARG;
  if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) || cdiv(rows, 4) > 0x7fffffff) return U2_ERR_ARG;
  ProfScope ps(PROF_ROWOP, 0, stream, (double)rows * C * 4.0);
  dim3 grid((unsigned)cdiv(rows, 4));
#define U2_RN(NC) hipLaunchKernelGGL((rmsnorm_kernel<NC>), grid, dim3(256), 0, stream, x, w, y, rows, C, ldx, ldy, eps)
  if (C <= 1024) U2_RN(2);
  else if (C <= 2048) U2_RN(4);
  else if (C <= 4096) U2_RN(8);
  else U2_RN(16);
#undef U2_RN
  return launch_status();
}

// ------------------------------------------------------------------------------------------------ q / k: head norm + rotary
// In place on the q and k column blocks of the packed projection output qkv[rows][(Hq + 2 Hkv) D]: for every (row, head)
//   x <- RMSNorm_D(x) * w_q|k          (Qwen3Attention.q_norm / k_norm; skipped when the weight pointer is null: Llama)
//   x <- x * cos + rotate_half(x) * sin (apply_rotary_pos_emb; cos / sin: [rows][D] fp32 or bf16 as HF hands them out)
// One wave per (row, head); lane l < D/2 holds the pair (x[l], x[l + D/2]) rotate_half couples.
// kc / vc (optional): the KV cache's own layout, [batch][kv head][S][D] dense (HF DynamicLayer: (B, H_kv, S, D)); the finished
// k heads are written there as well and the v heads copied, so the cache takes the tensors as they are (row = batch * S + s).
template <int D, typename CS>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(bf16_t* __restrict__ qkv, const bf16_t* __restrict__ wq,
                                                           const bf16_t* __restrict__ wk, const CS* __restrict__ cosp,
                                                           const CS* __restrict__ sinp, int64_t rows, int Hq, int Hkv,
                                                           int64_t ld, int64_t cs_ld, float eps, bf16_t* __restrict__ kc,
                                                           bf16_t* __restrict__ vc, int S, int64_t kvs, int s_off) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nh = Hq + Hkv + (kc ? Hkv : 0);
  if (item >= rows * nh) return;
  const int64_t row = item / nh;
  const int hh = (int)(item - row * nh);
  bf16_t* p = qkv + row * ld + (int64_t)hh * D;  // k heads follow the q heads in the packed row, v heads the k heads
  constexpr int HALF = D / 2;
  const bool on = lane < HALF;
  bf16_t* cdst = nullptr;
  if (kc && hh >= Hq) {
    const int64_t bi = row / S, si = row - bi * S;
    const int hk = hh - Hq;
    // (kvs: elements between consecutive (batch, kv head) entries -- S * D dense, or the capacity of an append-in-place cache,
    //  whose next free position is s_off)
    cdst = (hk < Hkv ? kc + (bi * Hkv + hk) * kvs + (int64_t)(s_off + si) * D : vc + (bi * Hkv + (hk - Hkv)) * kvs + (int64_t)(s_off + si) * D);
    if (hk >= Hkv) {  // a v head: copy
      if (on) {
        cdst[lane] = p[lane];
        cdst[lane + HALF] = p[lane + HALF];
      }
      return;
    }
  }
  const bf16_t* w = hh < Hq ? wq : wk;
  float a = 0.f, b = 0.f;
  if (on) {
    a = bf16_to_f32(p[lane]);
    b = bf16_to_f32(p[lane + HALF]);
  }
  if (w) {
    const float rstd = rsqrtf(wave_sum(a * a + b * b) / (float)D + eps);
    if (on) {
      // (the reference rounds the normalised value to bf16 before the weight product: keep that rounding point)
      a = bf16_to_f32(f32_to_bf16(a * rstd)) * bf16_to_f32(w[lane]);
      b = bf16_to_f32(f32_to_bf16(b * rstd)) * bf16_to_f32(w[lane + HALF]);
      a = bf16_to_f32(f32_to_bf16(a));
      b = bf16_to_f32(f32_to_bf16(b));
    }
  }
  if (on) {
    const CS* cr = cosp + row * cs_ld;
    const CS* sr = sinp + row * cs_ld;
    const float c0 = (float)cr[lane], c1 = (float)cr[lane + HALF], s0 = (float)sr[lane], s1 = (float)sr[lane + HALF];
    const bf16_t r0 = f32_to_bf16(a * c0 - b * s0), r1 = f32_to_bf16(b * c1 + a * s1);  // rotate_half(x) = (-x2, x1)
    p[lane] = r0;
    p[lane + HALF] = r1;
    if (cdst) {
      cdst[lane] = r0;
      cdst[lane + HALF] = r1;
    }
  }
}

struct Bf16Val {  // bf16 cos / sin tables
  bf16_t v;
  __device__ explicit operator float() const { return bf16_to_f32(v); }
};

int qk_norm_rope(bf16_t* qkv, const bf16_t* wq, const bf16_t* wk, const void* cosp, const void* sinp, int cs_is_f32,
                 int64_t rows, int Hq, int Hkv, int D, int64_t ld, int64_t cs_ld, float eps, bf16_t* kc, bf16_t* vc, int S,
                 int64_t kv_stride, int s_off, hipStream_t stream) {
  if (!qkv || !cosp || !sinp || rows <= 0 || Hq <= 0 || Hkv <= 0 || (D != 64 && D != 128) || (!wq) != (!wk)) return U2_ERR_ARG;
  if ((!kc) != (!vc) || (kc && (S <= 0 || rows % S))) return U2_ERR_ARG;
  if (kv_stride == 0) kv_stride = (int64_t)S * D;
  if (kc && (s_off < 0 || kv_stride < (int64_t)(s_off + S) * D)) return U2_ERR_ARG;
  const int64_t items = rows * (Hq + Hkv + (kc ? Hkv : 0));
  if (cdiv(items, 4) > 0x7fffffff) return U2_ERR_ARG;
  ProfScope ps(PROF_ROWOP, 0, stream, (double)items * D * 4.0);
  dim3 grid((unsigned)cdiv(items, 4));
#define U2_QK(D_, T_)                                                                                                     \
  hipLaunchKernelGGL((qk_norm_rope_kernel<D_, T_>), grid, dim3(256), 0, stream, qkv, wq, wk, reinterpret_cast<const T_*>(cosp), \
                     reinterpret_cast<const T_*>(sinp), rows, Hq, Hkv, ld, cs_ld, eps, kc, vc, S, kv_stride, s_off)
  if (D == 128 && cs_is_f32) U2_QK(128, float);
  else if (D == 128) U2_QK(128, Bf16Val);
  else if (cs_is_f32) U2_QK(64, float);
  else U2_QK(64, Bf16Val);
#undef U2_QK
  return launch_status();
}

// ------------------------------------------------------------------------------------------------ SwiGLU
// out[r][i] = silu(gu[r][i]) * gu[r][I + i]   (LlamaMLP / Qwen3MLP: down_proj(act_fn(gate_proj(x)) * up_proj(x)) with the
// gate | up projections packed into one GEMM); the reference rounds silu(gate) to bf16 before the product: kept.
__global__ __launch_bounds__(256) void swiglu_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ out, int64_t rows, int I,
                                                     int64_t ld_in, int64_t ld_out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per = I >> 3;
  if (idx >= rows * per) return;
  const int64_t r = idx / per;
  const int c = (int)(idx - r * per) * 8;
  const uint4 g = *reinterpret_cast<const uint4*>(gu + r * ld_in + c);
  const uint4 u = *reinterpret_cast<const uint4*>(gu + r * ld_in + I + c);
  auto f = [](float gate, float up) {
    const float s = gate / (1.0f + __expf(-gate));
    return bf16_to_f32(f32_to_bf16(s)) * up;
  };
  *reinterpret_cast<uint4*>(out + r * ld_out + c) =
      uint4{pack2_bf16(f(bf16lo(g.x), bf16lo(u.x)), f(bf16hi(g.x), bf16hi(u.x))),
            pack2_bf16(f(bf16lo(g.y), bf16lo(u.y)), f(bf16hi(g.y), bf16hi(u.y))),
            pack2_bf16(f(bf16lo(g.z), bf16lo(u.z)), f(bf16hi(g.z), bf16hi(u.z))),
            pack2_bf16(f(bf16lo(g.w), bf16lo(u.w)), f(bf16hi(g.w), bf16hi(u.w)))};
}

int swiglu_bf16(const bf16_t* gu, bf16_t* out, int64_t rows, int I, int64_t ld_in, int64_t ld_out, hipStream_t stream) {
  if (!gu || !out || rows <= 0 || I <= 0 || (I & 7) || (ld_in & 7) || (ld_out & 7) || (((uintptr_t)gu | (uintptr_t)out) & 15))
    return U2_ERR_ARG;
  const int64_t total = rows * (I >> 3);
  if (cdiv(total, 256) > 0x7fffffff) return U2_ERR_ARG;
  ProfScope ps(PROF_ROWOP, 0, stream, (double)rows * I * 6.0);
  hipLaunchKernelGGL(swiglu_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, gu, out, rows, I, ld_in, ld_out);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------ one decode step of a layer
// The two halves of a decoder layer's decode step (B <= 16 new tokens against the KV cache), each ONE call for the host: the
// step `generate` repeats up to 768 times per report is bound by the host's launch rate when every kernel is its own Python
// call (7.9 ms per step of a 36-layer decoder, ~4 ms of kernels).  Between the halves the host appends k / v to its cache
// (HF DynamicCache: a torch.cat) and hands back the dense (B, H_kv, T, D) tensors.
static int dec_linear(const bf16_t* x, int64_t ldx, const bf16_t* w, const bf16_t* b, bf16_t* y, int64_t ldy, int rows, int in,
                      int out, const bf16_t* R, int64_t ldr, hipStream_t st) {
  GemmDesc g;
  g.A = x; g.B = w; g.C = y; g.bias = b; g.R = R;
  g.M = rows; g.N = out; g.K = in;
  g.lda = ldx; g.ldb = in; g.ldc = ldy; g.ldr = ldr;
  g.flags = (b ? GEMM_BIAS_N : 0) | (R ? GEMM_RESIDUAL : 0);
  return gemm_bf16(g, st);
}

size_t decoder_decode_workspace_bytes(const DecodeCfg& c, int T) {
  const size_t rows = (size_t)c.B;
  size_t n = rows * ((size_t)3 * c.E + (size_t)c.Hq * c.D + (size_t)3 * c.I) * sizeof(bf16_t);  // xn, h, hn | ctx | gu (2 I), act
  n = (n + 255) & ~(size_t)255;
  return n + tok_attention_workspace_bytes(c.Hkv, c.Hq / c.Hkv, 1, T, c.D) + 256;
}

// input RMSNorm -> q|k|v projection -> per-head RMSNorm + rotary; qkv (B, (Hq + 2 Hkv) D) keeps the finished queries, kc / vc
// (B, Hkv, 1, D) receive the new cache entries
int decoder_decode_pre(const DecodeCfg& c, const bf16_t* x, const bf16_t* w_in_norm, const bf16_t* Wqkv, const bf16_t* bqkv,
                       const bf16_t* wq_norm, const bf16_t* wk_norm, const void* cosp, const void* sinp, int cs_is_f32,
                       int64_t cs_ld, bf16_t* qkv, bf16_t* kc, bf16_t* vc, int64_t kv_stride, int s_off, void* ws, size_t ws_bytes,
                       hipStream_t st) {
  if (c.B <= 0 || c.B > 16 || c.Hkv <= 0 || c.Hq % c.Hkv || !x || !w_in_norm || !Wqkv || !qkv || !kc || !vc || !ws) return U2_ERR_ARG;
  if (ws_bytes < (size_t)c.B * c.E * sizeof(bf16_t)) return U2_ERR_WORKSPACE;
  bf16_t* xn = reinterpret_cast<bf16_t*>(ws);
  const int nq = (c.Hq + 2 * c.Hkv) * c.D;
  int e = rmsnorm_bf16(x, w_in_norm, xn, c.B, c.E, c.E, c.E, c.eps, st);
  if (e != U2_OK) return e;
  e = dec_linear(xn, c.E, Wqkv, bqkv, qkv, nq, c.B, c.E, nq, nullptr, 0, st);
  if (e != U2_OK) return e;
  return qk_norm_rope(qkv, wq_norm, wk_norm, cosp, sinp, cs_is_f32, c.B, c.Hq, c.Hkv, c.D, nq, cs_ld, c.qk_eps, kc, vc, 1, kv_stride,
                      s_off, st);
}

// attention over the cache (keys split over workgroups) -> out projection + residual -> RMSNorm -> gate|up -> SwiGLU -> down
// projection + residual.  K / V: (B, Hkv, T, D) with kv_stride elements between (batch, kv head) entries (0: dense); out (B, E).
int decoder_decode_post(const DecodeCfg& c, const bf16_t* x, const bf16_t* qkv, const bf16_t* K, const bf16_t* V, int T,
                        int64_t kv_stride, const bf16_t* Wo, const bf16_t* bo, const bf16_t* w_post_norm, const bf16_t* Wgu, const bf16_t* bgu,
                        const bf16_t* Wdown, const bf16_t* bdown, bf16_t* out, void* ws, size_t ws_bytes, hipStream_t st) {
  if (c.B <= 0 || c.B > 16 || T <= 0 || !x || !qkv || !K || !V || !Wo || !w_post_norm || !Wgu || !Wdown || !out || !ws) return U2_ERR_ARG;
  if (ws_bytes < decoder_decode_workspace_bytes(c, T)) return U2_ERR_WORKSPACE;
  const int g = c.Hq / c.Hkv, nq = (c.Hq + 2 * c.Hkv) * c.D, qd = c.Hq * c.D;
  bf16_t* p = reinterpret_cast<bf16_t*>(ws);
  bf16_t* h = p + (size_t)c.B * c.E;          // (xn of the first half lives at p)
  bf16_t* hn = h + (size_t)c.B * c.E;
  bf16_t* ctx = hn + (size_t)c.B * c.E;
  bf16_t* gu = ctx + (size_t)c.B * qd;
  bf16_t* act = gu + (size_t)c.B * 2 * c.I;
  size_t used = ((size_t)c.B * ((size_t)3 * c.E + qd + (size_t)3 * c.I) * sizeof(bf16_t) + 255) & ~(size_t)255;
  char* aws = reinterpret_cast<char*>(ws) + used;
  const size_t aws_bytes = ws_bytes - used;
  const float scale = c.scale;
  if (kv_stride == 0) kv_stride = (int64_t)T * c.D;
  if (kv_stride < (int64_t)T * c.D || (kv_stride & 7)) return U2_ERR_ARG;
  for (int b = 0; b < c.B; ++b) {  // entries of one batch element: its kv heads; the g query heads of a group are the "heads"
    const int e = attention_ex(qkv + (size_t)b * nq, K + (size_t)b * c.Hkv * kv_stride, V + (size_t)b * c.Hkv * kv_stride,
                               ctx + (size_t)b * qd, c.Hkv, 1, T, g, 1, c.D, /*ldq*/ (int64_t)g * c.D, /*ldk*/ c.D, /*ldv*/ c.D,
                               /*ldo*/ (int64_t)g * c.D, /*q_bs*/ (int64_t)g * c.D, /*k_bs*/ kv_stride, /*v_bs*/ kv_stride,
                               /*o_bs*/ (int64_t)g * c.D, scale, nullptr, 0, 0, 0, aws, aws_bytes, st);
    if (e != U2_OK) return e;
  }
  int e = dec_linear(ctx, qd, Wo, bo, h, c.E, c.B, qd, c.E, x, c.E, st);
  if (e != U2_OK) return e;
  e = rmsnorm_bf16(h, w_post_norm, hn, c.B, c.E, c.E, c.E, c.eps, st);
  if (e != U2_OK) return e;
  if (!bgu && !(c.E & 63) && !(c.I & 15)) {  // SiLU(gate) * up in the epilogue of the pair product (gemm_rows16_kernel<., true>)
    GemmDesc g;
    g.A = hn; g.B = Wgu; g.C = act;
    g.M = c.B; g.N = 2 * c.I; g.K = c.E;
    g.lda = c.E; g.ldb = c.E; g.ldc = c.I;
    g.flags = GEMM_SWIGLU;
    e = gemm_bf16(g, st);
    if (e != U2_OK) return e;
  } else {
    e = dec_linear(hn, c.E, Wgu, bgu, gu, 2 * c.I, c.B, c.E, 2 * c.I, nullptr, 0, st);
    if (e != U2_OK) return e;
    e = swiglu_bf16(gu, act, c.B, c.I, 2 * c.I, c.I, st);
    if (e != U2_OK) return e;
  }
  return dec_linear(act, c.I, Wdown, bdown, out, c.E, c.B, c.I, c.E, h, c.E, st);
}

}  // namespace u2
