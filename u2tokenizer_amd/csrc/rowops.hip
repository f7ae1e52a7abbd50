// HBM-bound row kernels: LayerNorm(+residual), row softmax (+scale, +Toeplitz relative bias), bf16
// transpose.  One wave64 per row, 16-byte vector accesses, fp32 arithmetic.
#include "kernels.h"

namespace u2 {

// ---------------------------------------------------------------- LayerNorm
// Reference: nn.LayerNorm(hidden) in MONAI TransformerBlock norm1/norm2, ViT.norm (vit.py:107,124) and
// TextConditionTokenAttMap.norm_self/norm_cross_v/norm_cross_t (tta.py:78-79,88,96,100,103) where the
// residual sum is formed first: LN(q + attn(q)).
template <int NC>  // chunks (8 bf16) per lane held in registers: C <= NC * 512
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                        const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
                                                        bf16_t* __restrict__ y, int nb, int rows, int C,
                                                        int64_t x_bs, int64_t x_ld, int64_t res_bs, int64_t res_ld,
                                                        int64_t y_bs, int64_t y_ld, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)nb * rows) return;
  const int b = (int)(row / rows), r = (int)(row - (int64_t)b * rows);
  const bf16_t* xp = x + b * x_bs + r * x_ld;
  const bf16_t* rp = res ? res + b * res_bs + r * res_ld : nullptr;
  bf16_t* yp = y + b * y_bs + r * y_ld;
  const int nchunk = C >> 3;
  float v[NC][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = i * 64 + lane;
    if (c < nchunk) {
      const uint4 u = *reinterpret_cast<const uint4*>(xp + c * 8);
      v[i][0] = bf16lo(u.x); v[i][1] = bf16hi(u.x); v[i][2] = bf16lo(u.y); v[i][3] = bf16hi(u.y);
      v[i][4] = bf16lo(u.z); v[i][5] = bf16hi(u.z); v[i][6] = bf16lo(u.w); v[i][7] = bf16hi(u.w);
      if (rp) {
        const uint4 q = *reinterpret_cast<const uint4*>(rp + c * 8);
        v[i][0] += bf16lo(q.x); v[i][1] += bf16hi(q.x); v[i][2] += bf16lo(q.y); v[i][3] += bf16hi(q.y);
        v[i][4] += bf16lo(q.z); v[i][5] += bf16hi(q.z); v[i][6] += bf16lo(q.w); v[i][7] += bf16hi(q.w);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    if (i * 64 + lane < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float dlt = v[i][j] - mean; sq += dlt * dlt; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = i * 64 + lane;
    if (c < nchunk) {
      const uint4 uw = *reinterpret_cast<const uint4*>(w + c * 8);
      const uint4 ub = *reinterpret_cast<const uint4*>(bias + c * 8);
      float o[8];
      o[0] = (v[i][0] - mean) * rstd * bf16lo(uw.x) + bf16lo(ub.x);
      o[1] = (v[i][1] - mean) * rstd * bf16hi(uw.x) + bf16hi(ub.x);
      o[2] = (v[i][2] - mean) * rstd * bf16lo(uw.y) + bf16lo(ub.y);
      o[3] = (v[i][3] - mean) * rstd * bf16hi(uw.y) + bf16hi(ub.y);
      o[4] = (v[i][4] - mean) * rstd * bf16lo(uw.z) + bf16lo(ub.z);
      o[5] = (v[i][5] - mean) * rstd * bf16hi(uw.z) + bf16hi(ub.z);
      o[6] = (v[i][6] - mean) * rstd * bf16lo(uw.w) + bf16lo(ub.w);
      o[7] = (v[i][7] - mean) * rstd * bf16hi(uw.w) + bf16hi(ub.w);
      *reinterpret_cast<uint4*>(yp + c * 8) =
          uint4{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]), pack2_bf16(o[4], o[5]), pack2_bf16(o[6], o[7])};
    }
  }
}

// Long rows (C >= 2048: the tokenizer's hidden size), few of them (2048 SVR tokens, 256 TTA queries): one WORKGROUP per row -- four waves
// share the row's loads (2 KB in flight per wave instead of 8, four times the waves), the two reductions cross the waves through LDS.
// (option ln_wide, default 1; 0: one wave per row like the narrow rows)
template <int NC>  // chunks (8 elements) per THREAD: C <= NC * 2048
__global__ __launch_bounds__(256) void layernorm_wide_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                             const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
                                                             bf16_t* __restrict__ y, int nb, int rows, int C,
                                                             int64_t x_bs, int64_t x_ld, int64_t res_bs, int64_t res_ld,
                                                             int64_t y_bs, int64_t y_ld, float eps) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  const int b = (int)(row / rows), r = (int)(row - (int64_t)b * rows);
  const bf16_t* xp = x + b * x_bs + r * x_ld;
  const bf16_t* rp = res ? res + b * res_bs + r * res_ld : nullptr;
  bf16_t* yp = y + b * y_bs + r * y_ld;
  const int nchunk = C >> 3;
  float v[NC][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = i * 256 + tid;
    if (c < nchunk) {
      const uint4 u = *reinterpret_cast<const uint4*>(xp + c * 8);
      v[i][0] = bf16lo(u.x); v[i][1] = bf16hi(u.x); v[i][2] = bf16lo(u.y); v[i][3] = bf16hi(u.y);
      v[i][4] = bf16lo(u.z); v[i][5] = bf16hi(u.z); v[i][6] = bf16lo(u.w); v[i][7] = bf16hi(u.w);
      if (rp) {
        const uint4 q = *reinterpret_cast<const uint4*>(rp + c * 8);
        v[i][0] += bf16lo(q.x); v[i][1] += bf16hi(q.x); v[i][2] += bf16lo(q.y); v[i][3] += bf16hi(q.y);
        v[i][4] += bf16lo(q.z); v[i][5] += bf16hi(q.z); v[i][6] += bf16lo(q.w); v[i][7] += bf16hi(q.w);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) red[0][wave] = sum;
  __syncthreads();
  const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    if (i * 256 + tid < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float dlt = v[i][j] - mean; sq += dlt * dlt; }
    }
  }
  sq = wave_sum(sq);
  if (lane == 0) red[1][wave] = sq;
  __syncthreads();
  const float rstd = rsqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = i * 256 + tid;
    if (c < nchunk) {
      const uint4 uw = *reinterpret_cast<const uint4*>(w + c * 8);
      const uint4 ub = *reinterpret_cast<const uint4*>(bias + c * 8);
      float o[8];
      o[0] = (v[i][0] - mean) * rstd * bf16lo(uw.x) + bf16lo(ub.x);
      o[1] = (v[i][1] - mean) * rstd * bf16hi(uw.x) + bf16hi(ub.x);
      o[2] = (v[i][2] - mean) * rstd * bf16lo(uw.y) + bf16lo(ub.y);
      o[3] = (v[i][3] - mean) * rstd * bf16hi(uw.y) + bf16hi(ub.y);
      o[4] = (v[i][4] - mean) * rstd * bf16lo(uw.z) + bf16lo(ub.z);
      o[5] = (v[i][5] - mean) * rstd * bf16hi(uw.z) + bf16hi(ub.z);
      o[6] = (v[i][6] - mean) * rstd * bf16lo(uw.w) + bf16lo(ub.w);
      o[7] = (v[i][7] - mean) * rstd * bf16hi(uw.w) + bf16hi(ub.w);
      *reinterpret_cast<uint4*>(yp + c * 8) =
          uint4{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]), pack2_bf16(o[4], o[5]), pack2_bf16(o[6], o[7])};
    }
  }
}

int layernorm_bf16(const bf16_t* x, const bf16_t* res, const bf16_t* w, const bf16_t* bias, bf16_t* y,
                   int nb, int rows, int C, int64_t x_bs, int64_t x_ld, int64_t res_bs, int64_t res_ld,
                   int64_t y_bs, int64_t y_ld, float eps, hipStream_t stream) {
  if (!x || !w || !bias || !y || nb <= 0 || rows <= 0 || C <= 0) return U2_ERR_ARG;
  if ((C & 7) || (x_ld & 7) || (y_ld & 7) || (x_bs & 7) || (y_bs & 7) || C > 8192) return U2_ERR_ARG;
  if (res && ((res_ld & 7) || (res_bs & 7))) return U2_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)res) & 15) return U2_ERR_ARG;
  const int64_t total = (int64_t)nb * rows;
  dim3 grid((unsigned)cdiv(total, 4));
  ProfScope ps(PROF_ROWOP, 0, stream, (double)total * C * 2.0 * (res ? 3.0 : 2.0));
  if (opts().ln_wide && C >= 2048 && total <= 65535 * 4) {   // long rows: a workgroup per row
    dim3 gw((unsigned)total);
#define U2_LNW(NC)                                                                                              \
  hipLaunchKernelGGL((layernorm_wide_kernel<NC>), gw, dim3(256), 0, stream, x, res, w, bias, y, nb, rows, C,    \
                     x_bs, x_ld, res_bs, res_ld, y_bs, y_ld, eps)
    if (C <= 2048) U2_LNW(1);
    else if (C <= 4096) U2_LNW(2);
    else U2_LNW(4);
#undef U2_LNW
    return launch_status();
  }
#define U2_LN(NC)                                                                                          \
  hipLaunchKernelGGL((layernorm_kernel<NC>), grid, dim3(256), 0, stream, x, res, w, bias, y, nb, rows, C,  \
                     x_bs, x_ld, res_bs, res_ld, y_bs, y_ld, eps)
  if (C <= 1024) U2_LN(2);
  else if (C <= 2048) U2_LN(4);
  else if (C <= 4096) U2_LN(8);
  else U2_LN(16);
#undef U2_LN
  return launch_status();
}

// ---------------------------------------------------------------- row softmax
// Reference: rma.py:60-72 (scores / sqrt(depth) + relative_bias[j - i + max_len - 1][head], softmax(-1)),
// tta.py:55-57 (no bias), svr.py:108 (DiffTS softmax over tokens with temperature).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, bf16_t* __restrict__ P,
                                                           int nz, int rows, int n, int64_t lds_, int64_t ldp,
                                                           int64_t s_zs, int64_t p_zs, float scale,
                                                           const bf16_t* __restrict__ rel_bias, int H, int max_len) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)nz * rows) return;
  const int z = (int)(row / rows), r = (int)(row - (int64_t)z * rows);
  const float* sp = S + z * s_zs + r * lds_;
  bf16_t* pp = P + z * p_zs + r * ldp;
  const bf16_t* bp = rel_bias ? rel_bias + (int64_t)(max_len - 1 - r) * H + (z % H) : nullptr;
  float m = -INFINITY;
  for (int c = lane; c < n; c += 64) {
    float s = sp[c] * scale;
    if (bp) s += bf16_to_f32(bp[(int64_t)c * H]);
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < n; c += 64) {
    float s = sp[c] * scale;
    if (bp) s += bf16_to_f32(bp[(int64_t)c * H]);
    sum += __expf(s - m);
  }
  const float inv = 1.f / wave_sum(sum);
  for (int c = lane; c < (int)ldp; c += 64) {
    float p = 0.f;
    if (c < n) {
      float s = sp[c] * scale;
      if (bp) s += bf16_to_f32(bp[(int64_t)c * H]);
      p = __expf(s - m) * inv;
    }
    pp[c] = f32_to_bf16(p);
  }
}

// Vector form for the shapes of the path (n <= 256 * NCH, rows 16-byte aligned): one wave per row, a lane owns 4
// consecutive columns of every 256-column chunk; S is read ONCE (float4), the row lives in registers, P is written as
// 8-byte bf16x4.  (The scalar kernel above re-reads S three times with 4-byte loads: 1.5 TB/s on a 25 MB problem.)
template <int NCH>
__global__ __launch_bounds__(256) void softmax_rows_vec_kernel(const float* __restrict__ S, bf16_t* __restrict__ P, int nz,
                                                               int rows, int n, int64_t lds_, int64_t ldp, int64_t s_zs,
                                                               int64_t p_zs, float scale,
                                                               const bf16_t* __restrict__ rel_bias, int H, int max_len) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)nz * rows) return;
  const int z = (int)(row / rows), r = (int)(row - (int64_t)z * rows);
  const float* sp = S + z * s_zs + r * lds_;
  bf16_t* pp = P + z * p_zs + r * ldp;
  const bf16_t* bp = rel_bias ? rel_bias + (int64_t)(max_len - 1 - r) * H + (z % H) : nullptr;
  float v[NCH][4];
  float m = -INFINITY;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * 256 + lane * 4;
    float4 x = float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (c0 + 3 < n) {
      x = *reinterpret_cast<const float4*>(sp + c0);
    } else if (c0 < n) {  // n % 4 != 0 cannot happen here (launcher), kept for safety
      x.x = sp[c0];
      if (c0 + 1 < n) x.y = sp[c0 + 1];
      if (c0 + 2 < n) x.z = sp[c0 + 2];
    }
    v[ch][0] = x.x * scale; v[ch][1] = x.y * scale; v[ch][2] = x.z * scale; v[ch][3] = x.w * scale;
    if (bp) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + e < n) v[ch][e] += bf16_to_f32(bp[(int64_t)(c0 + e) * H]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) m = fmaxf(m, v[ch][e]);
  }
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[ch][e] = __expf(v[ch][e] - m);  // exp(-inf) = 0 for the columns past n
      sum += v[ch][e];
    }
  const float inv = 1.f / wave_sum(sum);
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * 256 + lane * 4;
    if (c0 < (int)ldp)  // ldp % 4 == 0 (launcher); columns in [n, ldp) get exact zeros
      *reinterpret_cast<uint2*>(pp + c0) = uint2{pack2_bf16(v[ch][0] * inv, v[ch][1] * inv),
                                                 pack2_bf16(v[ch][2] * inv, v[ch][3] * inv)};
  }
}

int softmax_rows(const float* S, bf16_t* P, int nz, int rows, int n, int64_t lds_, int64_t ldp, int64_t s_zs,
                 int64_t p_zs, float scale, const bf16_t* rel_bias, int H, int max_len, hipStream_t stream) {
  if (!S || !P || nz <= 0 || rows <= 0 || n <= 0 || ldp < n || lds_ < n) return U2_ERR_ARG;
  if (rel_bias && (H <= 0 || n > max_len || rows > max_len)) return U2_ERR_ARG;  // rma.py: seq_len <= max_seq_len
  const int64_t total = (int64_t)nz * rows;
  ProfScope ps(PROF_ROWOP, 0, stream, (double)total * (4.0 * n + 2.0 * ldp));
  const bool vec = !(n & 3) && !(lds_ & 3) && !(ldp & 3) && !(s_zs & 3) && !(p_zs & 3) && !((uintptr_t)S & 15) &&
                   !((uintptr_t)P & 7) && ldp <= 2048;
#define U2_SMV(NCH)                                                                                              \
  hipLaunchKernelGGL((softmax_rows_vec_kernel<NCH>), dim3((unsigned)cdiv(total, 4)), dim3(256), 0, stream, S, P, nz, \
                     rows, n, lds_, ldp, s_zs, p_zs, scale, rel_bias, H > 0 ? H : 1, max_len)
  if (vec && ldp <= 256) U2_SMV(1);
  else if (vec && ldp <= 512) U2_SMV(2);
  else if (vec && ldp <= 1024) U2_SMV(4);
  else if (vec) U2_SMV(8);
  else
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv(total, 4)), dim3(256), 0, stream, S, P, nz, rows, n,
                       lds_, ldp, s_zs, p_zs, scale, rel_bias, H > 0 ? H : 1, max_len);
#undef U2_SMV
  return launch_status();
}

// ---------------------------------------------------------------- transpose
// 64x64 bf16 tiles through LDS (padded rows -> conflict-free column reads), coalesced both sides.
// perm16: inside every group of 16 output columns the order becomes [0-3, 8-11, 4-7, 12-15] -- the k-slot order
// of the flash-attention P fragments (attn.hip), so V^T fragments are single 16-byte LDS reads.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R,
                                                        int C, int64_t ld_in, int64_t ld_out, int64_t in_zs,
                                                        int64_t out_zs, int perm16) {
  __shared__ bf16_t tile[64][66];
  const int z = blockIdx.z;
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const bf16_t* ip = in + z * in_zs;
  bf16_t* op = out + z * out_zs;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + i * 4, c = c0 + tx;
    tile[ty + i * 4][tx] = (r < R && c < C) ? ip[(int64_t)r * ld_in + c] : (bf16_t)0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + i * 4, r = r0 + tx;
    int rs = tx;  // source row (inside the tile) that lands in output column r
    if (perm16) {
      const int qd = (tx >> 2) & 3;
      rs = (tx & ~12) | ((qd == 1 ? 2 : (qd == 2 ? 1 : qd)) << 2);
    }
    if (c < C && r < ld_out) op[(int64_t)c * ld_out + r] = tile[rs][ty + i * 4];
  }
}

// Vector form (all strides multiples of 4 elements, 8-byte aligned bases): 8-byte global loads and stores, 64x64 tile.
__global__ __launch_bounds__(256) void transpose_vec_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R,
                                                            int C, int64_t ld_in, int64_t ld_out, int64_t in_zs,
                                                            int64_t out_zs, int perm16) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][68];  // 136-byte rows: 8-byte aligned, odd multiple of 8 B
  const int z = blockIdx.z;
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const bf16_t* ip = in + z * in_zs;
  bf16_t* op = out + z * out_zs;
  const int q = threadIdx.x & 15, w = threadIdx.x >> 4;  // q: group of 4 columns, w: row 0..15 (+16 i)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + w + i * 16, c = c0 + q * 4;
    uint2 v = {0u, 0u};
    if (r < R) {
      if (c + 3 < C) {
        v = *reinterpret_cast<const uint2*>(ip + (int64_t)r * ld_in + c);
      } else if (c < C) {  // ragged right edge (C % 4 != 0 never gets here: launcher)
        const bf16_t* p = ip + (int64_t)r * ld_in + c;
        const uint32_t e0 = p[0], e1 = c + 1 < C ? p[1] : 0u, e2 = c + 2 < C ? p[2] : 0u;
        v = uint2{e0 | (e1 << 16), e2};
      }
    }
    *reinterpret_cast<uint2*>(&tile[w + i * 16][q * 4]) = v;
  }
  __syncthreads();
  // output row = input column cc, output columns = 4 consecutive input rows (the source quad is permuted for perm16)
  int sq = q;
  if (perm16) {
    const int qd = q & 3;
    sq = (q & ~3) | (qd == 1 ? 2 : (qd == 2 ? 1 : qd));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cc = w + i * 16;
    const int c = c0 + cc, r = r0 + q * 4;
    if (c < C && r < ld_out) {
      const uint32_t e0 = tile[sq * 4 + 0][cc], e1 = tile[sq * 4 + 1][cc], e2 = tile[sq * 4 + 2][cc],
                     e3 = tile[sq * 4 + 3][cc];
      *reinterpret_cast<uint2*>(op + (int64_t)c * ld_out + r) = uint2{e0 | (e1 << 16), e2 | (e3 << 16)};
    }
  }
}

int transpose_bf16(const bf16_t* in, bf16_t* out, int nz, int R, int C, int64_t ld_in, int64_t ld_out,
                   int64_t in_zs, int64_t out_zs, int perm16, hipStream_t stream) {
  if (!in || !out || nz <= 0 || nz > 65535 || R <= 0 || C <= 0 || ld_in < C || ld_out < R) return U2_ERR_ARG;
  if (perm16 && (ld_out & 15)) return U2_ERR_ARG;
  dim3 grid((unsigned)cdiv(ld_out, 64), (unsigned)cdiv(C, 64), nz);
  if (grid.y > 65535) return U2_ERR_ARG;
  ProfScope ps(PROF_MOVE, 0, stream, (double)nz * C * (2.0 * R + 2.0 * ld_out));
  const bool vec = !(C & 3) && !(ld_in & 3) && !(ld_out & 3) && !(in_zs & 3) && !(out_zs & 3) &&
                   !(((uintptr_t)in | (uintptr_t)out) & 7);
  if (vec)
    hipLaunchKernelGGL(transpose_vec_kernel, grid, dim3(256), 0, stream, in, out, R, C, ld_in, ld_out, in_zs, out_zs,
                       perm16);
  else
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, in, out, R, C, ld_in, ld_out, in_zs, out_zs,
                       perm16);
  return launch_status();
}

}  // namespace u2
