// Backward-pass kernels of the path: everything that is not GEMM-shaped.  The GEMM-shaped parts
// of the backward (dX = dY W, dW = dY^T X, dQ / dK / dV / dP of the attention cores) run on the forward's MFMA kernels
// (gemm.hip / gemm_bt.hip) through u2tok_gemm_bf16; the host side (u2tokenizer_amd/autograd.py) sequences them.
//
//   gelu_fwd / gelu_bwd        MONAI MLPBlock GELU (vit.py:100-105) and the SPP MLP (spatial_pooling_projector.py:22-28)
//   colsum_partial / _finish   bias gradients, LayerNorm weight / bias gradients (deterministic two-stage column sums)
//   layernorm_bwd              nn.LayerNorm (+ residual) of the ViT blocks and the TTA post-norms (tta.py:96,100,103)
//   softmax_bwd                dS = P * (dP - rowsum(P * dP)) for rma.py:72 / tta.py:57 / svr.py:108
//   relbias_grad               gradient of the Toeplitz relative-bias table (rma.py:64-70)
//   rowdot                     D_i = dO_i . O_i
// All HBM-bound: one wave per row or 16-byte vectors per thread, fp32 arithmetic, bf16 storage.
#include "kernels.h"

namespace u2 {

namespace {

__device__ __forceinline__ void unpack8(const uint4 u, float (&v)[8]) {
  v[0] = bf16lo(u.x); v[1] = bf16hi(u.x); v[2] = bf16lo(u.y); v[3] = bf16hi(u.y);
  v[4] = bf16lo(u.z); v[5] = bf16hi(u.z); v[6] = bf16lo(u.w); v[7] = bf16hi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  return uint4{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
}

// d/dz [ 0.5 z (1 + erf(z / sqrt 2)) ] = Phi(z) + z phi(z)
__device__ __forceinline__ float gelu_grad(float z) {
  const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);
  return cdf + z * pdf;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ GELU
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ z, bf16_t* __restrict__ y, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  float v[8];
  unpack8(*reinterpret_cast<const uint4*>(z + i * 8), v);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = gelu_fast(v[j]);
  *reinterpret_cast<uint4*>(y + i * 8) = pack8(v);
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ z, const bf16_t* __restrict__ dy,
                                                       bf16_t* __restrict__ dz, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  float v[8], g[8];
  unpack8(*reinterpret_cast<const uint4*>(z + i * 8), v);
  unpack8(*reinterpret_cast<const uint4*>(dy + i * 8), g);
#pragma unroll
  for (int j = 0; j < 8; ++j) g[j] *= gelu_grad(v[j]);
  *reinterpret_cast<uint4*>(dz + i * 8) = pack8(g);
}

int gelu_fwd(const bf16_t* z, bf16_t* y, int64_t n, hipStream_t st) {
  if (!z || !y || n <= 0 || (n & 7) || (((uintptr_t)z | (uintptr_t)y) & 15)) return U2_ERR_ARG;
  ProfScope ps(PROF_ROWOP, 0, st, 4.0 * n);
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)cdiv(n / 8, 256)), dim3(256), 0, st, z, y, n / 8);
  return launch_status();
}
int gelu_bwd(const bf16_t* z, const bf16_t* dy, bf16_t* dz, int64_t n, hipStream_t st) {
  if (!z || !dy || !dz || n <= 0 || (n & 7) || (((uintptr_t)z | (uintptr_t)dy | (uintptr_t)dz) & 15)) return U2_ERR_ARG;
  ProfScope ps(PROF_ROWOP, 0, st, 6.0 * n);
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)cdiv(n / 8, 256)), dim3(256), 0, st, z, dy, dz, n / 8);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------ column sums
// out[c] = sum_r x[r][c] (optionally of x[r][c] * y[r][c]) in fp32, in a fixed order: slab s sums rows
// [s * R, (s + 1) * R) into part[s][c]; the finish kernel adds the slabs in a fixed order.  R is chosen per call so that
// ~1000 workgroups are in flight (16392 x 768: 37 + 50 us with 128-row slabs and a serial finish -> see DESIGN section 7).
static int colsum_slab_rows(int rows, int C) {
  const int64_t groups = cdiv(C, 512);
  int64_t R = (int64_t)rows * groups / 1024;
  R = (R / 4) * 4;
  return (int)std::max<int64_t>(4, std::min<int64_t>(128, R));
}

// Vector form (C % 8 == 0, 16-byte aligned rows): a lane owns 8 columns, wave w of the workgroup walks rows r0 + w, + 4, ...
// of the slab with four rows in flight; the four waves combine through LDS in a fixed order.
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                                                 float* __restrict__ part, int rows, int C, int64_t ldx,
                                                                 int64_t ldy, int R) {
  __shared__ float red[3][512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int chunk = blockIdx.x * 64 + lane;
  const bool on = chunk * 8 < C;
  const int r0 = blockIdx.y * R, r1 = min(rows, r0 + R);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (on) {
    const bf16_t* xp = x + chunk * 8;
    const bf16_t* yp = y ? y + chunk * 8 : nullptr;
#pragma unroll 4
    for (int r = r0 + wv; r < r1; r += 4) {
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(xp + (int64_t)r * ldx), v);
      if (yp) {
        float q[8];
        unpack8(*reinterpret_cast<const uint4*>(yp + (int64_t)r * ldy), q);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= q[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
  }
  if (wv > 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wv - 1][lane * 8 + j] = a[j];
  }
  __syncthreads();
  if (wv == 0 && on) {
    float* p = part + (int64_t)blockIdx.y * C + chunk * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = ((a[j] + red[0][lane * 8 + j]) + red[1][lane * 8 + j]) + red[2][lane * 8 + j];
  }
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                                             float* __restrict__ part, int rows, int C, int64_t ldx,
                                                             int64_t ldy, int R) {
  const int c2 = blockIdx.x * 256 + threadIdx.x;  // column pair
  if (c2 * 2 >= C) return;
  const int r0 = blockIdx.y * R, r1 = min(rows, r0 + R);
  float a0 = 0.f, a1 = 0.f;
  for (int r = r0; r < r1; ++r) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(x + (int64_t)r * ldx + c2 * 2);
    float v0 = bf16lo(u), v1 = bf16hi(u);
    if (y) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(y + (int64_t)r * ldy + c2 * 2);
      v0 *= bf16lo(w);
      v1 *= bf16hi(w);
    }
    a0 += v0;
    a1 += v1;
  }
  float* p = part + (int64_t)blockIdx.y * C + c2 * 2;
  p[0] = a0;
  p[1] = a1;
}

// out[c] (+)= sum_s part[s][c]; out_bf16 (optional) receives the rounded result as well.  A workgroup owns 32 columns;
// its 8 thread groups add slabs g, g + 8, ... (two interleaved chains each) and the partial sums are added in a fixed order
// (bit-repeatable).  blockIdx.y selects one of up to two problems of the same shape (dw and db of a LayerNorm backward).
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part_a, float* __restrict__ out_a,
                                                            const float* __restrict__ part_b, float* __restrict__ out_b,
                                                            bf16_t* __restrict__ out_bf16, int nslab, int C, int accumulate) {
  __shared__ float red[8][32];
  const float* __restrict__ part = blockIdx.y ? part_b : part_a;
  float* __restrict__ out = blockIdx.y ? out_b : out_a;
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    int s = g;
    for (; s + 8 < nslab; s += 16) {
      a0 += part[(int64_t)s * C + c];
      a1 += part[(int64_t)(s + 8) * C + c];
    }
    if (s < nslab) a0 += part[(int64_t)s * C + c];
  }
  red[g][cl] = a0 + a1;
  __syncthreads();
  if (g == 0 && c < C) {
    float t = red[0][cl];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += red[i][cl];
    if (accumulate) t += out[c];
    if (out) out[c] = t;
    if (out_bf16 && blockIdx.y == 0) out_bf16[c] = f32_to_bf16(t);
  }
}

static void colsum_finish(const float* part, float* out, bf16_t* out_bf16, int nslab, int C, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)cdiv(C, 32), 1), dim3(256), 0, st, part, out,
                     (const float*)nullptr, (float*)nullptr, out_bf16, nslab, C, accumulate);
}
static void colsum_finish2(const float* part_a, float* out_a, const float* part_b, float* out_b, int nslab, int C,
                           int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)cdiv(C, 32), 2), dim3(256), 0, st, part_a, out_a, part_b, out_b,
                     (bf16_t*)nullptr, nslab, C, accumulate);
}

size_t colsum_workspace_bytes(int rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)cdiv(rows, colsum_slab_rows(rows, C)) * C * sizeof(float);
}

int colsum_bf16(const bf16_t* x, const bf16_t* y, float* out, bf16_t* out_bf16, int rows, int C, int64_t ldx, int64_t ldy,
                float* ws, int accumulate, hipStream_t st) {
  if (!x || (!out && !out_bf16) || !ws || rows <= 0 || C <= 0 || (C & 1) || (ldx & 1) || (y && (ldy & 1))) return U2_ERR_ARG;
  if ((((uintptr_t)x | (uintptr_t)y) & 3) || (accumulate && !out)) return U2_ERR_ARG;
  const int R = colsum_slab_rows(rows, C);
  const int nslab = (int)cdiv(rows, R);
  if (nslab > 65535) return U2_ERR_ARG;
  ProfScope ps(PROF_ROWOP, 0, st, (double)rows * C * (y ? 4.0 : 2.0));
  const bool vec = !(C & 7) && !(ldx & 7) && !(y && (ldy & 7)) && !(((uintptr_t)x | (uintptr_t)y) & 15);
  if (vec)
    hipLaunchKernelGGL(colsum_partial_vec_kernel, dim3((unsigned)cdiv(C, 512), nslab), dim3(256), 0, st, x, y, ws, rows, C,
                       ldx, ldy, R);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)cdiv(C / 2, 256), nslab), dim3(256), 0, st, x, y, ws, rows, C,
                       ldx, ldy, R);
  colsum_finish(ws, out, out_bf16, nslab, C, accumulate, st);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// v = x (+ res);  xhat = (v - mean) rstd;  y = xhat w + b.   g = dy w;
//   dv = rstd (g - mean_c(g) - xhat mean_c(g xhat));   dw += dy xhat;   db += dy   (per-slab partial sums, fp32)
// One wave per row, the row in registers (as the forward kernel); a workgroup walks 4 x rpw rows so that each lane can
// keep its columns' dw / db partials in registers; waves combine through LDS, one partial row per workgroup.  Rows per wave
// (1..16) are chosen per call so that >= ~512 workgroups are in flight (2048 x 4096 with 64 rows per workgroup = 32
// workgroups: 205 us, latency-bound).
static int lnb_rows_per_wave(int rows) { return std::max(1, std::min(16, rows / 2048)); }

template <int NC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                            const bf16_t* __restrict__ w, const bf16_t* __restrict__ dy,
                                                            bf16_t* __restrict__ dv, float* __restrict__ part_w,
                                                            float* __restrict__ part_b, int rows, int C, float eps, int rpw) {
  __shared__ float red[3][NC * 512];  // waves 1..3 -> wave 0
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nchunk = C >> 3;
  float aw[NC][8], ab[NC][8];
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { aw[i][j] = 0.f; ab[i][j] = 0.f; }
  float wr[NC][8];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = i * 64 + lane;
    if (c < nchunk) unpack8(*reinterpret_cast<const uint4*>(w + c * 8), wr[i]);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) wr[i][j] = 0.f;
    }
  }
  const int r_begin = (blockIdx.x * 4 + wv) * rpw;
  for (int rr = 0; rr < rpw; ++rr) {
    const int r = r_begin + rr;
    if (r >= rows) break;  // wave-uniform
    const bf16_t* xp = x + (int64_t)r * C;
    const bf16_t* rp = res ? res + (int64_t)r * C : nullptr;
    const bf16_t* gp = dy + (int64_t)r * C;
    float v[NC][8], g[NC][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = i * 64 + lane;
      if (c < nchunk) {
        unpack8(*reinterpret_cast<const uint4*>(xp + c * 8), v[i]);
        if (rp) {
          float q[8];
          unpack8(*reinterpret_cast<const uint4*>(rp + c * 8), q);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][j] += q[j];
        }
        unpack8(*reinterpret_cast<const uint4*>(gp + c * 8), g[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[i][j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[i][j] = 0.f; g[i][j] = 0.f; }
      }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (i * 64 + lane < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
      }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;  // sum_c g, sum_c g xhat  (g = dy w)
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (i * 64 + lane < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (v[i][j] - mean) * rstd;
          v[i][j] = xh;
          aw[i][j] += g[i][j] * xh;
          ab[i][j] += g[i][j];
          g[i][j] *= wr[i][j];
          s1 += g[i][j];
          s2 += g[i][j] * xh;
        }
      }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = i * 64 + lane;
      if (c < nchunk) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (g[i][j] - s1 - v[i][j] * s2);
        *reinterpret_cast<uint4*>(dv + (int64_t)r * C + c * 8) = pack8(o);
      }
    }
  }
  // partial dw / db of this workgroup: waves 1..3 hand theirs to wave 0 through LDS, fixed order
  for (int pass = 0; pass < 2; ++pass) {
    if (wv > 0) {
#pragma unroll
      for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wv - 1][(i * 64 + lane) * 8 + j] = pass ? ab[i][j] : aw[i][j];
    }
    __syncthreads();
    if (wv == 0) {
      float* dst = (pass ? part_b : part_w) + (int64_t)blockIdx.x * C;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = i * 64 + lane;
        if (c < nchunk) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float a = pass ? ab[i][j] : aw[i][j];
            a += red[0][c * 8 + j];
            a += red[1][c * 8 + j];
            a += red[2][c * 8 + j];
            dst[c * 8 + j] = a;
          }
        }
      }
    }
    __syncthreads();
  }
}

size_t layernorm_bwd_workspace_bytes(int rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)2 * cdiv(rows, 4 * lnb_rows_per_wave(rows)) * C * sizeof(float);
}

// dv: gradient w.r.t. x (and, identically, w.r.t. res); dw / db (fp32, C each) are overwritten unless accumulate.
int layernorm_bwd(const bf16_t* x, const bf16_t* res, const bf16_t* w, const bf16_t* dy, bf16_t* dv, float* dw, float* db,
                  int rows, int C, float eps, float* ws, int accumulate, hipStream_t st) {
  if (!x || !w || !dy || !dv || !dw || !db || !ws || rows <= 0 || C <= 0 || (C & 7) || C > 4096) return U2_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)res | (uintptr_t)w | (uintptr_t)dy | (uintptr_t)dv) & 15) return U2_ERR_ARG;
  const int rpw = lnb_rows_per_wave(rows);
  const int nwg = (int)cdiv(rows, 4 * rpw);
  float* pw = ws;
  float* pb = ws + (size_t)nwg * C;
  ProfScope ps(PROF_ROWOP, 0, st, (double)rows * C * 2.0 * (res ? 4.0 : 3.0));
#define U2_LNB(NC)                                                                                                   \
  hipLaunchKernelGGL((layernorm_bwd_kernel<NC>), dim3(nwg), dim3(256), 0, st, x, res, w, dy, dv, pw, pb, rows, C, eps, rpw)
  if (C <= 1024) U2_LNB(2);
  else if (C <= 2048) U2_LNB(4);
  else U2_LNB(8);
#undef U2_LNB
  colsum_finish2(pw, dw, pb, db, nwg, C, accumulate, st);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------ softmax backward
// dS[z][r][c] = P[z][r][c] * (dP[z][r][c] - sum_c' P[z][r][c'] dP[z][r][c']);  columns [n, ldp) of dS are zeroed.
// (Gradient w.r.t. the logits s = raw * scale + bias: d raw = dS * scale is folded into the following GEMMs' alpha,
// d bias = dS.)  One wave per row.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ P, const float* __restrict__ dP,
                                                          bf16_t* __restrict__ dS, int64_t nrows, int n, int64_t ldp,
                                                          int64_t lddp) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const bf16_t* pp = P + row * ldp;
  const float* gp = dP + row * lddp;
  bf16_t* op = dS + row * ldp;
  float dot = 0.f;
  for (int c = lane; c < n; c += 64) dot += bf16_to_f32(pp[c]) * gp[c];
  dot = wave_sum(dot);
  for (int c = lane; c < (int)ldp; c += 64) {
    float v = 0.f;
    if (c < n) v = bf16_to_f32(pp[c]) * (gp[c] - dot);
    op[c] = f32_to_bf16(v);
  }
}

int softmax_bwd(const bf16_t* P, const float* dP, bf16_t* dS, int64_t nrows, int n, int64_t ldp, int64_t lddp,
                hipStream_t st) {
  if (!P || !dP || !dS || nrows <= 0 || n <= 0 || ldp < n || lddp < n) return U2_ERR_ARG;
  if (cdiv(nrows, 4) > 0x7fffffff) return U2_ERR_ARG;
  ProfScope ps(PROF_ROWOP, 0, st, (double)nrows * n * 8.0);
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)cdiv(nrows, 4)), dim3(256), 0, st, P, dP, dS, nrows, n, ldp, lddp);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------ relative-bias gradient
// bias[j - i + L - 1][h] is added to the logits of every (z, i, j) with z % H == h (rma.py:64-70):
//   dtable[d + L - 1][h] = sum_{z % H == h} sum_{i, j = i + d in range} dS[z][i][j]
// One workgroup per (diagonal d, head h), fixed summation order (threads stride over (zb, i), tree in LDS).
__global__ __launch_bounds__(256) void relbias_grad_kernel(const bf16_t* __restrict__ dS, float* __restrict__ dtable, int nz,
                                                           int S, int H, int64_t ldp, int max_len) {
  __shared__ float red[256];
  const int d = (int)blockIdx.x - (S - 1), h = blockIdx.y;
  const int i0 = d < 0 ? -d : 0, cnt = S - (d < 0 ? -d : d);  // i in [i0, i0 + cnt), j = i + d
  const int nzb = nz / H;
  float a = 0.f;
  for (int t = threadIdx.x; t < nzb * cnt; t += 256) {
    const int zb = t / cnt, i = i0 + (t - zb * cnt);
    a += bf16_to_f32(dS[((int64_t)(zb * H + h) * S + i) * ldp + (i + d)]);
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) dtable[(int64_t)(d + max_len - 1) * H + h] += red[0];
}

// dtable: fp32 [2 max_len - 1][H], ACCUMULATED into (the caller zeroes it once per backward pass)
int relbias_grad(const bf16_t* dS, float* dtable, int nz, int S, int H, int64_t ldp, int max_len, hipStream_t st) {
  if (!dS || !dtable || nz <= 0 || S <= 0 || H <= 0 || nz % H || S > max_len || ldp < S) return U2_ERR_ARG;
  ProfScope ps(PROF_ROWOP, 0, st, (double)nz * S * S * 2.0);
  hipLaunchKernelGGL(relbias_grad_kernel, dim3(2 * S - 1, H), dim3(256), 0, st, dS, dtable, nz, S, H, ldp, max_len);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------ row dot
// out[r] = sum_c a[r][c] * b[r][c]   (fp32; one wave per row)
__global__ __launch_bounds__(256) void rowdot_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                     float* __restrict__ out, int64_t rows, int C, int64_t lda, int64_t ldb) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) acc += bf16_to_f32(a[row * lda + c]) * bf16_to_f32(b[row * ldb + c]);
  acc = wave_sum(acc);
  if (lane == 0) out[row] = acc;
}

int rowdot_bf16(const bf16_t* a, const bf16_t* b, float* out, int64_t rows, int C, int64_t lda, int64_t ldb, hipStream_t st) {
  if (!a || !b || !out || rows <= 0 || C <= 0 || cdiv(rows, 4) > 0x7fffffff) return U2_ERR_ARG;
  ProfScope ps(PROF_ROWOP, 0, st, (double)rows * C * 4.0);
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st, a, b, out, rows, C, lda, ldb);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------ sharded AdamW
// One rank's piece of a ZeRO-1 bucket (u2tokenizer_amd/dp.py; config/ds_config.json:27-41, optim adamw_torch): fp32 master
// weights + moments updated in place from the reduce-scattered bf16 gradient SUM, the bf16 rounding of the new master
// written for the all-gather.  One pass over 14 bytes in / 14 bytes out per element instead of ~10 elementwise torch
// kernels over fp32 temporaries.  torch.optim.AdamW arithmetic:
//   g = grad * gscale (* *gcoef: the clipping coefficient, computed on the device);  p *= 1 - lr wd;
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / c1) m / (sqrt(v) / sqrt(c2) + eps)
// group[i] (may be null: group 0) selects (lr, wd) from the per-group tables (param groups: biases / LayerNorm without decay).
__device__ __forceinline__ void adamw_one(float& p, float& mi, float& vi, float g, float lr, float wd, const AdamWArgs& a) {
  p *= 1.0f - lr * wd;
  mi = a.b1 * mi + (1.0f - a.b1) * g;
  vi = a.b2 * vi + (1.0f - a.b2) * g * g;
  p -= (lr * a.inv_c1) * mi / (sqrtf(vi) * a.inv_sqrt_c2 + a.eps);
}

// VEC: every pointer 16-byte aligned (group: 8-byte) -- a lane owns 8 consecutive elements: 2 x 16-byte loads / stores per fp32
// array, one 16-byte load of gradients, one 16-byte store of parameters (29 bytes per element each way at full-width accesses;
// the scalar form of round 3's first version ran at 2.2 TB/s).  Tail elements (n % 8) and unaligned pieces: one element per lane.
template <bool VEC>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    const bf16_t* __restrict__ grad, const uint8_t* __restrict__ group,
                                                    bf16_t* __restrict__ out, int64_t n, int64_t first, AdamWArgs a) {
  const float gs = a.gscale * (a.gcoef ? *a.gcoef : 1.0f);
  if constexpr (VEC) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= n) return;  // (n is a multiple of 8 here)
    float4 p0 = *reinterpret_cast<const float4*>(master + i0), p1 = *reinterpret_cast<const float4*>(master + i0 + 4);
    float4 m0 = *reinterpret_cast<const float4*>(m + i0), m1 = *reinterpret_cast<const float4*>(m + i0 + 4);
    float4 v0 = *reinterpret_cast<const float4*>(v + i0), v1 = *reinterpret_cast<const float4*>(v + i0 + 4);
    const uint4 g4 = *reinterpret_cast<const uint4*>(grad + i0);
    uint2 gi2 = uint2{0u, 0u};
    if (group) gi2 = *reinterpret_cast<const uint2*>(group + i0);
    float pp[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
    float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    const float gg[8] = {bf16lo(g4.x), bf16hi(g4.x), bf16lo(g4.y), bf16hi(g4.y), bf16lo(g4.z), bf16hi(g4.z), bf16lo(g4.w), bf16hi(g4.w)};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int gi = (int)(((r < 4 ? gi2.x : gi2.y) >> (8 * (r & 3))) & 255u);
      adamw_one(pp[r], mm[r], vv[r], gg[r] * gs, a.lr[gi], a.wd[gi], a);
    }
    *reinterpret_cast<float4*>(master + i0) = float4{pp[0], pp[1], pp[2], pp[3]};
    *reinterpret_cast<float4*>(master + i0 + 4) = float4{pp[4], pp[5], pp[6], pp[7]};
    *reinterpret_cast<float4*>(m + i0) = float4{mm[0], mm[1], mm[2], mm[3]};
    *reinterpret_cast<float4*>(m + i0 + 4) = float4{mm[4], mm[5], mm[6], mm[7]};
    *reinterpret_cast<float4*>(v + i0) = float4{vv[0], vv[1], vv[2], vv[3]};
    *reinterpret_cast<float4*>(v + i0 + 4) = float4{vv[4], vv[5], vv[6], vv[7]};
    *reinterpret_cast<uint4*>(out + i0) = uint4{pack2_bf16(pp[0], pp[1]), pack2_bf16(pp[2], pp[3]), pack2_bf16(pp[4], pp[5]),
                                                pack2_bf16(pp[6], pp[7])};
  } else {
    const int64_t i = first + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int gi = group ? group[i] : 0;
    float p = master[i], mi = m[i], vi = v[i];
    adamw_one(p, mi, vi, bf16_to_f32(grad[i]) * gs, a.lr[gi], a.wd[gi], a);
    m[i] = mi; v[i] = vi; master[i] = p;
    out[i] = f32_to_bf16(p);
  }
}

int adamw_step(float* master, float* m, float* v, const bf16_t* grad, const uint8_t* group, bf16_t* out, int64_t n,
               const AdamWArgs& a, hipStream_t st) {
  if (!master || !m || !v || !grad || !out || n <= 0 || cdiv(n, 256) > 0x7fffffff) return U2_ERR_ARG;
  ProfScope ps(PROF_ROWOP, 0, st, (double)n * 28.0);
  const bool aligned = !((((uintptr_t)master | (uintptr_t)m | (uintptr_t)v | (uintptr_t)grad | (uintptr_t)out) & 15) ||
                         ((uintptr_t)group & 7));
  const int64_t nv = aligned ? (n & ~(int64_t)7) : 0;
  if (nv)
    hipLaunchKernelGGL(adamw_kernel<true>, dim3((unsigned)cdiv(nv, 2048)), dim3(256), 0, st, master, m, v, grad, group, out, nv,
                       (int64_t)0, a);
  if (nv < n)
    hipLaunchKernelGGL(adamw_kernel<false>, dim3((unsigned)cdiv(n - nv, 256)), dim3(256), 0, st, master, m, v, grad, group, out,
                       n, nv, a);
  return launch_status();
}

}  // namespace u2
