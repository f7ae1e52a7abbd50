// Token selection and multi-scale pooling of SpatioTemporalVisualTokenRefinerModel (svr.py:153-188):
// hard top-k (TokenSelection, svr.py:64-91) with exactly reproducible scores and a canonical tie rule,
// row gather, and the {1,2,4} average pooling with optional DynamicMultiScalePooling gates
// (svr.py:119-151, 173-184).
#include "kernels.h"

namespace u2 {

// ---------------------------------------------------------------- score GEMV (E -> 1)
// score_net = nn.Linear(E, 1) (svr.py:67,78).  The top-k ORDER is an integer output of the path, so the
// scores must not depend on summation order: bf16 x bf16 products are exact in fp64 and the fp64 sum of
// E of them is (to ~2^-50) exact, hence the single rounding to fp32 is reproducible by any oracle that
// also sums in fp64.
__global__ __launch_bounds__(256) void score_gemv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                         const bf16_t* __restrict__ bias, float* __restrict__ scores,
                                                         int rows, int E) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xp = x + row * E;
  double acc = 0.0;
  for (int c = lane; c < (E >> 3); c += 64) {
    const uint4 a = *reinterpret_cast<const uint4*>(xp + c * 8);
    const uint4 b = *reinterpret_cast<const uint4*>(w + c * 8);
    acc += (double)bf16lo(a.x) * (double)bf16lo(b.x);
    acc += (double)bf16hi(a.x) * (double)bf16hi(b.x);
    acc += (double)bf16lo(a.y) * (double)bf16lo(b.y);
    acc += (double)bf16hi(a.y) * (double)bf16hi(b.y);
    acc += (double)bf16lo(a.z) * (double)bf16lo(b.z);
    acc += (double)bf16hi(a.z) * (double)bf16hi(b.z);
    acc += (double)bf16lo(a.w) * (double)bf16lo(b.w);
    acc += (double)bf16hi(a.w) * (double)bf16hi(b.w);
  }
  acc = wave_sum_f64(acc);
  if (lane == 0) scores[row] = (float)(acc + (bias ? (double)bf16_to_f32(bias[0]) : 0.0));
}

int score_gemv(const bf16_t* x, const bf16_t* w, const bf16_t* bias, float* scores, int rows, int E,
               hipStream_t stream) {
  if (!x || !w || !scores || rows <= 0 || E <= 0 || (E & 7)) return U2_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)w) & 15) return U2_ERR_ARG;
  ProfScope ps(PROF_ROWOP, 0, stream);
  hipLaunchKernelGGL(score_gemv_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream, x, w, bias, scores, rows,
                     E);
  return launch_status();
}

// ---------------------------------------------------------------- top-k, sorted, canonical ties
// torch.topk(scores, k, dim=1) (svr.py:82) returns values sorted descending; its order among EQUAL
// scores is unspecified.  Canonical rule here (and in oracle/): descending score, ties by ascending
// index (== torch.sort(stable=True, descending=True)); -0.0 == +0.0.  One workgroup bitonic-sorts the
// 64-bit keys (orderable score bits << 32 | ~index) of one batch row in LDS.
__device__ __forceinline__ uint64_t topk_key(float f, uint32_t idx) {
  if (f == 0.f) f = 0.f;  // -0.0 -> +0.0
  uint32_t u = __float_as_uint(f);
  u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
  return ((uint64_t)u << 32) | (uint64_t)(0xffffffffu - idx);
}

__global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ scores, int64_t* __restrict__ idx, int n,
                                                    int k, int np2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  const int b = blockIdx.x;
  const float* sp = scores + (int64_t)b * n;
  for (int i = threadIdx.x; i < np2; i += blockDim.x) keys[i] = i < n ? topk_key(sp[i], (uint32_t)i) : 0ull;
  __syncthreads();
  for (int size = 2; size <= np2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (np2 >> 1); t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);  // final pass (size == np2): every pair sorts descending
        const uint64_t a = keys[lo], c = keys[hi];
        if ((a < c) == desc) { keys[lo] = c; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < k; i += blockDim.x)
    idx[(int64_t)b * k + i] = (int64_t)(0xffffffffu - (uint32_t)(keys[i] & 0xffffffffull));
}

int topk_sorted(const float* scores, int64_t* idx, int B, int n, int k, hipStream_t stream) {
  if (!scores || !idx || B <= 0 || n <= 0 || k <= 0 || k > n || n > 8192) return U2_ERR_ARG;
  int np2 = 2;
  while (np2 < n) np2 <<= 1;
  const int threads = np2 / 2 < 1024 ? (np2 / 2 < 64 ? 64 : np2 / 2) : 1024;
  ProfScope ps(PROF_ROWOP, 0, stream);
  hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(threads), (size_t)np2 * 8, stream, scores, idx, n, k, np2);
  return launch_status();
}

// ---------------------------------------------------------------- gather
// topk_tokens = x[arange(b)[:, None], idx // n, idx % n]  (svr.py:85-89) == rows of the flattened (t n) axis.
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ x, const int64_t* __restrict__ idx,
                                                          bf16_t* __restrict__ out, int B, int n, int k, int E) {
  const int e8n = E >> 3;
  const int64_t total = (int64_t)B * k * e8n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int e8 = (int)(i % e8n);
    const int64_t bk = i / e8n;
    const int b = (int)(bk / k);
    int64_t src = idx[bk];
    src = src < 0 ? 0 : (src >= n ? n - 1 : src);
    *reinterpret_cast<uint4*>(out + bk * E + e8 * 8) =
        *reinterpret_cast<const uint4*>(x + ((int64_t)b * n + src) * E + e8 * 8);
  }
}

int gather_rows(const bf16_t* x, const int64_t* idx, bf16_t* out, int B, int n, int k, int E, hipStream_t stream) {
  if (!x || !idx || !out || B <= 0 || n <= 0 || k <= 0 || (E & 7)) return U2_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)out) & 15) return U2_ERR_ARG;
  const int64_t total = (int64_t)B * k * (E >> 3);
  const unsigned blocks = (unsigned)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
  ProfScope ps(PROF_MOVE, 0, stream);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, stream, x, idx, out, B, n, k, E);
  return launch_status();
}

// ---------------------------------------------------------------- multi-scale pooling (+ DMTP gates)
// gate_s = gate_fc(mean_tokens(avg_pool1d(x, s)))  (svr.py:133-138).  The token mean of the pooled block
// equals the mean of the first floor(k/s)*s tokens, accumulated here in fp32 per column.
// ws[(b*3 + s)*ncg + cg] = sum over the 256 columns of group cg of colmean_s[e] * gate_w[e].
constexpr int DMTP_SLABS = 16;  // token slabs of the gate reduction (ws: B * 3 * DMTP_SLABS * ceil(E / 256) floats)

__global__ __launch_bounds__(256) void dmtp_gate_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gate_w,
                                                                float* __restrict__ ws, int k, int E, int ncg) {
  // block = 256 columns x one of DMTP_SLABS token slabs of one batch element (round 1 ran one block per 256 columns: 16
  // workgroups for 8 MB, 0.19 TB/s); thread (cgp = tid & 31, ty = tid >> 5) sums 8 columns over the slab's tokens
  // ty, ty + 8, ...: every wave reads 2 x 512 contiguous bytes per token row.  The three gate logits are linear in the
  // column sums, so every block leaves its own partial logits; multiscale_pool_kernel adds them in a fixed order.
  __shared__ float red[8][32][9];
  __shared__ float fin[3][4];
  const int b = blockIdx.y, cg = blockIdx.x, slab = blockIdx.z;
  const int cgp = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int e0 = cg * 256 + cgp * 8;
  const int lim2 = (k / 2) * 2, lim4 = (k / 4) * 4;
  const int per = (lim4 + DMTP_SLABS - 1) / DMTP_SLABS;
  const int t_begin = slab * per, t_end = min(lim4, t_begin + per);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bf16_t* xb = x + (int64_t)b * k * E;
  if (e0 < E) {
    for (int t = t_begin + ty; t < t_end; t += 8) {
      const uint4 u = *reinterpret_cast<const uint4*>(xb + (int64_t)t * E + e0);
      acc[0] += bf16lo(u.x); acc[1] += bf16hi(u.x); acc[2] += bf16lo(u.y); acc[3] += bf16hi(u.y);
      acc[4] += bf16lo(u.z); acc[5] += bf16hi(u.z); acc[6] += bf16lo(u.w); acc[7] += bf16hi(u.w);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ty][cgp][j] = acc[j];
  __syncthreads();
  // one thread per column: fixed-order reduction over the 8 token lanes, then the (<= 3) tail tokens
  const int col = threadIdx.x, e = cg * 256 + col;
  float s1 = 0.f, s2 = 0.f, s4 = 0.f;
  if (e < E) {
    float c4 = 0.f;
#pragma unroll
    for (int y = 0; y < 8; ++y) c4 += red[y][col >> 3][col & 7];
    float c2 = c4, c1 = c4;
    for (int t = lim4; t < k && slab == 0; ++t) {  // the (<= 3) tail tokens belong to slab 0
      const float v = bf16_to_f32(xb[(int64_t)t * E + e]);
      c1 += v;
      if (t < lim2) c2 += v;
    }
    const float gw = bf16_to_f32(gate_w[e]);
    s1 = c1 / (float)k * gw;
    s2 = lim2 ? c2 / (float)lim2 * gw : 0.f;
    s4 = lim4 ? c4 / (float)lim4 * gw : 0.f;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2); s4 = wave_sum(s4);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { fin[0][wv] = s1; fin[1][wv] = s2; fin[2][wv] = s4; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float* r = fin[threadIdx.x];
    ws[(((int64_t)b * 3 + threadIdx.x) * DMTP_SLABS + slab) * ncg + cg] = (r[0] + r[1]) + (r[2] + r[3]);
  }
}

__global__ __launch_bounds__(256) void multiscale_pool_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int k,
                                                              int E, const float* __restrict__ ws, const bf16_t* __restrict__ gate_b,
                                                              int ncg, int use_gate) {
  float wsm[3];                   // the three gate weights (every thread evaluates them from the summed logits)
  __shared__ float part[3][4];    // the summed partial gate logits of this batch element (ncg * DMTP_SLABS <= 512: E <= 8192)
  const int b = blockIdx.y;
  const int L1 = k, L2 = k / 2, L4 = k / 4;
  const int Lout = L1 + L2 + L4;
  // Round 6: every workgroup used to have its thread 0 read the 3 x 256 partial logits one by one from global memory before anybody
  // could start (63 us for a 23 MB kernel); now the workgroup fetches them together and one WAVE per scale adds them (lane sums of a
  // 64-stride, then the xor butterfly: a fixed order)
  const int np = ncg * DMTP_SLABS, ns_ = 1 + (k >= 2) + (k >= 4);
  if (use_gate) {
    // wave `sc` adds the partial logits of scale `sc`: lane sums of a 64-stride (all of a lane's loads in flight together), then the xor
    // butterfly -- a fixed order, the gates repeat bit for bit; ONE barrier, then every thread evaluates the three-way softmax itself
    const int sc = threadIdx.x >> 6, ln = threadIdx.x & 63;
    if (sc < ns_) {
      float pv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pv[u] = (ln + 64 * u < np) ? ws[((int64_t)b * 3 + sc) * np + ln + 64 * u] : 0.f;
      float a = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (ln + 64 * u < np) a += pv[u];
      a = wave_sum(a);
      if (ln == 0) part[sc][0] = a;
    }
    __syncthreads();
    float g[3], m = -INFINITY, den = 0.f;
    for (int s = 0; s < ns_; ++s) {
      g[s] = part[s][0] + bf16_to_f32(gate_b[0]);
      m = fmaxf(m, g[s]);
    }
    for (int s = 0; s < ns_; ++s) { g[s] = __expf(g[s] - m); den += g[s]; }
    for (int s = 0; s < 3; ++s) wsm[s] = s < ns_ ? g[s] / den : 1.f;
  } else {
    wsm[0] = wsm[1] = wsm[2] = 1.f;
  }
  const int e8n = E >> 3;
  const bf16_t* xb = x + (int64_t)b * k * E;
  bf16_t* ob = out + (int64_t)b * Lout * E;
  // Round 6: a work item = 8 columns x FOUR consecutive tokens 4 g .. 4 g + 3: the four 16-byte loads leave together, x is read once
  // (not once per scale), and the item writes its 4 + 2 + 1 output rows (the old loop walked the 1792 output rows with 1 / 2 / 4 dependent
  // loads each and two 64-bit divisions per item: 24 us for 23 MB).  Per output element the same additions in the same order.
  const float f1 = wsm[0], f2 = wsm[1] / 2.f, f4 = wsm[2] / 4.f;
  const int ngrp = k >> 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < ngrp * e8n; i += gridDim.x * 256) {
    const int e8 = i % e8n, g = i / e8n;
    float v[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const uint4 u = *reinterpret_cast<const uint4*>(xb + (int64_t)(4 * g + a) * E + e8 * 8);
      v[a][0] = bf16lo(u.x); v[a][1] = bf16hi(u.x); v[a][2] = bf16lo(u.y); v[a][3] = bf16hi(u.y);
      v[a][4] = bf16lo(u.z); v[a][5] = bf16hi(u.z); v[a][6] = bf16lo(u.w); v[a][7] = bf16hi(u.w);
    }
    auto put = [&](int j, const float (&r)[8], float f) {
      *reinterpret_cast<uint4*>(ob + (int64_t)j * E + e8 * 8) =
          uint4{pack2_bf16(r[0] * f, r[1] * f), pack2_bf16(r[2] * f, r[3] * f), pack2_bf16(r[4] * f, r[5] * f), pack2_bf16(r[6] * f, r[7] * f)};
    };
#pragma unroll
    for (int a = 0; a < 4; ++a) put(4 * g + a, v[a], f1);
    float p0[8], p1[8], q[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      p0[c] = 0.f + v[0][c] + v[1][c];            // (0 + a + b, then + c + d: the order of the row loop this replaces)
      p1[c] = 0.f + v[2][c] + v[3][c];
      q[c] = 0.f + v[0][c] + v[1][c] + v[2][c] + v[3][c];
    }
    put(L1 + 2 * g, p0, f2);
    put(L1 + 2 * g + 1, p1, f2);
    put(L1 + L2 + g, q, f4);
  }
  // the (<= 3) tokens past the last group of four: their scale-1 rows and, if there are two, one scale-2 row
  const int t_rest = ngrp * 4, jrows = (k - t_rest) + (L2 - 2 * ngrp);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < jrows * e8n; i += gridDim.x * 256) {
    const int e8 = i % e8n, jj = i / e8n;
    const bool one = jj < k - t_rest;
    const int t0 = one ? t_rest + jj : t_rest, s = one ? 1 : 2, j = one ? t_rest + jj : L1 + 2 * ngrp;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < s; ++a) {
      const uint4 u = *reinterpret_cast<const uint4*>(xb + (int64_t)(t0 + a) * E + e8 * 8);
      acc[0] += bf16lo(u.x); acc[1] += bf16hi(u.x); acc[2] += bf16lo(u.y); acc[3] += bf16hi(u.y);
      acc[4] += bf16lo(u.z); acc[5] += bf16hi(u.z); acc[6] += bf16lo(u.w); acc[7] += bf16hi(u.w);
    }
    const float f = one ? f1 : f2;
    *reinterpret_cast<uint4*>(ob + (int64_t)j * E + e8 * 8) =
        uint4{pack2_bf16(acc[0] * f, acc[1] * f), pack2_bf16(acc[2] * f, acc[3] * f),
              pack2_bf16(acc[4] * f, acc[5] * f), pack2_bf16(acc[6] * f, acc[7] * f)};
  }
}

int multiscale_pool(const bf16_t* x, bf16_t* out, int B, int k, int E, const bf16_t* gate_w, const bf16_t* gate_b,
                    float* ws, hipStream_t stream) {
  if (!x || !out || B <= 0 || B > 65535 || k <= 0 || (E & 7)) return U2_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)out) & 15) return U2_ERR_ARG;
  const int ncg = (int)cdiv(E, 256);
  const int use_gate = gate_w != nullptr;
  ProfScope ps(PROF_ROWOP, 0, stream);
  if (use_gate) {
    if (!gate_b || !ws || ncg * DMTP_SLABS > 512) return U2_ERR_ARG;   // (E <= 8192: the pooling kernel stages the partial logits in LDS)
    hipLaunchKernelGGL(dmtp_gate_partial_kernel, dim3(ncg, B, DMTP_SLABS), dim3(256), 0, stream, x, gate_w, ws, k, E, ncg);
    if (launch_status() != U2_OK) return U2_ERR_LAUNCH;
  }
  if ((int64_t)k * (E >> 3) >= (1ll << 31)) return U2_ERR_ARG;
  const int64_t total = (int64_t)std::max(k >> 2, 1) * (E >> 3);   // items of four tokens x 8 columns
  const unsigned blocks = (unsigned)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
  hipLaunchKernelGGL(multiscale_pool_kernel, dim3(blocks, B), dim3(256), 0, stream, x, out, k, E, ws, gate_b, ncg,
                     use_gate);
  return launch_status();
}

}  // namespace u2
