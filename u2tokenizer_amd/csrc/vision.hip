// HBM-bound data-movement kernels at the two ends of the hot path: 3-D patch im2col (volume -> GEMM
// operand), SPP average pooling, embedding lookup + visual-token splice.
#include "kernels.h"

namespace u2 {

// ---------------------------------------------------------------- im2col
// Reference: MONAI PatchEmbeddingBlock(pos_embed="perceptron") as instantiated at vit.py:90-99:
//   Rearrange("b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)")  followed by Linear.
// One workgroup stages the p1*p2 voxel rows of one (chunk, h, w) cell -- each row is W contiguous
// voxels = nd patches x p3 -- in LDS with fully coalesced 16-byte reads, converts to bf16 (RNE), and
// writes the nd token rows (p1*p2*p3 features each, contiguous) with coalesced 16-byte stores.
template <int DT>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ vol, bf16_t* __restrict__ out, int D, int H,
                                                     int W, int p1, int p2, int p3, int nh, int nw, int nd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int row_bytes = W * 2 + 16;  // +16 B pad: feature-group reads of one token hit distinct banks
  const int cell = blockIdx.x;       // (chunk, h, w)
  const int w_i = cell % nw, h_i = (cell / nw) % nh, chunk = cell / (nw * nh);
  const int nrows = p1 * p2, gpr = W >> 3;  // 8-voxel groups per row
  const int64_t vbase = (int64_t)chunk * D * H * W;
  for (int g = threadIdx.x; g < nrows * gpr; g += 256) {
    const int rr = g / gpr, col = (g - rr * gpr) * 8;
    const int p1i = rr / p2, p2i = rr - p1i * p2;
    const int64_t off = vbase + ((int64_t)(h_i * p1 + p1i) * H + (w_i * p2 + p2i)) * W + col;
    uint4 o;
    if constexpr (DT == (U2_ELEM_IS_F16 ? VOL_F16 : VOL_BF16)) {  // voxels already in the element type of the build
      o = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(vol) + off);
    } else if constexpr (DT == VOL_BF16) {
      const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(vol) + off);
      const uint16_t* hp = reinterpret_cast<const uint16_t*>(&u);
      o = uint4{pack2_bf16(voxel_bf16_to_f32(hp[0]), voxel_bf16_to_f32(hp[1])), pack2_bf16(voxel_bf16_to_f32(hp[2]), voxel_bf16_to_f32(hp[3])),
                pack2_bf16(voxel_bf16_to_f32(hp[4]), voxel_bf16_to_f32(hp[5])), pack2_bf16(voxel_bf16_to_f32(hp[6]), voxel_bf16_to_f32(hp[7]))};
    } else if constexpr (DT == VOL_F16) {
      const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const _Float16*>(vol) + off);
      const _Float16* hp = reinterpret_cast<const _Float16*>(&u);
      o = uint4{pack2_bf16((float)hp[0], (float)hp[1]), pack2_bf16((float)hp[2], (float)hp[3]),
                pack2_bf16((float)hp[4], (float)hp[5]), pack2_bf16((float)hp[6], (float)hp[7])};
    } else {
      const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(vol) + off);
      const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(vol) + off + 4);
      o = uint4{pack2_bf16(a.x, a.y), pack2_bf16(a.z, a.w), pack2_bf16(b.x, b.y), pack2_bf16(b.z, b.w)};
    }
    *reinterpret_cast<uint4*>(smem + rr * row_bytes + col * 2) = o;
  }
  __syncthreads();
  const int K = p1 * p2 * p3, gpt = K >> 3;  // 8-feature groups per token
  const int64_t tok0 = ((int64_t)chunk * nh * nw + (int64_t)h_i * nw + w_i) * nd;
  for (int g = threadIdx.x; g < nd * gpt; g += 256) {
    const int d_i = g / gpt, f0 = (g - d_i * gpt) * 8;
    const int rr = f0 / p3, p3i = f0 - rr * p3;
    const uint4 o = *reinterpret_cast<const uint4*>(smem + rr * row_bytes + (d_i * p3 + p3i) * 2);
    *reinterpret_cast<uint4*>(out + (tok0 + d_i) * K + f0) = o;
  }
}

int im2col_patches(const void* vol, int vol_dtype, bf16_t* out, int nchunk, int D, int H, int W, int p1, int p2,
                   int p3, hipStream_t stream) {
  if (!vol || !out || nchunk <= 0 || p1 <= 0 || p2 <= 0 || p3 <= 0) return U2_ERR_ARG;
  if (D % p1 || H % p2 || W % p3 || (p3 & 7) || (W & 7)) return U2_ERR_ARG;
  if (((uintptr_t)vol | (uintptr_t)out) & 15) return U2_ERR_ARG;
  const int nh = D / p1, nw = H / p2, nd = W / p3;
  const size_t smem = (size_t)p1 * p2 * (W * 2 + 16);
  if (smem > 64 * 1024) return U2_ERR_ARG;
  dim3 grid((unsigned)((int64_t)nchunk * nh * nw));
  ProfScope ps(PROF_MOVE, 0, stream, (double)nchunk * D * H * W * ((vol_dtype == VOL_F32 ? 4.0 : 2.0) + 2.0));
#define U2_IM2COL(DT) \
  hipLaunchKernelGGL((im2col_kernel<DT>), grid, dim3(256), smem, stream, vol, out, D, H, W, p1, p2, p3, nh, nw, nd)
  if (vol_dtype == VOL_F16) U2_IM2COL(VOL_F16);
  else if (vol_dtype == VOL_BF16) U2_IM2COL(VOL_BF16);
  else if (vol_dtype == VOL_F32) U2_IM2COL(VOL_F32);
  else return U2_ERR_ARG;
#undef U2_IM2COL
  return launch_status();
}

// ---------------------------------------------------------------- SPP pooling
// Reference: spatial_pooling_projector.py:38-41 -- tokens to a (g1,g2,g3) grid, F.avg_pool3d(k=s=ps)
// (floor semantics: trailing cells that do not fill a window are dropped), back to a token sequence;
// pooling_type "sequence" (spatial_pooling_projector.py:42-45) is the (1,1,ntok) grid with a (1,1,ps^3) window.
__global__ __launch_bounds__(256) void avgpool3d_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int nb, int g1,
                                                        int g2, int g3, int w1, int w2, int w3, int C) {
  const int o1n = g1 / w1, o2n = g2 / w2, o3n = g3 / w3, c8n = C >> 3;
  const int64_t total = (int64_t)nb * o1n * o2n * o3n * c8n;
  const float inv = 1.f / (float)(w1 * w2 * w3);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % c8n);
    int64_t t = i / c8n;
    const int o3 = (int)(t % o3n); t /= o3n;
    const int o2 = (int)(t % o2n); t /= o2n;
    const int o1 = (int)(t % o1n);
    const int b = (int)(t / o1n);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < w1; ++a)
      for (int bb = 0; bb < w2; ++bb)
        for (int cc = 0; cc < w3; ++cc) {
          const int64_t tok = ((int64_t)(o1 * w1 + a) * g2 + (o2 * w2 + bb)) * g3 + (o3 * w3 + cc);
          const uint4 u = *reinterpret_cast<const uint4*>(x + ((int64_t)b * g1 * g2 * g3 + tok) * C + c8 * 8);
          acc[0] += bf16lo(u.x); acc[1] += bf16hi(u.x); acc[2] += bf16lo(u.y); acc[3] += bf16hi(u.y);
          acc[4] += bf16lo(u.z); acc[5] += bf16hi(u.z); acc[6] += bf16lo(u.w); acc[7] += bf16hi(u.w);
        }
    const int64_t otok = ((int64_t)b * o1n + o1) * o2n * o3n + (int64_t)o2 * o3n + o3;
    *reinterpret_cast<uint4*>(y + otok * C + c8 * 8) =
        uint4{pack2_bf16(acc[0] * inv, acc[1] * inv), pack2_bf16(acc[2] * inv, acc[3] * inv),
              pack2_bf16(acc[4] * inv, acc[5] * inv), pack2_bf16(acc[6] * inv, acc[7] * inv)};
  }
}

int avgpool3d_tokens(const bf16_t* x, bf16_t* y, int nb, int g1, int g2, int g3, int w1, int w2, int w3, int C,
                     hipStream_t stream) {
  if (!x || !y || nb <= 0 || w1 <= 0 || w2 <= 0 || w3 <= 0 || g1 < w1 || g2 < w2 || g3 < w3 || (C & 7)) return U2_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)y) & 15) return U2_ERR_ARG;
  const int64_t total = (int64_t)nb * (g1 / w1) * (g2 / w2) * (g3 / w3) * (C >> 3);
  const unsigned blocks = (unsigned)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
  ProfScope ps(PROF_MOVE, 0, stream);
  hipLaunchKernelGGL(avgpool3d_kernel, dim3(blocks), dim3(256), 0, stream, x, y, nb, g1, g2, g3, w1, w2, w3, C);
  return launch_status();
}

// ---------------------------------------------------------------- broadcast one row into a strided slot
// cls_token.expand(b, -1, -1) + cat (vit.py:116-118); query_tokens.expand(B, -1, -1) (u2Tokenizer.py:43).
__global__ __launch_bounds__(256) void fill_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int nb,
                                                        int64_t n, int64_t dst_bs) {
  const int64_t total = (int64_t)nb * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / n, e = i - b * n;
    dst[b * dst_bs + e] = src[e];
  }
}

int fill_rows(const bf16_t* src, bf16_t* dst, int nb, int64_t n, int64_t dst_bs, hipStream_t stream) {
  if (!src || !dst || nb <= 0 || n <= 0) return U2_ERR_ARG;
  const int64_t total = (int64_t)nb * n;
  const unsigned blocks = (unsigned)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
  ProfScope ps(PROF_MOVE, 0, stream);
  hipLaunchKernelGGL(fill_rows_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, nb, n, dst_bs);
  return launch_status();
}

// ---------------------------------------------------------------- rotary position embedding
// Reference: rope.py:6-13,33-40,77-80 (rotate-half form, base 10000, cos/sin cached in fp32 and cast to
// the activation dtype).  Rows are indexed (outer, s, inner) with position s; heads are d-wide column slices.
// sgn = -1 applies the inverse rotation: the backward of y = x cos + rotate_half(x) sin is dx = dy cos - rotate_half(dy) sin.
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ x, int64_t n_outer, int S, int n_inner, int H, int d,
                                                   int64_t ld, float sgn) {
  const int half = d >> 1;
  const int64_t total = n_outer * S * n_inner * H * half;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % half);
    int64_t t = i / half;
    const int h = (int)(t % H);
    const int64_t row = t / H;
    const int s = (int)((row / n_inner) % S);
    const float inv_freq = 1.0f / powf(10000.0f, (float)(2 * j) / (float)d);
    const float ang = (float)s * inv_freq;
    const float c = bf16_to_f32(f32_to_bf16(cosf(ang))), sn = sgn * bf16_to_f32(f32_to_bf16(sinf(ang)));
    bf16_t* p = x + row * ld + (int64_t)h * d + j;
    const float a = bf16_to_f32(p[0]), b = bf16_to_f32(p[half]);
    p[0] = f32_to_bf16(a * c - b * sn);
    p[half] = f32_to_bf16(b * c + a * sn);
  }
}

int rope_apply(bf16_t* x, int64_t n_outer, int S, int n_inner, int H, int d, int64_t ld, int max_len, int inverse,
               hipStream_t stream) {
  if (!x || n_outer <= 0 || S <= 0 || n_inner <= 0 || H <= 0 || d <= 0 || (d & 1) || S > max_len) return U2_ERR_ARG;
  const int64_t total = n_outer * S * n_inner * H * (d >> 1);
  const unsigned blocks = (unsigned)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
  ProfScope ps(PROF_ROWOP, 0, stream);
  hipLaunchKernelGGL(rope_kernel, dim3(blocks), dim3(256), 0, stream, x, n_outer, S, n_inner, H, d, ld, inverse ? -1.0f : 1.0f);
  return launch_status();
}

// ---------------------------------------------------------------- embedding lookup + splice
// Reference: u2_arch.py:109 (embed_tokens(question_ids)) and u2_arch.py:113-116
//   cat(embeds[:, :1], image_features, embeds[:, image_features.shape[1] + 1:]).
__global__ __launch_bounds__(256) void embed_splice_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ ids,
                                                           const bf16_t* __restrict__ feats, bf16_t* __restrict__ out, int B,
                                                           int S, int E, int nfeat, int64_t vocab) {
  const int e8n = E >> 3;
  const int64_t total = (int64_t)B * S * e8n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int e8 = (int)(i % e8n);
    const int64_t bs = i / e8n;
    const int s = (int)(bs % S);
    const int b = (int)(bs / S);
    const bf16_t* src;
    if (s >= 1 && s <= nfeat) {
      src = feats + ((int64_t)b * nfeat + (s - 1)) * E;
    } else {
      int64_t id = ids[bs];
      id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // never read outside the table
      src = table + id * E;
    }
    *reinterpret_cast<uint4*>(out + bs * E + e8 * 8) = *reinterpret_cast<const uint4*>(src + e8 * 8);
  }
}

int embed_splice(const bf16_t* table, const int64_t* ids, const bf16_t* feats, bf16_t* out, int B, int S, int E,
                 int nfeat, int64_t vocab, hipStream_t stream) {
  if (!table || !ids || !out || B <= 0 || S <= 0 || (E & 7) || vocab <= 0) return U2_ERR_ARG;
  if (nfeat < 0 || (nfeat > 0 && (!feats || nfeat + 1 > S))) return U2_ERR_ARG;
  if (((uintptr_t)table | (uintptr_t)out | (uintptr_t)feats) & 15) return U2_ERR_ARG;
  const int64_t total = (int64_t)B * S * (E >> 3);
  const unsigned blocks = (unsigned)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
  ProfScope ps(PROF_MOVE, 0, stream);
  hipLaunchKernelGGL(embed_splice_kernel, dim3(blocks), dim3(256), 0, stream, table, ids, feats, out, B, S, E, nfeat,
                     vocab);
  return launch_status();
}

}  // namespace u2
