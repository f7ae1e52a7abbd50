// GPU form of u2Transform.adaptive_resize (reference src/utils/u2Transform.py:62-122, validation transforms :46-54):
// the step that PRODUCES the (8, 32, 256, 256) tensor the hot path consumes.  In the reference it is MONAI on CPU in
// the DataLoader workers (and on the training process itself for DPO, dpo_u2trainer.py:19,160), ~1-2 s per volume; at
// 80+ volumes/s per GPU that cannot feed the path.  Stages (all HBM-bound, nothing synchronises: every data-dependent
// quantity -- percentiles, crop box, output geometry, filter taps -- lives in a device-side parameter block, and the
// kernels are launched over the worst-case extents):
//
//   1. ScaleIntensityRangePercentiles(0.5, 99.5, b_min 0, b_max 1, clip)      u2Transform.py:51
//        exact order statistics by a 3-pass radix select on the sortable fp32 bits (11 + 11 + 10 bits, four ranks
//        at once), numpy's linear interpolation in fp64
//   2. CropForeground (select x > 0 after scaling == x > a_min)               u2Transform.py:52
//        bounding box by atomicMin / atomicMax
//   3. resize to (int(H r), int(W r), min(D, 256)), r = min(256/H, 256/W), trilinear, align_corners, anti-aliased:
//        per-axis Gaussian (sigma = max(0, (in/out - 1)/2), MONAI's erf taps, zero padding), then linear
//        interpolation                                                         u2Transform.py:75-112
//   4. zero-pad to 256^3, viewed as (8, 32, 256, 256)                          u2Transform.py:93-94,117-120
//
// Working layout is the input's own [d][h][w] (w fastest) from start to finish: the reference's permutes
// (u2Transform.py:71,117) are pure index relabelling and the final (8, 32, 256, 256) view IS [d][h][w].
#include "kernels.h"

namespace u2 {

namespace {

constexpr int PP_MAX_TAIL = 16;  // Gaussian taps kept per axis: 2 * tail + 1 <= 33 (sigma <= 4, i.e. 9x downsampling)

struct PreParams {  // device-side parameter block
  // --- radix select state: 4 ranks (lower lo/hi, upper lo/hi)
  unsigned long long rank[4];   // remaining rank inside the current prefix
  unsigned prefix[4];           // key prefix found so far
  // --- results of the passes
  double a_min, a_max, inv_range;  // percentiles; 1 / (a_max - a_min) (0 when degenerate)
  int degenerate;                  // a_max == a_min: MONAI returns x - a_min unclipped
  int box_lo[3], box_hi[3];        // foreground box [lo, hi) over (d, h, w)
  int in_sz[3], out_sz[3];         // cropped size, resized size (d, h, w)
  float scale[3];                  // align_corners source step (in - 1) / (out - 1)
  int tail[3];
  float taps[3][2 * PP_MAX_TAIL + 1];
  int status;                      // 0 ok, 1 empty foreground, 2 filter too wide
  // --- training-time augmentation (u2Transform.py:37-42), applied to the cropped volume before the resize
  int crop_sz[3];                  // foreground box size (d, h, w); in_sz = the same after the rotation
  int rot_k;                       // RandRotate90 over (h, w): 0..3 quarter turns (torch.rot90)
  int flip[3];                     // RandFlip along d / h / w of the rotated volume
  float aug_mul, aug_add;          // RandScaleIntensity (1 + factor), RandShiftIntensity offset
};

__device__ __forceinline__ unsigned sortable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unsortable(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// pass 0: bits 31..21 (2048 bins), pass 1: bits 20..10 (2048), pass 2: bits 9..0 (1024)
template <int PASS>
__global__ __launch_bounds__(256) void pre_hist_kernel(const float* __restrict__ x, int64_t n, const PreParams* __restrict__ pp,
                                                       unsigned* __restrict__ hist /* [4][2048] */) {
  constexpr int SHIFT = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
  constexpr int BINS = PASS == 2 ? 1024 : 2048;
  constexpr int NH = PASS == 0 ? 1 : 4;  // pass 0: no prefix yet, the four ranks share one histogram
  __shared__ unsigned lh[NH][BINS];
  for (int i = threadIdx.x; i < NH * BINS; i += 256) (&lh[0][0])[i] = 0;
  unsigned pre[4] = {0, 0, 0, 0};
  if (PASS > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) pre[r] = pp->prefix[r];
  }
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const unsigned k = sortable(x[i]);
    if (PASS == 0) {
      atomicAdd(&lh[0][k >> 21], 1u);
    } else {
      const unsigned top = PASS == 1 ? (k >> 21) : (k >> 10);
      const unsigned dg = (k >> SHIFT) & (BINS - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (top == pre[r]) atomicAdd(&lh[r][dg], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NH * BINS; i += 256) {
    const unsigned v = (&lh[0][0])[i];
    if (v) atomicAdd(&hist[(i / BINS) * 2048 + (i % BINS)], v);
  }
}

// one workgroup: for every rank, the bin in which the cumulative count passes the rank; refine prefix, reduce rank
template <int PASS>
__global__ __launch_bounds__(256) void pre_select_kernel(PreParams* pp, unsigned* hist) {
  constexpr int BINS = PASS == 2 ? 1024 : 2048;
  constexpr int BITS = PASS == 2 ? 10 : 11;
  __shared__ unsigned long long part[256];
  for (int r = 0; r < 4; ++r) {
    const unsigned* h = hist + (PASS == 0 ? 0 : r) * 2048;
    constexpr int PER = BINS / 256;
    unsigned long long s = 0;
    for (int j = 0; j < PER; ++j) s += h[threadIdx.x * PER + j];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long rk = pp->rank[r], acc = 0;
      int t = 0;
      while (t < 255 && acc + part[t] <= rk) acc += part[t++];
      int b = t * PER;
      while (b < BINS - 1 && acc + h[b] <= rk) acc += h[b++];
      pp->rank[r] = rk - acc;
      pp->prefix[r] = PASS == 0 ? (unsigned)b : ((pp->prefix[r] << BITS) | (unsigned)b);
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 4 * 2048; i += 256) hist[i] = 0;  // ready for the next pass
}

__global__ void pre_init_kernel(PreParams* pp, unsigned* hist, int64_t n, double lower, double upper, int D, int H, int W,
                                PreAugment aug) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    pp->rot_k = aug.rot_k & 3;
    for (int a = 0; a < 3; ++a) pp->flip[a] = aug.flip[a] != 0;
    pp->aug_mul = aug.mul;
    pp->aug_add = aug.add;
    // np.percentile(method="linear"): virtual index q/100 * (n - 1), neighbours floor / floor + 1 (clamped)
    const double qs[2] = {lower, upper};
    for (int i = 0; i < 2; ++i) {
      const double pos = qs[i] / 100.0 * (double)(n - 1);
      long long lo = (long long)floor(pos);
      long long hi = lo + 1 < n ? lo + 1 : n - 1;
      pp->rank[2 * i] = (unsigned long long)lo;
      pp->rank[2 * i + 1] = (unsigned long long)hi;
    }
    for (int r = 0; r < 4; ++r) pp->prefix[r] = 0;
    pp->box_lo[0] = D; pp->box_lo[1] = H; pp->box_lo[2] = W;
    pp->box_hi[0] = pp->box_hi[1] = pp->box_hi[2] = 0;
    pp->status = 0;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 4 * 2048; i += gridDim.x * blockDim.x) hist[i] = 0;
}

__global__ void pre_percentile_kernel(PreParams* pp, int64_t n, double lower, double upper) {
  if (threadIdx.x || blockIdx.x) return;
  double res[2];
  const double qs[2] = {lower, upper};
  for (int i = 0; i < 2; ++i) {
    const double a = (double)unsortable(pp->prefix[2 * i]), b = (double)unsortable(pp->prefix[2 * i + 1]);
    const double pos = qs[i] / 100.0 * (double)(n - 1);
    const double t = pos - floor(pos);
    // numpy _lerp: a + (b - a) t, computed from the far end for t >= 0.5
    const double d = b - a;
    res[i] = t >= 0.5 ? b - d * (1.0 - t) : a + d * t;
  }
  pp->a_min = res[0];
  pp->a_max = res[1];
  pp->degenerate = (res[1] - res[0] == 0.0) ? 1 : 0;
  pp->inv_range = pp->degenerate ? 0.0 : 1.0 / (res[1] - res[0]);
}

// ScaleIntensityRange (a_min, a_max, 0, 1, clip=True), result in float32 as MONAI returns it
__device__ __forceinline__ float pre_scale(float x, const PreParams& p) {
  if (p.degenerate) return (float)((double)x - p.a_min);
  const double y = ((double)x - p.a_min) * p.inv_range;
  return (float)(y < 0.0 ? 0.0 : (y > 1.0 ? 1.0 : y));
}

__global__ __launch_bounds__(256) void pre_bbox_kernel(const float* __restrict__ x, PreParams* pp, int D, int H, int W) {
  __shared__ int s_lo[3], s_hi[3];
  if (threadIdx.x < 3) { s_lo[threadIdx.x] = 0x7fffffff; s_hi[threadIdx.x] = 0; }
  __syncthreads();
  const PreParams p = *pp;
  const int64_t n = (int64_t)D * H * W;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (pre_scale(x[i], p) > 0.f) {  // CropForeground's default select_fn: img > 0
      const int w = (int)(i % W), h = (int)((i / W) % H), d = (int)(i / ((int64_t)W * H));
      lo[0] = min(lo[0], d); hi[0] = max(hi[0], d + 1);
      lo[1] = min(lo[1], h); hi[1] = max(hi[1], h + 1);
      lo[2] = min(lo[2], w); hi[2] = max(hi[2], w + 1);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (lo[a] != 0x7fffffff) { atomicMin(&s_lo[a], lo[a]); atomicMax(&s_hi[a], hi[a]); }
  }
  __syncthreads();
  if (threadIdx.x < 3 && s_lo[threadIdx.x] != 0x7fffffff) {
    atomicMin(&pp->box_lo[threadIdx.x], s_lo[threadIdx.x]);
    atomicMax(&pp->box_hi[threadIdx.x], s_hi[threadIdx.x]);
  }
}

// output geometry, interpolation steps and Gaussian taps from the crop box (one thread; double / float arithmetic in the
// order Python / torch perform it, so int() truncations agree)
__global__ void pre_geometry_kernel(PreParams* pp, int target, int depth_pad) {
  if (threadIdx.x || blockIdx.x) return;
  PreParams& p = *pp;
  for (int a = 0; a < 3; ++a) p.crop_sz[a] = p.in_sz[a] = p.box_hi[a] - p.box_lo[a];
  if (p.in_sz[0] <= 0 || p.in_sz[1] <= 0 || p.in_sz[2] <= 0) { p.status = 1; return; }
  if (p.rot_k & 1) { p.in_sz[1] = p.crop_sz[2]; p.in_sz[2] = p.crop_sz[1]; }  // a quarter turn swaps h and w
  // ratio = min(target / H, target / W); scaling_shape = [int(H * ratio), int(W * ratio)]   (u2Transform.py:75-76)
  const double rh = (double)target / (double)p.in_sz[1], rw = (double)target / (double)p.in_sz[2];
  const double ratio = rh < rw ? rh : rw;
  p.out_sz[1] = (int)((double)p.in_sz[1] * ratio);
  p.out_sz[2] = (int)((double)p.in_sz[2] * ratio);
  p.out_sz[0] = depth_pad >= p.in_sz[0] ? p.in_sz[0] : depth_pad;  // :80-81 / :97-98
  bool aa = false;
  for (int a = 0; a < 3; ++a) aa = aa || p.out_sz[a] < p.in_sz[a];
  for (int a = 0; a < 3; ++a) {
    // F.interpolate(align_corners=True): step (in - 1) / (out - 1) in float, 0 when out == 1
    p.scale[a] = p.out_sz[a] > 1 ? (float)(p.in_sz[a] - 1) / (float)(p.out_sz[a] - 1) : 0.f;
    // anti-aliasing (monai resize): factors = in / out (float32), sigma = max(0, (factors - 1) / 2)
    float sigma = 0.f;
    if (aa) {
      const float f = (float)p.in_sz[a] / (float)p.out_sz[a];
      sigma = fmaxf(0.f, (f - 1.f) / 2.f);
    }
    // gaussian_1d(sigma, truncated=4.0, approx="erf", normalize=False): MONAI's GaussianFilter.forward calls it with the
    // default normalize=False, so the taps are NOT divided by their sum (they miss the truncated tail mass, ~6e-5)
    const int tail = (int)(fmaxf(sigma * 4.0f, 0.5f) + 0.5f);
    if (tail > PP_MAX_TAIL) { p.status = 2; return; }
    p.tail[a] = aa ? tail : 0;
    if (!aa) {
      p.taps[a][0] = 1.f;
      continue;
    }
    const float t = 0.70710678f / fabsf(sigma);  // inf for sigma == 0 -> taps {0, 1, 0}
    for (int i = 0; i <= 2 * tail; ++i) {
      const float xx = (float)(i - tail);
      float v = 0.5f * (erff(t * (xx + 0.5f)) - erff(t * (xx - 0.5f)));
      v = v < 0.f ? 0.f : v;
      p.taps[a][i] = v;
    }
  }
}

// scaled + cropped volume, compact [dc][hc][wc]
__global__ __launch_bounds__(256) void pre_scale_crop_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const PreParams* __restrict__ pp, int D, int H, int W) {
  const PreParams& p = *pp;
  if (p.status) return;
  const int64_t n = (int64_t)D * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i % W), h = (int)((i / W) % H), d = (int)(i / ((int64_t)W * H));
    if (d >= p.box_lo[0] && d < p.box_hi[0] && h >= p.box_lo[1] && h < p.box_hi[1] && w >= p.box_lo[2] && w < p.box_hi[2]) {
      // cropped coordinates -> torch.rot90(k, (h, w)) -> flips: r = rot90(x): k = 1: r[i][j] = x[j][W-1-i], k = 2:
      // r[i][j] = x[H-1-i][W-1-j], k = 3: r[i][j] = x[H-1-j][i]  (H, W = cropped sizes)
      const int Hc = p.crop_sz[1], Wc = p.crop_sz[2];
      const int is = h - p.box_lo[1], js = w - p.box_lo[2];
      int dd = d - p.box_lo[0], ii, jj;
      switch (p.rot_k) {
        case 1: ii = Wc - 1 - js; jj = is; break;
        case 2: ii = Hc - 1 - is; jj = Wc - 1 - js; break;
        case 3: ii = js; jj = Hc - 1 - is; break;
        default: ii = is; jj = js; break;
      }
      if (p.flip[0]) dd = p.in_sz[0] - 1 - dd;
      if (p.flip[1]) ii = p.in_sz[1] - 1 - ii;
      if (p.flip[2]) jj = p.in_sz[2] - 1 - jj;
      y[((int64_t)dd * p.in_sz[1] + ii) * p.in_sz[2] + jj] = pre_scale(x[i], *pp) * p.aug_mul + p.aug_add;
    }
  }
}

// one axis of the separable Gaussian, zero padding (monai separable_filtering, mode "zeros")
template <int AXIS>
__global__ __launch_bounds__(256) void pre_conv_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                       const PreParams* __restrict__ pp, int64_t n_max) {
  const PreParams& p = *pp;
  if (p.status) return;
  const int Dc = p.in_sz[0], Hc = p.in_sz[1], Wc = p.in_sz[2];
  const int64_t n = (int64_t)Dc * Hc * Wc;
  const int tail = p.tail[AXIS];
  const int len = AXIS == 0 ? Dc : (AXIS == 1 ? Hc : Wc);
  const int64_t stride = AXIS == 0 ? (int64_t)Hc * Wc : (AXIS == 1 ? Wc : 1);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n && i < n_max; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i % Wc), h = (int)((i / Wc) % Hc), d = (int)(i / ((int64_t)Wc * Hc));
    const int c = AXIS == 0 ? d : (AXIS == 1 ? h : w);
    float acc = 0.f;
    for (int j = -tail; j <= tail; ++j) {
      const int cc = c + j;
      if (cc >= 0 && cc < len) acc = __builtin_fmaf(p.taps[AXIS][j + tail], in[i + (int64_t)j * stride], acc);
    }
    out[i] = acc;
  }
}

// trilinear, align_corners = True (torch upsample_trilinear3d index rule), zero pad to T^3, written as fp16 / bf16 / fp32
template <int DT>
__global__ __launch_bounds__(256) void pre_resize_kernel(const float* __restrict__ in, void* __restrict__ out,
                                                         const PreParams* __restrict__ pp, int T, int TD) {
  const PreParams& p = *pp;
  const int64_t n = (int64_t)TD * T * T;
  const int Hc = p.in_sz[1], Wc = p.in_sz[2];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i % T), h = (int)((i / T) % T), d = (int)(i / ((int64_t)T * T));
    float v = 0.f;
    if (!p.status && d < p.out_sz[0] && h < p.out_sz[1] && w < p.out_sz[2]) {
      int i0[3], i1[3];
      float l1[3];
      const int o[3] = {d, h, w};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float src = p.scale[a] * (float)o[a];
        i0[a] = (int)src;
        if (i0[a] > p.in_sz[a] - 1) i0[a] = p.in_sz[a] - 1;
        i1[a] = i0[a] + (i0[a] < p.in_sz[a] - 1 ? 1 : 0);
        l1[a] = src - (float)i0[a];
      }
      const float l0d = 1.f - l1[0], l0h = 1.f - l1[1], l0w = 1.f - l1[2];
      auto at = [&](int dd, int hh, int ww) { return in[((int64_t)dd * Hc + hh) * Wc + ww]; };
      v = l0d * (l0h * (l0w * at(i0[0], i0[1], i0[2]) + l1[2] * at(i0[0], i0[1], i1[2])) +
                 l1[1] * (l0w * at(i0[0], i1[1], i0[2]) + l1[2] * at(i0[0], i1[1], i1[2]))) +
          l1[0] * (l0h * (l0w * at(i1[0], i0[1], i0[2]) + l1[2] * at(i1[0], i0[1], i1[2])) +
                   l1[1] * (l0w * at(i1[0], i1[1], i0[2]) + l1[2] * at(i1[0], i1[1], i1[2])));
    }
    if constexpr (DT == VOL_F32) reinterpret_cast<float*>(out)[i] = v;
    else if constexpr (DT == VOL_BF16) reinterpret_cast<uint16_t*>(out)[i] = f32_to_voxel_bf16(v);
    else reinterpret_cast<_Float16*>(out)[i] = (_Float16)v;
  }
}

__global__ void pre_info_kernel(const PreParams* pp, int32_t* info) {
  if (threadIdx.x || blockIdx.x || !info) return;
  const PreParams& p = *pp;
  info[0] = p.status;
  for (int a = 0; a < 3; ++a) { info[1 + a] = p.box_lo[a]; info[4 + a] = p.box_hi[a]; info[7 + a] = p.out_sz[a]; }
  reinterpret_cast<float*>(info)[10] = (float)p.a_min;
  reinterpret_cast<float*>(info)[11] = (float)p.a_max;
}

}  // namespace

size_t preprocess_workspace_bytes(int D, int H, int W) {
  if (D <= 0 || H <= 0 || W <= 0) return 0;
  const size_t n = (size_t)D * H * W;
  return 2 * ((n * 4 + 255) & ~(size_t)255) + 4 * 2048 * 4 + sizeof(PreParams) + 1024;
}

int preprocess_volume(const float* vol, void* out, int32_t* info, int D, int H, int W, int target, int depth_pad,
                      float lower_pct, float upper_pct, int out_dtype, const PreAugment* aug_in, void* ws, size_t ws_bytes,
                      hipStream_t st) {
  PreAugment aug;  // identity unless the caller asks for the training-time augmentations
  if (aug_in) {
    aug = *aug_in;
    if (aug.rot_k < 0 || aug.rot_k > 3) return U2_ERR_ARG;
  }
  if (!vol || !out || !ws || D <= 0 || H <= 0 || W <= 0 || target <= 0 || depth_pad <= 0) return U2_ERR_ARG;
  if (!(lower_pct >= 0.f && lower_pct < upper_pct && upper_pct <= 100.f)) return U2_ERR_ARG;
  if (out_dtype != VOL_F16 && out_dtype != VOL_BF16 && out_dtype != VOL_F32) return U2_ERR_ARG;
  if (ws_bytes < preprocess_workspace_bytes(D, H, W) || ((uintptr_t)ws & 255)) return U2_ERR_WORKSPACE;
  const int64_t n = (int64_t)D * H * W;
  const size_t nb = ((size_t)n * 4 + 255) & ~(size_t)255;
  char* base = reinterpret_cast<char*>(ws);
  float* bufA = reinterpret_cast<float*>(base);
  float* bufB = reinterpret_cast<float*>(base + nb);
  unsigned* hist = reinterpret_cast<unsigned*>(base + 2 * nb);
  PreParams* pp = reinterpret_cast<PreParams*>(base + 2 * nb + 4 * 2048 * 4);
  const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(n, 256), 256 * 16);
  ProfScope ps(PROF_MOVE, 0, st, (double)n * 4.0 * 9.0 + (double)depth_pad * target * target * 2.0);
  hipLaunchKernelGGL(pre_init_kernel, dim3(32), dim3(256), 0, st, pp, hist, n, (double)lower_pct, (double)upper_pct, D, H, W, aug);
  hipLaunchKernelGGL(pre_hist_kernel<0>, dim3(blocks), dim3(256), 0, st, vol, n, pp, hist);
  hipLaunchKernelGGL(pre_select_kernel<0>, dim3(1), dim3(256), 0, st, pp, hist);
  hipLaunchKernelGGL(pre_hist_kernel<1>, dim3(blocks), dim3(256), 0, st, vol, n, pp, hist);
  hipLaunchKernelGGL(pre_select_kernel<1>, dim3(1), dim3(256), 0, st, pp, hist);
  hipLaunchKernelGGL(pre_hist_kernel<2>, dim3(blocks), dim3(256), 0, st, vol, n, pp, hist);
  hipLaunchKernelGGL(pre_select_kernel<2>, dim3(1), dim3(256), 0, st, pp, hist);
  hipLaunchKernelGGL(pre_percentile_kernel, dim3(1), dim3(64), 0, st, pp, n, (double)lower_pct, (double)upper_pct);
  hipLaunchKernelGGL(pre_bbox_kernel, dim3(blocks), dim3(256), 0, st, vol, pp, D, H, W);
  hipLaunchKernelGGL(pre_geometry_kernel, dim3(1), dim3(64), 0, st, pp, target, depth_pad);
  hipLaunchKernelGGL(pre_scale_crop_kernel, dim3(blocks), dim3(256), 0, st, vol, bufA, pp, D, H, W);
  hipLaunchKernelGGL(pre_conv_kernel<2>, dim3(blocks), dim3(256), 0, st, bufA, bufB, pp, n);
  hipLaunchKernelGGL(pre_conv_kernel<1>, dim3(blocks), dim3(256), 0, st, bufB, bufA, pp, n);
  hipLaunchKernelGGL(pre_conv_kernel<0>, dim3(blocks), dim3(256), 0, st, bufA, bufB, pp, n);
  const int64_t no = (int64_t)depth_pad * target * target;
  const unsigned oblocks = (unsigned)std::min<int64_t>(cdiv(no, 256), 256 * 16);
  if (out_dtype == VOL_F32) hipLaunchKernelGGL(pre_resize_kernel<VOL_F32>, dim3(oblocks), dim3(256), 0, st, bufB, out, pp, target, depth_pad);
  else if (out_dtype == VOL_BF16) hipLaunchKernelGGL(pre_resize_kernel<VOL_BF16>, dim3(oblocks), dim3(256), 0, st, bufB, out, pp, target, depth_pad);
  else hipLaunchKernelGGL(pre_resize_kernel<VOL_F16>, dim3(oblocks), dim3(256), 0, st, bufB, out, pp, target, depth_pad);
  hipLaunchKernelGGL(pre_info_kernel, dim3(1), dim3(64), 0, st, pp, info);
  return launch_status();
}

}  // namespace u2
