// The arithmetic of the few-rows product (M <= 16 rows against N x K weights), shared by gemm_rows16_kernel (gemm.hip) and by the
// big-tile kernel's in-launch tail (gemm_bt.hip, round 5: the ViT's 8 cls rows ride in the launch of the 16384 patch rows instead
// of a launch of their own) -- ONE definition, so that a row's value does not depend on which of the two computed it:
//   * K is cut into NW contiguous slices of `per` 32-wide steps; a slice is ONE accumulator chain of v_mfma_f32_16x16x32 in step
//     order (weights as the A operand: a lane ends up with 4 consecutive output columns of one row);
//   * the NW partial 16 x 16 tiles are added in slice order 0 .. NW - 1, then the epilogue of the product is applied.
#pragma once
#include "kernels.h"

namespace u2 {

// steps [s0, s1) of one slice; wp / xp = the lane's weight / activation row + 8 g elements (16-byte fragments, 64 contiguous bytes per
// 16 rows and step); 8 steps = 16 loads per lane in flight
__device__ __forceinline__ f32x4 rows16_slice(const bf16_t* wp, const bf16_t* xp, int s0, int s1) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;
  for (int sb = s0; sb < s1; sb += U) {
    bf16x8 wf[U], xf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int st = min(sb + u, s1 - 1);
      wf[u] = *reinterpret_cast<const bf16x8*>(wp + st * 32);
      xf[u] = *reinterpret_cast<const bf16x8*>(xp + st * 32);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (sb + u < s1) acc = mfma16(wf[u], xf[u], acc);
  }
  return acc;
}

// epilogue for the lane's 4 columns n = n_first .. n_first + 3 of row m (row index inside C / R / bias_m as handed over)
__device__ __forceinline__ void rows16_store(const GemmDesc& d, const float (&v)[4], int m, int n_first, char* C, const bf16_t* R) {
  const bool out_f32 = d.flags & GEMM_OUT_F32;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n_first + r;
    if (n >= d.N) break;
    float x = v[r] * (n < d.nsplit ? d.alpha_lo : d.alpha);
    if (d.flags & GEMM_BIAS_M) x += bf16_to_f32(d.bias[m]);
    if (d.flags & GEMM_BIAS_N) x += bf16_to_f32(d.bias[n]);
    if (d.flags & GEMM_GELU) x = gelu_epi(x);
    if (d.flags & GEMM_RESIDUAL) x += bf16_to_f32(R[(int64_t)m * d.ldr + n]);
    if (out_f32) reinterpret_cast<float*>(C)[(int64_t)m * d.ldc + n] = x;
    else reinterpret_cast<bf16_t*>(C)[(int64_t)m * d.ldc + n] = f32_to_bf16(x);
  }
}

__host__ __device__ inline int rows16_slices(int nsteps) { return nsteps >= 64 ? 16 : nsteps >= 16 ? 8 : 4; }  // (the launcher's choice of NW)

}  // namespace u2
