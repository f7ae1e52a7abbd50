// 64 < M <= 256 rows against a wide COLD weight: the 22 query-side nn.Linear products of the tokenizer's TextConditionTokenAttMap chain
// (/root/reference/src/model/u2tokenizer/tta.py:93-103: self-attention out-projection, the q projections and dense layers of the visual
// and the text cross attention, four layers; 256 queries x 4096 x 4096 at the benchmark's size, 33.5 MB of weights each, streamed once).
//
// Rounds 1-5 ran them as 64 x 64 tiles x 4 K slices (1024 workgroups, fp32 partial sums, a reduce launch): 25.5 + 7.5 us in the pipeline;
// unsliced on the same two-stage kernel the product takes 53 us (one 16 KB K tile in flight per workgroup against the latency of a cold
// weight).  Round 6, after two dead ends (below): the decomposition the vendor library picks for this shape
// (profiles/r06_vendor_kernel_names.csv: MT64x64x128, no split-K, two tiles prefetched: 22.9 us) -- 64 x 64 tiles over the WHOLE K,
// 4 row tiles x 64 column strips = 256 workgroups, one per CU, no partial sums, no reduce launch (22 launches per volume fewer).  Four
// waves (2 x 2, one v_mfma_f32_32x32x16 accumulator each); a K tile is 128 wide: 64 rows x 256 bytes per operand (32 KB per stage), moved
// by 1 KB LDS-DMA pieces of 4 rows x 256 B -- the longer the row piece, the better the load path does (profiles/r01_stage_bw.log: 49 B/clk
// at 128-byte rows, 31 at 64) --, XOR-swizzled by row & 15 (the sixteen 16-byte positions of a 256-byte row are the 64 banks); three
// stages, two K tiles (64 KB per CU) in flight behind ONE counted wait (both operands have the same depth, so the wave's in-order
// vector-memory counter couples nothing) and one raw barrier per tile.  The row tiles of a strip sit on one XCD (one HBM read of the
// weights, three L2 hits).
//
// Measured (tools/skinny_probe.py, cold weights, 256 x 4096 x 4096): **23.8 us** against 30.0 for the round-5 pair of launches and 23.2 for
// the vendor library; in the pipeline 8.086 -> 7.960 ms per volume with two calls in flight, 8.601 -> 8.433 with one
// (profiles/r06_ab_skinny64.log); three, four and five stages: 8.482 / 8.477 / 8.526 ms (r06_ab_skinny64_stages.log).
//
// The dead ends (history: commit "Skinny M<=256 kernel (256 x 64 tiles ..."; profiles/r06_skinny_probe.log, r06_skinny_ablate.log,
// r06_ab_skinny.log): ALL 256 rows x 64 columns x K / 4 per workgroup (640 KB of operands per CU instead of 1 MB) with the K slices
// combined IN the launch by ticket (agent-scope release / acquire, slabs in the accumulators' own order, summed in slice order:
// bit-repeatable) -- 28.4 us standalone, 6 of them the combine, and slower than the round-5 path in the pipeline; the same with the weight
// pieces issued seven tiles ahead by waves of their own -- no different.  Its ablations (every part removable one at a time, >= 25 us
// left) said the loop ran at ~1.2 us per K tile whichever operand it waited for: fewer bytes per CU was not what this product needed.
#include <algorithm>
#include "kernels.h"

namespace u2 {
namespace {

typedef unsigned int sk_u32x4 __attribute__((ext_vector_type(4)));
constexpr int S64_STAGE = 32768, S64_NS = 3;

// MUBUF: the pieces leave as `buffer_load_dwordx4 ... lds` (descriptor + 32-bit lane offset, the K advance in the scalar offset) instead of
// `global_load_lds_dwordx4` (64-bit lane addresses, a VALU add per piece) -- tools/ubench/stage_bw.hip, profiles/r06_stage_bw.log: beside
// waves that issue MFMAs the FLAT-encoded form stages 14 B/clk per CU, the MUBUF form 44.
template <bool MUBUF>
__global__ __launch_bounds__(256, 1) void gemm_skinny64_kernel(GemmDesc d, int nstrips, int mtiles) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  // XCD-contiguous remap (workgroup b runs on XCD b & 7), then strip = id / mtiles, row tile = id % mtiles
  const int total = nstrips * mtiles;
  int id = blockIdx.x;
  {
    const int q = total >> 3, r = total & 7, xcd = id & 7, idx = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int strip = id / mtiles, mt = id - strip * mtiles;
  const int m0 = mt * 64, n0 = strip * 64;
  const int nk = d.K >> 7;
  // pieces: 4 rows x 256 B; lane -> (row lane >> 4, position lane & 15) reads global chunk position ^ (row & 15).  Wave w issues pieces
  // 4 w .. 4 w + 3 of the 16 of each operand (rows past M / N read the last row: never stored)
  const int pr = lane >> 4, pc = lane & 15;
  const bf16_t* pa[4];
  const bf16_t* pw[4];
  int va[4], vw[4];   // (MUBUF: byte offsets from the operand's base; the launcher keeps both operands under 2 GB)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * (4 * wave + i) + pr;
    const int64_t oa = (int64_t)min(m0 + row, d.M - 1) * d.lda + ((pc ^ (row & 15)) << 3);
    const int64_t ow = (int64_t)min(n0 + row, d.N - 1) * d.ldb + ((pc ^ (row & 15)) << 3);
    pa[i] = d.A + oa;
    pw[i] = d.B + ow;
    va[i] = (int)(oa * 2);
    vw[i] = (int)(ow * 2);
  }
  auto issue = [&](int t) {
    char* st = lds + (t % S64_NS) * S64_STAGE + (4 * wave) * 1024;
    if constexpr (MUBUF) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        lds_dma_mubuf16(d.A, st + i * 1024, va[i], t * 256);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        lds_dma_mubuf16(d.B, st + 16384 + i * 1024, vw[i], t * 256);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[i] + t * 128),
                                         (__attribute__((address_space(3))) void*)(st + i * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pw[i] + t * 128),
                                         (__attribute__((address_space(3))) void*)(st + 16384 + i * 1024), 16, 0, 0);
    }
  };
  // fragments: lane (l31, hi) reads row l31 of its wave's 32-row block, k chunk 2 kk + hi (kk = 0 .. 7), swizzled by the row
  const int fa = (32 * wm + l31) * 256, fw = 16384 + (32 * wn + l31) * 256;
  int foff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) foff[kk] = ((2 * kk + hi) ^ (l31 & 15)) << 4;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int t = 0; t < min(nk, S64_NS - 1); ++t) issue(t);
  for (int t = 0; t < nk; ++t) {
    // K tile t has landed when only the younger tile's pieces (8 per tile and wave) are outstanding (vector memory retires in order)
    if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // ... for every wave's pieces; and everybody is done with tile t - 1
    if (t + S64_NS - 1 < nk) issue(t + S64_NS - 1);        // into the stage tile t - 1 just left
    const char* sb = lds + (t % S64_NS) * S64_STAGE;
    bf16x8 xf[8], wf[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      xf[kk] = *reinterpret_cast<const bf16x8*>(sb + fa + foff[kk]);
      wf[kk] = *reinterpret_cast<const bf16x8*>(sb + fw + foff[kk]);
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) acc = mfma32(wf[kk], xf[kk], acc);   // (weights first: lane = row m, registers = columns n)
  }
  // epilogue: v_permlane32_swap gives a lane 8 consecutive columns of its row (gemm_bt.hip: pp_epilogue)
  const int m = m0 + 32 * wm + l31;
  const bool out_f32 = d.flags & GEMM_OUT_F32;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * t + e]), __float_as_uint(acc[8 * t + 4 + e]), false, false);
      v[e] = __uint_as_float(r[0]);
      v[4 + e] = __uint_as_float(r[1]);
    }
    const int n = n0 + 32 * wn + 16 * t + 8 * hi;
    if (m >= d.M || n >= d.N) continue;                    // (N % 64 == 0: whole 8-column groups)
    const float al = n < d.nsplit ? d.alpha_lo : d.alpha;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= al;
    if (d.flags & GEMM_BIAS_N) {
      const sk_u32x4 b4 = *reinterpret_cast<const sk_u32x4*>(d.bias + n);
      v[0] += bf16lo(b4.x); v[1] += bf16hi(b4.x); v[2] += bf16lo(b4.y); v[3] += bf16hi(b4.y);
      v[4] += bf16lo(b4.z); v[5] += bf16hi(b4.z); v[6] += bf16lo(b4.w); v[7] += bf16hi(b4.w);
    }
    if (d.flags & GEMM_GELU) {
#pragma unroll
      for (int e = 0; e < 8; e += 2) gelu_epi2(v[e], v[e + 1]);
    }
    if (d.flags & GEMM_RESIDUAL) {
      const sk_u32x4 r4 = *reinterpret_cast<const sk_u32x4*>(d.R + (int64_t)m * d.ldr + n);
      v[0] += bf16lo(r4.x); v[1] += bf16hi(r4.x); v[2] += bf16lo(r4.y); v[3] += bf16hi(r4.y);
      v[4] += bf16lo(r4.z); v[5] += bf16hi(r4.z); v[6] += bf16lo(r4.w); v[7] += bf16hi(r4.w);
    }
    if (out_f32) {
      float* cp = reinterpret_cast<float*>(d.C) + (int64_t)m * d.ldc + n;
      *reinterpret_cast<float4*>(cp) = float4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<float4*>(cp + 4) = float4{v[4], v[5], v[6], v[7]};
    } else {
      *reinterpret_cast<sk_u32x4*>(reinterpret_cast<bf16_t*>(d.C) + (int64_t)m * d.ldc + n) =
          sk_u32x4{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
    }
  }
}

}  // namespace

// Returns 1 when the product was launched here, 0 when it is not this kernel's (the caller goes on), < 0 on error.  `d` validated by
// gemm_bf16 (GEMM_VEC_OK resolved).  Taken: one batch entry, 64 < M <= 256 rows, row-major operands, K a multiple of 128 with >= 16 K
// tiles of 64, N = 2048 .. 4096 in whole 64-column strips (about one workgroup per CU; wider products fill the chip with the big-tile
// kernel's slices), 16-byte epilogue accesses.  Needs no scratch.
int gemm_skinny_try(const GemmDesc& d, hipStream_t stream) {
  const Options& o = opts();
  if (!o.gemm_skinny || o.gemm_tile != 0 || o.gemm_big != 0 || o.gemm_splitk != 0) return 0;   // (forced choices keep their kernels)
  if (d.nz != 1 || d.M <= 64 || d.M > 256 || (d.N & 63) || (d.K & 127) || d.ldbk) return 0;
  if (d.flags & (GEMM_BIAS_M | GEMM_A_KMAJOR | GEMM_B_KMAJOR | GEMM_SWIGLU) || !(d.flags & GEMM_VEC_OK) || d.vt) return 0;
  const bool f32 = d.flags & GEMM_OUT_F32;
  if (((uintptr_t)d.C & 15) || (d.ldc & (f32 ? 3 : 7))) return 0;
  if ((d.flags & GEMM_BIAS_N) && ((uintptr_t)d.bias & 15)) return 0;
  if ((d.flags & GEMM_RESIDUAL) && (((uintptr_t)d.R & 15) || (d.ldr & 7))) return 0;
  const int nstrips = d.N >> 6;
  if ((d.K >> 6) < 16 || nstrips < 32 || nstrips > 64) return 0;
  const int mtiles = (d.M + 63) >> 6;
  const bool mubuf = o.gemm_skinny == 2 && (int64_t)d.M * d.lda < (1ll << 29) && (int64_t)d.N * d.ldb < (1ll << 29);
  if (mubuf) hipLaunchKernelGGL(gemm_skinny64_kernel<true>, dim3(nstrips * mtiles), dim3(256), S64_NS * S64_STAGE, stream, d, nstrips, mtiles);
  else hipLaunchKernelGGL(gemm_skinny64_kernel<false>, dim3(nstrips * mtiles), dim3(256), S64_NS * S64_STAGE, stream, d, nstrips, mtiles);
  return launch_status() == U2_OK ? 1 : U2_ERR_LAUNCH;
}

}  // namespace u2
