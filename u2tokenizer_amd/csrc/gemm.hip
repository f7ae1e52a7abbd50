// bf16 MFMA GEMM for gfx950:  C[z][m][n] = epilogue( alpha * sum_k A[z][m][k] * B[z][n][k] )
//
// Both operands are K-contiguous ("NT" form): A is an activation matrix [M][K], B is a weight in
// nn.Linear layout [N][K] (or K/V^T/... for the attention products).  This is the workhorse of the
// hot path: every nn.Linear of the ViT blocks (reference vit.py:100-105 via MONAI SABlock/MLPBlock),
// the SPP MLP (spatial_pooling_projector.py:22-28), every wq/wk/wv/dense of the tokenizer
// (rma.py:13-16, tta.py:16-19), the DiffTS score/aggregation products (svr.py:105-115) and the
// QK^T / PV products of the tokenizer attention (rma.py:61,73; tta.py:56,59).
//
// Design (CDNA4):
//  * 256 threads = 4 wave64 in a 2x2 arrangement; block tile BMxBNx64, wave tile (BM/2)x(BN/2) built
//    from v_mfma_f32_16x16x32_bf16.  The MFMA is issued with the WEIGHT fragment as the A operand and
//    the ACTIVATION fragment as the B operand, so a lane ends up holding 4 consecutive output columns
//    n of one output row m -> 8-byte bf16 / 16-byte fp32 stores and vector bias/residual loads.
//  * LDS tile = [rows][64 bf16] (128 B per row), 16-byte chunks XOR-swizzled with (row & 7) so the
//    ds_read_b128 fragment reads of a 16-lane group touch 16 distinct 16-B slots (conflict-free).
//  * Two LDS stages; global->LDS by LDS-DMA (global_load_lds_dwordx4, swizzle applied on the per-lane
//    SOURCE address because the DMA destination is lane-linear), issued as one burst ahead of the MFMAs
//    of the current K tile.  (Register staging, DMA pieces spread between the MFMAs and 64-byte rows
//    (BK = 32) were measured 10-25 % slower in round 1 -- profiles/r01_gemm_pp_study.log -- and removed.)
//  * XCD-aware, M-grouped tile order so that the 8 private L2s each see a compact band of tiles.
#include <type_traits>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include "kernels.h"
#include "rows16.h"

namespace u2 {

__device__ uint4 g_zero16;  // zero-initialised; K-tail chunks of the LDS-DMA path read from here

// LDS chunk swizzle: XOR value for the 16-byte chunk index of `row` (128-B rows, 8 chunks): row & 7.
template <int BK>
__device__ __forceinline__ int lds_swz(int row) {
  static_assert(BK == 64, "BK");
  return row & 7;
}

// K-major operands (TA / TB; the weight-gradient and input-gradient products of the training path, whose operands would
// otherwise be transposed in HBM first): the operand is stored [K][rows] (row index = k, the M resp. N index contiguous).
// Its LDS tile is [64 k][BX] with 16-byte chunks (8 consecutive m) rotated inside the k row by kmaj_rot(k), and the MFMA
// fragment (8 consecutive k of one m) is gathered by two ds_read_b64_tr_b16: a 16-lane group reads a [4 k][16 m] block --
// lane a supplies the address of row a >> 2, 8-byte piece a & 3 and receives column a of the four rows
// (tools/ubench/tr_read.hip prints the instruction's mapping) -- so a lane ends up with k = 8 (lane >> 4) + {0..7} of column
// lane & 15, the same k-slot order as a K-contiguous fragment (the two forms mix freely as A and B).
// The rotation keeps the 16 chunks that the two 16-lane groups of a half wave touch (rows k..k+3 and k+8..k+11, two chunks
// each) in 16 different 16-byte slots of the 256-byte bank row.
template <int CPRX>  // chunks per k row: 16 (128-wide tile) or 8 (64-wide tile, two k rows per bank row)
__device__ __forceinline__ int kmaj_rot(int k) {
  if constexpr (CPRX == 16) return 2 * (k & 3) + 8 * ((k >> 3) & 1);
  else return 2 * ((k >> 1) & 1) + 4 * ((k >> 3) & 1);
}

typedef short v4s_t __attribute__((ext_vector_type(4)));

template <int BM, int BN, bool TA = false, bool TB = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(GemmDesc d) {
  constexpr int BK = 64;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int CPR = BK / 8;                              // 16-B chunks per tile row
  constexpr int ROWB = BK * 2;                             // bytes per tile row
  constexpr int CA = BM * CPR / 256, CB = BN * CPR / 256;  // 16-B chunks per thread per tile
  constexpr int KSTEPS = BK / 32;
  constexpr int STAGE = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile order: XCD-contiguous bands, then groups of 8 m-tiles ----
  const int ntiles = d.tiles_m * d.tiles_n;
  int pid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * d.tiles_n;
  const int group = pid / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(d.tiles_m - first_m, GROUP_M);
  const int tm = first_m + (pid % per_group) % gsz;
  const int tn = (pid % per_group) / gsz;
  const int bm0 = tm * BM, bn0 = tn * BN;

  const int z = blockIdx.y;
  const int zb = z / d.nbh, zh = z - zb * d.nbh;
  const bf16_t* __restrict__ A = d.A + zb * d.sAb + zh * d.sAh;
  const bf16_t* __restrict__ B = d.B + zb * d.sBb + zh * d.sBh;

  // ---- per-thread global source pointers (row clamped, swizzled chunk folded in) ----
  // K-contiguous operand: chunk c of the tile = (row c / 8, k chunk c % 8); K-major operand: (k row c / CPRX, slot c % CPRX)
  // holding the 8 columns  ((slot - rot(k)) mod CPRX) * 8.  ka / kb: k offset inside a K tile (tail mask); oka / okb: the
  // chunk exists at all (columns past M / N of a K-major operand read the zero chunk).
  constexpr int CPRA = BM / 8, CPRB = BN / 8;
  const bf16_t* pa[CA];
  const bf16_t* pb[CB];
  int ka[CA], kb[CB];
  bool oka[CA], okb[CB];
  // MUBUF form (d.mubuf; round 6): byte offsets from this batch entry's operand base, one raw descriptor per operand, the K advance in the
  // scalar offset.  (Beside waves that issue MFMAs -- here: the other workgroup of the CU -- the FLAT-encoded global_load_lds stages a
  // third of what buffer_load ... lds does: tools/ubench/stage_bw.hip, profiles/r06_stage_bw.log.)  A chunk that does not exist (K tail,
  // columns past M / N of a K-major operand) takes the lane offset 2^31: past num_records whichever way the range check reads it -> zeros.
  int va[CA], vb[CB];
  const bool mubuf = d.mubuf != 0;
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int c = i * 256 + tid;
    if constexpr (TA) {
      const int krow = c / CPRA, cc = ((c % CPRA) - kmaj_rot<CPRA>(krow)) & (CPRA - 1);
      ka[i] = krow;
      oka[i] = bm0 + cc * 8 < d.M;
      pa[i] = A + (int64_t)krow * d.lda + bm0 + cc * 8;
    } else {
      const int row = c / CPR, gc = (c % CPR) ^ lds_swz<BK>(row);
      const int grow = min(bm0 + row, d.M - 1);
      ka[i] = gc * 8;
      oka[i] = true;
      pa[i] = A + (int64_t)grow * d.lda + gc * 8;
    }
    va[i] = (int)((pa[i] - A) * 2);
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int c = i * 256 + tid;
    if constexpr (TB) {
      const int krow = c / CPRB, cc = ((c % CPRB) - kmaj_rot<CPRB>(krow)) & (CPRB - 1);
      kb[i] = krow;
      okb[i] = bn0 + cc * 8 < d.N;
      pb[i] = B + (int64_t)krow * d.ldb + bn0 + cc * 8;
    } else {
      const int row = c / CPR, gc = (c % CPR) ^ lds_swz<BK>(row);
      const int grow = min(bn0 + row, d.N - 1);
      kb[i] = gc * 8;
      okb[i] = true;
      pb[i] = B + (int64_t)grow * d.ldb + gc * 8;
    }
    vb[i] = (int)((pb[i] - B) * 2);
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = base + (lane & 15) with bases multiples of 16, so swz(row) == swz(lane & 15)
  const int frow = (lane & 15) * ROWB;
  int foff[KSTEPS];
#pragma unroll
  for (int kk = 0; kk < KSTEPS; ++kk) foff[kk] = (((kk * 4 + (lane >> 4)) ^ lds_swz<BK>(lane & 15)) << 4);
  // K-major fragments: lane a = lane & 15 of its 16-lane group addresses k row 8 (lane >> 4) + (a >> 2) (+ 4 for the second
  // read, + 32 per k step), 8-byte piece a & 3 of the fragment's two chunks; rot() of that row does not depend on the +4 /
  // +32, so the per-lane part is one offset per fragment.
  int fta[TA ? MI : 1], ftb[TB ? NI : 1];
  {
    const int a = lane & 15, krow = 8 * (lane >> 4) + (a >> 2);
    if constexpr (TA) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int cc = (wm * WM + mi * 16) / 8 + ((a & 3) >> 1);
        fta[mi] = (krow * CPRA + ((cc + kmaj_rot<CPRA>(krow)) & (CPRA - 1))) * 16 + (a & 1) * 8;
      }
    }
    if constexpr (TB) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int cc = (wn * WN + ni * 16) / 8 + ((a & 3) >> 1);
        ftb[ni] = (krow * CPRB + ((cc + kmaj_rot<CPRB>(krow)) & (CPRB - 1))) * 16 + (a & 1) * 8;
      }
    }
  }

  // elements between consecutive K tiles of a B row: BK for row-major weights, N * BK for "K-tile-major" packed ones
  // ([K / 64][N][64]: the 64 x 64 tile a workgroup stages per K step is one contiguous 8 KB block); K-major: BK rows
  [[maybe_unused]] const int64_t kadv_a = BK * d.lda;
  const int64_t kadv_b = TB ? BK * d.ldb : (d.ldbk ? d.ldbk : BK);
  const int nkt_all = (d.K + BK - 1) / BK;
  const int kt0 = (d.ksplit > 1) ? (int)blockIdx.z * d.kt_per : 0;   // split-K: this slice's K tiles
  const int nkt = (d.ksplit > 1) ? min(nkt_all, kt0 + d.kt_per) : nkt_all;
  auto dma = [&](int kt, int buf) {
    const int k0 = kt * BK;
    char* s = lds + buf * STAGE;
    if (mubuf) {
      const int sa = (int)((TA ? (int64_t)kt * kadv_a : (int64_t)k0) * 2), sb = (int)((int64_t)kt * kadv_b * 2);
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const bool ok = (TA ? oka[i] : true) && k0 + ka[i] < d.K;
        lds_dma_mubuf16(A, s + (i * 256 + wave * 64) * 16, ok ? va[i] : (int)0x80000000, sa);
      }
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        const bool ok = (TB ? okb[i] : true) && k0 + kb[i] < d.K;
        lds_dma_mubuf16(B, s + BM * ROWB + (i * 256 + wave * 64) * 16, ok ? vb[i] : (int)0x80000000, sb);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const void* src;
      if constexpr (TA) src = (oka[i] && k0 + ka[i] < d.K) ? (const void*)(pa[i] + (int64_t)kt * kadv_a) : (const void*)&g_zero16;
      else src = (k0 + ka[i] < d.K) ? (const void*)(pa[i] + k0) : (const void*)&g_zero16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(s + (i * 256 + wave * 64) * 16),
                                       16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const void* src;
      if constexpr (TB) src = (okb[i] && k0 + kb[i] < d.K) ? (const void*)(pb[i] + (int64_t)kt * kadv_b) : (const void*)&g_zero16;
      else src = (k0 + kb[i] < d.K) ? (const void*)(pb[i] + (int64_t)kt * kadv_b) : (const void*)&g_zero16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(s + BM * ROWB + (i * 256 + wave * 64) * 16),
                                       16, 0, 0);
    }
  };
  auto compute = [&](int buf) {
    const char* tA = lds + buf * STAGE;                    // A tile base ([BM][64] or [64][BM])
    const char* tB = lds + buf * STAGE + BM * ROWB;        // B tile base (both forms are (BM resp. BN) * 128 bytes)
    const char* sA = tA + (wm * WM) * ROWB + frow;
    const char* sB = tB + (wn * WN) * ROWB + frow;
    typedef __attribute__((address_space(3))) v4s_t* lds_v4;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      bf16x8 xf[MI], wf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        if constexpr (TA) {
          const char* q = tA + fta[mi] + kk * (32 * CPRA * 16);
          const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)q);
          const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q + 4 * CPRA * 16));
          xf[mi] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        } else {
          xf[mi] = *reinterpret_cast<const bf16x8*>(sA + mi * 16 * ROWB + foff[kk]);
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if constexpr (TB) {
          const char* q = tB + ftb[ni] + kk * (32 * CPRB * 16);
          const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)q);
          const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q + 4 * CPRB * 16));
          wf[ni] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        } else {
          wf[ni] = *reinterpret_cast<const bf16x8*>(sB + ni * 16 * ROWB + foff[kk]);
        }
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = mfma16(wf[ni], xf[mi], acc[mi][ni]);
    }
  };

  dma(kt0, kt0 & 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = kt0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) dma(kt + 1, (kt + 1) & 1);
    compute(kt & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n0..n0+3], m = tile row (lane & 15), n0 = 4 * (lane >> 4) ----
  if (d.ksplit > 1) {  // split-K slice: raw fp32 sums, the reduce kernel does the rest
    float* P = d.partial + ((int64_t)blockIdx.z * d.nz + z) * (int64_t)d.M * d.N;
    const int m_b = bm0 + wm * WM + (lane & 15), n_b = bn0 + wn * WN + (lane >> 4) * 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m_b + mi * 16;
      if (m >= d.M) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n0 = n_b + ni * 16;
        float* pp = P + (int64_t)m * d.N + n0;
        if (n0 + 3 < d.N && (d.N & 3) == 0) {
          *reinterpret_cast<float4*>(pp) = float4{acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + r < d.N) pp[r] = acc[mi][ni][r];
        }
      }
    }
    return;
  }
  const bool out_f32 = d.flags & GEMM_OUT_F32;
  char* Cz = reinterpret_cast<char*>(d.C) + (zb * d.sCb + zh * d.sCh) * (out_f32 ? 4 : 2);
  const bf16_t* Rz = (d.flags & GEMM_RESIDUAL) ? d.R + zb * d.sRb + zh * d.sRh : nullptr;
  const int m_base = bm0 + wm * WM + (lane & 15);
  const int n_base = bn0 + wn * WN + (lane >> 4) * 4;

  // Fast path: every column group of this block is inside N and all vector accesses are legal.  The flag set is
  // resolved ONCE into compile-time booleans (the common combinations), so the unrolled body is branch-free.
  auto epi_vec = [&](auto BN_, auto GELU_, auto RES_, auto F32_) {
    float4 bn[NI];
    if constexpr (decltype(BN_)::value) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const uint2 b2 = *reinterpret_cast<const uint2*>(d.bias + n_base + ni * 16);
        bn[ni] = float4{bf16lo(b2.x), bf16hi(b2.x), bf16lo(b2.y), bf16hi(b2.y)};
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m_base + mi * 16;
      if (m >= d.M) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n0 = n_base + ni * 16;
        const float al = n0 < d.nsplit ? d.alpha_lo : d.alpha;
        float v0 = acc[mi][ni][0] * al, v1 = acc[mi][ni][1] * al, v2 = acc[mi][ni][2] * al, v3 = acc[mi][ni][3] * al;
        if constexpr (decltype(BN_)::value) { v0 += bn[ni].x; v1 += bn[ni].y; v2 += bn[ni].z; v3 += bn[ni].w; }
        if constexpr (decltype(GELU_)::value) { gelu_epi2(v0, v1); gelu_epi2(v2, v3); }
        if constexpr (decltype(RES_)::value) {
          const uint2 r2 = *reinterpret_cast<const uint2*>(Rz + (int64_t)m * d.ldr + n0);
          v0 += bf16lo(r2.x); v1 += bf16hi(r2.x); v2 += bf16lo(r2.y); v3 += bf16hi(r2.y);
        }
        if constexpr (decltype(F32_)::value)
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cz) + (int64_t)m * d.ldc + n0) = float4{v0, v1, v2, v3};
        else
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cz) + (int64_t)m * d.ldc + n0) =
              uint2{pack2_bf16(v0, v1), pack2_bf16(v2, v3)};
      }
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  const int ef = d.flags & (GEMM_BIAS_N | GEMM_BIAS_M | GEMM_GELU | GEMM_RESIDUAL | GEMM_OUT_F32);
  if ((d.flags & GEMM_VEC_OK) && bn0 + BN <= d.N && !(ef & GEMM_BIAS_M)) {
    switch (ef) {
      case 0: epi_vec(F_{}, F_{}, F_{}, F_{}); return;
      case GEMM_OUT_F32: epi_vec(F_{}, F_{}, F_{}, T_{}); return;
      case GEMM_BIAS_N: epi_vec(T_{}, F_{}, F_{}, F_{}); return;
      case GEMM_BIAS_N | GEMM_GELU: epi_vec(T_{}, T_{}, F_{}, F_{}); return;
      case GEMM_BIAS_N | GEMM_RESIDUAL: epi_vec(T_{}, F_{}, T_{}, F_{}); return;
      case GEMM_RESIDUAL: epi_vec(F_{}, F_{}, T_{}, F_{}); return;
      default: break;
    }
  }
  // Generic path (tails, unaligned, bias along M, rare flag sets): scalar, fully predicated.  The loops must stay
  // fully unrolled: a runtime index into acc[][] would move the accumulators to scratch memory.
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m_base + mi * 16;
    if (m >= d.M) continue;
    const float bm_v = (d.flags & GEMM_BIAS_M) ? bf16_to_f32(d.bias[m]) : 0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_base + ni * 16 + r;
        if (n >= d.N) break;
        float x = acc[mi][ni][r] * (n < d.nsplit ? d.alpha_lo : d.alpha) + bm_v;
        if (d.flags & GEMM_BIAS_N) x += bf16_to_f32(d.bias[n]);
        if (d.flags & GEMM_GELU) x = gelu_epi(x);
        if (Rz) x += bf16_to_f32(Rz[(int64_t)m * d.ldr + n]);
        if (out_f32) reinterpret_cast<float*>(Cz)[(int64_t)m * d.ldc + n] = x;
        else reinterpret_cast<bf16_t*>(Cz)[(int64_t)m * d.ldc + n] = f32_to_bf16(x);
      }
    }
  }
}

// out[z][m][n] = epilogue(alpha * sum_s partial[s][z][m][n]); one thread per 4 consecutive n
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmDesc d) {
  const int64_t n4 = (d.N + 3) >> 2;
  const int64_t total = (int64_t)d.nz * d.M * n4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n0 = (int)(idx % n4) * 4;
  const int64_t zm = idx / n4;
  const int m = (int)(zm % d.M), z = (int)(zm / d.M);
  const int64_t slice = (int64_t)d.nz * d.M * d.N;
  const float* P = d.partial + ((int64_t)z * d.M + m) * d.N + n0;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const bool full = (n0 + 3 < d.N) && (d.N & 3) == 0;
  for (int s = 0; s < d.ksplit; ++s) {
    if (full) {
      const float4 t = *reinterpret_cast<const float4*>(P + s * slice);
      v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    } else {
      for (int r = 0; r < 4; ++r)
        if (n0 + r < d.N) v[r] += P[s * slice + r];
    }
  }
  const int zb = z / d.nbh, zh = z - zb * d.nbh;
  const bool out_f32 = d.flags & GEMM_OUT_F32;
  const float bm_v = (d.flags & GEMM_BIAS_M) ? bf16_to_f32(d.bias[m]) : 0.f;
  const bf16_t* Rz = (d.flags & GEMM_RESIDUAL) ? d.R + zb * d.sRb + zh * d.sRh + (int64_t)m * d.ldr : nullptr;
  char* Cz = reinterpret_cast<char*>(d.C) + (zb * d.sCb + zh * d.sCh + (int64_t)m * d.ldc) * (out_f32 ? 4 : 2);
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + r;
    if (n >= d.N) break;
    float x = v[r] * (n < d.nsplit ? d.alpha_lo : d.alpha) + bm_v;
    if (d.flags & GEMM_BIAS_N) x += bf16_to_f32(d.bias[n]);
    if (d.flags & GEMM_GELU) x = gelu_epi(x);
    if (Rz) x += bf16_to_f32(Rz[n]);
    if (out_f32) reinterpret_cast<float*>(Cz)[n] = x;
    else reinterpret_cast<bf16_t*>(Cz)[n] = f32_to_bf16(x);
  }
}

// ------------------------------------------------------------------------------------------------ M <= 16 rows
// The ViT keeps its 8 cls rows behind the 16384 patch rows (pipeline.hip); the big-tile kernel takes the 64 full 256-row
// tiles and leaves those 8 rows to a second launch -- 36 of them per volume (qkv, out-proj, fc2 of 12 blocks).  On the
// 64 x 64 tile kernel such a product is a chain of K / 64 dependent stage-and-wait steps on a few dozen workgroups:
// ~20 us each, latency-bound.  Here a workgroup owns 16 output columns; its NW waves split K, every wave loads its
// weight fragments (the MFMA A operand: 16 rows x 64 contiguous bytes per instruction) and activation fragments straight
// from memory -- all loads of a wave are issued before its first MFMA, so the whole product costs about one memory latency
// -- and the waves' partial tiles are added through LDS in a fixed order (bit-repeatable).  v_mfma_f32_16x16x32_bf16 with
// the weights as A: a lane ends up with 4 consecutive output columns of one row, as in the tile kernels.
// PAIR (GEMM_SWIGLU: B = [gate rows | up rows], N = 2 I, C (M, I) = bf16(silu(gate)) * up -- the form the big-tile kernel has for
// the prefill): the workgroup's 16 weight rows are 8 gate rows and the SAME 8 up rows, so the lanes of column groups 0 / 1 hold
// the gates and those of groups 2 / 3 (32 lanes further) the matching ups; the decode step's SwiGLU launch goes away.
template <int NW, bool PAIR = false>
__global__ __launch_bounds__(NW * 64) void gemm_rows16_kernel(GemmDesc d) {
  __shared__ float red[NW][64][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nsteps = d.K >> 5;  // K % 32 == 0 (launcher)
  const int per = (nsteps + NW - 1) / NW, s0 = wv * per, s1 = min(nsteps, s0 + per);
  const int I2 = d.N >> 1;
  const int nrow = PAIR ? min((l15 < 8 ? 0 : I2) + (int)blockIdx.x * 8 + (l15 & 7), d.N - 1) : min(n0 + l15, d.N - 1);
  const int mrow = min(l15, d.M - 1);
  const bf16_t* wp = d.B + (int64_t)nrow * d.ldb + g * 8;
  const bf16_t* xp = d.A + (int64_t)mrow * d.lda + g * 8;
  const f32x4 acc = rows16_slice(wp, xp, s0, s1);   // (rows16.h: shared with the big-tile kernel's in-launch tail)
  // lane holds C[m = l15][n = n0 + 4 g + r]
  red[wv][lane][0] = acc[0]; red[wv][lane][1] = acc[1]; red[wv][lane][2] = acc[2]; red[wv][lane][3] = acc[3];
  __syncthreads();
  if (wv != 0 || (!PAIR && l15 >= d.M)) return;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < NW; ++w)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += red[w][lane][r];
  const int m = l15;
  if constexpr (PAIR) {  // (wave 0, all 64 lanes: lanes of rows >= M carry copies of row M - 1 and write nothing)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gate = bf16_to_f32(f32_to_bf16(v[r] * d.alpha));
      const float up = bf16_to_f32(f32_to_bf16(__shfl_xor(v[r], 32, 64) * d.alpha));  // column group g + 2 of the same row
      const int n = (int)blockIdx.x * 8 + 4 * g + r;
      if (g < 2 && l15 < d.M && n < I2) {
        const float sg = gate / (1.0f + __expf(-gate));
        reinterpret_cast<bf16_t*>(d.C)[(int64_t)m * d.ldc + n] = f32_to_bf16(bf16_to_f32(f32_to_bf16(sg)) * up);
      }
    }
    return;
  }
  rows16_store(d, v, m, n0 + 4 * g, reinterpret_cast<char*>(d.C), d.R);
}

// M <= 16, one batch entry, K-contiguous operands, K % 32 == 0: returns 1 when launched, 0 when not applicable
static int gemm_rows16_try(const GemmDesc& d, hipStream_t stream) {
  if (d.M > 16 || d.nz != 1 || (d.K & 31) || d.ldbk || (d.flags & (GEMM_A_KMAJOR | GEMM_B_KMAJOR))) return 0;
  const int nsteps = d.K >> 5;
  dim3 grid((unsigned)cdiv(d.N, 16));
  if (d.flags & GEMM_SWIGLU) {  // (validated by gemm_bf16: alone, N = 2 I)
    if (nsteps >= 64) hipLaunchKernelGGL((gemm_rows16_kernel<16, true>), grid, dim3(1024), 0, stream, d);
    else if (nsteps >= 16) hipLaunchKernelGGL((gemm_rows16_kernel<8, true>), grid, dim3(512), 0, stream, d);
    else hipLaunchKernelGGL((gemm_rows16_kernel<4, true>), grid, dim3(256), 0, stream, d);
    return launch_status() == U2_OK ? 1 : U2_ERR_LAUNCH;
  }
  if (nsteps >= 64) hipLaunchKernelGGL((gemm_rows16_kernel<16>), grid, dim3(1024), 0, stream, d);
  else if (nsteps >= 16) hipLaunchKernelGGL((gemm_rows16_kernel<8>), grid, dim3(512), 0, stream, d);
  else hipLaunchKernelGGL((gemm_rows16_kernel<4>), grid, dim3(256), 0, stream, d);
  return launch_status() == U2_OK ? 1 : U2_ERR_LAUNCH;
}

template <int BM, int BN>
static int launch_tile(GemmDesc d, hipStream_t stream) {
  d.tiles_m = (int)cdiv(d.M, BM);
  d.tiles_n = (int)cdiv(d.N, BN);
  dim3 grid(d.tiles_m * d.tiles_n, d.nz, d.ksplit > 1 ? d.ksplit : 1);
  constexpr int smem = 2 * (BM + BN) * 64 * 2;
  const bool ta = d.flags & GEMM_A_KMAJOR, tb = d.flags & GEMM_B_KMAJOR;
  {  // MUBUF pieces when every byte offset of a batch entry's operands (K tile advance included) stays below 2^31
    const int64_t ktiles = cdiv(d.K, 64) + 1;
    const int64_t ea = (ta ? ktiles * 64 * d.lda + d.M : (int64_t)d.M * d.lda + ktiles * 64) * 2;
    const int64_t eb = (tb ? ktiles * 64 * d.ldb + d.N : d.ldbk ? ktiles * d.ldbk + (int64_t)d.N * d.ldb : (int64_t)d.N * d.ldb + ktiles * 64) * 2;
    d.mubuf = (opts().gemm_mubuf && ea < (1ll << 31) - 65536 && eb < (1ll << 31) - 65536) ? 1 : 0;
  }
  if (ta) hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM, BN, true, true>), grid, dim3(256), smem, stream, d);
  else if (tb) hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM, BN, false, true>), grid, dim3(256), smem, stream, d);
  else hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM, BN>), grid, dim3(256), smem, stream, d);
  return launch_status();
}

int gemm_bf16(GemmDesc d, hipStream_t stream) {
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.nz <= 0 || d.nz > 65535) return U2_ERR_ARG;
  if (!d.A || !d.B || !d.C) return U2_ERR_ARG;
  if (d.nsplit < 0 || (d.nsplit & 15) || (d.nsplit && (d.flags & GEMM_SWIGLU))) return U2_ERR_ARG;
  if (d.nbh <= 0) d.nbh = 1;
  static const bool trace = getenv("U2TOK_GEMM_TRACE") != nullptr;  // diagnostics: one line per product on stderr
  if (trace) fprintf(stderr, "gemm M=%d N=%d K=%d nz=%d flags=0x%x lda=%d ldb=%d ldc=%d\n", d.M, d.N, d.K, d.nz, d.flags, (int)d.lda, (int)d.ldb, (int)d.ldc);
  // 16-byte chunked loads along the contiguous dimension of each operand (K, or M / N of a K-major one): that dimension,
  // leading dims and batch strides must keep every chunk aligned
  if (d.flags & GEMM_SWIGLU) {  // gate | up pair product with SiLU(gate) * up in the epilogue: the 256 x 192-tile kernel only
    const int64_t I = d.N >> 1;
    if ((d.flags & ~GEMM_SWIGLU) || d.nz != 1 || d.ldbk || (d.N & 1) || (I & 15) || (d.K & 63) || d.ldc < I || (d.ldc & 7) ||
        (d.lda & 7) || (d.ldb & 7) || (((uintptr_t)d.A | (uintptr_t)d.B | (uintptr_t)d.C) & 15))
      return U2_ERR_ARG;
    ProfScope ps(PROF_GEMM, 2.0 * d.M * d.N * d.K, stream, 2.0 * d.M * d.K + 2.0 * d.N * d.K + 2.0 * d.M * I);
    if (d.M <= 16 && opts().gemm_tile == 0) {  // a few rows (decode steps): the few-rows kernel has the pair form too
      const int r = gemm_rows16_try(d, stream);
      if (r != 0) return r > 0 ? U2_OK : r;
    }
    const int big = gemm_big_try(d, stream);
    return big > 0 ? U2_OK : (big < 0 ? big : U2_ERR_ARG);
  }
  const bool ta = d.flags & GEMM_A_KMAJOR, tb = d.flags & GEMM_B_KMAJOR;
  if ((ta && !tb) || (tb && d.ldbk)) return U2_ERR_ARG;
  if ((ta ? d.M : d.K) & 7) return U2_ERR_ARG;
  if ((tb ? d.N : d.K) & 7) return U2_ERR_ARG;
  if ((ta && d.lda < d.M) || (tb && d.ldb < d.N)) return U2_ERR_ARG;
  if ((d.lda & 7) || (d.ldb & 7) || (d.sAb & 7) || (d.sAh & 7) || (d.sBb & 7) || (d.sBh & 7)) return U2_ERR_ARG;
  if (((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15)) return U2_ERR_ARG;
  if ((d.flags & (GEMM_BIAS_N | GEMM_BIAS_M)) && !d.bias) return U2_ERR_ARG;
  if ((d.flags & GEMM_RESIDUAL) && !d.R) return U2_ERR_ARG;
  const bool out_f32 = d.flags & GEMM_OUT_F32;
  bool vec = (d.ldc % 4 == 0) && (d.sCb % 4 == 0) && (d.sCh % 4 == 0) &&
             (((uintptr_t)d.C & (out_f32 ? 15 : 7)) == 0);
  if (d.flags & GEMM_BIAS_N) vec = vec && (((uintptr_t)d.bias & 7) == 0);
  if (d.flags & GEMM_RESIDUAL)
    vec = vec && (d.ldr % 4 == 0) && (d.sRb % 4 == 0) && (d.sRh % 4 == 0) && (((uintptr_t)d.R & 7) == 0);
  d.flags = vec ? (d.flags | GEMM_VEC_OK) : (d.flags & ~GEMM_VEC_OK);

  const double zA = (d.sAb || d.sAh) ? d.nz : 1, zB = (d.sBb || d.sBh) ? d.nz : 1;  // a batch-shared operand is read once
  ProfScope ps(PROF_GEMM, 2.0 * d.M * d.N * d.K * d.nz, stream,
               2.0 * d.M * d.K * zA + 2.0 * d.N * d.K * zB +
                   d.nz * ((out_f32 ? 4.0 : 2.0) * d.M * d.N + ((d.flags & GEMM_RESIDUAL) ? 2.0 * d.M * d.N : 0.0)));
  if (d.ldbk == 0) {  // (the 256-wide-tile kernels read row-major B only)
    const int sk = gemm_skinny_try(d, stream);  // <= 256 rows against a cold E x E weight: all rows x 64 columns per workgroup, slices combined in the launch
    if (sk != 0) return sk > 0 ? U2_OK : sk;
    const int big = gemm_big_try(d, stream);  // large products: the big-tile kernel (gemm_bt.hip)
    if (big != 0) return big > 0 ? U2_OK : big;
  }
  if (d.vt) return U2_ERR_ARG;  // (a transposed side output only exists in the big-tile kernel: ask gemm_vt_supported first)
  return gemm_classic(d, stream);
}

// 128^2 / 64^2 tile kernel above; `d` already validated (GEMM_VEC_OK resolved).
int gemm_classic(GemmDesc d, hipStream_t stream) {
  const Options& o = opts();
  if (o.gemm_tile == 0) {  // (a forced tile keeps the tile kernels: tests of their row tails)
    const int r = gemm_rows16_try(d, stream);
    if (r != 0) return r > 0 ? U2_OK : r;
    // <= 16 rows past a multiple of 128 in a many-row product (the ViT's GELU product: M = 16384 + 8 cls rows, 24 column
    // tiles): one more row of 128 x 128 tiles is 24 workgroups that start a SEVENTH round after six full ones (+ 16 %);
    // the few-rows kernel takes them instead
    const int rem = d.M & 127;
    if (d.nz == 1 && d.M >= 2048 && rem != 0 && rem <= 16 && !(d.K & 31) && !d.ldbk &&
        !(d.flags & (GEMM_A_KMAJOR | GEMM_B_KMAJOR | GEMM_BIAS_M))) {
      const bool f32 = d.flags & GEMM_OUT_F32;
      GemmDesc main = d, tail = d;
      main.M = d.M - rem;
      tail.M = rem;
      tail.A = d.A + (int64_t)main.M * d.lda;
      tail.C = reinterpret_cast<char*>(d.C) + (int64_t)main.M * d.ldc * (f32 ? 4 : 2);
      if (d.flags & GEMM_RESIDUAL) tail.R = d.R + (int64_t)main.M * d.ldr;
      const int e = gemm_classic(main, stream);
      if (e != U2_OK) return e;
      const int r2 = gemm_rows16_try(tail, stream);
      return r2 > 0 ? U2_OK : (r2 < 0 ? r2 : U2_ERR_ARG);
    }
  }
  int tile = o.gemm_tile;
  // "long K": weight-gradient products of the training path (dW = dY^T X: a small output, K = the 16392 token rows of the
  // ViT).  64 x 64 tiles fill the CUs there but run at ~0.35-0.4 PF/s (197 us for 3072 x 768 x 16392); 128 x 128 tiles with
  // K sliced over 5-8 workgroups keep the better tile and fill the machine.  No inference product has K >= 8192.
  bool longk = false;
  if (tile != 64 && tile != 128) {
    const int64_t big = cdiv(d.M, 128) * cdiv(d.N, 128) * d.nz;
    tile = (big >= 192) ? 128 : 64;  // fill 256 CUs; small-M weight-streaming shapes get 64^2 tiles
    if (tile == 64 && o.gemm_splitk == 0 && d.nz == 1 && d.M >= 512 && d.N >= 512 && d.K >= 8192) {
      tile = 128;
      longk = true;
    }
  }
  // split-K: a product with fewer workgroups than ~2 per CU runs one single-stage-prefetch K loop per CU and is
  // latency-bound (M = 256, N = K = 4096: 36 us, 0.24 PF/s).  Slicing K puts several workgroups on every CU.
  d.ksplit = 1;
  if (o.gemm_splitk >= 0) {
    const int64_t wgs = cdiv(d.M, tile) * cdiv(d.N, tile) * d.nz;
    const int nkt = (int)cdiv(d.K, 64);
    int s = o.gemm_splitk > 1 ? o.gemm_splitk : 0;
    // (128 x 128 tiles: two workgroups per CU are resident -> aim at 512; the decoder prefill's M = 1024 out / down
    //  projections are 256 tiles with K = 4096 / 12288: one K loop per CU with nothing to overlap it otherwise)
    if (s == 0 && d.nz == 1 && (wgs <= 320 || longk) && nkt >= 16)
      s = (int)std::min<int64_t>(8, std::min<int64_t>(nkt / 4, cdiv(longk ? 640 : (tile == 128 ? 512 : 1024), wgs)));
    if (s > 1) {
      const Scratch sc = ctx().scratch_of(stream);
      const size_t slice = (size_t)d.nz * d.M * d.N * sizeof(float);
      // partial sums cost HBM traffic: capped at 24 MB unless the K loop is long enough to dwarf it
      if (o.gemm_splitk <= 1)
        s = (int)std::min<size_t>(s, (longk ? sc.bytes : std::min<size_t>(sc.bytes, (tile == 128 ? 40u : 24u) << 20)) / slice);
      const size_t need = (size_t)s * slice;
      if (s > 1 && sc.p && need <= sc.bytes && d.nz <= 65535) {
        d.ksplit = s;
        d.kt_per = (int)cdiv(nkt, s);
        d.ksplit = (int)cdiv(nkt, d.kt_per);  // no empty slices
        d.partial = reinterpret_cast<float*>(sc.p);
      }
    }
    if (d.ksplit <= 1) d.ksplit = 1;
  }
  if (longk && d.ksplit == 1) tile = 64;  // no scratch for the slices: the tile that fills the CUs
  const int e = tile == 128 ? launch_tile<128, 128>(d, stream) : launch_tile<64, 64>(d, stream);
  if (e != U2_OK || d.ksplit == 1) return e;
  return gemm_splitk_reduce(d, stream);
}

int gemm_splitk_reduce(const GemmDesc& d, hipStream_t stream) {
  const int64_t total = (int64_t)d.nz * d.M * ((d.N + 3) >> 2);
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, d);
  return launch_status();
}

}  // namespace u2
