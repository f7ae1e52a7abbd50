// Big-tile bf16 MFMA GEMM for the large products of the hot path (same contract as gemm.hip:
//   C[z][m][n] = epilogue(alpha * sum_k A[z][m][k] * B[z][n][k]),  A = activations [M][K], B = nn.Linear weight [N][K]).
//
// 256 x 256 x 64 (NJ = 4) or 256 x 192 x 64 (NJ = 3) tiles, ONE persistent workgroup of 4 waves per CU, each wave a
// 128 x (32 NJ) output tile = 4 x NJ accumulators of v_mfma_f32_32x32x16_bf16 in AccVGPRs; the K loop is one generated
// asm block (gemm_bt_asm.inc, written by tools/gen_gemm_bt_asm.py, which documents the slot schedule).  With one wave
// per SIMD and a hand-placed instruction stream the matrix pipe only drains at the single barrier per K tile, every
// LDS-DMA piece and fragment read sits in the shadow of an MFMA, and a 256^2 tile needs half the LDS fill bandwidth
// per flop of the 128^2 kernel (the measured limit of the CU's L2 -> LDS path, ~50 B/clk, profiles/r01_stage_bw.log).
// The 8-wave "ping-pong" kernels this file grew out of (measured slower on every shape, profiles/r01_gemm_pp_study.log)
// were removed from the product in round 2; they are in the history at fc1ca7f.
//  * XCD-aware tile order (workgroup b runs on XCD b % 8): every XCD gets a compact band of tiles per round.
//  * persistent: grid = min(#tiles, #CUs); a workgroup walks tiles b, b + grid, ... and its K loop runs on from one
//    output tile into the next (the last iterations stage the next tile's first K tiles).
//  * round 4: THREE-stage forms of the same loop -- the ring form (NJ = 2: 256 x 128 tiles) and the deep forms (DEEP = 1 / 2: three
//    stages for A / for B, two for the other operand) -- because what bounds the two-stage K loop is the latency of its own
//    LDS-DMA stream ("Round 4: the big-tile GEMM"); gemm_big_try below says which form a product takes.
// Requirements checked by the launcher: K % 64 == 0, operands addressable with 32-bit byte offsets.
#include <algorithm>
#include <type_traits>
#include "kernels.h"
#include "rows16.h"

namespace u2 {

// tile geometry handed to the epilogue
template <int BN_>
struct BTCfg {
  static constexpr int BN = BN_;
  static constexpr int NI = BN / 64;  // 32-wide n fragments per wave (wave tile 128 x BN/2 as two 64-row halves)
};

// round r of workgroup `bid` -> (z, bm0, bn0).  Tiles of one round are remapped so that XCD x (= bid & 7) works
// on a contiguous range of logical ids; logical ids walk groups of 8 m-tiles column by column.
template <int BN>
__device__ __forceinline__ void pp_tile(const GemmDesc& d, int round, int gd, int bid, int total, int tiles_mn, int& z,
                                        int& bm0, int& bn0) {
  const int base = round * gd;
  const int n_r = min(gd, total - base);
  const int q = n_r >> 3, r = n_r & 7, xcd = bid & 7, idx = bid >> 3;
  const int pid = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  z = pid / tiles_mn;
  const int rem = pid - z * tiles_mn;
  const int GROUP_M = d.group_m;
  const int per_group = GROUP_M * d.tiles_n;
  const int group = rem / per_group, in_g = rem - group * per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(d.tiles_m - first_m, GROUP_M);
  bm0 = (first_m + in_g % gsz) * 256;
  bn0 = (in_g / gsz) * BN;
}

typedef unsigned int bt_u32x4 __attribute__((ext_vector_type(4)));

// Fast form of the epilogue below for the cases the big products are (tile inside C, bf16 output, bias[n] and / or residual):
// the bias vectors of the wave's columns were loaded BEFORE the K loop (`bpre`, NI x 2 x 16 bytes: their latency used to be
// exposed once per column group), all residual loads of a 64-row half are issued before the first use (they were one
// load -> wait -> store chain per 8 values: C and R may alias -- the ViT's residual products are in place --, so the compiler
// kept that order).  (Non-temporal stores were tried and are wrong here: 32-byte row pieces that bypass the L2's write
// combining -- the q|k|v product went 73 -> 106 us.)  profiles/r03_bt_epilogue_ablation.log has what the epilogue cost:
// 26 of 76 us of the ViT's q|k|v product, 13.5 of 35 us of its out-projection.
template <int NI, bool BIAS, bool RES>
__device__ __forceinline__ void bt_epilogue_fast(const GemmDesc& d, f32x16 (&acc)[2][NI], bf16_t* Cz, const bf16_t* Rz, int m_base,
                                                 int n_base, const bt_u32x4 (&bpre)[NI * 2]) {
  // one 32-row block (mi) at a time: its NI x 2 residual vectors in flight together, then per 8-column group
  // swap -> scale / bias / residual -> pack -> store straight from the accumulator registers (8 live values, no write-back)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    bt_u32x4 rr[NI][2];
    if constexpr (RES) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          rr[ni][t] = *reinterpret_cast<const bt_u32x4*>(Rz + (int64_t)(m_base + mi * 32) * d.ldr + n_base + ni * 32 + t * 16);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float v[8];
        const float al = n_base + ni * 32 + t * 16 < d.nsplit ? d.alpha_lo : d.alpha;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mi][ni][8 * t + e]),
                                                          __float_as_uint(acc[mi][ni][8 * t + 4 + e]), false, false);
          v[e] = __uint_as_float(r[0]) * al;
          v[4 + e] = __uint_as_float(r[1]) * al;
        }
        if constexpr (BIAS) {
          const bt_u32x4 b4 = bpre[ni * 2 + t];
          v[0] += bf16lo(b4.x); v[1] += bf16hi(b4.x); v[2] += bf16lo(b4.y); v[3] += bf16hi(b4.y);
          v[4] += bf16lo(b4.z); v[5] += bf16hi(b4.z); v[6] += bf16lo(b4.w); v[7] += bf16hi(b4.w);
        }
        if constexpr (RES) {
          const bt_u32x4 r4 = rr[ni][t];
          v[0] += bf16lo(r4.x); v[1] += bf16hi(r4.x); v[2] += bf16lo(r4.y); v[3] += bf16hi(r4.y);
          v[4] += bf16lo(r4.z); v[5] += bf16hi(r4.z); v[6] += bf16lo(r4.w); v[7] += bf16hi(r4.w);
        }
        const bt_u32x4 pk = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
        *reinterpret_cast<bt_u32x4*>(Cz + (int64_t)(m_base + mi * 32) * d.ldc + n_base + ni * 32 + t * 16) = pk;
      }
  }
}

// Epilogue of one 256 x BN tile for the calling wave.  The accumulator fragment of v_mfma_f32_32x32x16 leaves a lane
// with 4 consecutive n (register quad q) and its partner lane (+32) with the next 4; v_permlane32_swap exchanges
// quads between the half-waves so that every lane owns 8 CONSECUTIVE n of one row m: 16-byte bias / residual loads
// and bf16 stores, two float4 stores for fp32 output.
//   lane (l31, hi), pair t of fragment (mi, ni):  m = m_base + 32 mi,  n = n_tile + 32 ni + 16 t + 8 hi + [0, 8)
template <class CFG, int G, bool PAIR = false>
__device__ __forceinline__ void pp_epilogue(const GemmDesc& d, f32x16 (&acc)[2][CFG::NI], int z, int bm0, int bn0, int wm2,
                                            int wn2, int lane, const bt_u32x4 (&bpre)[CFG::NI * 2], bool fast) {
  constexpr int NI = CFG::NI, BN = CFG::BN;
  const int hi = lane >> 5, l31 = lane & 31;
  const bool out_f32 = d.flags & GEMM_OUT_F32;
  const int zb = z / d.nbh, zh = z - zb * d.nbh;
  char* Cz = reinterpret_cast<char*>(d.C) + (zb * d.sCb + zh * d.sCh) * (out_f32 ? 4 : 2);
  const bf16_t* Rz = (d.flags & GEMM_RESIDUAL) ? d.R + zb * d.sRb + zh * d.sRh : nullptr;
  const int m_base = bm0 + G * 128 + wm2 * 64 + l31;
  const int n_base = bn0 + wn2 * (BN / 2) + 8 * hi;
  if constexpr (!PAIR && NI == 3) {
    if (fast) {  // (uniform: the caller checked tile-inside-C, bf16 output and the flag set)
      bf16_t* C16 = reinterpret_cast<bf16_t*>(Cz);
      switch (d.flags & (GEMM_BIAS_N | GEMM_RESIDUAL)) {
        case GEMM_BIAS_N: bt_epilogue_fast<NI, true, false>(d, acc, C16, Rz, m_base, n_base, bpre); break;
        case GEMM_BIAS_N | GEMM_RESIDUAL: bt_epilogue_fast<NI, true, true>(d, acc, C16, Rz, m_base, n_base, bpre); break;
        case GEMM_RESIDUAL: bt_epilogue_fast<NI, false, true>(d, acc, C16, Rz, m_base, n_base, bpre); break;
        default: bt_epilogue_fast<NI, false, false>(d, acc, C16, Rz, m_base, n_base, bpre); break;
      }
      return;
    }
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mi][ni][8 * t + e]),
                                                          __float_as_uint(acc[mi][ni][8 * t + 4 + e]), false, false);
          acc[mi][ni][8 * t + e] = __uint_as_float(r[0]);
          acc[mi][ni][8 * t + 4 + e] = __uint_as_float(r[1]);
        }
  if constexpr (PAIR) {
    // SwiGLU-pair form (see the kernel): block ni of this wave holds gate columns (t = 0) and up columns (t = 1) of output
    // columns (bn0 / 2) + 16 (NI wn2 + ni) + 8 hi + [0, 8).  Rounding points of HF's LlamaMLP on bf16 tensors, the ones
    // swiglu_kernel (decoder.hip) keeps: gate, up and silu(gate) are bf16 values.
    const int I = (int)(d.N >> 1);
    bf16_t* C16 = reinterpret_cast<bf16_t*>(Cz);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int c0 = (bn0 >> 1) + 16 * (NI * wn2 + ni) + 8 * hi;
      if (c0 >= I) continue;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int m = m_base + mi * 32;
        if (m >= d.M) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float g = bf16_to_f32(f32_to_bf16(acc[mi][ni][e] * d.alpha));
          const float u = bf16_to_f32(f32_to_bf16(acc[mi][ni][8 + e] * d.alpha));
          const float sg = g / (1.0f + __expf(-g));
          o[e] = bf16_to_f32(f32_to_bf16(sg)) * u;
        }
        *reinterpret_cast<uint4*>(C16 + (int64_t)m * d.ldc + c0) =
            uint4{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]), pack2_bf16(o[4], o[5]), pack2_bf16(o[6], o[7])};
      }
    }
    return;
  }
  // FULL: the tile lies inside C (no row / column predicates); flags resolved at compile time.
  auto body = [&](auto FULL_, auto BIAS_, auto GELU_, auto RES_, auto F32_) {
    constexpr bool FULL = decltype(FULL_)::value;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n0 = n_base + ni * 32 + t * 16;
        if (!FULL && n0 >= d.N) continue;
        float bv[8];
        if constexpr (decltype(BIAS_)::value) {
          const uint4 b4 = *reinterpret_cast<const uint4*>(d.bias + n0);
          bv[0] = bf16lo(b4.x); bv[1] = bf16hi(b4.x); bv[2] = bf16lo(b4.y); bv[3] = bf16hi(b4.y);
          bv[4] = bf16lo(b4.z); bv[5] = bf16hi(b4.z); bv[6] = bf16lo(b4.w); bv[7] = bf16hi(b4.w);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int m = m_base + mi * 32;
          if (!FULL && m >= d.M) continue;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[mi][ni][8 * t + e] * (n0 < d.nsplit ? d.alpha_lo : d.alpha);
          if constexpr (decltype(BIAS_)::value) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[e];
          }
          if constexpr (decltype(GELU_)::value) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) gelu_epi2(v[e], v[e + 1]);
          }
          if constexpr (decltype(RES_)::value) {
            const uint4 r4 = *reinterpret_cast<const uint4*>(Rz + (int64_t)m * d.ldr + n0);
            v[0] += bf16lo(r4.x); v[1] += bf16hi(r4.x); v[2] += bf16lo(r4.y); v[3] += bf16hi(r4.y);
            v[4] += bf16lo(r4.z); v[5] += bf16hi(r4.z); v[6] += bf16lo(r4.w); v[7] += bf16hi(r4.w);
          }
          if constexpr (decltype(F32_)::value) {
            float* cp = reinterpret_cast<float*>(Cz) + (int64_t)m * d.ldc + n0;
            *reinterpret_cast<float4*>(cp) = float4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4*>(cp + 4) = float4{v[4], v[5], v[6], v[7]};
          } else {
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(Cz) + (int64_t)m * d.ldc + n0) =
                uint4{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
          }
        }
      }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  const int ef = d.flags & (GEMM_BIAS_N | GEMM_GELU | GEMM_RESIDUAL | GEMM_OUT_F32);
  if (bm0 + 256 <= d.M && bn0 + BN <= d.N) {
    switch (ef) {
      case 0: body(T_{}, F_{}, F_{}, F_{}, F_{}); break;
      case GEMM_OUT_F32: body(T_{}, F_{}, F_{}, F_{}, T_{}); break;
      case GEMM_BIAS_N: body(T_{}, T_{}, F_{}, F_{}, F_{}); break;
      case GEMM_BIAS_N | GEMM_OUT_F32: body(T_{}, T_{}, F_{}, F_{}, T_{}); break;
      case GEMM_BIAS_N | GEMM_GELU: body(T_{}, T_{}, T_{}, F_{}, F_{}); break;
      case GEMM_BIAS_N | GEMM_RESIDUAL: body(T_{}, T_{}, F_{}, T_{}, F_{}); break;
      case GEMM_RESIDUAL: body(T_{}, F_{}, F_{}, T_{}, F_{}); break;
      default: break;  // the launcher never sends other combinations here
    }
  } else {
    switch (ef) {  // border tiles: same bodies with row / column predicates
      case 0: body(F_{}, F_{}, F_{}, F_{}, F_{}); break;
      case GEMM_OUT_F32: body(F_{}, F_{}, F_{}, F_{}, T_{}); break;
      case GEMM_BIAS_N: body(F_{}, T_{}, F_{}, F_{}, F_{}); break;
      case GEMM_BIAS_N | GEMM_OUT_F32: body(F_{}, T_{}, F_{}, F_{}, T_{}); break;
      case GEMM_BIAS_N | GEMM_GELU: body(F_{}, T_{}, T_{}, F_{}, F_{}); break;
      case GEMM_BIAS_N | GEMM_RESIDUAL: body(F_{}, T_{}, F_{}, T_{}, F_{}); break;
      case GEMM_RESIDUAL: body(F_{}, F_{}, F_{}, T_{}, F_{}); break;
      default: break;
    }
  }
}

// Epilogue of a TRANSPOSED tile (the K loop ran with the MFMA operands exchanged): lane (l31, hi) holds column
//   n = n_wave + 32 ni + l31,  rows m = m_wave + 64 h + 32 mi + (r & 3) + 8 (r >> 2) + 4 hi  in register r of acc[h][mi][ni].
// Target: d.vt[chunk][n - vt_n0][key] with m = chunk * vt_rows + key, the V^T operand of the flash kernel, whose key order inside
// every group of 16 is [0-3, 8-11, 4-7, 12-15] (kernels.h: transpose_bf16, perm16) -- which is the order the registers have:
// registers 0..7 of a lane are keys {0-3, 8-11} (hi = 0) / {4-7, 12-15} (hi = 1) of the group = positions 8 hi .. 8 hi + 7, registers
// 8..15 the same of the next group.  So a lane stores two 16-byte vectors per accumulator; no permute, no bias, no residual.
template <int NJ>
__device__ __forceinline__ void vt_epilogue(const GemmDesc& d, f32x16 (&acc)[2][2][NJ], int m_wave, int n_wave, int lane) {
  const int hi = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m0 = m_wave + 64 * h + 32 * mi;
      const int chunk = m0 / d.vt_rows, key0 = m0 - chunk * d.vt_rows;
#pragma unroll
      for (int ni = 0; ni < NJ; ++ni) {
        const int n = n_wave + 32 * ni + l31 - d.vt_n0;
        bf16_t* p = d.vt + (int64_t)chunk * d.vt_bs + (int64_t)n * d.vt_ld + key0 + 8 * hi;
        const f32x16& a = acc[h][mi][ni];
#pragma unroll
        for (int g = 0; g < 2; ++g)
          *reinterpret_cast<bt_u32x4*>(p + 16 * g) =
              bt_u32x4{pack2_bf16(a[8 * g + 0] * d.alpha, a[8 * g + 1] * d.alpha), pack2_bf16(a[8 * g + 2] * d.alpha, a[8 * g + 3] * d.alpha),
                       pack2_bf16(a[8 * g + 4] * d.alpha, a[8 * g + 5] * d.alpha), pack2_bf16(a[8 * g + 6] * d.alpha, a[8 * g + 7] * d.alpha)};
      }
    }
}

// "Big tile" kernel (variant 20): 256 x 256 x 64 tiles, ONE workgroup of 4 waves per CU, each wave a 128 x 128 output
// tile (4 x 4 MFMA accumulators = 256 AccVGPRs), its K loop one generated asm block (gemm_bt_asm.inc, written by
// tools/gen_gemm_bt_asm.py, which documents the slot schedule).  What it is after: with one wave per SIMD and a
// hand-placed instruction stream the matrix pipe only drains at the single barrier per K tile, every LDS-DMA piece
// and fragment read sits in the shadow of an MFMA, and a 256^2 tile needs half the LDS fill bandwidth per flop of
// the 128^2 kernel (the measured limit of the CU's L2 -> LDS path, ~50 B/clk, profiles/r01_stage_bw.log).
// Requirements checked by the launcher: K % 64 == 0, operands addressable with 32-bit byte offsets.
#if U2_ELEM_IS_F16
#include "build_f16/gemm_bt_asm.inc"  // derived at build time: tools/asm_elem_f16.py
#else
#include "gemm_bt_asm.inc"
#endif
typedef int bt_i32x4 __attribute__((ext_vector_type(4)));

// NJ = 2 is the RING form (256 x 128 tiles, variant 22): three LDS stages of 48 KB, K tile t + 2 issued during iteration t and
// waited for with a counted vmcnt one iteration later (tools/gen_gemm_bt_asm.py, gen_ring) -- the tile for products whose
// 256-wide tiles would leave half of the CUs without work (M = 2048 / 1024 against E x E and E x 2E weights: 256 tiles).
// DEEP = 2 (variant 24 at 256 x 192, 26 at 256 x 256): the B operand (the weights) streams through THREE LDS stages on the ring's
// schedule, A keeps two (gen_deep of the generator has the schedule and the in-order argument; the A-deep twins of round 4 never won
// a shape and are gone).  VT (256 x 192, deep): output tiles at columns >= d.vt_n0 run the loop with the MFMA operands exchanged and
// leave their TRANSPOSED tile in d.vt (vt_epilogue) instead of C -- the ViT's q|k|v product writes V^T for the flash kernel itself.
// TAIL: d.tail_rows rows behind the M rows of the tiles (the cls rows of the ViT: 8 behind 16384) are computed IN this launch, before the
// first tile, by the few-rows arithmetic of rows16.h -- workgroup b takes the 16-column blocks b, b + grid, ...; its four waves run the
// slices the few-rows kernel's 4 / 8 / 16 waves would, so the values are those of a gemm_rows16 launch bit for bit, which it replaces
// (48 launches of ~7-10 us per volume; the workgroups concerned start their tiles ~2 us later).
template <int NWV>
__device__ __forceinline__ void bt_tail_block(const GemmDesc& d, int n0, float (*red)[64][4], int wv, int lane) {
  const int l15 = lane & 15, g = lane >> 4;
  const bool f32 = d.flags & GEMM_OUT_F32;
  const bf16_t* At = d.A + (int64_t)d.M * d.lda;
  char* Ct = reinterpret_cast<char*>(d.C) + (int64_t)d.M * d.ldc * (f32 ? 4 : 2);
  const bf16_t* Rt = (d.flags & GEMM_RESIDUAL) ? d.R + (int64_t)d.M * d.ldr : nullptr;
  const int nsteps = d.K >> 5, per = (nsteps + NWV - 1) / NWV;
  const int nrow = min(n0 + l15, d.N - 1), mrow = min(l15, d.tail_rows - 1);
  const bf16_t* wp = d.B + (int64_t)nrow * d.ldb + g * 8;
  const bf16_t* xp = At + (int64_t)mrow * d.lda + g * 8;
  for (int vw = wv; vw < NWV; vw += 4) {   // the slices of "waves" wv, wv + 4, ...
    const int s0 = vw * per, s1 = min(nsteps, s0 + per);
    const f32x4 acc = rows16_slice(wp, xp, s0, s1);
    red[vw][lane][0] = acc[0]; red[vw][lane][1] = acc[1]; red[vw][lane][2] = acc[2]; red[vw][lane][3] = acc[3];
  }
  __syncthreads();
  if (wv == 0 && l15 < d.tail_rows) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NWV; ++w)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += red[w][lane][r];
    rows16_store(d, v, l15, n0 + 4 * g, Ct, Rt);
  }
  __syncthreads();  // red is free again
}

template <int NJ, bool PAIR = false, bool SPLIT = false, int DEEP = 0, bool VT = false, bool TAIL = false>  // 32-column blocks per wave: tile = 256 x (64 NJ)
__global__ __launch_bounds__(256, 1) void gemm_bt_kernel(GemmDesc d) {
  static_assert(!TAIL || (!PAIR && !SPLIT), "the in-launch tail exists for the plain forms");
  static_assert(DEEP == 0 || (DEEP == 2 && NJ >= 3 && !PAIR && !SPLIT), "the deep forms are plain 256 x 192 / 256 x 256 kernels");
  static_assert(!VT || (DEEP == 2 && NJ == 3), "the transposed-tile form exists for 256 x 192 deep tiles");
  static_assert(!(PAIR && SPLIT), "the pair form is not sliced");
  static_assert(!PAIR || NJ == 3, "the pair form exists for 256 x 192 tiles");
  static_assert(NJ == 2 || NJ == 3 || NJ == 4, "256 x 128 (ring) / 256 x 192 / 256 x 256 tiles");
  static_assert(NJ != 2 || !PAIR, "no SwiGLU-pair ring form");
  constexpr int BN = 64 * NJ;
  constexpr bool RING = NJ == 2;
  using CFG = BTCfg<BN>;
  // two stages: [stage][A tile 32 KB | B tile <= 32 KB];  ring: three stages of [A tile 32 KB | B tile 16 KB]
  // deep forms: [A stages of 32 KB][B stages of 8 NJ KB], three of the deep operand and two of the other
  constexpr int OFFB = DEEP == 2 ? 2 * 32768 : 32768;
  constexpr int LDS_BYTES = DEEP == 2 ? 2 * 32768 + 3 * NJ * 8192 : RING ? 147456 : 131072;
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const int tiles_mn = d.tiles_m * d.tiles_n;
  // split-K (nz == 1 then): the K slices take the place of the batch index -- "z" of a tile is its slice, which leaves raw
  // fp32 sums in partial[slice][m][n] for gemm_splitk_reduce_kernel (gemm.hip), epilogue flags and all
  constexpr bool split = SPLIT;
  const int total = tiles_mn * (split ? d.ksplit : d.nz);
  const int gd = gridDim.x, bid = blockIdx.x;
  if constexpr (TAIL) {
    if (d.tail_rows > 0) {  // (uniform) before the first DMA of the K loop touches the LDS
      float (*red)[64][4] = reinterpret_cast<float (*)[64][4]>(lds);
      const int nblk = (d.N + 15) >> 4, nsl = rows16_slices(d.K >> 5);
      for (int blk = bid; blk < nblk; blk += gd) {
        if (nsl == 16) bt_tail_block<16>(d, blk * 16, red, wave, lane);
        else if (nsl == 8) bt_tail_block<8>(d, blk * 16, red, wave, lane);
        else bt_tail_block<4>(d, blk * 16, red, wave, lane);
      }
    }
  }
  const int my_tiles = (total - bid + gd - 1) / gd;  // >= 1: grid <= total
  // DMA pieces of this wave: rows [64 w, 64 w + 64) of the A tile and [16 NJ w, ..) of the B tile, 8 rows x 128 B per
  // piece; LDS position p of row r holds global chunk p ^ ((r >> 1) & 7): even / odd pieces differ by 4 in that term.
  // Lane offsets are relative to the tile origin; the origin (and the K tile) travel in the scalar offset.
  const int pr = lane >> 3, sw0 = (lane >> 4) & 3, pc = lane & 7;
  const int ra = wave * 64 + pr;
  const int va0 = (ra * (int)d.lda + ((pc ^ sw0) << 3)) * 2;
  const int va1 = ((ra + 8) * (int)d.lda + ((pc ^ sw0 ^ 4) << 3)) * 2;
  // PAIR (GEMM_SWIGLU: B = [gate rows | up rows], N = 2 I, C[m][j] = silu(gate_j) * up_j; NJ = 3): tile row
  // R = 32 b + 16 half + i  <->  weight row half * I + (bn0 / 2) + 16 b + i, so every 32-column MFMA block holds 16 gate columns
  // and THE SAME 16 up columns, which the epilogue finds in one lane (t = 0 / 1).  The lane offsets then address a row inside a
  // 16-row group and the group's offset is a scalar per (wave, group) handed to the K loop.
  const int rb = (PAIR ? 0 : wave * (16 * NJ)) + pr;
  const int vb0 = (rb * (int)d.ldb + ((pc ^ sw0) << 3)) * 2;
  const int vb1 = ((rb + 8) * (int)d.ldb + ((pc ^ sw0 ^ 4) << 3)) * 2;
  [[maybe_unused]] int rowb[3] = {0, 0, 0};
  if constexpr (PAIR) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int g = NJ * wave + q;  // 16-row group of the tile
      rowb[q] = __builtin_amdgcn_readfirstlane(((g & 1) * (int)(d.N >> 1) + 16 * (g >> 1)) * (int)d.ldb * 2);
    }
  }
  const uint32_t lds_u32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&lds[0];  // 0: the only LDS object
  const uint32_t abk0 = (uint32_t)(l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4));
  const int lda16 = 16 * (int)d.lda * 2, ldb16 = 16 * (int)d.ldb * 2;
  const int nkt_all = d.K >> 6;
  const bool chain = d.nz == 1;  // one K loop runs on from tile to tile (same descriptors)
  uint32_t st0 = 0;              // LDS stage that holds K tile 0 of the current output tile (byte offset 0 / 0x10000; ring: index 0..2;
                                 // deep forms: K tiles consumed so far mod 6 = the (A stage, B stage) pair)
  int z, bm0, bn0;
  pp_tile<BN>(d, 0, gd, bid, total, tiles_mn, z, bm0, bn0);
  for (int r = 0; r < my_tiles; ++r) {
    int zn = z, bm0n = bm0, bn0n = bn0;  // next output tile of this workgroup (itself after the last one: its K
    if (r + 1 < my_tiles) pp_tile<BN>(d, r + 1, gd, bid, total, tiles_mn, zn, bm0n, bn0n);  // loop prefetches in-bounds garbage)
    const int zq = split ? 0 : z;
    const int zb = zq / d.nbh, zh = zq - zb * d.nbh;
    const bf16_t* A = d.A + zb * d.sAb + zh * d.sAh;
    const bf16_t* B = d.B + zb * d.sBb + zh * d.sBh;
    // K range of this tile (split-K: slice z, the last one may be shorter; >= 2 K tiles each, the launcher sees to it)
    const int kt0 = split ? z * d.kt_per : 0, kt0n = split ? zn * d.kt_per : 0;
    const int nkt = __builtin_amdgcn_readfirstlane(split ? min(d.kt_per, nkt_all - kt0) : nkt_all);
    // MUBUF descriptors: rows past M / N read as zero
    const uint64_t aaddr = (uint64_t)(uintptr_t)A, baddr = (uint64_t)(uintptr_t)B;
    bt_i32x4 rsa, rsb;
    rsa[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)aaddr);
    rsa[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(aaddr >> 32));
    rsa[2] = (int)((((int64_t)d.M - 1) * d.lda + d.K) * 2);
    rsa[3] = 0x00020000;
    rsb[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)baddr);
    rsb[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(baddr >> 32));
    rsb[2] = (int)((((int64_t)d.N - 1) * d.ldb + d.K) * 2);
    rsb[3] = 0x00020000;
    const int first = (r == 0 || !chain) ? 1 : 0;
    if (first) st0 = 0;
    const uint32_t aa0 = (lds_u32 + wm * 16384 + abk0) ^ ((RING || DEEP) ? 0u : st0);  // (ring / deep: stage 0's addresses, the asm adds the stage)
    const uint32_t ab0 = (lds_u32 + OFFB + wn * (NJ * 4096) + abk0) ^ ((RING || DEEP) ? 0u : st0);
    // (readfirstlane: the values are uniform, but hipcc keeps loop-carried tile coordinates in VGPRs)
    const int base_a = __builtin_amdgcn_readfirstlane((bm0 * (int)d.lda + kt0 * 64) * 2);
    const int base_b = __builtin_amdgcn_readfirstlane(((PAIR ? bn0 >> 1 : bn0) * (int)d.ldb + kt0 * 64) * 2);
    const int nbase_a = __builtin_amdgcn_readfirstlane(((chain ? bm0n : bm0) * (int)d.lda + (chain ? kt0n : kt0) * 64) * 2);
    const int nbase_b = __builtin_amdgcn_readfirstlane(
        ((PAIR ? (chain ? bn0n : bn0) >> 1 : (chain ? bn0n : bn0)) * (int)d.ldb + (chain ? kt0n : kt0) * 64) * 2);
    const int st0_s = __builtin_amdgcn_readfirstlane((int)st0), first_s = __builtin_amdgcn_readfirstlane(first);
    // fast epilogue (bt_epilogue_fast): tile inside C, bf16 output, bias[n] / residual only; its bias vectors are fetched here,
    // under the K loop
    f32x16 acc[2][2][NJ];  // [64-row half of the wave tile][32-row block][32-column block]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NJ; ++ni)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[h][mi][ni][q] = 0.f;
#define BT_ACC3(h_, m_) [c##h_##m_##0] "+a"(acc[h_][m_][0]), [c##h_##m_##1] "+a"(acc[h_][m_][1]), [c##h_##m_##2] "+a"(acc[h_][m_][2])
#define BT_IN                                                                                                         \
  [va0] "v"(va0), [va1] "v"(va1), [vb0] "v"(vb0), [vb1] "v"(vb1), [aa0] "v"(aa0), [ab0] "v"(ab0), [rsa] "s"(rsa),     \
      [rsb] "s"(rsb), [lda16] "s"(lda16), [nkt] "s"(nkt), [wave] "s"(wave), [st0] "s"(st0_s), [first] "s"(first_s),   \
      [base_a] "s"(base_a), [base_b] "s"(base_b), [nbase_a] "s"(nbase_a), [nbase_b] "s"(nbase_b)
    [[maybe_unused]] const bool vt_tile = VT && bn0 >= d.vt_n0;  // (uniform) this tile's accumulators come out transposed
    if constexpr (DEEP != 0 && NJ == 3) {
      if (VT && vt_tile)
        asm volatile(GEMM_BT_ASM_TEXT_NJ3_DB_T
                     : BT_ACC3(0, 0), BT_ACC3(0, 1), BT_ACC3(1, 0), BT_ACC3(1, 1)
                     : BT_IN, [ldb16] "s"(ldb16)
                     : GEMM_BT_ASM_CLOBBERS_NJ3_DEEP);
      else
        asm volatile(GEMM_BT_ASM_TEXT_NJ3_DB
                     : BT_ACC3(0, 0), BT_ACC3(0, 1), BT_ACC3(1, 0), BT_ACC3(1, 1)
                     : BT_IN, [ldb16] "s"(ldb16)
                     : GEMM_BT_ASM_CLOBBERS_NJ3_DEEP);
    } else if constexpr (DEEP != 0) {
      asm volatile(GEMM_BT_ASM_TEXT_NJ4_DB
                   : BT_ACC3(0, 0), [c003] "+a"(acc[0][0][3]), BT_ACC3(0, 1), [c013] "+a"(acc[0][1][3]), BT_ACC3(1, 0),
                     [c103] "+a"(acc[1][0][3]), BT_ACC3(1, 1), [c113] "+a"(acc[1][1][3])
                   : BT_IN, [ldb16] "s"(ldb16)
                   : GEMM_BT_ASM_CLOBBERS_NJ4_DEEP);
    } else if constexpr (NJ == 2) {
      asm volatile(GEMM_BT_ASM_TEXT_NJ2_RING
                   : [c000] "+a"(acc[0][0][0]), [c001] "+a"(acc[0][0][1]), [c010] "+a"(acc[0][1][0]), [c011] "+a"(acc[0][1][1]),
                     [c100] "+a"(acc[1][0][0]), [c101] "+a"(acc[1][0][1]), [c110] "+a"(acc[1][1][0]), [c111] "+a"(acc[1][1][1])
                   : BT_IN, [ldb16] "s"(ldb16)
                   : GEMM_BT_ASM_CLOBBERS_RING);
    } else if constexpr (NJ == 4) {
      asm volatile(GEMM_BT_ASM_TEXT_NJ4
                   : BT_ACC3(0, 0), [c003] "+a"(acc[0][0][3]), BT_ACC3(0, 1), [c013] "+a"(acc[0][1][3]), BT_ACC3(1, 0),
                     [c103] "+a"(acc[1][0][3]), BT_ACC3(1, 1), [c113] "+a"(acc[1][1][3])
                   : BT_IN, [ldb16] "s"(ldb16)
                   : GEMM_BT_ASM_CLOBBERS);
    } else if constexpr (PAIR) {
      asm volatile(GEMM_BT_ASM_TEXT_NJ3_PAIR
                   : BT_ACC3(0, 0), BT_ACC3(0, 1), BT_ACC3(1, 0), BT_ACC3(1, 1)
                   : BT_IN, [rowb0] "s"(rowb[0]), [rowb1] "s"(rowb[1]), [rowb2] "s"(rowb[2])
                   : GEMM_BT_ASM_CLOBBERS);
    } else {
      asm volatile(GEMM_BT_ASM_TEXT_NJ3
                   : BT_ACC3(0, 0), BT_ACC3(0, 1), BT_ACC3(1, 0), BT_ACC3(1, 1)
                   : BT_IN, [ldb16] "s"(ldb16)
                   : GEMM_BT_ASM_CLOBBERS);
    }
#undef BT_ACC3
#undef BT_IN
    // pp_epilogue's row base is bm0 + 128 G + 64 wm2: G = 0 with the wave's 128-row offset folded into bm0 (its "tile
    // inside C" fast-path test then only errs towards the predicated path)
    if constexpr (split) {
      GemmDesc ds = d;  // this slice's raw sums: partial[z][m][n], dense fp32
      ds.C = d.partial + (int64_t)z * d.M * d.N;
      ds.ldc = d.N;
      ds.alpha = 1.f;
      ds.nsplit = 0;      // the column-range scale (alpha_lo below nsplit) belongs to the reduce that applies the epilogue,
      ds.alpha_lo = 1.f;  // not to the raw sums (it would have been applied twice)
      ds.flags = GEMM_OUT_F32 | GEMM_VEC_OK;
      const bt_u32x4 bnone[NJ * 2] = {};
      pp_epilogue<CFG, 0, false>(ds, acc[0], 0, bm0 + wm * 128, bn0, 0, wn, lane, bnone, false);
      pp_epilogue<CFG, 0, false>(ds, acc[1], 0, bm0 + wm * 128, bn0, 1, wn, lane, bnone, false);
    } else if (VT && vt_tile) {
      vt_epilogue<NJ>(d, acc, bm0 + wm * 128, bn0 + wn * (BN / 2), lane);
    } else {
      auto fast_tile = [&]() {  // (evaluated again after the K loop instead of living in an SGPR across it: none to spare)
        // (256 x 192 tiles only: with 256 accumulator registers the 256-wide form has no room for the residual batch)
        return NJ == 3 && !PAIR && !SPLIT && !(d.flags & (GEMM_OUT_F32 | GEMM_GELU)) && bm0 + 256 <= d.M && bn0 + BN <= d.N;
      };
      bt_u32x4 bpre[NJ * 2];
#pragma unroll
      for (int j = 0; j < NJ * 2; ++j) bpre[j] = bt_u32x4{0u, 0u, 0u, 0u};
      if (fast_tile() && (d.flags & GEMM_BIAS_N)) {
#pragma unroll
        for (int j = 0; j < NJ * 2; ++j)
          bpre[j] = *reinterpret_cast<const bt_u32x4*>(d.bias + bn0 + wn * (BN / 2) + 8 * hi + (j >> 1) * 32 + (j & 1) * 16);
      }
      const bool fast = fast_tile();
      pp_epilogue<CFG, 0, PAIR>(d, acc[0], z, bm0 + wm * 128, bn0, 0, wn, lane, bpre, fast);
      pp_epilogue<CFG, 0, PAIR>(d, acc[1], z, bm0 + wm * 128, bn0, 1, wn, lane, bpre, fast);
    }
    if constexpr (DEEP != 0) st0 = (st0 + (uint32_t)nkt) % 6u;
    else if constexpr (RING) st0 = (st0 + (uint32_t)nkt) % 3u;
    else st0 ^= (uint32_t)(nkt & 1) << 16;
    z = zn; bm0 = bm0n; bn0 = bn0n;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last K loop's prefetch into LDS
}


// ---- round 6: the DRAIN form of the deep 256 x 192 kernel (variant 27): tile i's epilogue under the K loop of tile i + 1 ----------------
// (tools/gen_gemm_bt_asm.py, "drain forms", has the mechanism and the register map.)  For products whose workgroups walk SEVERAL output
// tiles -- the ViT's q|k|v (three rounds of tiles; MONAI SABlock.qkv, /root/reference/src/model/multimodal_encoder/vit.py:100-105) and
// fc1 + bias + GELU (MLPBlock.linear1: four rounds of 192-wide tiles) -- the asm statement of tile r >= 1 opens with CONVERT (tile r - 1's
// accumulators -> 96 registers of packed 16-bit elements, ~1 us) and stores them, 16 bytes per lane and MFMA slot, from the first twelve
// iterations of tile r's K loop; the GELU form evaluates gelu_fast2 on the held values in those iterations' spare issue slots.  Only the
// workgroup's LAST tile still pays an exposed epilogue (drain_final_epilogue, the same arithmetic in HIP).
// The accumulator file is pinned (operands "+{a[32 k : 32 k + 31]}": the asm text names a0..a191 literally); nothing else lives in a
// clobbered register across two statements.  Requirements (bt_drain_ok): nz = 1, K a multiple of 384 and >= 768 (the loop's twelve drain
// bodies start at LDS stage pair 0), whole 256 x 192 tiles, bf16 output with alpha / alpha_lo, bias[n] and GELU only, nsplit a multiple
// of 192, N <= 6144 with a bias (its fp32 copy sits in the 24 KB of LDS behind the stages).
// GELU rounding point: the pre-activation is rounded to the element type BEFORE the GELU (gelu_epi, common.h: the rule of every GEMM
// epilogue of this library since round 6, and the reference's own rounding point).
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef int bt_i32x16 __attribute__((ext_vector_type(16)));

template <bool GELU>
__device__ __forceinline__ void drain_final_epilogue(const GemmDesc& d, f32x16 (&acc)[4][3], int m_wave, int n_wave, int lane, float al,
                                                     const float* lds_bias) {
  const int hi = lane >> 5, l31 = lane & 31;
  bf16_t* C = reinterpret_cast<bf16_t*>(d.C);
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int ni = 0; ni < 3; ++ni)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float v[8], raw[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[rb][ni][8 * t + e]),
                                                          __float_as_uint(acc[rb][ni][8 * t + 4 + e]), false, false);
          raw[e] = __uint_as_float(r[0]);
          raw[4 + e] = __uint_as_float(r[1]);
          v[e] = raw[e] * al;
          v[4 + e] = raw[4 + e] * al;
        }
        const int n0 = n_wave + ni * 32 + t * 16 + 8 * hi;
        if (lds_bias) {   // (acc * alpha + bias as one fused multiply-add, like CONVERT and like the other forms' compiled epilogues)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(raw[e], al, lds_bias[n0 + e]);
        }
        if constexpr (GELU) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) gelu_epi2(v[e], v[e + 1]);
        }
        *reinterpret_cast<bt_u32x4*>(C + (int64_t)(m_wave + rb * 32 + l31) * d.ldc + n0) =
            bt_u32x4{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
      }
}

template <bool GELU, bool VT, bool TAIL>
__global__ __launch_bounds__(256, 1) void gemm_bt_drain_kernel(GemmDesc d) {
  static_assert(!(GELU && VT), "the GELU form has no transposed tiles");
  constexpr int NJ = 3, BN = 192, OFFB = 2 * 32768, STAGES = 2 * 32768 + 3 * NJ * 8192, LDS_BYTES = 163840;
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const int tiles_mn = d.tiles_m * d.tiles_n;
  const int total = tiles_mn;
  const int gd = gridDim.x, bid = blockIdx.x;
  if constexpr (TAIL) {
    if (d.tail_rows > 0) {
      float (*red)[64][4] = reinterpret_cast<float (*)[64][4]>(lds);
      const int nblk = (d.N + 15) >> 4, nsl = rows16_slices(d.K >> 5);
      for (int blk = bid; blk < nblk; blk += gd) {
        if (nsl == 16) bt_tail_block<16>(d, blk * 16, red, wave, lane);
        else if (nsl == 8) bt_tail_block<8>(d, blk * 16, red, wave, lane);
        else bt_tail_block<4>(d, blk * 16, red, wave, lane);
      }
    }
  }
  // fp32 copy of the bias behind the stages (CONVERT reads it with two ds_read_b128 per group): requested here, 8 elements per lane and
  // piece (N <= 6144: three pieces), written to the LDS behind the first tile's K loop, which hides the latency
  float* lds_bias = reinterpret_cast<float*>(lds + STAGES);
  const bool have_bias = (d.flags & GEMM_BIAS_N) != 0;
  bt_u32x4 braw[3] = {};
  if (have_bias) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (tid * 8 + q * 2048 < d.N) braw[q] = *reinterpret_cast<const bt_u32x4*>(d.bias + tid * 8 + q * 2048);
  }
  const int my_tiles = (total - bid + gd - 1) / gd;
  const int pr = lane >> 3, sw0 = (lane >> 4) & 3, pc = lane & 7;
  const int ra = wave * 64 + pr;
  const int va0 = (ra * (int)d.lda + ((pc ^ sw0) << 3)) * 2;
  const int va1 = ((ra + 8) * (int)d.lda + ((pc ^ sw0 ^ 4) << 3)) * 2;
  const int rb_ = wave * (16 * NJ) + pr;
  const int vb0 = (rb_ * (int)d.ldb + ((pc ^ sw0) << 3)) * 2;
  const int vb1 = ((rb_ + 8) * (int)d.ldb + ((pc ^ sw0 ^ 4) << 3)) * 2;
  const uint32_t lds_u32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&lds[0];
  const uint32_t abk0 = (uint32_t)(l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4));
  const int lda16 = 16 * (int)d.lda * 2, ldb16 = 16 * (int)d.ldb * 2;
  const int nkt = __builtin_amdgcn_readfirstlane(d.K >> 6);
  const uint32_t aa0 = lds_u32 + wm * 16384 + abk0;
  const uint32_t ab0 = lds_u32 + OFFB + wn * (NJ * 4096) + abk0;
  // lane parts of the store addresses of a drained tile: row-major C / transposed V^T
  const int voff_c = (l31 * (int)d.ldc + 8 * hi) * 2;
  [[maybe_unused]] const int voff_t = (l31 * (int)d.vt_ld + 8 * hi) * 2;
  // MUBUF descriptors: operands (rows past M / N read as zero) and the two store targets
  bt_i32x4 rsa, rsb;
  {
    const uint64_t aaddr = (uint64_t)(uintptr_t)d.A, baddr = (uint64_t)(uintptr_t)d.B;
    rsa[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)aaddr);
    rsa[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(aaddr >> 32));
    rsa[2] = (int)((((int64_t)d.M - 1) * d.lda + d.K) * 2);
    rsa[3] = 0x00020000;
    rsb[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)baddr);
    rsb[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(baddr >> 32));
    rsb[2] = (int)((((int64_t)d.N - 1) * d.ldb + d.K) * 2);
    rsb[3] = 0x00020000;
  }
  const uint64_t caddr = (uint64_t)(uintptr_t)d.C, taddr = (uint64_t)(uintptr_t)d.vt;
  const int c_bytes = (int)((((int64_t)d.M - 1) * d.ldc + d.N) * 2);
  [[maybe_unused]] const int t_bytes = VT ? (int)((int64_t)(d.M / d.vt_rows) * d.vt_bs * 2) : 0;
  // accumulator (i, j) = 32-row block i of the wave's 128 rows x 32-column block j: registers a[16 (3 i + j) : +15], named literally by
  // the statements below and listed as their clobbers -- no C++ object holds them (tests/test_generated_asm.py audits the compiled
  // kernel: no compiler instruction touches an accumulator register between the first statement and the read-out)
  asm volatile(GEMM_BT_ASM_TEXT_ACC192_ZERO ::: GEMM_BT_ASM_CLOBBERS_ACC192);
  int z, bm0, bn0;
  pp_tile<BN>(d, 0, gd, bid, total, tiles_mn, z, bm0, bn0);
  // the tile whose accumulators are waiting (drained by the next statement): (pbm0, pbn0) -- its parameter block is built from
  // wave-uniform scalars right in front of the statement (a loop-carried vector would live in VGPRs)
  int pbm0 = 0, pbn0 = 0;
  for (int r = 0; r < my_tiles; ++r) {
    int zn = z, bm0n = bm0, bn0n = bn0;
    if (r + 1 < my_tiles) pp_tile<BN>(d, r + 1, gd, bid, total, tiles_mn, zn, bm0n, bn0n);
    const int first = r == 0 ? 1 : 0;
    const int base_a = __builtin_amdgcn_readfirstlane((bm0 * (int)d.lda) * 2);
    const int base_b = __builtin_amdgcn_readfirstlane((bn0 * (int)d.ldb) * 2);
    const int nbase_a = __builtin_amdgcn_readfirstlane((bm0n * (int)d.lda) * 2);
    const int nbase_b = __builtin_amdgcn_readfirstlane((bn0n * (int)d.ldb) * 2);
    int st0_s, first_s;  // (opaque to the optimiser: a folded literal is not a legal operand everywhere the text uses them)
    asm volatile("s_mov_b32 %0, 0\n\ts_mov_b32 %1, %2" : "=&s"(st0_s), "=&s"(first_s) : "s"(__builtin_amdgcn_readfirstlane(first)));
    [[maybe_unused]] const bool vt_tile = VT && bn0 >= d.vt_n0;
    // parameters of the PREVIOUS tile (garbage for r = 0: not read)
    const int pm_wave = __builtin_amdgcn_readfirstlane(pbm0 + wm * 128), pn_wave = __builtin_amdgcn_readfirstlane(pbn0 + wn * (BN / 2));
    const bool pvt = VT && pbn0 >= d.vt_n0;
    const float pal = pvt ? d.alpha : (pbn0 < d.nsplit ? d.alpha_lo : d.alpha);
    int p0, p1, p2, p4, p5, p6;
    if (pvt) {
      const int chunk = pm_wave / d.vt_rows, key0 = pm_wave - chunk * d.vt_rows;
      p0 = (int)(uint32_t)taddr; p1 = (int)(uint32_t)(taddr >> 32); p2 = t_bytes;
      p4 = (int)(((int64_t)chunk * d.vt_bs + (int64_t)(pn_wave - d.vt_n0) * d.vt_ld + key0) * 2);
      p5 = 64; p6 = (int)(32 * d.vt_ld * 2);
    } else {
      p0 = (int)(uint32_t)caddr; p1 = (int)(uint32_t)(caddr >> 32); p2 = c_bytes;
      p4 = (int)(((int64_t)pm_wave * d.ldc + pn_wave) * 2);
      p5 = (int)(32 * d.ldc * 2); p6 = 64;
    }
    const int p7 = (int)__float_as_uint(pal);
    const int p8 = (pvt ? 1 : 0) | (pal != 1.0f ? 2 : 0) | ((have_bias && !pvt) ? 4 : 0);
#define U2_RFL(x) __builtin_amdgcn_readfirstlane(x)
    const bt_i32x16 prm = {U2_RFL(p0), U2_RFL(p1), U2_RFL(p2), 0x00020000, U2_RFL(p4), U2_RFL(p5), U2_RFL(p6), U2_RFL(p7),
                           U2_RFL(p8), 0, 0, 0, 0, 0, 0, 0};
#undef U2_RFL
    const int voff_prev = pvt ? voff_t : voff_c;
    const int vbias_prev = (int)(lds_u32 + STAGES + (pbn0 + wn * (BN / 2)) * 4 + 16 * hi);
#define DR_IN                                                                                                               \
  [va0] "v"(va0), [va1] "v"(va1), [vb0] "v"(vb0), [vb1] "v"(vb1), [aa0] "v"(aa0), [ab0] "v"(ab0), [rsa] "s"(rsa),           \
      [rsb] "s"(rsb), [lda16] "s"(lda16), [ldb16] "s"(ldb16), [nkt] "s"(nkt), [wave] "s"(wave), [st0] "s"(st0_s),           \
      [first] "s"(first_s), [base_a] "s"(base_a), [base_b] "s"(base_b), [nbase_a] "s"(nbase_a), [nbase_b] "s"(nbase_b)
#define DR_PREV [voff] "v"(voff_prev), [vbias] "v"(vbias_prev), [prm] "{s[72:87]}"(prm)
    if (r == 0) {
      if (VT && vt_tile) asm volatile(GEMM_BT_ASM_TEXT_NJ3_DB_FX_T : : DR_IN : GEMM_BT_ASM_CLOBBERS_NJ3_DEEP, GEMM_BT_ASM_CLOBBERS_ACC192);
      else asm volatile(GEMM_BT_ASM_TEXT_NJ3_DB_FX : : DR_IN : GEMM_BT_ASM_CLOBBERS_NJ3_DEEP, GEMM_BT_ASM_CLOBBERS_ACC192);
    } else if constexpr (GELU) {
      asm volatile(GEMM_BT_ASM_TEXT_NJ3_DB_DRAIN_GELU : : DR_IN, DR_PREV : GEMM_BT_ASM_CLOBBERS_DRAIN, GEMM_BT_ASM_CLOBBERS_ACC192);
    } else {
      if (VT && vt_tile) asm volatile(GEMM_BT_ASM_TEXT_NJ3_DB_DRAIN_T : : DR_IN, DR_PREV : GEMM_BT_ASM_CLOBBERS_DRAIN, GEMM_BT_ASM_CLOBBERS_ACC192);
      else asm volatile(GEMM_BT_ASM_TEXT_NJ3_DB_DRAIN : : DR_IN, DR_PREV : GEMM_BT_ASM_CLOBBERS_DRAIN, GEMM_BT_ASM_CLOBBERS_ACC192);
    }
#undef DR_IN
#undef DR_PREV
    if (r == 0) {
      if (have_bias) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if (tid * 8 + q * 2048 < d.N) {
            float* bp = lds_bias + tid * 8 + q * 2048;
            bp[0] = bf16lo(braw[q].x); bp[1] = bf16hi(braw[q].x); bp[2] = bf16lo(braw[q].y); bp[3] = bf16hi(braw[q].y);
            bp[4] = bf16lo(braw[q].z); bp[5] = bf16hi(braw[q].z); bp[6] = bf16lo(braw[q].w); bp[7] = bf16hi(braw[q].w);
          }
      }
      __syncthreads();   // (the statement's prefetched K tiles are LDS-DMA in flight: the fence waits for them, as the next loop would)
    }
    if (r + 1 == my_tiles) {
      // the workgroup's last tile: nothing left to hide it under
      const int m_wave = bm0 + wm * 128, n_wave = bn0 + wn * (BN / 2);
      const float al = vt_tile ? d.alpha : (bn0 < d.nsplit ? d.alpha_lo : d.alpha);
      f32x16 a16[4][3];
      asm volatile(GEMM_BT_ASM_TEXT_ACC192_READ
                   : "={v[56:71]}"(a16[0][0]), "={v[72:87]}"(a16[0][1]), "={v[88:103]}"(a16[0][2]), "={v[104:119]}"(a16[1][0]),
                     "={v[120:135]}"(a16[1][1]), "={v[136:151]}"(a16[1][2]), "={v[152:167]}"(a16[2][0]), "={v[168:183]}"(a16[2][1]),
                     "={v[184:199]}"(a16[2][2]), "={v[200:215]}"(a16[3][0]), "={v[216:231]}"(a16[3][1]), "={v[232:247]}"(a16[3][2]));
      if (VT && vt_tile) {
        f32x16 (&av)[2][2][3] = reinterpret_cast<f32x16 (&)[2][2][3]>(a16);
        vt_epilogue<NJ>(d, av, m_wave, n_wave, lane);
      } else {
        drain_final_epilogue<GELU>(d, a16, m_wave, n_wave, lane, al, have_bias ? lds_bias : nullptr);
      }
    }
    pbm0 = bm0; pbn0 = bn0;
    z = zn; bm0 = bm0n; bn0 = bn0n;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// May the deep 256 x 192 launch of `d` (many-row part: M a multiple of 256) run as the drain form?
static bool bt_drain_ok(const GemmDesc& d) {
  if (!opts().gemm_big_drain || !opts().gemm_big_deep || d.nz != 1 || d.ksplit > 1) return false;
  if ((d.flags & GEMM_GELU) && opts().gemm_big_drain == 2) return false;  // (2: the forms whose arithmetic is that of the plain deep form only)
  if (d.flags & ~(GEMM_VEC_OK | GEMM_BIAS_N | GEMM_GELU)) return false;
  if (d.K % 384 || d.K < 768 || d.M % 256 || d.N % 192 || d.nsplit % 192) return false;
  if ((d.flags & GEMM_BIAS_N) && d.N > 6144) return false;
  if ((d.flags & GEMM_GELU) && d.vt) return false;
  if (((int64_t)d.M - 1) * d.ldc + d.N >= (1ll << 30)) return false;
  if (d.vt && (int64_t)(d.M / d.vt_rows) * d.vt_bs >= (1ll << 30)) return false;
  const int64_t tiles = (int64_t)(d.M / 256) * (d.N / 192);
  return tiles >= 2 * (int64_t)opts().gemm_big_grid;   // every workgroup has a tile to hide the previous one under
}

// Row tiles per group of the tile walk (pp_tile).  An XCD runs 32 consecutive ids = group_m row tiles x 32 / group_m column tiles at a time.
static int bt_group_m(const GemmDesc& d) {
  const int o = opts().gemm_big_group_m;
  return o > 0 ? o : 8;
}

static int bt_launch_drain(GemmDesc d, hipStream_t stream) {
  d.tiles_m = d.M / 256;
  d.tiles_n = d.N / 192;
  d.group_m = bt_group_m(d);
  const int64_t total = (int64_t)d.tiles_m * d.tiles_n;
  const int grid = (int)std::min<int64_t>(total, opts().gemm_big_grid);
  const bool tail = d.tail_rows > 0;
#define U2_DRAIN(G_, V_, T_) hipLaunchKernelGGL((gemm_bt_drain_kernel<G_, V_, T_>), dim3(grid), dim3(256), 0, stream, d)
  if (d.flags & GEMM_GELU) { if (tail) U2_DRAIN(true, false, true); else U2_DRAIN(true, false, false); }
  else if (d.vt) { if (tail) U2_DRAIN(false, true, true); else U2_DRAIN(false, true, false); }
  else { if (tail) U2_DRAIN(false, false, true); else U2_DRAIN(false, false, false); }
#undef U2_DRAIN
  return launch_status();
}

template <int NJ, int DEEP>
static int bt_launch_deep(GemmDesc d, hipStream_t stream) {
  if (d.ksplit > 1) return U2_ERR_ARG;
  d.tiles_m = (int)cdiv(d.M, 256);
  d.tiles_n = (int)cdiv(d.N, 64 * NJ);
  d.group_m = bt_group_m(d);
  const int64_t total = (int64_t)d.tiles_m * d.tiles_n * d.nz;
  if (total > 0x3fffffff) return U2_ERR_ARG;
  const int grid = (int)std::min<int64_t>(total, opts().gemm_big_grid);
  if constexpr (NJ == 3 && DEEP == 2) {
    if (d.vt) {
      if (d.tail_rows) hipLaunchKernelGGL((gemm_bt_kernel<3, false, false, 2, true, true>), dim3(grid), dim3(256), 0, stream, d);
      else hipLaunchKernelGGL((gemm_bt_kernel<3, false, false, 2, true>), dim3(grid), dim3(256), 0, stream, d);
      return launch_status();
    }
    if (d.tail_rows) {
      hipLaunchKernelGGL((gemm_bt_kernel<3, false, false, 2, false, true>), dim3(grid), dim3(256), 0, stream, d);
      return launch_status();
    }
  }
  if (d.vt || d.tail_rows) return U2_ERR_ARG;
  hipLaunchKernelGGL((gemm_bt_kernel<NJ, false, false, DEEP>), dim3(grid), dim3(256), 0, stream, d);
  return launch_status();
}

template <int NJ, bool PAIR = false>
static int bt_launch(GemmDesc d, hipStream_t stream) {
  if (d.vt) return U2_ERR_ARG;  // only the deep 256 x 192 form leaves transposed tiles (bt_launch_deep): never drop the request silently
  d.tiles_m = (int)cdiv(d.M, 256);
  d.tiles_n = (int)cdiv(d.N, 64 * NJ);
  d.group_m = bt_group_m(d);
  if (d.ksplit > 1 && (d.nz != 1 || PAIR || !d.partial)) return U2_ERR_ARG;
  const int64_t total = (int64_t)d.tiles_m * d.tiles_n * (d.ksplit > 1 ? d.ksplit : d.nz);
  if (total > 0x3fffffff) return U2_ERR_ARG;
  const int grid = (int)std::min<int64_t>(total, opts().gemm_big_grid);
  if constexpr (!PAIR) {
    if (d.ksplit > 1) {
      hipLaunchKernelGGL((gemm_bt_kernel<NJ, false, true>), dim3(grid), dim3(256), 0, stream, d);
      return gemm_splitk_reduce(d, stream);
    }
  }
  if constexpr (NJ == 4 && !PAIR) {
    if (d.tail_rows) {
      hipLaunchKernelGGL((gemm_bt_kernel<4, false, false, 0, false, true>), dim3(grid), dim3(256), 0, stream, d);
      return launch_status();
    }
  }
  if (d.tail_rows) return U2_ERR_ARG;
  hipLaunchKernelGGL((gemm_bt_kernel<NJ, PAIR>), dim3(grid), dim3(256), 0, stream, d);
  return launch_status();
}

// 64-wide K tiles only (at least two); 32-bit byte offsets into A and B (per z)
static bool bt_legal(const GemmDesc& d) {
  return !(d.K & 63) && d.K >= 128 && (int64_t)d.M * d.lda < (1ll << 30) && (int64_t)d.N * d.ldb < (1ll << 30);
}

// K slices for a product of `tiles` output tiles: fill the 256 CUs, keep >= 4 K tiles per slice (and >= 2 in the last one:
// the K loop's pipeline), within the stream's scratch.  0 / 1 = unsplit.
static int bt_slices(GemmDesc& d, int64_t tiles, int want, hipStream_t stream) {
  d.ksplit = 1;
  const int nkt = d.K >> 6;
  if (want <= 1 || d.nz != 1 || nkt < 8) return 1;
  const Scratch sc = ctx().scratch_of(stream);
  const size_t slice = (size_t)d.M * d.N * sizeof(float);
  if (!sc.p || sc.bytes < 2 * slice) return 1;
  int s = (int)std::min<int64_t>(std::min<int64_t>(want, nkt / 4), (int64_t)(sc.bytes / slice));
  while (s > 1) {
    const int per = (int)cdiv(nkt, s), used = (int)cdiv(nkt, per);
    if (nkt - (used - 1) * per >= 2) {  // last slice long enough
      d.ksplit = used;
      d.kt_per = per;
      d.partial = reinterpret_cast<float*>(sc.p);
      return used > 1 ? used : (d.ksplit = 1);
    }
    --s;
  }
  (void)tiles;
  return 1;
}

// variants: 20 = 256 x 256, 21 = 256 x 192, 22 = 256 x 128 tiles (ring form); 24 = 256 x 192 with B deep, 26 = 256 x 256 with B deep,
// 27 = 24 as the drain form (tile i's epilogue under tile i + 1's K loop)
static int bt_launch_variant(int v, const GemmDesc& d, hipStream_t stream) {
  switch (v) {
    case 20: return bt_launch<4>(d, stream);
    case 22: return bt_launch<2>(d, stream);
    case 23: case 25: return U2_ERR_ARG;  // (the A-deep twins of round 4: removed)
    case 24: return bt_launch_deep<3, 2>(d, stream);
    case 26: return bt_launch_deep<4, 2>(d, stream);
    case 27: return bt_drain_ok(d) ? bt_launch_drain(d, stream) : U2_ERR_ARG;
    default: return bt_launch<3>(d, stream);
  }
}

// The heuristic's choice (20 / 21) as launched: the deep form of the same tile width with B as the three-stage operand (26 / 24) --
// tools/bt_sweep.py / bt_epilogue_probe.py, cold operands, us two-stage -> deep B (deep A beside it), profiles/r04_bt_deep_*.log:
//   ViT q|k|v 67.6 -> 64.2 (65.6), out-projection 32.1 -> 30.6 (31.0), fc2 72.2 -> 67.4 (67.3); 2048 x 12288 x 4096 191.4 -> 163.0 (166.3);
//   1792 x 8192 x 4096 114.9 -> 102.6 (102.9); 4096^3 117.8 -> 110.6 (110.4); 8192^3 890 -> 816 (831) = 1.35 PF/s.
//   (GELU products keep two stages: fc1 + bias + GELU 112.7 us against 123.6 deep, profiles/r04_bt_gelu_forms.log)
static int bt_deep_of(int v, int flags) {
  return (!opts().gemm_big_deep || (flags & GEMM_GELU)) ? v : v == 20 ? 26 : v == 21 ? 24 : v;
}

// Which tile (tools/gpu_check.py ppperf on MI355X, random operands; DESIGN.md section 3 has the tables): the kernel
// runs its K loop at ~50 % of the MFMA peak (8192^3: 1.23-1.29 PF/s; gemm.hip's 128 x 128 tiles: 0.9) but nothing
// overlaps its prologue and epilogue, and a product is as slow as its last round of tiles: it is taken when the tiles
// fill their rounds of 256 workgroups to >= 70 %, with the tile width that needs the fewest (work-weighted) rounds --
// the ViT's N = 2304 / 768 projections are exactly 3 / 1 rounds of 192-wide tiles, N = 3072 exactly 3 rounds of
// 256-wide ones.
static int bt_pick(const GemmDesc& d) {
  if (!bt_legal(d) || d.K < 256) return 0;
  // the GELU epilogue is 256 values per lane of VALU work that the 128 x 128 kernel hides under its second workgroup
  // per CU (fc1 of the ViT: 112 us there, 120 us here in round 1).  Round 4 (packed-math GELU, gelu_fast2): 128 -> 113 us with cold
  // operands (profiles/r04_bt_gelu_forms.log), pipeline 8.98 -> 8.93 ms per volume (r04_ab_gelu_pipeline.log): default on;
  // option "gemm_big_gelu" = 0 keeps GELU products on the 128 x 128 kernel
  if ((d.flags & GEMM_GELU) && !opts().gemm_big_gelu) return 0;
  const int gmax = opts().gemm_big_grid;
  const int64_t tm = cdiv(d.M, 256) * d.nz;
  const int64_t t4 = tm * cdiv(d.N, 256), t3 = tm * cdiv(d.N, 192);
  const int64_t r4 = cdiv(t4, gmax), r3 = cdiv(t3, gmax);
  const double fill4 = (double)d.M * d.N * d.nz / ((double)r4 * gmax * 65536.0);
  const double fill3 = (double)d.M * d.N * d.nz / ((double)r3 * gmax * 49152.0);
  const double c4 = (double)r4, c3 = 0.9 * (double)r3;  // a 192-wide tile takes ~0.9 of the time of a 256-wide one
  if (c3 < c4) return fill3 >= 0.7 ? 21 : (fill4 >= 0.7 ? 20 : 0);
  return fill4 >= 0.7 ? 20 : (fill3 >= 0.7 ? 21 : 0);
}

// The ring form (256 x 128 tiles, variant 22) for products whose 256- and 192-wide tiles leave the CUs a partial round
// (bt_pick: fill < 70 %) while the 128-wide ones make ONE round that is at least three-quarters full: M = 2048 rows against an
// E x E weight (the SVR's output projections), 1024 against 2E x E (the TTA's text k | v), 2048 x 4096 x 6144 (projector) --
// 256 tiles each.  tools/bt_sweep.py, cold weights, us (128^2 kernel -> here; vendor library beside it): 89.7 -> 69.4 (65.9),
// 82.0 -> 64.4 (62.9), 115.1 -> 95.5 (profiles/r04_bt_ring_sweep.log); 192 tiles (prefill q|k|v, 1024 x 6144 x 4096): 71.2 with two K
// slices of 256 x 192 tiles -> 57.5 (r04_bt_counted_waits_ab.log).  With 128 tiles (1024 x 4096 x 4096) it ties the 128^2 kernel.
static int bt_pick_ring(const GemmDesc& d) {
  if (!bt_legal(d) || d.K < 512) return 0;
  if ((d.flags & GEMM_GELU) && !opts().gemm_big_gelu) return 0;
  const int gmax = opts().gemm_big_grid;
  const int64_t t2 = cdiv(d.M, 256) * cdiv(d.N, 128) * d.nz;
  return (t2 * 4 >= (int64_t)gmax * 3 && t2 <= gmax) ? 22 : 0;
}

// Products that leave the 256 CUs a partial round of big tiles, sliced along K so that (tiles x slices) fills them -- the
// cases tools/bt_sweep.py measured ahead of the 128 x 128 kernel with its own split-K (profiles/r03_bt_sweep.log, cold weights,
// us: 128^2 kernel -> here):
//   (a) 129..256 rows against a wide weight (the TTA self-attention's packed q|k|v, 256 x 12288 x 4096): 71.5 -> 51.0
//       (256 x 192 tiles, 4 slices); the 256 x 4096 x 4096 products of the same chain tie at 30 us and stay where they are
//   (b) 512..1024 rows, K >= 8192 (decoder down-projection at prefill, 1024 x 4096 x 12288): 131.7 -> 111.0 (256 x 256, 4)
//   (c) 512..1024 rows whose 192-wide tiles make exactly half a round (prefill q|k|v, 1024 x 6144 x 4096): 77.1 -> 71.6
//       (256 x 192, 2)
// Returns variant | slices << 8, or 0.  The caller falls back to the small-tile kernel when the scratch cannot hold the slices.
static int bt_pick_sliced(const GemmDesc& d) {
  if (!bt_legal(d) || d.nz != 1 || (d.flags & GEMM_GELU)) return 0;
  const int64_t tm = cdiv(d.M, 256);
  if (d.M > 128 && d.M <= 256 && d.N >= 8192 && d.K >= 2048) return 21 | (4 << 8);
  if (d.M >= 512 && d.M <= 1024 && (d.M & 255) == 0) {
    if (d.K >= 8192 && d.N >= 2048 && tm * cdiv(d.N, 256) <= 64) return 20 | (4 << 8);
    const int64_t t3 = tm * (d.N / 192);
    if (d.N % 192 == 0 && d.K >= 4096 && t3 >= 112 && t3 <= 128) return 21 | (2 << 8);
  }
  return 0;
}

// Can the product leave its columns [vt_n0, N) TRANSPOSED in d.vt (GemmDesc::vt) instead of C?  Only the 256 x 192 deep form does that,
// so: the heuristic must pick it for the many-row part of the product (a cls-row tail goes to the few-rows kernel and is written to
// C as usual), nothing forced / sliced, plain bf16 output without bias or residual, whole tiles on both sides of vt_n0, and 256-row
// tiles that do not straddle a chunk of vt_rows keys.  The pipeline asks before it sets d.vt (and runs transpose_bf16 otherwise).
bool gemm_vt_supported(const GemmDesc& d, int vt_n0, int vt_rows) {
  if (opts().gemm_big != 0 || !opts().gemm_big_deep || d.nz != 1 || d.flags & ~GEMM_VEC_OK) return false;
  const int rem = d.M & 255, Mm = d.M - rem;
  if (rem > 64 || Mm < 512 || d.N < 256 || d.K < 128) return false;
  if (vt_n0 % 192 || (d.N - vt_n0) % 192 || vt_n0 <= 0 || vt_n0 >= d.N || vt_rows % 256 || Mm % vt_rows || (d.nsplit > vt_n0)) return false;
  GemmDesc m = d;
  m.M = Mm;
  if (opts().gemm_big_ring && bt_pick(m) == 0 && bt_pick_ring(m) == 22) return false;
  if (opts().gemm_big_skinny && rem == 0 && bt_pick_sliced(d) != 0) return false;  // gemm_big_try takes the sliced forms first
  return bt_pick(m) == 21;
}

// Returns 1 when the product was launched here, 0 when the caller should use gemm.hip's kernel, < 0 on error.
// `d` has been validated by gemm_bf16 (alignment of A / B, GEMM_VEC_OK resolved).
int gemm_big_try(const GemmDesc& d, hipStream_t stream) {
  if (d.vt && !gemm_vt_supported(d, d.vt_n0, d.vt_rows)) return U2_ERR_ARG;  // (the caller did not ask first)
  if (d.flags & GEMM_SWIGLU) {  // only this kernel has the pair form (256 x 192 tiles = 96 output columns); gemm_bf16 validated
    if (!bt_legal(d)) return U2_ERR_ARG;
    const int e = bt_launch<3, true>(d, stream);
    return e == U2_OK ? 1 : e;
  }
  const int mode = opts().gemm_big;
  if (mode < 0) return 0;
  if (!(d.flags & GEMM_VEC_OK) || (d.flags & (GEMM_BIAS_M | GEMM_A_KMAJOR | GEMM_B_KMAJOR)) || (d.N & 7)) return 0;
  switch (d.flags & (GEMM_BIAS_N | GEMM_GELU | GEMM_RESIDUAL | GEMM_OUT_F32)) {
    case 0: case GEMM_OUT_F32: case GEMM_BIAS_N: case GEMM_BIAS_N | GEMM_OUT_F32: case GEMM_BIAS_N | GEMM_GELU:
    case GEMM_BIAS_N | GEMM_RESIDUAL: case GEMM_RESIDUAL: break;
    default: return 0;
  }
  // 16-byte epilogue accesses (8 consecutive n per lane)
  const bool f32 = d.flags & GEMM_OUT_F32;
  if (((uintptr_t)d.C & 15) || (d.ldc & (f32 ? 3 : 7)) || (d.sCb & (f32 ? 3 : 7)) || (d.sCh & (f32 ? 3 : 7))) return 0;
  if ((d.flags & GEMM_BIAS_N) && ((uintptr_t)d.bias & 15)) return 0;
  if ((d.flags & GEMM_RESIDUAL) && (((uintptr_t)d.R & 15) || (d.ldr & 7) || (d.sRb & 7) || (d.sRh & 7))) return 0;
  if (mode > 0) {  // forced (tests, measurements)
    if (!bt_legal(d)) return 0;
    GemmDesc ds = d;
    if (mode <= 22) bt_slices(ds, 0, opts().gemm_big_splitk, stream);
    if (mode == 27 && !bt_drain_ok(ds)) return 0;
    const int e = bt_launch_variant(mode, ds, stream);
    return e == U2_OK ? 1 : e;
  }
  // A few rows past a multiple of 256 (the ViT's cls rows: M = 2049 per chunk, 8 * 2049 per volume) would cost a whole extra
  // row of tiles: they go through the few-rows / small-tile kernel -- for EVERY form, so that a chunk's rows are computed
  // by the same arithmetic whatever the number of chunks in the call (tests/test_gpu_path.py::test_vit_full_size_properties).
  const int rem = d.M & 255;
  const bool split_tail = d.nz == 1 && rem != 0 && rem <= 64 && d.M > 256;
  GemmDesc main = d, tail = d;
  if (split_tail) {
    main.M = d.M - rem;
    tail.M = rem;
    tail.A = d.A + (int64_t)main.M * d.lda;
    tail.C = reinterpret_cast<char*>(d.C) + (int64_t)main.M * d.ldc * (f32 ? 4 : 2);
    if (d.flags & GEMM_RESIDUAL) tail.R = d.R + (int64_t)main.M * d.ldr;
  }
  auto launch = [&](int v) {
    // <= 16 tail rows ride in the launch of the plain 256 x 256 (20) and deep 256 x 192 (24) forms -- what the ViT's products run
    const bool in_launch = split_tail && rem <= 16 && !(d.K & 31) && !(d.flags & GEMM_BIAS_M) && (v == 20 || v == 24 || v == 27) &&
                           opts().gemm_tail_fused && main.ksplit <= 1;
    if (in_launch) main.tail_rows = rem;
    int e = bt_launch_variant(v, main, stream);
    main.tail_rows = 0;
    if (e == U2_OK && split_tail && !in_launch) e = gemm_classic(tail, stream);
    return e == U2_OK ? 1 : e;
  };
  // (ahead of the sliced forms: 192 ring tiles beat 2 x 128 sliced ones)
  if (opts().gemm_big_ring && main.M >= 512 && bt_pick(main) == 0 && bt_pick_ring(main) == 22) return launch(22);
  if (opts().gemm_big_skinny && !split_tail) {
    const int v = bt_pick_sliced(d);
    if (v > 0) {
      GemmDesc ds = d;
      if (bt_slices(ds, 0, v >> 8, stream) == (v >> 8)) {  // (fewer slices than wanted: the small-tile kernel is the better one)
        const int e = bt_launch_variant(v & 0xff, ds, stream);
        return e == U2_OK ? 1 : e;
      }
    }
  }
  if (d.M < 512 || d.N < 256 || d.K < 128) return 0;
  const int v = bt_pick(main);
  if (v == 0) return 0;
  int vv = bt_deep_of(v, d.flags);
  // round 6, the drain form: products whose workgroups walk two or more 256 x 192 tiles hide tile i's epilogue under tile i + 1's K
  // loop.  A GELU product prefers it over the 256-wide two-stage form whenever its 192-wide tiles fill their rounds (fc1 of the ViT:
  // four whole rounds instead of three with 256 GELUs per lane exposed in each).
  if (bt_drain_ok(main)) {
    const int gmax = opts().gemm_big_grid;
    const int64_t t3 = (int64_t)(main.M / 256) * (main.N / 192);
    const double fill3 = (double)t3 / ((double)cdiv(t3, gmax) * gmax);
    if (vv == 24 || ((d.flags & GEMM_GELU) && fill3 >= 0.7)) vv = 27;
  }
  return launch(vv);
}

}  // namespace u2
