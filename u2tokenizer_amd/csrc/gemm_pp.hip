// "Ping-pong" bf16 MFMA GEMM for the large products of the hot path (same contract as gemm.hip:
//   C[z][m][n] = epilogue(alpha * sum_k A[z][m][k] * B[z][n][k]),  A = activations [M][K], B = nn.Linear weight [N][K]).
//
// Why a second kernel: gemm.hip's 128x128 tile with one barrier pair per K step tops out where every
// "load -> barrier -> MFMA" structure does on CDNA4 (the matrix pipe drains at each barrier while all waves
// fetch their fragments at once).  Here one persistent workgroup per CU owns a 256 x BN output tile and its
// 8 waves form two groups of 4 that alternate roles on every barrier:
//
//      barrier #  0      1      2      3      4      5   ...
//      group 0    | L(0) | M(0) | L(1) | M(1) | L(2) | ...      L(t): issue LDS-DMA for K tile t+NS-1, read the
//      group 1    |  --  | L(0) | M(0) | L(1) | M(1) | ...            MFMA fragments of K tile t from LDS
//                                                                M(t): MI*NI*KSTEPS back-to-back MFMAs
//
// Waves w and w+4 share a SIMD (CDNA4 places a workgroup's waves on SIMDs cyclically), so each SIMD always
// has one wave in an M segment: the matrix pipe stays busy while the partner wave does the LDS / global
// traffic.  Global->LDS goes by LDS-DMA (global_load_lds_dwordx4) into a ring of NS stages with COUNTED
// vmcnt waits, so NS-1 K tiles stay in flight across the barriers; barriers are raw s_barrier.
//
//  * group g owns rows [128 g, 128 g + 128) of the tile, its 4 waves a 2 x 2 grid of 64 x BN/2 wave tiles made
//    of v_mfma_f32_32x32x16_bf16 (weight fragment as the A operand: a lane ends up with 4 consecutive n).
//  * LDS stage = [256 + BN rows][BK bf16]; 16-byte chunks XOR-swizzled (applied to the per-lane SOURCE
//    address, the DMA destination is lane-linear) so the 32x32 fragment reads are conflict-free ds_read_b128.
//  * persistent: grid = min(#tiles, #CUs); a workgroup walks tiles b, b + grid, ... as ONE flattened K loop, so
//    the next tile's first K tiles are already in flight while the previous tile's epilogue stores drain; the
//    epilogue of a group runs at the head of its next L segment, under the other group's MFMAs.
//  * XCD-aware tile order (workgroup b runs on XCD b % 8): every XCD gets a compact band of tiles per round.
#include <algorithm>
#include <type_traits>
#include "kernels.h"

namespace u2 {

__device__ uint4 g_pp_zero16;  // zero source for K-tail chunks
// diagnostics (ABL bit 3 builds only): per (workgroup, wave) 8 uint64: sums of s_memtime deltas [L tail (vmcnt wait),
// wait at the barrier closing L, M work, wait at the barrier closing M], segment count, [L issue part, L fragment reads]
__device__ unsigned long long* g_pp_dbg;

// INM: issue the LDS-DMA of a K tile from the M segment (between the MFMAs) instead of the head of the L segment.
// ABL: measurement-only ablations (bit 0: no MFMAs, bit 1: no DMA after the prologue, bit 2: no fragment reads).
// SPLIT: segments per K tile and group (2: a K tile is consumed as two half-depth L/M pairs, which is what lets a
//        2-stage ring of full 128-byte rows (BK = 64) keep a tile in flight for >= 2 segments).
template <int BN_, int BK_, int NS_, bool INM_ = false, int ABL_ = 0, int SPLIT_ = 1>
struct PPCfg {
  static constexpr int BM = 256, BN = BN_, BK = BK_, NS = NS_, ABL = ABL_, SPLIT = SPLIT_;
  static constexpr bool INM = INM_;
  static constexpr int ROWB = BK * 2;            // bytes per LDS row
  static constexpr int CPR = BK / 8;             // 16-byte chunks per row
  static constexpr int RPP = 64 / CPR;           // rows covered by one 1-KiB DMA piece (one wave instruction)
  static constexpr int ROWS = BM + BN;
  static constexpr int STAGE = ROWS * ROWB;
  static constexpr int NP = STAGE / 1024;        // pieces per K tile
  static constexpr int PG0 = (NP / 4 + 1) / 2;   // pieces per wave of group 0 / group 1
  static constexpr int PG1 = NP / 4 - PG0;
  static constexpr int KSTEPS = BK / 16;
  static constexpr int NI = BN / 64;             // 32-wide n fragments per wave (wave tile 64 x BN/2)
  static constexpr int LDS_BYTES = NS * STAGE;
  static_assert(BK == 32 || BK == 64, "BK");
  static_assert(BN % 64 == 0 && NP % 4 == 0 && PG1 >= 1, "piece split");
  static_assert(NS >= ((INM && SPLIT == 1) ? 3 : 2) && LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert((SPLIT == 1 || SPLIT == 2) && KSTEPS % SPLIT == 0 && (SPLIT == 2 || NS >= 3 || !INM), "split");
};

template <int BK>
__device__ __forceinline__ int pp_swz(int row) {
  if constexpr (BK == 64) return (row >> 1) & 7;
  else return (row >> 2) & 3;
}

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
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * d.tiles_n;
  const int group = rem / per_group, in_g = rem - group * per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(d.tiles_m - first_m, GROUP_M);
  bm0 = (first_m + in_g % gsz) * 256;
  bn0 = (in_g / gsz) * BN;
}

// Epilogue of one 256 x BN tile for the calling wave.  The accumulator fragment of v_mfma_f32_32x32x16 leaves a lane
// with 4 consecutive n (register quad q) and its partner lane (+32) with the next 4; v_permlane32_swap exchanges
// quads between the half-waves so that every lane owns 8 CONSECUTIVE n of one row m: 16-byte bias / residual loads
// and bf16 stores, two float4 stores for fp32 output.
//   lane (l31, hi), pair t of fragment (mi, ni):  m = m_base + 32 mi,  n = n_tile + 32 ni + 16 t + 8 hi + [0, 8)
template <class CFG, int G>
__device__ __forceinline__ void pp_epilogue(const GemmDesc& d, f32x16 (&acc)[2][CFG::NI], int z, int bm0, int bn0, int wm2,
                                            int wn2, int lane) {
  constexpr int NI = CFG::NI, BN = CFG::BN;
  const int hi = lane >> 5, l31 = lane & 31;
  const bool out_f32 = d.flags & GEMM_OUT_F32;
  const int zb = z / d.nbh, zh = z - zb * d.nbh;
  char* Cz = reinterpret_cast<char*>(d.C) + (zb * d.sCb + zh * d.sCh) * (out_f32 ? 4 : 2);
  const bf16_t* Rz = (d.flags & GEMM_RESIDUAL) ? d.R + zb * d.sRb + zh * d.sRh : nullptr;
  const int m_base = bm0 + G * 128 + wm2 * 64 + l31;
  const int n_base = bn0 + wn2 * (BN / 2) + 8 * hi;
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
          for (int e = 0; e < 8; ++e) v[e] = acc[mi][ni][8 * t + e] * d.alpha;
          if constexpr (decltype(BIAS_)::value) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[e];
          }
          if constexpr (decltype(GELU_)::value) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
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

#define PP_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define PP_BARRIER()                        \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    __builtin_amdgcn_s_barrier();           \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)

template <class CFG, int G>
__device__ __forceinline__ void pp_group(const GemmDesc& d, char* lds, const int j, const int lane) {
  constexpr int BN = CFG::BN, BK = CFG::BK, NS = CFG::NS, ROWB = CFG::ROWB, CPR = CFG::CPR, RPP = CFG::RPP;
  constexpr int STAGE = CFG::STAGE, KSTEPS = CFG::KSTEPS, NI = CFG::NI;
  constexpr int PG = (G == 0) ? CFG::PG0 : CFG::PG1;  // DMA pieces this wave issues per K tile
  constexpr int P0 = (G == 0) ? 0 : 4 * CFG::PG0;
  constexpr bool INM = CFG::INM;
  constexpr int ABL = CFG::ABL;
  // pieces this wave issued AFTER the K tile that must have landed at its wait point
  constexpr int WAITN = (NS - 2) * PG;
  constexpr int SPLIT = CFG::SPLIT, KH = KSTEPS / SPLIT;
  constexpr bool LATE = INM && SPLIT == 1;  // group 1's wait precedes its issue of the same iteration
  constexpr int WAITN_L = LATE ? (NS - 3) * PG : WAITN;
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm2 = j >> 1, wn2 = j & 1;

  const int tiles_mn = d.tiles_m * d.tiles_n;
  const int total = tiles_mn * d.nz;
  const int gd = gridDim.x, bid = blockIdx.x;
  const int nkt = (d.K + BK - 1) / BK;
  const int my_tiles = (total - bid + gd - 1) / gd;  // >= 1: grid <= total
  const int nit = my_tiles * nkt;                     // flattened (tile, K tile) iterations

  // ---------------- issue cursor: DMA source pointers of the tile being staged
  const bf16_t* src[PG];
  int kc[PG];
  int is_kt = 0, is_round = 0, is_stage = 0;
#define PP_SETUP_ISSUE(round_)                                                                   \
  {                                                                                              \
    int z_, bm0_, bn0_;                                                                          \
    pp_tile<BN>(d, (round_), gd, bid, total, tiles_mn, z_, bm0_, bn0_);                          \
    const int zb_ = z_ / d.nbh, zh_ = z_ - zb_ * d.nbh;                                          \
    const bf16_t* A_ = d.A + zb_ * d.sAb + zh_ * d.sAh;                                          \
    const bf16_t* B_ = d.B + zb_ * d.sBb + zh_ * d.sBh;                                          \
    _Pragma("unroll") for (int i = 0; i < PG; ++i) {                                             \
      const int r_ = (P0 + i * 4 + j) * RPP + lane / CPR;                                        \
      const int gc_ = (lane % CPR) ^ pp_swz<BK>(r_);                                             \
      kc[i] = gc_ * 8;                                                                           \
      if (r_ < 256) src[i] = A_ + (int64_t)min(bm0_ + r_, d.M - 1) * d.lda + gc_ * 8;            \
      else src[i] = B_ + (int64_t)min(bn0_ + r_ - 256, d.N - 1) * d.ldb + gc_ * 8;               \
    }                                                                                            \
  }
#define PP_ISSUE_PIECE(i_)                                                                                      \
  {                                                                                                             \
    const int k0_ = is_kt * BK;                                                                                 \
    const void* g_ = (k0_ + kc[i_] < d.K) ? (const void*)(src[i_] + k0_) : (const void*)&g_pp_zero16;           \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_,                         \
                                     (__attribute__((address_space(3))) void*)(lds + is_stage * STAGE +         \
                                                                               (P0 + j) * 1024 + (i_) * 4096),  \
                                     16, 0, 0);                                                                 \
  }
#define PP_ISSUE_ADVANCE()                                                                                      \
  {                                                                                                             \
    is_stage = (is_stage + 1 == NS) ? 0 : is_stage + 1;                                                         \
    if (++is_kt == nkt) {                                                                                       \
      is_kt = 0;                                                                                                \
      if (++is_round < my_tiles) PP_SETUP_ISSUE(is_round)                                                       \
    }                                                                                                           \
  }
#define PP_ISSUE()                                                          \
  {                                                                         \
    _Pragma("unroll") for (int i = 0; i < PG; ++i) PP_ISSUE_PIECE(i)        \
    PP_ISSUE_ADVANCE()                                                      \
  }

  // ---------------- compute cursor
  f32x16 acc[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  int koff[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) koff[ks] = ((ks * 2 + hi) ^ pp_swz<BK>(l31)) << 4;
  const int a_base = (G * 128 + wm2 * 64 + l31) * ROWB;
  const int b_base = (256 + wn2 * (BN / 2) + l31) * ROWB;
  int c_stage = 0, c_kt = 0, c_round = 0, pend_round = -1;

#define PP_EPILOGUE()                                                                 \
  {                                                                                   \
    int z_, bm0_, bn0_;                                                               \
    pp_tile<BN>(d, pend_round, gd, bid, total, tiles_mn, z_, bm0_, bn0_);             \
    pp_epilogue<CFG, G>(d, acc, z_, bm0_, bn0_, wm2, wn2, lane);                      \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                  \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;          \
    pend_round = -1;                                                                  \
  }

  // ---------------- prologue: NS-1 K tiles in flight, the first one landed before barrier #0
  PP_SETUP_ISSUE(0)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nit) PP_ISSUE()
  if (nit >= NS - 1) PP_WAIT_VM(WAITN);
  else PP_WAIT_VM(0);
  PP_BARRIER();
  if constexpr (G == 1) PP_BARRIER();  // group 1 runs one barrier behind group 0

  bf16x8 xf[KH][2], wf[KH][NI];
  if constexpr (ABL & 4) {  // "no fragment reads": operands are whatever the registers hold
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) asm volatile("" : "=v"(xf[ks][mi]));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) asm volatile("" : "=v"(wf[ks][ni]));
    }
  }
  for (int it = 0;; ++it) {
    if (pend_round >= 0) PP_EPILOGUE()  // head of an L segment: runs under the other group's MFMAs
    if (it == nit) break;
    const bool more = it + NS - 1 < nit;
#pragma unroll
    for (int h = 0; h < SPLIT; ++h) {
      // ======== L segment (the other group is in its M segment)
      if constexpr (!INM && !(ABL & 2)) {
        if (h == 0 && more) PP_ISSUE()
      }
      if constexpr (!(ABL & 4)) {
        const char* sb = lds + c_stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            xf[ks][mi] = *reinterpret_cast<const bf16x8*>(sb + a_base + mi * 32 * ROWB + koff[h * KH + ks]);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            wf[ks][ni] = *reinterpret_cast<const bf16x8*>(sb + b_base + ni * 32 * ROWB + koff[h * KH + ks]);
        }
      }
      // fragment reads retired before the barrier: the stage may be overwritten by DMA issued after it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (G == 1) {  // K tile it+1 (this wave's share) landed before group 0 reads it
        if (h == SPLIT - 1) {
          if (LATE ? (it + NS - 2 < nit) : more) PP_WAIT_VM(WAITN_L);
          else PP_WAIT_VM(0);
        }
      }
      PP_BARRIER();
      // ======== M segment
      __builtin_amdgcn_s_setprio(1);
      if constexpr (ABL & 1) {  // "no MFMAs": keep the fragments live so their reads stay
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) asm volatile("" ::"v"(xf[ks][mi]));
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(wf[ks][ni]));
        }
        if constexpr (INM && !(ABL & 2)) {
          if (h == 0 && more) PP_ISSUE()
        }
      } else {
        constexpr int NM = KH * 2 * NI;                      // MFMAs of the segment
        constexpr int EVERY = NM / PG > 0 ? NM / PG : 1;     // INM: one DMA piece after every EVERY MFMAs
#pragma unroll
        for (int ks = 0; ks < KH; ++ks)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][ni], xf[ks][mi], acc[mi][ni], 0, 0, 0);
              if constexpr (INM && !(ABL & 2)) {
                const int m_idx = (ks * 2 + mi) * NI + ni;
                if (h == 0 && m_idx % EVERY == EVERY - 1 && m_idx / EVERY < PG) {
                  if (more) PP_ISSUE_PIECE(m_idx / EVERY)
                }
              }
            }
        if constexpr (INM && !(ABL & 2)) {
          if (h == 0 && more) {
#pragma unroll
            for (int i = NM / EVERY; i < PG; ++i) PP_ISSUE_PIECE(i)
            PP_ISSUE_ADVANCE()
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (G == 0) {
        if (h == SPLIT - 1) {
          if (more) PP_WAIT_VM(WAITN);
          else PP_WAIT_VM(0);
        }
      }
      PP_BARRIER();
    }
    c_stage = (c_stage + 1 == NS) ? 0 : c_stage + 1;
    if (++c_kt == nkt) {
      c_kt = 0;
      pend_round = c_round++;
    }
  }
  if constexpr (G == 0) PP_BARRIER();
#undef PP_SETUP_ISSUE
#undef PP_ISSUE
#undef PP_ISSUE_PIECE
#undef PP_ISSUE_ADVANCE
#undef PP_EPILOGUE
}

// ------------------------------------------------------------------------------------------------ "SB" schedule
// Two LDS stages of full 128-byte rows (BK = 64), every K tile consumed as two half-depth L/M pairs, and the refill
// of a stage cut into its four row blocks, each issued in the first L segment after the block's last reader:
//
//   seg (mod 4 of K tile t)      0 = G0 L(t,0)        1 = G1 L(t,0)        2 = G0 L(t,1)        3 = G1 L(t,1)
//   DMA batch issued there       B rows lo of t+1     B rows hi of t+1     A rows 128.. of t+1  A rows 0..127 of t+2
//
// (G0 = group 0 reads A rows 0..127 and all B rows, G1 reads A rows 128..255 and all B rows.)  One batch (16 KB at
// BN = 256) goes out per segment, i.e. the DMA runs continuously at 64 KB per K tile instead of in one burst per tile,
// every batch has >= 2 segments of flight before the barrier that precedes its first reader, and a wave only ever
// waits with vmcnt(size of the batch it has just issued).
//
// (A register-staged form of the same schedule -- global_load_dwordx4 -> VGPR -> ds_write_b128 one L segment later --
// was measured too: no faster, see profiles/r01_gemm_pp_study.log; it is not kept.)
template <class CFG, int G>
__device__ __forceinline__ void pp_group_sb(const GemmDesc& d, char* lds, const int j, const int lane) {
  constexpr int BN = CFG::BN, BK = CFG::BK, ROWB = CFG::ROWB, STAGE = CFG::STAGE, NI = CFG::NI, ABL = CFG::ABL;
  static_assert(BK == 64 && CFG::NS == 2 && CFG::SPLIT == 2, "SB schedule: BK 64, 2 stages, split K tile");
  constexpr int KH = 2;                     // k steps of 16 per segment
  constexpr int PA = 128 / 8 / 4;           // pieces per wave of an A row block (128 rows, 8 rows per piece, 4 waves)
  constexpr int PB = (BN / 2) / 8 / 4;      // ... of a B row block (BN/2 rows)
  static_assert((BN / 2) % 32 == 0, "B row block");
  // batch X is issued in L(.,0), batch Y in L(.,1)
  constexpr int PX = PB, PY = PA;
  constexpr int ROWX = 256 + (G == 0 ? 0 : BN / 2);   // first stage row of batch X (B lo for G0, B hi for G1)
  constexpr int ROWY = (G == 0 ? 128 : 0);            // first stage row of batch Y (A hi for G0, A lo for G1)
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm2 = j >> 1, wn2 = j & 1;

  const int tiles_mn = d.tiles_m * d.tiles_n;
  const int total = tiles_mn * d.nz;
  const int gd = gridDim.x, bid = blockIdx.x;
  const int nkt = (d.K + BK - 1) / BK;
  const int my_tiles = (total - bid + gd - 1) / gd;
  const int nit = my_tiles * nkt;

  // ---------------- two issue cursors (batch X and batch Y run on different K tiles for group 1)
  const bf16_t* sx[PX];
  const bf16_t* sy[PY];
  int kx[PX], ky[PY];
  int x_kt = 0, x_round = 0, x_it = 0, y_kt = 0, y_round = 0, y_it = 0;
#define SB_SETUP(src_, kc_, np_, row0_, round_)                                                   \
  {                                                                                               \
    int z_, bm0_, bn0_;                                                                           \
    pp_tile<BN>(d, (round_), gd, bid, total, tiles_mn, z_, bm0_, bn0_);                           \
    const int zb_ = z_ / d.nbh, zh_ = z_ - zb_ * d.nbh;                                           \
    const bf16_t* A_ = d.A + zb_ * d.sAb + zh_ * d.sAh;                                           \
    const bf16_t* B_ = d.B + zb_ * d.sBb + zh_ * d.sBh;                                           \
    _Pragma("unroll") for (int i = 0; i < (np_); ++i) {                                           \
      const int r_ = (row0_) + (i * 4 + j) * 8 + (lane >> 3);                                     \
      const int gc_ = (lane & 7) ^ pp_swz<BK>(r_);                                                \
      kc_[i] = gc_ * 8;                                                                           \
      if ((row0_) < 256) src_[i] = A_ + (int64_t)min(bm0_ + r_, d.M - 1) * d.lda + gc_ * 8;       \
      else src_[i] = B_ + (int64_t)min(bn0_ + r_ - 256, d.N - 1) * d.ldb + gc_ * 8;               \
    }                                                                                             \
  }
// The source pointers RUN (advanced by BK after the batch): every piece reads its own, already valid address
// registers, so the pieces issue back to back.  (Computing "base + k0" into one temporary per piece makes each
// address computation wait until the previous global_load_lds has read that register pair.)  K % 64 == 0 here.
#define SB_ISSUE(src_, kc_, np_, row0_, c_kt_, c_round_, c_it_)                                                       \
  {                                                                                                                   \
    char* s_ = lds + (c_it_ & 1) * STAGE + ((row0_) + j * 8) * ROWB;                                                  \
    _Pragma("unroll") for (int i = 0; i < (np_); ++i)                                                                 \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_[i],                        \
                                       (__attribute__((address_space(3))) void*)(s_ + i * 32 * ROWB), 16, 0, 0);      \
    ++c_it_;                                                                                                          \
    if (++c_kt_ == nkt) {                                                                                             \
      c_kt_ = 0;                                                                                                      \
      if (++c_round_ < my_tiles) SB_SETUP(src_, kc_, np_, row0_, c_round_)                                            \
    } else {                                                                                                          \
      _Pragma("unroll") for (int i = 0; i < (np_); ++i) src_[i] += BK;                                                \
    }                                                                                                                 \
  }
#define SB_ISSUE_X() SB_ISSUE(sx, kx, PX, ROWX, x_kt, x_round, x_it)
#define SB_ISSUE_Y() SB_ISSUE(sy, ky, PY, ROWY, y_kt, y_round, y_it)

  f32x16 acc[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = ((ks * 2 + hi) ^ pp_swz<BK>(l31)) << 4;
  const int a_base = (G * 128 + wm2 * 64 + l31) * ROWB;
  const int b_base = (256 + wn2 * (BN / 2) + l31) * ROWB;
  int c_kt = 0, c_round = 0, pend_round = -1;

#define SB_EPILOGUE()                                                                 \
  {                                                                                   \
    int z_, bm0_, bn0_;                                                               \
    pp_tile<BN>(d, pend_round, gd, bid, total, tiles_mn, z_, bm0_, bn0_);             \
    pp_epilogue<CFG, G>(d, acc, z_, bm0_, bn0_, wm2, wn2, lane);                      \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                  \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;          \
    pend_round = -1;                                                                  \
  }

  // ---------------- prologue: K tile 0 complete (each group its own two batches), plus A rows 0..127 of K tile 1
  SB_SETUP(sx, kx, PX, ROWX, 0)
  SB_SETUP(sy, ky, PY, ROWY, 0)
  SB_ISSUE_X()
  SB_ISSUE_Y()
  if constexpr (G == 1) {
    if (1 < nit) {
      SB_ISSUE_Y()
      PP_WAIT_VM(PY);
    } else {
      PP_WAIT_VM(0);
    }
  } else {
    PP_WAIT_VM(0);
  }
  PP_BARRIER();
  if constexpr (G == 1) PP_BARRIER();

  bf16x8 xf[KH][2], wf[KH][NI];
  unsigned long long tsum[4] = {0, 0, 0, 0}, tsum4 = 0, tsum5 = 0, tprev = 0, nseg = 0;
  if constexpr (ABL & 8) tprev = __builtin_amdgcn_s_memtime();
  for (int it = 0;; ++it) {
    if (pend_round >= 0) SB_EPILOGUE()
    if (it == nit) break;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // ======== L segment: one DMA batch, then this half's fragments
      if constexpr (ABL & 8) {  // time since the previous stamp = wait at the barrier that closed the last M segment
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        tsum[3] += t - tprev; tprev = t;
      }
      bool issued = false;
      if constexpr (!(ABL & 2)) {
        if (h == 0) {
          if (x_it < nit) { SB_ISSUE_X() issued = true; }
        } else {
          if (y_it < nit) { SB_ISSUE_Y() issued = true; }
        }
      }
      if constexpr (ABL & 8) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        tsum4 += t - tprev; tprev = t;
      }
      {
        const char* sb = lds + (it & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            xf[ks][mi] = *reinterpret_cast<const bf16x8*>(sb + a_base + mi * 32 * ROWB + koff[h * KH + ks]);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            wf[ks][ni] = *reinterpret_cast<const bf16x8*>(sb + b_base + ni * 32 * ROWB + koff[h * KH + ks]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (ABL & 8) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        tsum5 += t - tprev; tprev = t;
      }
      // everything this wave issued before the batch of this segment has landed
      if (issued) {
        if (h == 0) PP_WAIT_VM(PX);
        else PP_WAIT_VM(PY);
      } else {
        PP_WAIT_VM(0);
      }
      if constexpr (ABL & 8) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        tsum[0] += t - tprev; tprev = t;
      }
      PP_BARRIER();
      // ======== M segment
      if constexpr (ABL & 8) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        tsum[1] += t - tprev; tprev = t;
      }
      if constexpr (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);
      if constexpr (ABL & 1) {
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) asm volatile("" ::"v"(xf[ks][mi]));
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(wf[ks][ni]));
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < KH; ++ks)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][ni], xf[ks][mi], acc[mi][ni], 0, 0, 0);
      }
      if constexpr (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);
      if constexpr (ABL & 8) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        tsum[2] += t - tprev; tprev = t;
        ++nseg;
      }
      PP_BARRIER();
    }
    if (++c_kt == nkt) {
      c_kt = 0;
      pend_round = c_round++;
    }
  }
  if constexpr (G == 0) PP_BARRIER();
  if constexpr (ABL & 8) {
    if (lane == 0 && g_pp_dbg) {
      unsigned long long* o = g_pp_dbg + ((size_t)blockIdx.x * 8 + G * 4 + j) * 8;
      o[0] = tsum[0]; o[1] = tsum[1]; o[2] = tsum[2]; o[3] = tsum[3]; o[4] = nseg; o[5] = tsum4; o[6] = tsum5;
    }
  }
#undef SB_SETUP
#undef SB_ISSUE
#undef SB_ISSUE_X
#undef SB_ISSUE_Y
#undef SB_EPILOGUE
}

template <int BN, int ABL>
__global__ __launch_bounds__(512) void gemm_pp_sb_kernel(GemmDesc d) {
  using CFG = PPCfg<BN, 64, 2, false, ABL, 2>;
  __shared__ __attribute__((aligned(16))) char lds[CFG::LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) pp_group_sb<CFG, 0>(d, lds, wave, lane);
  else pp_group_sb<CFG, 1>(d, lds, wave - 4, lane);
}

template <int BN, int BK, int NS, bool INM, int ABL, int SPLIT>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmDesc d) {
  using CFG = PPCfg<BN, BK, NS, INM, ABL, SPLIT>;
  __shared__ __attribute__((aligned(16))) char lds[CFG::LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) pp_group<CFG, 0>(d, lds, wave, lane);
  else pp_group<CFG, 1>(d, lds, wave - 4, lane);
}

// ------------------------------------------------------------------------------------------------ launcher
static int g_pp_mode = 0;      // 0: heuristic, -1: never, v > 0: force variant v where the shape allows it
static int g_pp_max_grid = 256;

int gemm_pp_set_debug_buffer(void* p) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pp_dbg), &q, sizeof(q)) == hipSuccess ? U2_OK : U2_ERR_LAUNCH;
}

void gemm_pp_set_options(int mode, int max_grid) {
  if (mode >= -1) g_pp_mode = mode;
  if (max_grid > 0) g_pp_max_grid = max_grid;
}

template <int BN, int BK, int NS, bool INM = false, int ABL = 0, int SPLIT = 1>
static int pp_launch(GemmDesc d, hipStream_t stream) {
  d.tiles_m = (int)cdiv(d.M, 256);
  d.tiles_n = (int)cdiv(d.N, BN);
  const int64_t total = (int64_t)d.tiles_m * d.tiles_n * d.nz;
  if (total > 0x3fffffff) return U2_ERR_ARG;
  const int grid = (int)std::min<int64_t>(total, g_pp_max_grid);
  hipLaunchKernelGGL((gemm_pp_kernel<BN, BK, NS, INM, ABL, SPLIT>), dim3(grid), dim3(512), 0, stream, d);
  return launch_status();
}

template <int BN, int ABL = 0>
static int pp_launch_sb(GemmDesc d, hipStream_t stream) {
  if (d.K % 64) return pp_launch<256, 64, 2, true, 0, 2>(d, stream);  // SB streams whole 64-wide K tiles only
  d.tiles_m = (int)cdiv(d.M, 256);
  d.tiles_n = (int)cdiv(d.N, BN);
  const int64_t total = (int64_t)d.tiles_m * d.tiles_n * d.nz;
  if (total > 0x3fffffff) return U2_ERR_ARG;
  const int grid = (int)std::min<int64_t>(total, g_pp_max_grid);
  hipLaunchKernelGGL((gemm_pp_sb_kernel<BN, ABL>), dim3(grid), dim3(512), 0, stream, d);
  return launch_status();
}

// ----------------------------------------------------------------------------------------------------------------
// "Big tile" kernel (variant 20): 256 x 256 x 64 tiles, ONE workgroup of 4 waves per CU, each wave a 128 x 128 output
// tile (4 x 4 MFMA accumulators = 256 AccVGPRs), its K loop one generated asm block (gemm_bt_asm.inc, written by
// tools/gen_gemm_bt_asm.py, which documents the slot schedule).  What it is after: with one wave per SIMD and a
// hand-placed instruction stream the matrix pipe only drains at the single barrier per K tile, every LDS-DMA piece
// and fragment read sits in the shadow of an MFMA, and a 256^2 tile needs half the LDS fill bandwidth per flop of
// the 128^2 kernel (the measured limit of the CU's L2 -> LDS path, ~50 B/clk, profiles/r01_stage_bw.log).
// Requirements checked by the launcher: K % 64 == 0, operands addressable with 32-bit byte offsets.
#include "gemm_bt_asm.inc"
typedef int bt_i32x4 __attribute__((ext_vector_type(4)));

template <int NJ>  // 32-column blocks per wave: tile = 256 x (64 NJ)
__global__ __launch_bounds__(256, 1) void gemm_bt_kernel(GemmDesc d) {
  constexpr int BN = 64 * NJ;
  using CFG = PPCfg<BN, 64, 2>;
  __shared__ __attribute__((aligned(1024))) char lds[131072];  // [stage][A tile 32 KB | B tile <= 32 KB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const int tiles_mn = d.tiles_m * d.tiles_n;
  const int total = tiles_mn * d.nz;
  const int gd = gridDim.x, bid = blockIdx.x;
  const int my_tiles = (total - bid + gd - 1) / gd;  // >= 1: grid <= total
  // DMA pieces of this wave: rows [64 w, 64 w + 64) of the A tile and [16 NJ w, ..) of the B tile, 8 rows x 128 B per
  // piece; LDS position p of row r holds global chunk p ^ ((r >> 1) & 7): even / odd pieces differ by 4 in that term.
  // Lane offsets are relative to the tile origin; the origin (and the K tile) travel in the scalar offset.
  const int pr = lane >> 3, sw0 = (lane >> 4) & 3, pc = lane & 7;
  const int ra = wave * 64 + pr, rb = wave * (16 * NJ) + pr;
  const int va0 = (ra * (int)d.lda + ((pc ^ sw0) << 3)) * 2;
  const int va1 = ((ra + 8) * (int)d.lda + ((pc ^ sw0 ^ 4) << 3)) * 2;
  const int vb0 = (rb * (int)d.ldb + ((pc ^ sw0) << 3)) * 2;
  const int vb1 = ((rb + 8) * (int)d.ldb + ((pc ^ sw0 ^ 4) << 3)) * 2;
  const uint32_t lds_u32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&lds[0];  // 0: the only LDS object
  const uint32_t abk0 = (uint32_t)(l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4));
  const int lda16 = 16 * (int)d.lda * 2, ldb16 = 16 * (int)d.ldb * 2;
  const int nkt = d.K >> 6;
  const bool chain = d.nz == 1;  // one K loop runs on from tile to tile (same descriptors)
  uint32_t st0 = 0;              // LDS stage that holds K tile 0 of the current output tile
  int z, bm0, bn0;
  pp_tile<BN>(d, 0, gd, bid, total, tiles_mn, z, bm0, bn0);
  for (int r = 0; r < my_tiles; ++r) {
    int zn = z, bm0n = bm0, bn0n = bn0;  // next output tile of this workgroup (itself after the last one: its K
    if (r + 1 < my_tiles) pp_tile<BN>(d, r + 1, gd, bid, total, tiles_mn, zn, bm0n, bn0n);  // loop prefetches in-bounds garbage)
    const int zb = z / d.nbh, zh = z - zb * d.nbh;
    const bf16_t* A = d.A + zb * d.sAb + zh * d.sAh;
    const bf16_t* B = d.B + zb * d.sBb + zh * d.sBh;
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
    const uint32_t aa0 = (lds_u32 + wm * 16384 + abk0) ^ st0, ab0 = (lds_u32 + 32768 + wn * (NJ * 4096) + abk0) ^ st0;
    // (readfirstlane: the values are uniform, but hipcc keeps loop-carried tile coordinates in VGPRs)
    const int base_a = __builtin_amdgcn_readfirstlane(bm0 * (int)d.lda * 2);
    const int base_b = __builtin_amdgcn_readfirstlane(bn0 * (int)d.ldb * 2);
    const int nbase_a = __builtin_amdgcn_readfirstlane((chain ? bm0n : bm0) * (int)d.lda * 2);
    const int nbase_b = __builtin_amdgcn_readfirstlane((chain ? bn0n : bn0) * (int)d.ldb * 2);
    const int st0_s = __builtin_amdgcn_readfirstlane((int)st0), first_s = __builtin_amdgcn_readfirstlane(first);
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
      [rsb] "s"(rsb), [lda16] "s"(lda16), [ldb16] "s"(ldb16), [nkt] "s"(nkt), [wave] "s"(wave), [st0] "s"(st0_s),     \
      [first] "s"(first_s), [base_a] "s"(base_a), [base_b] "s"(base_b), [nbase_a] "s"(nbase_a), [nbase_b] "s"(nbase_b)
    if constexpr (NJ == 4) {
      asm volatile(GEMM_BT_ASM_TEXT_NJ4
                   : BT_ACC3(0, 0), [c003] "+a"(acc[0][0][3]), BT_ACC3(0, 1), [c013] "+a"(acc[0][1][3]), BT_ACC3(1, 0),
                     [c103] "+a"(acc[1][0][3]), BT_ACC3(1, 1), [c113] "+a"(acc[1][1][3])
                   : BT_IN
                   : GEMM_BT_ASM_CLOBBERS);
    } else {
      asm volatile(GEMM_BT_ASM_TEXT_NJ3 : BT_ACC3(0, 0), BT_ACC3(0, 1), BT_ACC3(1, 0), BT_ACC3(1, 1) : BT_IN : GEMM_BT_ASM_CLOBBERS);
    }
#undef BT_ACC3
#undef BT_IN
    // pp_epilogue's row base is bm0 + 128 G + 64 wm2: G = 0 with the wave's 128-row offset folded into bm0 (its "tile
    // inside C" fast-path test then only errs towards the predicated path)
    pp_epilogue<CFG, 0>(d, acc[0], z, bm0 + wm * 128, bn0, 0, wn, lane);
    pp_epilogue<CFG, 0>(d, acc[1], z, bm0 + wm * 128, bn0, 1, wn, lane);
    st0 ^= (uint32_t)(nkt & 1) << 16;
    z = zn; bm0 = bm0n; bn0 = bn0n;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last K loop's prefetch into LDS
}

template <int NJ>
static int bt_launch(GemmDesc d, hipStream_t stream) {
  // 64-wide K tiles only (at least two); 32-bit byte offsets into A and B (per z)
  if ((d.K & 63) || d.K < 128 || (int64_t)d.M * d.lda >= (1ll << 30) || (int64_t)d.N * d.ldb >= (1ll << 30))
    return pp_launch<256, 64, 2, true, 0, 2>(d, stream);
  d.tiles_m = (int)cdiv(d.M, 256);
  d.tiles_n = (int)cdiv(d.N, 64 * NJ);
  const int64_t total = (int64_t)d.tiles_m * d.tiles_n * d.nz;
  if (total > 0x3fffffff) return U2_ERR_ARG;
  const int grid = (int)std::min<int64_t>(total, g_pp_max_grid);
  hipLaunchKernelGGL(gemm_bt_kernel<NJ>, dim3(grid), dim3(256), 0, stream, d);
  return launch_status();
}

// Variant ids (u2tok_set_option("gemm_pp", id) forces one; every id computes the same C unless marked otherwise).
// The many other (BN, BK, NS, schedule) points that were measured on MI355X are recorded in DESIGN.md section 3.
static int pp_launch_variant(int v, const GemmDesc& d, hipStream_t stream) {
  switch (v) {
    case 1: return pp_launch<256, 64, 2, true, 0, 2>(d, stream);   // 256x256, DMA issued between the MFMAs
    case 2: return pp_launch<192, 64, 2, false, 0, 2>(d, stream);  // 256x192, DMA issued at the head of L
    case 3: return pp_launch<128, 64, 3, true, 0, 2>(d, stream);   // 256x128, 3 stages
    case 4: return pp_launch<256, 32, 4>(d, stream);               // 256x256, 64-byte rows, 4 stages
    case 5: return pp_launch_sb<256>(d, stream);                   // row-block ("SB") DMA schedule
    case 6: return pp_launch_sb<192>(d, stream);
    case 7: return pp_launch_sb<128>(d, stream);
    case 20: return bt_launch<4>(d, stream);                       // 256x256 big tile, asm K loop, 1 workgroup of 4 waves per CU
    case 21: return bt_launch<3>(d, stream);                       // 256x192 big tile
    // measurement-only builds (results of 10..13 are wrong by construction; 14..17 are correct but slower)
    case 10: return pp_launch<256, 64, 2, true, 1, 2>(d, stream);  // variant 1 without MFMAs
    case 11: return pp_launch<256, 64, 2, true, 2, 2>(d, stream);  // ... without DMA after the prologue
    case 12: return pp_launch<256, 64, 2, true, 6, 2>(d, stream);  // ... MFMAs + barriers only
    case 13: return pp_launch<256, 64, 2, true, 5, 2>(d, stream);  // ... DMA + barriers only
    case 14: return pp_launch_sb<256, 8>(d, stream);               // variant 5, s_memtime instrumented
    case 15: return pp_launch_sb<256, 9>(d, stream);               // variant 5 instrumented, no MFMAs (wrong C)
    case 16: return pp_launch_sb<256, 24>(d, stream);              // variant 5 instrumented, no s_setprio
    case 17: return pp_launch_sb<256, 10>(d, stream);              // variant 5 instrumented, no DMA (wrong C)
    default: return U2_ERR_ARG;
  }
}

// Which kernel (tools/gpu_check.py ppperf on MI355X, random operands; DESIGN.md section 3 has the tables):
//  * the big-tile kernel (variants 20 / 21: 256 x 256 / 256 x 192 tiles, one 4-wave workgroup per CU, asm K loop) runs
//    its K loop at ~50 % of the MFMA peak (8192^3: 1.23-1.29 PF/s; gemm.hip's 128 x 128 tiles: 0.9) but nothing
//    overlaps its prologue and epilogue, and a product is as slow as its last round of tiles: it is taken when the
//    tiles fill their rounds of 256 workgroups to >= 70 %, with the tile width that needs the fewest (work-weighted)
//    rounds -- the ViT's N = 2304 / 768 projections are exactly 3 / 1 rounds of 192-wide tiles, N = 3072 exactly 3
//    rounds of 256-wide ones.  Not with the GELU epilogue (256 values per lane of VALU work that the 128 x 128
//    kernel hides under its second workgroup per CU: 112 us either way for the fc1 product).
//  * the 8-wave ping-pong kernel (variant 1) is behind both on every shape measured since; it stays selectable.
static int pp_pick(const GemmDesc& d) {
  if ((d.K & 63) || d.K < 256 || (d.flags & GEMM_GELU)) return 0;
  if ((int64_t)d.M * d.lda >= (1ll << 30) || (int64_t)d.N * d.ldb >= (1ll << 30)) return 0;
  const int64_t tm = cdiv(d.M, 256) * d.nz;
  const int64_t t4 = tm * cdiv(d.N, 256), t3 = tm * cdiv(d.N, 192);
  const int64_t r4 = cdiv(t4, g_pp_max_grid), r3 = cdiv(t3, g_pp_max_grid);
  const double fill4 = (double)d.M * d.N * d.nz / ((double)r4 * g_pp_max_grid * 65536.0);
  const double fill3 = (double)d.M * d.N * d.nz / ((double)r3 * g_pp_max_grid * 49152.0);
  const double c4 = (double)r4, c3 = 0.9 * (double)r3;  // a 192-wide tile takes ~0.9 of the time of a 256-wide one
  if (c3 < c4) return fill3 >= 0.7 ? 21 : (fill4 >= 0.7 ? 20 : 0);
  return fill4 >= 0.7 ? 20 : (fill3 >= 0.7 ? 21 : 0);
}

// Returns 1 when the product was launched here, 0 when the caller should use gemm.hip's kernel, < 0 on error.
// `d` has been validated by gemm_bf16 (alignment of A / B, GEMM_VEC_OK resolved).
int gemm_pp_try(const GemmDesc& d, hipStream_t stream) {
  if (g_pp_mode < 0) return 0;
  if (!(d.flags & GEMM_VEC_OK) || (d.flags & GEMM_BIAS_M) || (d.N & 7)) return 0;
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
  if (g_pp_mode > 0) {
    const int e = pp_launch_variant(g_pp_mode, d, stream);
    return e == U2_OK ? 1 : e;
  }
  if (d.M < 512 || d.N < 256 || d.K < 128) return 0;
  // A few rows past a multiple of 256 (the ViT's 8 cls rows: M = 8 * 2049) would cost a whole extra round of
  // 256-row tiles: run them through the small-tile kernel instead.
  const int rem = d.M & 255;
  if (d.nz == 1 && rem != 0 && rem <= 64) {
    GemmDesc main = d, tail = d;
    main.M = d.M - rem;
    const int v = pp_pick(main);
    if (v == 0) return 0;
    tail.M = rem;
    tail.A = d.A + (int64_t)main.M * d.lda;
    tail.C = reinterpret_cast<char*>(d.C) + (int64_t)main.M * d.ldc * (f32 ? 4 : 2);
    if (d.flags & GEMM_RESIDUAL) tail.R = d.R + (int64_t)main.M * d.ldr;
    int e = pp_launch_variant(v, main, stream);
    if (e != U2_OK) return e;
    e = gemm_classic(tail, stream);
    return e == U2_OK ? 1 : e;
  }
  const int v = pp_pick(d);
  if (v == 0) return 0;
  const int e = pp_launch_variant(v, d, stream);
  return e == U2_OK ? 1 : e;
}

}  // namespace u2
