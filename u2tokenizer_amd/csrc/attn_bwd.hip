// Backward of the d = 64 self-attention of the ViT blocks (MONAI SABlock, vit.py:100-105):
//   out = softmax(q k^T * scale) v   ->   dq, dk, dv   from q, k, v, out, d_out
// without the (S x S) probability / score-gradient tensors the unfused backward round-trips through HBM (1.6 GB fp32 +
// 0.8 GB bf16, several times per layer at S = 2049).  Two kernels, no atomics, bit-repeatable:
//
//   flash_bwd_dq_kernel   one workgroup per (batch, head, 128 query rows), 4 waves x 32 queries, two sweeps over the keys:
//                         sweep 1 rebuilds the row statistics (lse = log2 sum exp2(s c), c = scale log2 e) and
//                         D = rowsum(d_out * out); sweep 2 computes  S^T = K Q^T,  dP^T = V dO^T,
//                         dS^T = P^T (dP^T - D)  and  dQ^T += K^T dS^T.  A lane owns one QUERY (column of the 32x32 MFMA
//                         result) and 16 keys per block, so lse / D are lane scalars and dS^T feeds the next MFMA from
//                         the lane's own registers: its k slots carry the keys of a 16-group in the order
//                         [0-3, 8-11 | 4-7, 12-15] (lane halves), and the K^T fragment of the other operand is gathered
//                         in exactly that order from the row-major K tile by two ds_read_b64_tr_b16 (a 16-lane group
//                         reads a [4 keys][16 d] block; lane a supplies row a >> 2, piece a & 3, receives column a).
//   flash_bwd_dkv_kernel  one workgroup per (batch, head, 128 keys), 4 waves x 32 keys, one sweep over the queries in the
//                         other orientation: S = Q K^T, dP = dO V^T with a lane owning one KEY and 16 queries per block;
//                         dV^T += dO^T P,  dK^T += Q^T dS  again from registers, the Q^T / dO^T fragments transpose-read
//                         from the same row-major Q / dO tiles that feed S and dP.  lse / D come from the first kernel
//                         through the workspace.
//
// Cost: 8 matmul units of 2 S^2 64 flop per head (the unfused form has 5 plus ~16 GB of HBM traffic per ViT layer);
// HBM: q, k, v, out, d_out read, dq, dk, dv written, 2 x 4 bytes per (head, row) of statistics -- no transposed copies.
#include "kernels.h"

namespace u2 {

namespace {

struct FlashBwdArgs {
  const bf16_t *q, *k, *v, *o, *dout;  // row-major views, head h at column h*64
  bf16_t *dq, *dk, *dv;
  float *lse, *dsum;  // (nb*H, S_pad)
  const float* lse_in;  // optional row statistics of the forward kernel, (nb*H, lse_ld)
  int64_t lse_ld;
  int S, H, S_pad, nblk, nwg;  // nblk: 128-row blocks per (batch, head); nwg = nb * H * nblk
  int64_t ld_qkv, bs_qkv, ld_o, bs_o, ld_d, bs_d;
  float scale, scale_log2e;
};

// [64][64] bf16 tiles, 128-byte rows, 16-byte chunks XOR-swizzled with the BIT-REVERSED row pair index rev3((row >> 1) & 7).
// Two access patterns share a tile: the 32 x 32 row fragments (ds_read_b128: 16 consecutive rows at one chunk -> any
// bijection of the 8 row pairs onto the 8 slots is conflict-free, as attn.hip's plain (row >> 1) & 7) and the transpose
// reads (4 consecutive rows x 4 consecutive chunks per half wave): rows r and r + 2 must land in different 64-byte groups,
// i.e. their XOR values must differ in bit 2 -- rev3(p) ^ rev3(p + 1) = 4 for even p.  With the plain swizzle the
// transpose reads were 2-way conflicts (15-20 % of the LDS cycles, profiles/r02_kernel_pmc.json).
__device__ __forceinline__ uint32_t tile_off(int row, int chunk) {
  const int p = (row >> 1) & 7;
  const int x = ((p & 1) << 2) | (p & 2) | ((p >> 2) & 1);
  return (uint32_t)(row * 128 + ((chunk ^ x) << 4));
}

__device__ __forceinline__ float dot8(const uint4 a, const uint4 b) {
  float s = 0.f;
  s = __builtin_fmaf(bf16lo(a.x), bf16lo(b.x), s); s = __builtin_fmaf(bf16hi(a.x), bf16hi(b.x), s);
  s = __builtin_fmaf(bf16lo(a.y), bf16lo(b.y), s); s = __builtin_fmaf(bf16hi(a.y), bf16hi(b.y), s);
  s = __builtin_fmaf(bf16lo(a.z), bf16lo(b.z), s); s = __builtin_fmaf(bf16hi(a.z), bf16hi(b.z), s);
  s = __builtin_fmaf(bf16lo(a.w), bf16lo(b.w), s); s = __builtin_fmaf(bf16hi(a.w), bf16hi(b.w), s);
  return s;
}

// XCD-aware order (as attn.hip): workgroup w runs on XCD w % 8; every XCD gets a contiguous range of logical ids so that
// the blocks of one (batch, head) share that XCD's L2 copy of the operands they all stream (K, V, K^T resp. Q, dO, Q^T,
// dO^T): with the plain (block, head) grid every head was fetched by all eight L2s -- 0.93 GB of fabric-side reads per
// launch against 0.15 GB of operands (profiles/r02_kernel_pmc.json).
__device__ __forceinline__ int xcd_order(int w, int nwg) {
  const int qn = nwg >> 3, rn = nwg & 7;
  const int xcd = w & 7, idx = w >> 3;
  return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
}

union Frag {
  bf16x8 v;
  uint4 q;
  uint32_t u[4];
};

typedef short v4s_t __attribute__((ext_vector_type(4)));

// A operand "X^T" (32 d rows x 16 k) of v_mfma_f32_32x32x16_bf16 from a row-major [64 rows][64 d] tile (rows = keys or
// queries = the contraction index), in the k-slot order of an accumulator fed back as the B operand: lane (d = lane & 31,
// hi = lane >> 5) needs rows r0 + 4 hi + {0..3} and r0 + 8 + 4 hi + {0..3} of column d.  Two transpose reads; `lane_off` is
// trf_lane_off() (the part that depends on the lane and on nb only), r0 = 32 blk + 16 ks2 a compile-time constant.
__device__ __forceinline__ int trf_row(int lane) { return 4 * (lane >> 5) + ((lane & 15) >> 2); }
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int lane, int nb, int r0) {
  typedef __attribute__((address_space(3))) v4s_t* lds_v4;
  const int a = lane & 15;
  const int chunk = 4 * nb + 2 * ((lane >> 4) & 1) + ((a & 3) >> 1), sub = (a & 1) * 8;
  const int row1 = r0 + trf_row(lane), row2 = row1 + 8;
  const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + tile_off(row1, chunk) + sub));
  const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + tile_off(row2, chunk) + sub));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// ----------------------------------------------------------------------------------------------------------- dQ
template <bool HAVE_LSE>  // the forward kernel's row statistics are given: no sweep 1
__global__ __launch_bounds__(256, 2) void flash_bwd_dq_kernel(const FlashBwdArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[2][2][8192];  // [stage][K | V]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int S = a.S, S_pad = a.S_pad;
  const int vid = xcd_order(blockIdx.x, a.nwg);
  const int bh = vid / a.nblk, b = bh / a.H, h = bh - b * a.H;
  const float c = a.scale_log2e;
  const int wrow0 = (vid - bh * a.nblk) * 128 + wv * 32;
  const bool wave_active = wrow0 < S;
  const int qrow = min(wrow0 + l31, S - 1);
  const bf16_t* kb_ = a.k + (int64_t)b * a.bs_qkv + h * 64;
  const bf16_t* vb_ = a.v + (int64_t)b * a.bs_qkv + h * 64;
  const int64_t ld = a.ld_qkv;

  // Q / dO fragments (B operands: lane = query column, 8 d values per k16 step) and D = rowsum(dO * O)
  Frag qf[4], dof[4];
  float dpart = 0.f;
  {
    const bf16_t* qp = a.q + (int64_t)b * a.bs_qkv + (int64_t)qrow * ld + h * 64 + hi * 8;
    const bf16_t* op = a.o + (int64_t)b * a.bs_o + (int64_t)qrow * a.ld_o + h * 64 + hi * 8;
    const bf16_t* gp = a.dout + (int64_t)b * a.bs_o + (int64_t)qrow * a.ld_o + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks].q = *reinterpret_cast<const uint4*>(qp + ks * 16);
      dof[ks].q = *reinterpret_cast<const uint4*>(gp + ks * 16);
      const uint4 ov = *reinterpret_cast<const uint4*>(op + ks * 16);
      dpart += dot8(ov, dof[ks].q);
    }
  }
  const float Dq = dpart + __shfl_xor(dpart, 32, 64);

  const int srow0 = tid >> 3, srow1 = 32 + (tid >> 3), sch = tid & 7;
  const uint32_t soff0 = tile_off(srow0, sch), soff1 = tile_off(srow1, sch);
  uint4 rk0, rk1, rv0, rv1;
  rv0 = rv1 = uint4{0u, 0u, 0u, 0u};
#define U2_DQ_GLOAD(t_, full_)                                                                  \
  do {                                                                                          \
    const int kv0_ = (t_) * 64;                                                                 \
    const int64_t r0_ = (int64_t)min(kv0_ + srow0, S - 1) * ld + sch * 8;                       \
    const int64_t r1_ = (int64_t)min(kv0_ + srow1, S - 1) * ld + sch * 8;                       \
    rk0 = *reinterpret_cast<const uint4*>(kb_ + r0_);                                           \
    rk1 = *reinterpret_cast<const uint4*>(kb_ + r1_);                                           \
    if (full_) {                                                                                \
      rv0 = *reinterpret_cast<const uint4*>(vb_ + r0_);                                         \
      rv1 = *reinterpret_cast<const uint4*>(vb_ + r1_);                                         \
    }                                                                                           \
  } while (0)
#define U2_DQ_LSTORE(st_, full_)                                       \
  do {                                                                 \
    *reinterpret_cast<uint4*>(&lds[st_][0][soff0]) = rk0;              \
    *reinterpret_cast<uint4*>(&lds[st_][0][soff1]) = rk1;              \
    if (full_) {                                                       \
      *reinterpret_cast<uint4*>(&lds[st_][1][soff0]) = rv0;            \
      *reinterpret_cast<uint4*>(&lds[st_][1][soff1]) = rv1;            \
    }                                                                  \
  } while (0)

  const int ntile = (S + 63) >> 6;
  const bool ragged = (S & 63) != 0;

  float m_run = -INFINITY, l_run = 0.f;
  if constexpr (!HAVE_LSE) {
    // ---------------- sweep 1: lse (log2 units) of the lane's query row
    U2_DQ_GLOAD(0, false);
    U2_DQ_LSTORE(0, false);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks].v), "+v"(dof[ks].v));  // loads done before the loops
    for (int t = 0; t < ntile; ++t) {
      const int st = t & 1;
      if (t + 1 < ntile) U2_DQ_GLOAD(t + 1, false);
      if (wave_active) {
        const char* sK = lds[st][0];
        f32x16 sc[2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[kbk][r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + tile_off(kbk * 32 + l31, ks * 2 + hi));
            sc[kbk] = mfma32(kf, qf[ks].v, sc[kbk]);
          }
        }
        // lane owns keys t*64 + kbk*32 + (r&3) + 8*(r>>2) + 4*hi
        if (ragged && t == ntile - 1) {
          const int kvb = t * 64 + 4 * hi;
#pragma unroll
          for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (kvb + kbk * 32 + (r & 3) + 8 * (r >> 2) >= S) sc[kbk][r] = -INFINITY;
        }
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx[r & 3] = fmaxf(mx[r & 3], sc[kbk][r]);
        float mt = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * c;  // scale > 0
        const float m_new = fmaxf(m_run, mt);        // finite from tile 0 on: every tile but the last is full, S >= 1
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int r = 0; r < 16; ++r) ps[r & 3] += __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kbk][r], c, -m_new));
        l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + ((ps[0] + ps[1]) + (ps[2] + ps[3]));
        m_run = m_new;
      }
      if (t + 1 < ntile) U2_DQ_LSTORE((t + 1) & 1, false);
      __syncthreads();
    }
  }
  float lse = 0.f;
  if constexpr (HAVE_LSE) {
    lse = a.lse_in[(int64_t)bh * a.lse_ld + qrow];
  } else if (wave_active) {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    lse = m_run + __builtin_log2f(l_tot);
  }
  if (wave_active && hi == 0 && wrow0 + l31 < S) {  // for the dK / dV kernel
    a.lse[(int64_t)bh * S_pad + wrow0 + l31] = lse;
    a.dsum[(int64_t)bh * S_pad + wrow0 + l31] = Dq;
  }

  // ---------------- sweep 2: dQ^T
  f32x16 acc[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  U2_DQ_GLOAD(0, true);
  U2_DQ_LSTORE(0, true);
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks].v), "+v"(dof[ks].v));  // loads done before the loop
  for (int t = 0; t < ntile; ++t) {
    const int st = t & 1;
    if (t + 1 < ntile) U2_DQ_GLOAD(t + 1, true);
    if (wave_active) {
      const char* sK = lds[st][0];
      const char* sV = lds[st][1];
      f32x16 sc[2], dp[2];
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[kbk][r] = 0.f; dp[kbk][r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + tile_off(kbk * 32 + l31, ks * 2 + hi));
          sc[kbk] = mfma32(kf, qf[ks].v, sc[kbk]);
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + tile_off(kbk * 32 + l31, ks * 2 + hi));
          dp[kbk] = mfma32(vf, dof[ks].v, dp[kbk]);
        }
      }
      if (ragged && t == ntile - 1) {
        const int kvb = t * 64 + 4 * hi;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kvb + kbk * 32 + (r & 3) + 8 * (r >> 2) >= S) sc[kbk][r] = -INFINITY;  // p = 0 below
      }
      // dS^T = P^T (dP^T - D), unscaled (the scale is applied once in the epilogue)
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kbk][r], c, -lse));
          sc[kbk][r] = p * (dp[kbk][r] - Dq);
        }
      // dQ^T += K^T dS^T: k-slots jj of step (kbk, ks2) carry keys kbk*32 + 16*ks2 + 8*(jj>>2) + 4*hi + (jj&3); the K^T
      // fragment is transpose-read from the row-major K tile in that order
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          Frag pf;
#pragma unroll
          for (int j = 0; j < 4; ++j) pf.u[j] = pack2_bf16(sc[kbk][ks2 * 8 + 2 * j], sc[kbk][ks2 * 8 + 2 * j + 1]);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const bf16x8 tf = tr_frag(sK, lane, nb, kbk * 32 + ks2 * 16);
            acc[nb] = mfma32(tf, pf.v, acc[nb]);
          }
        }
    }
    if (t + 1 < ntile) U2_DQ_LSTORE((t + 1) & 1, true);
    __syncthreads();
  }
#undef U2_DQ_GLOAD
#undef U2_DQ_LSTORE
  // lane: q = lane & 31, d = nb*32 + 8*g + 4*hi + e
  if (wave_active && wrow0 + l31 < S) {
    bf16_t* op = a.dq + (int64_t)b * a.bs_d + (int64_t)(wrow0 + l31) * a.ld_d + h * 64;
    const float sc_ = a.scale;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<uint2*>(op + nb * 32 + 8 * g + 4 * hi) =
            uint2{pack2_bf16(acc[nb][4 * g] * sc_, acc[nb][4 * g + 1] * sc_),
                  pack2_bf16(acc[nb][4 * g + 2] * sc_, acc[nb][4 * g + 3] * sc_)};
  }
}

// ----------------------------------------------------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(256, 2) void flash_bwd_dkv_kernel(const FlashBwdArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[2][2][8192];  // [stage][Q | dO]
  __shared__ __attribute__((aligned(16))) float stat[2][2][64];  // [stage][lse | D][query of the tile]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int S = a.S, S_pad = a.S_pad;
  const int vid = xcd_order(blockIdx.x, a.nwg);
  const int bh = vid / a.nblk, b = bh / a.H, h = bh - b * a.H;
  const float c = a.scale_log2e;
  const int wkey0 = (vid - bh * a.nblk) * 128 + wv * 32;
  const bool wave_active = wkey0 < S;
  const int krow = min(wkey0 + l31, S - 1);
  const int64_t ld = a.ld_qkv;

  // K / V fragments (B operands: lane = key column)
  Frag kf[4], vf[4];
  {
    const bf16_t* kp = a.k + (int64_t)b * a.bs_qkv + (int64_t)krow * ld + h * 64 + hi * 8;
    const bf16_t* vp = a.v + (int64_t)b * a.bs_qkv + (int64_t)krow * ld + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks].q = *reinterpret_cast<const uint4*>(kp + ks * 16);
      vf[ks].q = *reinterpret_cast<const uint4*>(vp + ks * 16);
    }
  }
  const bf16_t* qb_ = a.q + (int64_t)b * a.bs_qkv + h * 64;
  const bf16_t* gb_ = a.dout + (int64_t)b * a.bs_o + h * 64;
  const float* lseb_ = a.lse + (int64_t)bh * S_pad;
  const float* dsb_ = a.dsum + (int64_t)bh * S_pad;

  const int srow0 = tid >> 3, srow1 = 32 + (tid >> 3), sch = tid & 7;
  const uint32_t soff0 = tile_off(srow0, sch), soff1 = tile_off(srow1, sch);
  uint4 rq0, rq1, rg0, rg1;
  float4 rs = {0.f, 0.f, 0.f, 0.f};
#define U2_DKV_GLOAD(t_)                                                                                   \
  do {                                                                                                     \
    const int q0_ = (t_) * 64;                                                                             \
    const int qr0_ = min(q0_ + srow0, S - 1), qr1_ = min(q0_ + srow1, S - 1);                              \
    rq0 = *reinterpret_cast<const uint4*>(qb_ + (int64_t)qr0_ * ld + sch * 8);                             \
    rq1 = *reinterpret_cast<const uint4*>(qb_ + (int64_t)qr1_ * ld + sch * 8);                             \
    rg0 = *reinterpret_cast<const uint4*>(gb_ + (int64_t)qr0_ * a.ld_o + sch * 8);                         \
    rg1 = *reinterpret_cast<const uint4*>(gb_ + (int64_t)qr1_ * a.ld_o + sch * 8);                         \
    if (tid < 32) {                                                                                        \
      const int qq_ = q0_ + (tid & 15) * 4;                                                                \
      rs = *reinterpret_cast<const float4*>((tid < 16 ? lseb_ : dsb_) + qq_);                              \
      /* queries past the end: lse = +inf makes their probabilities exactly 0 */                           \
      const float fill_ = tid < 16 ? INFINITY : 0.f;                                                       \
      if (qq_ + 0 >= S) rs.x = fill_;                                                                      \
      if (qq_ + 1 >= S) rs.y = fill_;                                                                      \
      if (qq_ + 2 >= S) rs.z = fill_;                                                                      \
      if (qq_ + 3 >= S) rs.w = fill_;                                                                      \
    }                                                                                                      \
  } while (0)
#define U2_DKV_LSTORE(st_)                                                                     \
  do {                                                                                         \
    *reinterpret_cast<uint4*>(&lds[st_][0][soff0]) = rq0;                                      \
    *reinterpret_cast<uint4*>(&lds[st_][0][soff1]) = rq1;                                      \
    *reinterpret_cast<uint4*>(&lds[st_][1][soff0]) = rg0;                                      \
    *reinterpret_cast<uint4*>(&lds[st_][1][soff1]) = rg1;                                      \
    if (tid < 32) *reinterpret_cast<float4*>(&stat[st_][tid >> 4][(tid & 15) * 4]) = rs;      \
  } while (0)

  f32x16 accV[2], accK[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { accV[nb][r] = 0.f; accK[nb][r] = 0.f; }

  const int ntile = (S + 63) >> 6;
  U2_DKV_GLOAD(0);
  U2_DKV_LSTORE(0);
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(kf[ks].v), "+v"(vf[ks].v));
  for (int t = 0; t < ntile; ++t) {
    const int st = t & 1;
    if (t + 1 < ntile) U2_DKV_GLOAD(t + 1);
    if (wave_active) {
      const char* sQ = lds[st][0];
      const char* sG = lds[st][1];
#pragma unroll
      for (int qbk = 0; qbk < 2; ++qbk) {
        // S = Q K^T, dP = dO V^T: lane = key column, rows = queries qbk*32 + (r&3) + 8*(r>>2) + 4*hi
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 qa = *reinterpret_cast<const bf16x8*>(sQ + tile_off(qbk * 32 + l31, ks * 2 + hi));
          s = mfma32(qa, kf[ks].v, s);
          const bf16x8 ga = *reinterpret_cast<const bf16x8*>(sG + tile_off(qbk * 32 + l31, ks * 2 + hi));
          dp = mfma32(ga, vf[ks].v, dp);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 l4 = *reinterpret_cast<const float4*>(&stat[st][0][qbk * 32 + 8 * g + 4 * hi]);
          const float4 d4 = *reinterpret_cast<const float4*>(&stat[st][1][qbk * 32 + 8 * g + 4 * hi]);
          const float le[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[4 * g + e], c, -le[e]));
            s[4 * g + e] = p;
            dp[4 * g + e] = p * (dp[4 * g + e] - de[e]);
          }
        }
        // dV^T += dO^T P, dK^T += Q^T dS: k-slots jj of step ks2 carry queries qbk*32 + 16*ks2 + 8*(jj>>2) + 4*hi + (jj&3);
        // the dO^T / Q^T fragments are transpose-read from the row-major tiles in that order
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          Frag pp, ps;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            pp.u[j] = pack2_bf16(s[ks2 * 8 + 2 * j], s[ks2 * 8 + 2 * j + 1]);
            ps.u[j] = pack2_bf16(dp[ks2 * 8 + 2 * j], dp[ks2 * 8 + 2 * j + 1]);
          }
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const bf16x8 gt = tr_frag(sG, lane, nb, qbk * 32 + ks2 * 16);
            accV[nb] = mfma32(gt, pp.v, accV[nb]);
            const bf16x8 qt = tr_frag(sQ, lane, nb, qbk * 32 + ks2 * 16);
            accK[nb] = mfma32(qt, ps.v, accK[nb]);
          }
        }
      }
    }
    if (t + 1 < ntile) U2_DKV_LSTORE((t + 1) & 1);
    __syncthreads();
  }
#undef U2_DKV_GLOAD
#undef U2_DKV_LSTORE
  // lane: key = lane & 31, d = nb*32 + 8*g + 4*hi + e
  if (wave_active && wkey0 + l31 < S) {
    bf16_t* kp = a.dk + (int64_t)b * a.bs_d + (int64_t)(wkey0 + l31) * a.ld_d + h * 64;
    bf16_t* vp = a.dv + (int64_t)b * a.bs_d + (int64_t)(wkey0 + l31) * a.ld_d + h * 64;
    const float sc_ = a.scale;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = nb * 32 + 8 * g + 4 * hi;
        *reinterpret_cast<uint2*>(kp + d0) = uint2{pack2_bf16(accK[nb][4 * g] * sc_, accK[nb][4 * g + 1] * sc_),
                                                   pack2_bf16(accK[nb][4 * g + 2] * sc_, accK[nb][4 * g + 3] * sc_)};
        *reinterpret_cast<uint2*>(vp + d0) = uint2{pack2_bf16(accV[nb][4 * g], accV[nb][4 * g + 1]),
                                                   pack2_bf16(accV[nb][4 * g + 2], accV[nb][4 * g + 3])};
      }
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

size_t flash_attention_d64_bwd_workspace_bytes(int nb, int S, int H) {
  if (nb <= 0 || S <= 0 || H <= 0) return 0;
  const size_t S_pad = ((size_t)S + 63) & ~(size_t)63;
  return 2 * align256((size_t)nb * H * S_pad * 4);
}

int flash_attention_d64_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld_qkv, int64_t bs_qkv, const bf16_t* o,
                            const bf16_t* dout, int64_t ld_o, int64_t bs_o, bf16_t* dq, bf16_t* dk, bf16_t* dv, int64_t ld_d,
                            int64_t bs_d, int nb, int S, int H, float scale, const float* lse_in, int64_t lse_ld,
                            void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!q || !k || !v || !o || !dout || !dq || !dk || !dv || !workspace) return U2_ERR_ARG;
  if (nb <= 0 || S <= 0 || H <= 0 || !(scale > 0.f) || (int64_t)nb * H > 65535) return U2_ERR_ARG;
  if ((int64_t)nb * H * ((S + 127) / 128) > 0x7fffffff) return U2_ERR_ARG;
  if ((ld_qkv & 7) || (bs_qkv & 7) || (ld_o & 7) || (bs_o & 7) || (ld_d & 3) || (bs_d & 3)) return U2_ERR_ARG;
  if (ld_qkv < (int64_t)H * 64 || ld_o < (int64_t)H * 64 || ld_d < (int64_t)H * 64) return U2_ERR_ARG;
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout) & 15) ||
      (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 7) || ((uintptr_t)workspace & 255))
    return U2_ERR_ARG;
  if (lse_in && (lse_ld < S || ((uintptr_t)lse_in & 3))) return U2_ERR_ARG;
  if (workspace_bytes < flash_attention_d64_bwd_workspace_bytes(nb, S, H)) return U2_ERR_WORKSPACE;
  const int S_pad = (S + 63) & ~63;
  const int E = H * 64;
  char* w = static_cast<char*>(workspace);
  const size_t sb = align256((size_t)nb * H * S_pad * 4);
  float* lse = reinterpret_cast<float*>(w);
  float* dsum = reinterpret_cast<float*>(w + sb);
  int e = U2_OK;
  FlashBwdArgs a;
  a.q = q; a.k = k; a.v = v; a.o = o; a.dout = dout;
  a.dq = dq; a.dk = dk; a.dv = dv; a.lse = lse; a.dsum = dsum;
  a.lse_in = lse_in; a.lse_ld = lse_ld;
  a.S = S; a.H = H; a.S_pad = S_pad;
  a.ld_qkv = ld_qkv; a.bs_qkv = bs_qkv; a.ld_o = ld_o; a.bs_o = bs_o; a.ld_d = ld_d; a.bs_d = bs_d;
  a.scale = scale;
  a.scale_log2e = scale * 1.44269504088896340736f;
  a.nblk = (S + 127) / 128;
  a.nwg = nb * H * a.nblk;
  const dim3 grid((unsigned)a.nwg);
  const double unit = 2.0 * (double)nb * H * (double)S * S * 64;
  {
    ProfScope ps(PROF_FLASH, (lse_in ? 3.0 : 4.0) * unit, stream, (double)nb * S * E * 2.0 * 6.0);
    if (lse_in) hipLaunchKernelGGL(flash_bwd_dq_kernel<true>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(flash_bwd_dq_kernel<false>, grid, dim3(256), 0, stream, a);
  }
  e = launch_status();
  if (e != U2_OK) return e;
  {
    ProfScope ps(PROF_FLASH, 4.0 * unit, stream, (double)nb * S * E * 2.0 * 8.0);
    hipLaunchKernelGGL(flash_bwd_dkv_kernel, grid, dim3(256), 0, stream, a);
  }
  return launch_status();
}

}  // namespace u2
