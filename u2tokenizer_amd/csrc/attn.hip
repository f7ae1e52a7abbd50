// Attention cores that do not go through the batched-GEMM path:
//   * temporal_attention : the across-chunk half of SpatioTemporalAttentionLayer (svr.py:31-36), sequence
//     length T = number of chunks (<= 16 here; longer sequences take the batched-GEMM path, pipeline.hip),
//     head dim d = E/8 in {64, 128, 256, 512}.  Reads q/k/v in the (b t n) row
//     order the projections produced them in, so neither of the reference's two permute().contiguous()
//     round trips (svr.py:32,36) touches HBM.
//   * flash_attention_d64: MONAI SABlock attention of the ViT blocks (vit.py:100-105): S = 2049 tokens,
//     12 heads x 64.  Online-softmax, never materialises the (S x S) probabilities the reference writes
//     (1.6 GB fp32 per layer per volume).
#include "kernels.h"

namespace u2 {

// ================================================================= temporal attention
// One wave64 per (b, n, head).  QK^T on the matrix core straight from HBM fragments (K as the MFMA A
// operand, Q as B: lane holds S[t1 = lane & 15][t2 = 4*(lane >> 4) + r]); softmax across the 4 lane
// groups with two xor-shuffles; P through a 1 KB LDS patch; PV on the VALU with each lane owning a
// contiguous d/64 slice of the head (coalesced 16-byte row reads of V, lane-local accumulation).
template <int VPL>  // bf16 elements of the head owned per lane: d = 64 * VPL
__global__ __launch_bounds__(256) void temporal_attention_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                 const bf16_t* __restrict__ v, bf16_t* __restrict__ out, int B,
                                                                 int T, int N, int H, int64_t ld_qkv, int64_t ld_out,
                                                                 float scale, const bf16_t* __restrict__ rel_bias, int max_len) {
  constexpr int d = 64 * VPL;
  __shared__ float pl[4][16][17];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t inst = (int64_t)blockIdx.x * 4 + wv;  // (b, n, h)
  const bool active = inst < (int64_t)B * N * H;
  const int h = active ? (int)(inst % H) : 0;
  const int64_t bn = active ? inst / H : 0;
  const int n = (int)(bn % N);
  const int b = (int)(bn / N);
  // row (b, t, n) -> ((b*T + t)*N + n)
  const int64_t row0 = (int64_t)b * T * N + n;
  const int64_t rstep = N;

  // ---- S^T tile = K Q^T (16x16, rows >= T are zero fragments)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  {
    const int t = lane & 15, g = lane >> 4;
    const bool valid = active && t < T;
    const bf16_t* qp = q + (row0 + (int64_t)t * rstep) * ld_qkv + h * d + g * 8;
    const bf16_t* kp = k + (row0 + (int64_t)t * rstep) * ld_qkv + h * d + g * 8;
#pragma unroll 4
    for (int ks = 0; ks < d / 32; ++ks) {
      bf16x8 qa = {0, 0, 0, 0, 0, 0, 0, 0}, ka = {0, 0, 0, 0, 0, 0, 0, 0};
      if (valid) {
        qa = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
        ka = *reinterpret_cast<const bf16x8*>(kp + ks * 32);
      }
      acc = mfma16(ka, qa, acc);
    }
  }
  // lane: t1 = lane & 15 (query), t2 = 4*(lane>>4) + r (key)
  {
    const int t1 = lane & 15, t2b = (lane >> 4) * 4;
    float s[4];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t2 = t2b + r;
      float x = acc[r] * scale;
      if (rel_bias && t1 < T && t2 < T) x += bf16_to_f32(rel_bias[(int64_t)(t2 - t1 + max_len - 1) * H + h]);
      s[r] = (t2 < T) ? x : -INFINITY;
      m = fmaxf(m, s[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[r] = __expf(s[r] - m); sum += s[r]; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int r = 0; r < 4; ++r) pl[wv][t1][t2b + r] = s[r] * inv;
  }
  __syncthreads();
  if (!active) return;
  // ---- O = P V on the VALU; lane owns columns [lane*VPL, lane*VPL + VPL) of the head
  float o[16][VPL];
#pragma unroll
  for (int t1 = 0; t1 < 16; ++t1)
#pragma unroll
    for (int j = 0; j < VPL; ++j) o[t1][j] = 0.f;
  const bf16_t* vp = v + row0 * ld_qkv + h * d + lane * VPL;
  for (int t2 = 0; t2 < T; ++t2) {
    float vv[VPL];
    const bf16_t* p = vp + (int64_t)t2 * rstep * ld_qkv;
    if constexpr (VPL == 8) {
      const uint4 u = *reinterpret_cast<const uint4*>(p);
      vv[0] = bf16lo(u.x); vv[1] = bf16hi(u.x); vv[2] = bf16lo(u.y); vv[3] = bf16hi(u.y);
      vv[4] = bf16lo(u.z); vv[5] = bf16hi(u.z); vv[6] = bf16lo(u.w); vv[7] = bf16hi(u.w);
    } else if constexpr (VPL == 4) {
      const uint2 u = *reinterpret_cast<const uint2*>(p);
      vv[0] = bf16lo(u.x); vv[1] = bf16hi(u.x); vv[2] = bf16lo(u.y); vv[3] = bf16hi(u.y);
    } else if constexpr (VPL == 2) {
      const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
      vv[0] = bf16lo(u); vv[1] = bf16hi(u);
    } else {
      vv[0] = bf16_to_f32(p[0]);
    }
#pragma unroll
    for (int t1 = 0; t1 < 16; ++t1) {
      const float pw = pl[wv][t1][t2];  // broadcast read
#pragma unroll
      for (int j = 0; j < VPL; ++j) o[t1][j] += pw * vv[j];
    }
  }
  bf16_t* op = out + row0 * ld_out + h * d + lane * VPL;
#pragma unroll
  for (int t1 = 0; t1 < 16; ++t1) {
    if (t1 < T) {
      bf16_t* p = op + (int64_t)t1 * rstep * ld_out;
      if constexpr (VPL == 8) {
        *reinterpret_cast<uint4*>(p) = uint4{pack2_bf16(o[t1][0], o[t1][1]), pack2_bf16(o[t1][2], o[t1][3]),
                                             pack2_bf16(o[t1][4], o[t1][5]), pack2_bf16(o[t1][6], o[t1][7])};
      } else if constexpr (VPL == 4) {
        *reinterpret_cast<uint2*>(p) = uint2{pack2_bf16(o[t1][0], o[t1][1]), pack2_bf16(o[t1][2], o[t1][3])};
      } else if constexpr (VPL == 2) {
        *reinterpret_cast<uint32_t*>(p) = pack2_bf16(o[t1][0], o[t1][1]);
      } else {
        p[0] = f32_to_bf16(o[t1][0]);
      }
    }
  }
}

int temporal_attention(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int B, int T, int N, int H,
                       int d, int64_t ld_qkv, int64_t ld_out, float scale, const bf16_t* rel_bias, int max_len,
                       hipStream_t stream) {
  if (!q || !k || !v || !out || B <= 0 || T <= 0 || T > 16 || N <= 0 || H <= 0) return U2_ERR_ARG;
  if (d != 64 && d != 128 && d != 256 && d != 512) return U2_ERR_ARG;
  if ((ld_qkv & 7) || (ld_out & 7) || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15)) return U2_ERR_ARG;
  if (rel_bias && T > max_len) return U2_ERR_ARG;
  const int64_t ninst = (int64_t)B * N * H;
  dim3 grid((unsigned)cdiv(ninst, 4));
  ProfScope ps(PROF_TEMPORAL, 4.0 * ninst * T * T * d, stream);
#define U2_TA(VPL)                                                                                                   \
  hipLaunchKernelGGL((temporal_attention_kernel<VPL>), grid, dim3(256), 0, stream, q, k, v, out, B, T, N, H, ld_qkv, \
                     ld_out, scale, rel_bias, max_len)
  if (d == 512) U2_TA(8);
  else if (d == 256) U2_TA(4);
  else if (d == 128) U2_TA(2);
  else U2_TA(1);
#undef U2_TA
  return launch_status();
}

// ================================================================= flash attention, head dim 64
// Sequence = S "main" rows per batch (contiguous, leading dim ld) + an optional single EXTRA row per batch stored
// elsewhere (the ViT keeps its cls token after all patch rows, so that the 2048 patch rows tile exactly and the
// GEMMs see M = 16384 + 8 instead of 8 x 2049).
//
// A pass = 4 waves x QB blocks of 32 query rows against all keys; KV tiles of 64 keys, v_mfma_f32_32x32x16_bf16.
//   S^T = K Q^T : A = K fragment (LDS), B = Q fragment (registers, loaded once).  Lane (q = lane&31, hi) then
//                 owns 16 keys of each 32-key block -> row max/sum are lane-local + one xor-32.
//   O^T = V^T P^T: A = V^T fragment (LDS, V pre-transposed in HBM), B = P^T = the lane's own exp'd scores packed
//                 to bf16 -- NO cross-lane movement: the k-slot -> key map of the MFMA is a free permutation as
//                 long as A and B agree, so V^T is stored in P's native order: inside every group of 16 keys the
//                 column order is [0-3, 8-11, 4-7, 12-15] (transpose_bf16(..., perm16 = 1)), which makes each
//                 lane's 8 k-slots one 16-byte read.
//   QB = 2 reads every K / V^T fragment once for TWO MFMAs: the QB = 1 form spends ~80 % of the LDS bandwidth
//   of a CU on fragment reads (each wave re-reads the whole 16 KB tile for 32 rows), QB = 2 halves that.
//   The online-softmax rescale is lane-local (O^T keeps q = lane&31 per lane) and deferred until some row's
//   running max grows by more than 2^8 (wave-uniform branch); the softmax scale is folded into one FMA per score;
//   key masking only in a partial last tile; the extra key is one VALU step after the tile loop.
// This plain pass (QB = 1, "mode 1") serves short sequences; S >= 512 takes the double pipeline further down
// ("mode 7").  The QB = 2 / mixed-unit / 8-wave ping-pong forms measured in round 1 (profiles/r01_flash_study.log) were
// removed from the product in round 2 (history: fc1ca7f).
// Extra query rows (one per head) are handled by small VALU workgroups at the end of the grid.
// LDS tiles ([64][64] bf16, 128 B rows): 16-byte chunks XOR-swizzled with (row>>1)&7 -> conflict-free
// ds_read_b128 for the 32x32 fragment pattern.
__device__ __forceinline__ uint32_t kt_off(int row, int chunk) {  // 16-B chunk index 0..7
  return (uint32_t)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

struct FlashArgs {
  const bf16_t *q, *k, *vt;
  bf16_t* out;
  const bf16_t *qx, *kx, *vx;
  bf16_t* outx;
  int S, H, nb, S_pad, n_extra, mode, n_main;
  int q_prescaled;  // q and qx already carry scale * log2 e (mode 7 only)
  int wide_out;     // out rows are 16-byte aligned: the double-pipeline forms store 16 bytes per lane
  int64_t ld_qk, q_bs, ld_out, out_bs, x_bs, ox_bs;
  float scale_log2e;
  float* lse;      // optional: lse[(b * H + h) * lse_ld + row] = log2 sum_k exp2(s_k scale log2e) per query row (the extra
  int64_t lse_ld;  // row at index S), what the fused backward (attn_bwd.hip) would otherwise rebuild in a sweep of its own
};

constexpr float FLASH_RESCALE_THR = 8.0f;

__device__ unsigned long long* g_flash_dbg;  // diagnostics: TIMED builds add s_memtime deltas per phase here

template <int QB, bool TIMED = false>
__device__ __forceinline__ void flash_pass(const FlashArgs& a, char (*lds)[16384], const int b, const int h, const int row0,
                                           const int tid) {
  unsigned long long ts[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define U2_STAMP(i_)                                                   \
  if constexpr (TIMED) {                                               \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
    ts[i_] += t_ - tprev;                                              \
    tprev = t_;                                                        \
  }
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int S = a.S;
  const bf16_t* qb_ = a.q + (int64_t)b * a.q_bs + h * 64;
  const bf16_t* kb_ = a.k + (int64_t)b * a.q_bs + h * 64;
  const bf16_t* vb_ = a.vt + ((int64_t)b * a.H + h) * 64 * a.S_pad;
  const int64_t ld_qk = a.ld_qk;
  const int S_pad = a.S_pad;
  const float scale_log2e = a.scale_log2e;

  // a wave whose rows are all past the sequence end only helps with staging
  const int wrow0 = row0 + wv * 32 * QB;
  const bool wave_active = wrow0 < S;
  bf16x8 qf[QB][4];  // Q fragments (B operand): Q[q][d = ks*16 + hi*8 .. +7]
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const bf16_t* qp = qb_ + (int64_t)min(wrow0 + qb * 32 + l31, S - 1) * ld_qk + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  // staging: K tile = 64 rows x 8 chunks, V^T tile = 64 rows x 8 chunks; 2 chunks of each per thread.
  // Named scalars + macros on purpose: arrays captured by lambdas ended up in scratch memory here.
  const int srow0 = tid >> 3, srow1 = 32 + (tid >> 3), sch = tid & 7;
  const bf16_t* vsrc0 = vb_ + (int64_t)srow0 * S_pad + sch * 8;
  const bf16_t* vsrc1 = vb_ + (int64_t)srow1 * S_pad + sch * 8;
  const uint32_t soff0 = kt_off(srow0, sch), soff1 = kt_off(srow1, sch);
  uint4 rk0, rk1, rv0, rv1;
#define U2_FLASH_GLOAD(t_)                                                                                       \
  do {                                                                                                           \
    const int kv0_ = (t_) * 64;                                                                                  \
    rk0 = *reinterpret_cast<const uint4*>(kb_ + (int64_t)min(kv0_ + srow0, S - 1) * ld_qk + sch * 8);           \
    rk1 = *reinterpret_cast<const uint4*>(kb_ + (int64_t)min(kv0_ + srow1, S - 1) * ld_qk + sch * 8);           \
    rv0 = *reinterpret_cast<const uint4*>(vsrc0 + kv0_);                                                         \
    rv1 = *reinterpret_cast<const uint4*>(vsrc1 + kv0_);                                                         \
  } while (0)
#define U2_FLASH_LSTORE(st_)                                                  \
  do {                                                                        \
    *reinterpret_cast<uint4*>(&lds[st_][soff0]) = rk0;                        \
    *reinterpret_cast<uint4*>(&lds[st_][soff1]) = rk1;                        \
    *reinterpret_cast<uint4*>(&lds[st_][8192 + soff0]) = rv0;                 \
    *reinterpret_cast<uint4*>(&lds[st_][8192 + soff1]) = rv1;                 \
  } while (0)

  f32x16 oacc[QB][2];
  float m_run[QB], l_run[QB];  // m_run in scaled (log2) units
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][nb][r] = 0.f;
  }

  const int ntile = (S + 63) >> 6;
  U2_FLASH_GLOAD(0);
  U2_FLASH_LSTORE(0);
  __syncthreads();
  // Pin the Q fragments as "arrived" here.  Without this hipcc sinks their loads into the loop pre-header and its
  // waitcnt pass then drains vmcnt to 0 in front of the first QK^T MFMAs of EVERY iteration -- i.e. it waits for
  // the K/V prefetch issued a few instructions earlier and exposes the full HBM latency per tile.
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qb][ks]));
  if constexpr (TIMED) tprev = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < ntile; ++t) {
    const int st = t & 1;
    U2_STAMP(6)  // loop overhead + wait at the barrier
    if (t + 1 < ntile) U2_FLASH_GLOAD(t + 1);
    U2_STAMP(0)  // global load issue
    if (wave_active) {
      const char* sK = lds[st];
      const char* sV = lds[st] + 8192;
      // ---- S^T = K Q^T for the two 32-key blocks; every K fragment feeds QB MFMAs
      f32x16 sc[QB][2];
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[qb][kbk][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + kt_off(kbk * 32 + l31, ks * 2 + hi));
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            sc[qb][kbk] = mfma32(kf, qf[qb][ks], sc[qb][kbk]);
        }
      }
      if constexpr (TIMED) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) asm volatile("" : "+v"(sc[qb][0]), "+v"(sc[qb][1]));  // S^T complete
      }
      U2_STAMP(1)  // K fragment reads + QK^T MFMAs
      // lane owns keys kv = t*64 + kbk*32 + (r&3) + 8*(r>>2) + 4*hi ; only the last tile can be partial
      if (t == ntile - 1 && (S & 63)) {
        const int kvb = t * 64 + 4 * hi;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (kvb + kbk * 32 + (r & 3) + 8 * (r >> 2) >= S) sc[qb][kbk][r] = -INFINITY;
      }
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        // four independent chains (a serial chain of 32 dependent ops is latency-, not throughput-bound)
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx[r & 3] = fmaxf(mx[r & 3], sc[qb][kbk][r]);
        float mt = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * scale_log2e;  // scale > 0: max commutes with it
        if (__any(mt > m_run[qb] + FLASH_RESCALE_THR)) {  // wave-uniform; always taken on the first tile
          const float m_new = fmaxf(m_run[qb], mt);
          const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
          m_run[qb] = m_new;
          l_run[qb] *= alpha;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][nb][r] *= alpha;
        }
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[qb][kbk][r], scale_log2e, -m_run[qb]));
            sc[qb][kbk][r] = p;
            ps[r & 3] += p;
          }
        l_run[qb] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
      }
      if constexpr (TIMED) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) asm volatile("" : "+v"(sc[qb][0]), "+v"(sc[qb][1]));
      }
      U2_STAMP(2)  // softmax
      // ---- O^T += V^T P^T ; k-slots jj of step (kbk, ks2) carry keys kbk*32 + 16*ks2 + 8*(jj>>2) + 4*hi + (jj&3),
      //      which is exactly 16-byte chunk (kbk*2 + ks2)*2 + hi of the permuted V^T row
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          union { bf16x8 v; uint32_t u[4]; } pf[QB];
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              pf[qb].u[j] = pack2_bf16(sc[qb][kbk][ks2 * 8 + 2 * j], sc[qb][kbk][ks2 * 8 + 2 * j + 1]);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + kt_off(nb * 32 + l31, (kbk * 2 + ks2) * 2 + hi));
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
              oacc[qb][nb] = mfma32(vf, pf[qb].v, oacc[qb][nb]);
          }
        }
    }
    if constexpr (TIMED) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) asm volatile("" : "+v"(oacc[qb][0]), "+v"(oacc[qb][1]));
    }
    U2_STAMP(3)  // V^T fragment reads + PV MFMAs
    if (t + 1 < ntile) U2_FLASH_LSTORE((t + 1) & 1);
    U2_STAMP(4)  // wait for the prefetched tile + LDS stores
    __syncthreads();
  }
  if constexpr (TIMED) {
    if ((tid & 63) == 0 && g_flash_dbg) {
      unsigned long long* o = g_flash_dbg + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 8 + (QB == 2 ? 0 : 0);
#pragma unroll
      for (int i = 0; i < 7; ++i) o[i] += ts[i];
      o[7] += (unsigned long long)ntile;
    }
  }
#undef U2_STAMP
#undef U2_FLASH_GLOAD
#undef U2_FLASH_LSTORE
  if (!wave_active) return;
  // ---- the extra key (one per batch): scores on the VALU from the Q fragments the lane already holds
  if (a.n_extra) {
    const bf16_t* kxp = a.kx + (int64_t)b * a.x_bs + h * 64 + hi * 8;
    const bf16_t* vxp = a.vx + (int64_t)b * a.x_bs + h * 64 + 4 * hi;
    uint4 kc[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kc[ks] = *reinterpret_cast<const uint4*>(kxp + ks * 16);
    uint2 vc[2][4];  // V[d], d = nb*32 + 8g + 4hi + e : the 32 rows of O^T this lane owns
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) vc[nb][g] = *reinterpret_cast<const uint2*>(vxp + nb * 32 + 8 * g);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float part = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        union { bf16x8 v; uint32_t u[4]; } qq;
        qq.v = qf[qb][ks];
        const uint32_t kw[4] = {kc[ks].x, kc[ks].y, kc[ks].z, kc[ks].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          part = __builtin_fmaf(bf16lo(qq.u[j]), bf16lo(kw[j]), part);
          part = __builtin_fmaf(bf16hi(qq.u[j]), bf16hi(kw[j]), part);
        }
      }
      const float mt = (part + __shfl_xor(part, 32, 64)) * scale_log2e;
      if (__any(mt > m_run[qb] + FLASH_RESCALE_THR)) {
        const float m_new = fmaxf(m_run[qb], mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        m_run[qb] = m_new;
        l_run[qb] *= alpha;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[qb][nb][r] *= alpha;
      }
      const float p = __builtin_amdgcn_exp2f(mt - m_run[qb]);
      l_run[qb] += 0.5f * p;  // both half-waves hold the same key: the xor-32 sum below counts it once
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          oacc[qb][nb][4 * g + 0] = __builtin_fmaf(p, bf16lo(vc[nb][g].x), oacc[qb][nb][4 * g + 0]);
          oacc[qb][nb][4 * g + 1] = __builtin_fmaf(p, bf16hi(vc[nb][g].x), oacc[qb][nb][4 * g + 1]);
          oacc[qb][nb][4 * g + 2] = __builtin_fmaf(p, bf16lo(vc[nb][g].y), oacc[qb][nb][4 * g + 2]);
          oacc[qb][nb][4 * g + 3] = __builtin_fmaf(p, bf16hi(vc[nb][g].y), oacc[qb][nb][4 * g + 3]);
        }
    }
  }
  // ---- epilogue: O^T[d][q] / l ; lane: q = lane&31, d = nb*32 + (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = 1.f / l_tot;
    const int qrow = wrow0 + qb * 32 + l31;
    if (a.lse && hi == 0 && qrow < S) a.lse[((int64_t)b * a.H + h) * a.lse_ld + qrow] = m_run[qb] + __builtin_log2f(l_tot);
    if (qrow < S) {
      bf16_t* op = a.out + (int64_t)b * a.out_bs + (int64_t)qrow * a.ld_out + h * 64;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = nb * 32 + 8 * g + 4 * hi;
          *reinterpret_cast<uint2*>(op + d0) =
              uint2{pack2_bf16(oacc[qb][nb][4 * g] * inv, oacc[qb][nb][4 * g + 1] * inv),
                    pack2_bf16(oacc[qb][nb][4 * g + 2] * inv, oacc[qb][nb][4 * g + 3] * inv)};
        }
    }
  }
}

// The extra QUERY row of head (b, h) against all S + 1 keys, on the VALU (256 threads): scores key-parallel,
// softmax through LDS, P V with 4 lanes per output column walking the permuted V^T rows.
template <int NT = 256>  // threads of the workgroup (256 or 512)
__device__ __forceinline__ void flash_extra_row(const FlashArgs& a, char* lds_raw, const int b, const int h, const int tid) {
  constexpr int NW = NT / 64;
  float* sbuf = reinterpret_cast<float*>(lds_raw);            // [S_pad] scaled scores -> probabilities
  float* red = reinterpret_cast<float*>(lds_raw) + a.S_pad;   // [2 * NW] block reductions
  const int S = a.S, S_pad = a.S_pad;
  const int lane = tid & 63, wv = tid >> 6;
  const float c = a.q_prescaled ? 1.0f : a.scale_log2e;
  uint4 qv[8];
  {
    const bf16_t* qp = a.qx + (int64_t)b * a.x_bs + h * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = *reinterpret_cast<const uint4*>(qp + i * 8);
  }
  auto dot_row = [&](const bf16_t* kp) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint4 kk = *reinterpret_cast<const uint4*>(kp + i * 8);
      acc = __builtin_fmaf(bf16lo(qv[i].x), bf16lo(kk.x), acc); acc = __builtin_fmaf(bf16hi(qv[i].x), bf16hi(kk.x), acc);
      acc = __builtin_fmaf(bf16lo(qv[i].y), bf16lo(kk.y), acc); acc = __builtin_fmaf(bf16hi(qv[i].y), bf16hi(kk.y), acc);
      acc = __builtin_fmaf(bf16lo(qv[i].z), bf16lo(kk.z), acc); acc = __builtin_fmaf(bf16hi(qv[i].z), bf16hi(kk.z), acc);
      acc = __builtin_fmaf(bf16lo(qv[i].w), bf16lo(kk.w), acc); acc = __builtin_fmaf(bf16hi(qv[i].w), bf16hi(kk.w), acc);
    }
    return acc;
  };
  const bf16_t* kb_ = a.k + (int64_t)b * a.q_bs + h * 64;
  const float sx = dot_row(a.kx + (int64_t)b * a.x_bs + h * 64) * c;  // the extra key (every thread, redundantly)
  float m = sx;
  for (int j = tid; j < S_pad; j += NT) {
    float s = -INFINITY;  // padding columns of V^T are zero: probability 0 there
    if (j < S) s = dot_row(kb_ + (int64_t)j * a.ld_qk) * c;
    sbuf[j] = s;
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
  float sum = 0.f;
  for (int j = tid; j < S_pad; j += NT) {
    const float p = __builtin_amdgcn_exp2f(sbuf[j] - m);
    sbuf[j] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[NW + wv] = sum;
  __syncthreads();
  const float px = __builtin_amdgcn_exp2f(sx - m);
  float l_tot = px;
#pragma unroll
  for (int w = 0; w < NW; ++w) l_tot += red[NW + w];
  if (a.lse && tid == 0) a.lse[((int64_t)b * a.H + h) * a.lse_ld + S] = m + __builtin_log2f(l_tot);
  if (tid >= 256) return;  // the P V walk below uses 4 lanes per output column = 256 threads
  // O[d] = sum_j p_j V[j][d]: thread (d = tid >> 2, part = tid & 3) walks 16-byte pieces of row d of V^T; inside a
  // group of 16 keys the stored order is [0-3, 8-11, 4-7, 12-15], i.e. piece (G, hh) holds keys 16G + 4hh + {0..3}
  // and 16G + 8 + 4hh + {0..3}.
  const int d = tid >> 2, part = tid & 3;
  const bf16_t* vrow = a.vt + (((int64_t)b * a.H + h) * 64 + d) * S_pad;
  float o = 0.f;
  for (int pc = part; pc < S_pad / 8; pc += 4) {
    const uint4 vv = *reinterpret_cast<const uint4*>(vrow + pc * 8);
    const int G = pc >> 1, hh = pc & 1;
    const float4 p0 = *reinterpret_cast<const float4*>(sbuf + 16 * G + 4 * hh);
    const float4 p1 = *reinterpret_cast<const float4*>(sbuf + 16 * G + 8 + 4 * hh);
    o = __builtin_fmaf(p0.x, bf16lo(vv.x), o); o = __builtin_fmaf(p0.y, bf16hi(vv.x), o);
    o = __builtin_fmaf(p0.z, bf16lo(vv.y), o); o = __builtin_fmaf(p0.w, bf16hi(vv.y), o);
    o = __builtin_fmaf(p1.x, bf16lo(vv.z), o); o = __builtin_fmaf(p1.y, bf16hi(vv.z), o);
    o = __builtin_fmaf(p1.z, bf16lo(vv.w), o); o = __builtin_fmaf(p1.w, bf16hi(vv.w), o);
  }
  o += __shfl_xor(o, 1, 64);
  o += __shfl_xor(o, 2, 64);
  if (part == 0) {
    o = __builtin_fmaf(px, bf16_to_f32(a.vx[(int64_t)b * a.x_bs + h * 64 + d]), o);
    a.outx[(int64_t)b * a.ox_bs + h * 64 + d] = f32_to_bf16(o / l_tot);
  }
}

// Plain form (mode 1): 128-row units, 4 waves x 32 query rows, three workgroups per CU.  Used below S = 512, where the
// double pipeline's 256-row units leave most of the machine idle.
__global__ __launch_bounds__(256, 3) void flash_d64_kernel(const FlashArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[2][16384];  // [stage][K tile 8 KB | V^T tile 8 KB]
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.n_main) {  // extra query rows: one workgroup per (batch, head)
    const int e = blockIdx.x - a.n_main;
    flash_extra_row(a, &lds[0][0], e / a.H, e % a.H, tid);
    return;
  }
  // XCD-aware order: workgroup w runs on XCD w % 8 (observed dispatch rule); give every XCD a contiguous range of
  // logical ids so that the units of one (batch, head) share that XCD's L2 copy of K and V^T.
  int bid;
  {
    const int nwg = a.n_main, qn = nwg >> 3, rn = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  const int nqt = (a.S + 127) >> 7;
  const int hh = bid / nqt;
  flash_pass<1, false>(a, lds, hh / a.H, hh % a.H, (bid % nqt) * 128, tid);
}

// ----------------------------------------------------------------------------------------------------------------
// Double-pipeline form (modes 7 / 8; the round-1 loop, "mode 5", was removed in round 4: history da95d2b).  The
// measurements behind it (profiles/r01_flash_study.log, tools/ubench/valu_rate.hip): at head dim 64 a 32 x 64 score block
// costs a SIMD 16 MFMAs (512 matrix-pipe cycles) against hundreds of cycles of softmax VALU issue (v_exp_f32 7.5,
// 3-operand VALU ~4, 2-operand ~2 cycles per wave64 instruction); the plain form above runs them one after the other and
// relies on a second, unrelated wave of the SIMD to fill the gaps, which it does only by chance (matrix pipe busy 28 %).
// Here ONE wave owns two 32-row query blocks and alternates phases over 32-key half tiles in which the MFMAs of one block
// are interleaved, slot by slot, with the softmax VALU of the other:
//     B(u):   S0(u+1) = K(u+1) Q0^T ; O0 += V(u) P0(u)      ||   P1(u)   = softmax step on S1(u)
//     A(u+1): S1(u+1) = K(u+1) Q1^T ; O1 += V(u) P1(u)      ||   P0(u+1) = softmax step on S0(u+1)
// (an in-order wave keeps issuing independent VALU while its own MFMA occupies the matrix pipe; the second wave of
// the SIMD runs the same mix).  K / V^T tiles (64 keys) arrive by LDS-DMA (buffer_load ... lds) into a ring of 4 slots,
// 3 tiles ahead, one s_barrier per tile.  4 waves x 64 rows = 256-row units, two workgroups per CU.
// The whole KV loop is ONE generated asm block (flash_dp2_asm.inc, written by tools/gen_flash_dp2_asm.py; register map
// and schedule are documented there): hipcc could not be made to keep the slot order AND the accumulators in place at
// 256 VGPRs -- through asm operands it reordered the slots, rotated the accumulators through extra tuples, copied them
// around pinned registers, or spilled (~290 VGPRs, one wave per SIMD).  With every loop register fixed by hand the
// kernel runs two waves per SIMD.
constexpr int FDP_SLOTS = 4;

struct FdpBlock {       // final state of one 32-row query block of a wave
  f32x16 oacc[2];       // O^T accumulators
  float m_run, l_run;
};

// the extra key (one per batch) and the output of one block
__device__ __forceinline__ void fdp_finish(const FlashArgs& a, FdpBlock& x, const bf16x8 (&qfx)[4], const int b, const int h,
                                           const int qrow, const int hi, const bool q_prescaled = false,
                                           const bool extra_done = false) {
  const float scale_log2e = q_prescaled ? 1.0f : a.scale_log2e;  // mode 7 carries scale * log2 e in its Q fragments
  if (a.n_extra && !extra_done) {  // (the round-4 loop opens its running sums with the extra key: extra_done)
    const bf16_t* kxp = a.kx + (int64_t)b * a.x_bs + h * 64 + hi * 8;
    const bf16_t* vxp = a.vx + (int64_t)b * a.x_bs + h * 64 + 4 * hi;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 kc = *reinterpret_cast<const uint4*>(kxp + ks * 16);
      union { bf16x8 v; uint32_t u[4]; } qq;
      qq.v = qfx[ks];
      const uint32_t kw[4] = {kc.x, kc.y, kc.z, kc.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        part = __builtin_fmaf(bf16lo(qq.u[j]), bf16lo(kw[j]), part);
        part = __builtin_fmaf(bf16hi(qq.u[j]), bf16hi(kw[j]), part);
      }
    }
    const float mt = (part + __shfl_xor(part, 32, 64)) * scale_log2e;
    if (__any(mt > x.m_run + FLASH_RESCALE_THR)) {
      const float m_new = fmaxf(x.m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f(x.m_run - m_new);
      x.m_run = m_new;
      x.l_run *= alpha;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) x.oacc[nb][r] *= alpha;
    }
    const float p = __builtin_amdgcn_exp2f(mt - x.m_run);
    x.l_run += 0.5f * p;  // both half-waves hold the same key: the xor-32 sum below counts it once
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2 vc = *reinterpret_cast<const uint2*>(vxp + nb * 32 + 8 * g);
        x.oacc[nb][4 * g + 0] = __builtin_fmaf(p, bf16lo(vc.x), x.oacc[nb][4 * g + 0]);
        x.oacc[nb][4 * g + 1] = __builtin_fmaf(p, bf16hi(vc.x), x.oacc[nb][4 * g + 1]);
        x.oacc[nb][4 * g + 2] = __builtin_fmaf(p, bf16lo(vc.y), x.oacc[nb][4 * g + 2]);
        x.oacc[nb][4 * g + 3] = __builtin_fmaf(p, bf16hi(vc.y), x.oacc[nb][4 * g + 3]);
      }
  }
  const float l_tot = x.l_run + __shfl_xor(x.l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (a.lse && hi == 0 && qrow < a.S) a.lse[((int64_t)b * a.H + h) * a.lse_ld + qrow] = x.m_run + __builtin_log2f(l_tot);
  bf16_t* op = a.out + (int64_t)b * a.out_bs + (int64_t)qrow * a.ld_out + h * 64;
  if (a.wide_out) {
    // 16-byte stores (guide T21): the lane owns d = 8 g + 4 hi + [0, 4) of its row, its partner lane (+32) the other half
    // of the same 8; v_permlane32_swap on the packed words of two neighbouring groups leaves 8 consecutive d per lane.
    // All 64 lanes take part in the swaps; rows past S are only not stored.
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int g0 = 2 * gp, g1 = 2 * gp + 1;
        const uint32_t a0 = pack2_bf16(x.oacc[nb][4 * g0] * inv, x.oacc[nb][4 * g0 + 1] * inv);
        const uint32_t a1 = pack2_bf16(x.oacc[nb][4 * g0 + 2] * inv, x.oacc[nb][4 * g0 + 3] * inv);
        const uint32_t b0 = pack2_bf16(x.oacc[nb][4 * g1] * inv, x.oacc[nb][4 * g1 + 1] * inv);
        const uint32_t b1 = pack2_bf16(x.oacc[nb][4 * g1 + 2] * inv, x.oacc[nb][4 * g1 + 3] * inv);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        if (qrow < a.S) *reinterpret_cast<uint4*>(op + nb * 32 + 8 * (g0 + hi)) = uint4{r0[0], r1[0], r0[1], r1[1]};
      }
  } else if (qrow < a.S) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = nb * 32 + 8 * g + 4 * hi;
        *reinterpret_cast<uint2*>(op + d0) =
            uint2{pack2_bf16(x.oacc[nb][4 * g] * inv, x.oacc[nb][4 * g + 1] * inv),
                  pack2_bf16(x.oacc[nb][4 * g + 2] * inv, x.oacc[nb][4 * g + 3] * inv)};
      }
  }
}

typedef int i32x4_t __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------------------------------
// Round 4 (mode 7): same units, ring and DMA as the round-1 loop; the KV loop is flash_dp2_asm.inc
// (tools/gen_flash_dp2_asm.py, where the schedule and the reasons are written down).  In short, the round-1 loop
// was VALU-issue bound at 59 SIMD cycles per MFMA slot; this one takes the scale FMAs, the row
// max, the address adds and the trans-use nops out of the slot: Q fragments carry scale * log2 e (one bf16
// rounding of q * c instead of q: the same relative error, a different rounding point than the reference's), the
// running max is subtracted by the matrix pipe (C operand of the first Q K^T MFMA = a tuple holding -m), m is only
// kept within 2^64 of the true running max (an out-of-line path restores that when a row-sum piece says so), and
// the tile loop is unrolled over the ring so fragment reads are lane base + immediate.
#if U2_ELEM_IS_F16
#include "build_f16/flash_dp2_asm.inc"  // derived at build time: tools/asm_elem_f16.py
#else
#include "flash_dp2_asm.inc"
#endif


// Extra query row of head (b, h) for the double pipeline of round 4 (NT threads).  The first form (flash_extra_row above)
// took ~40 us per row -- 64 dependent 16-byte loads per thread -- and sat on the kernel's tail.  Scores: 8 lanes per key
// (16-byte K chunks, fully coalesced rows), 16 keys per thread in flight; softmax through LDS; P V: 16 lanes per V^T
// row, 16 independent 16-byte loads per thread in flight.
__device__ __forceinline__ float dot2_bf16(const uint32_t a, const uint32_t b, const float c) {  // v_dot2c_f32_bf16 / _f16
  return dot2_elem(a, b, c);
}
template <int NT>
__device__ __forceinline__ void flash_extra_row2(const FlashArgs& a, float* sbuf, const int b, const int h, const int tid,
                                                 unsigned long long* tl = nullptr) {
  constexpr int NW = NT / 64;
  float* red = sbuf + a.S_pad;                                    // [2 NW] block reductions
  bf16_t* pb = reinterpret_cast<bf16_t*>(sbuf + a.S_pad + 64);    // [S_pad] probabilities, bf16, in V^T's key order
  const int lane = tid & 63, wv = tid >> 6;
  const int S = a.S, S_pad = a.S_pad;
  const float c = a.q_prescaled ? 1.0f : a.scale_log2e;
  const int kk = lane >> 3, ch = lane & 7;
  // the products run on v_dot2c_f32_bf16 (two bf16 x bf16 terms per instruction, fp32 sum, no unpacking): in the shadow of
  // the main waves, which are VALU-issue bound, this routine is paid for in issue slots, not in bytes
  const uint4 qw = *reinterpret_cast<const uint4*>(a.qx + (int64_t)b * a.x_bs + h * 64 + ch * 8);
  auto dot8 = [&](const uint4 u) {
    float acc = dot2_bf16(qw.x, u.x, 0.f);
    acc = dot2_bf16(qw.y, u.y, acc); acc = dot2_bf16(qw.z, u.z, acc); acc = dot2_bf16(qw.w, u.w, acc);
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
    return acc * c;
  };
  const bf16_t* kb_ = a.k + (int64_t)b * a.q_bs + h * 64 + ch * 8;
  const float sx = dot8(*reinterpret_cast<const uint4*>(a.kx + (int64_t)b * a.x_bs + h * 64 + ch * 8));  // the extra key
  float m = sx;
  // keys per thread in flight (8, 16 and 32 measure the same ~30 us beside the main waves: scores 18.6 + softmax 2 + P V 11.4 us
  // by s_memrealtime; moved into the V^T launch in front of the flash launch the rows took 24 us there against 12 us for the
  // transpose alone and bought the flash launch 2 us: removed again, profiles/r04_flash_tail_split_trial.log)
  constexpr int KU = 8;
  for (int j0 = wv * 8 + kk; j0 < S_pad; j0 += KU * 8 * NW) {
    uint4 u[KU];
#pragma unroll
    for (int i = 0; i < KU; ++i) u[i] = *reinterpret_cast<const uint4*>(kb_ + (int64_t)min(j0 + 8 * NW * i, S - 1) * a.ld_qk);
#pragma unroll
    for (int i = 0; i < KU; ++i) {
      const int j = j0 + 8 * NW * i;
      float sc = dot8(u[i]);
      if (j >= S) sc = -INFINITY;  // padding columns of V^T are zero: probability 0 there
      if (j < S_pad) {
        if (ch == 0) sbuf[j] = sc;
        m = fmaxf(m, sc);
      }
    }
  }
  m = wave_max(m);
  if (tl && lane == 0) tl[4] = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
  float sum = 0.f;
  for (int j = tid; j < S_pad; j += NT) {
    const float p = __builtin_amdgcn_exp2f(sbuf[j] - m);
    // inside a group of 16 keys V^T stores [0-3, 8-11, 4-7, 12-15]: the same permutation for P
    const int r = j & 15;
    pb[(j & ~15) + ((r & 3) | ((r & 4) << 1) | ((r & 8) >> 1))] = f32_to_bf16(p);
    sum += p;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[NW + wv] = sum;
  __syncthreads();
  if (tl && lane == 0) tl[5] = __builtin_amdgcn_s_memrealtime();
  const float px = __builtin_amdgcn_exp2f(sx - m);
  float l_tot = px;
#pragma unroll
  for (int w = 0; w < NW; ++w) l_tot += red[NW + w];
  if (a.lse && tid == 0) a.lse[((int64_t)b * a.H + h) * a.lse_ld + S] = m + __builtin_log2f(l_tot);
  // O[d] = sum_j p_j V[j][d]: 16 lanes per row d of V^T, 16-byte pieces against the 16 bytes of P at the same offset
  const int part = tid & 15;
  const int npc = S_pad >> 3;
  for (int d = tid >> 4; d < 64; d += NT / 16) {
    const bf16_t* vrow = a.vt + (((int64_t)b * a.H + h) * 64 + d) * S_pad;
    float o0 = 0.f, o1 = 0.f;
    constexpr int PU = 8;   // V^T pieces per thread in flight
    for (int p0 = part; p0 < npc; p0 += 16 * PU) {
      uint4 vv[PU];
#pragma unroll
      for (int i = 0; i < PU; ++i) {
        const int pc = p0 + 16 * i;
        vv[i] = pc < npc ? *reinterpret_cast<const uint4*>(vrow + pc * 8) : uint4{0, 0, 0, 0};
      }
#pragma unroll
      for (int i = 0; i < PU; ++i) {
        const uint4 pw = *reinterpret_cast<const uint4*>(pb + min(p0 + 16 * i, npc - 1) * 8);
        o0 = dot2_bf16(pw.x, vv[i].x, o0); o1 = dot2_bf16(pw.y, vv[i].y, o1);
        o0 = dot2_bf16(pw.z, vv[i].z, o0); o1 = dot2_bf16(pw.w, vv[i].w, o1);
      }
    }
    float o = o0 + o1;
    o += __shfl_xor(o, 1, 64); o += __shfl_xor(o, 2, 64); o += __shfl_xor(o, 4, 64); o += __shfl_xor(o, 8, 64);
    if (part == 0) {
      o = __builtin_fmaf(px, bf16_to_f32(a.vx[(int64_t)b * a.x_bs + h * 64 + d]), o);
      a.outx[(int64_t)b * a.ox_bs + h * 64 + d] = f32_to_bf16(o / l_tot);
    }
  }
}

// QMODE: 0 = q as the reference has it, every score multiplied by scale * log2 e in fp32 (flash_dp2_asm.inc, "_X" text);
//        1 = the caller's q / qx already carry scale * log2 e (the ViT's q|k|v product scales its q columns in the
//            epilogue, from the fp32 accumulator: one rounding, as for the unscaled q): two VALU per slot less.
// Grid = [n_main units of 256 query rows | one extra-row workgroup per (batch, head)].  768 units on 512 workgroup slots
// are 1.5 rounds; two ways of cutting the last half round in two key ranges (a persistent form, every workgroup 1.5 units
// in lock step; a tail form, the last 256 units as 512 half units with a ticket and an fp32 slab per pair) were built and
// measured in round 4 and removed again: the extra-row workgroups need ~30 us of a slot whatever they execute and only
// the half-empty second round has slots to spare, and the first round's exits are spread over 16 us, which the plain
// form absorbs for free (profiles/r04_flash_tail_split_trial.log; the commit before this form).
template <bool TIMED, int QMODE>
__global__ __launch_bounds__(256, 2) void flash_dp2_kernel(const FlashArgs a) {
  __shared__ __attribute__((aligned(1024))) char lds[FDP_SLOTS][16384];  // [slot][K tile 8 KB | V^T tile 8 KB]
  const int tid = threadIdx.x;
  // TIMED: wall-clock stamps (s_memrealtime, 100 MHz) of the workgroup's sections in a second region of the debug buffer
  unsigned long long* tl = nullptr;
  if constexpr (TIMED) {
    tl = g_flash_dbg + 65536 + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 8;
    if ((tid & 63) == 0) tl[0] = __builtin_amdgcn_s_memrealtime();
  }
  // XCD-aware order inside each region of the grid: workgroup w runs on XCD w % 8 (observed dispatch rule); every XCD
  // gets a contiguous range of logical ids so that the units of one (batch, head) share that XCD's L2 copy of K and V^T
  auto xcd_order = [](const int w, const int nwg) {
    const int qn = nwg >> 3, rn = nwg & 7, xcd = w & 7, idx = w >> 3;
    return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  };
  const int ntile = (a.S + 63) >> 6;
  int unit;
  {
    const int w = blockIdx.x;
    if (w < a.n_main) unit = xcd_order(w, a.n_main);
    else {
      // these four waves share their SIMDs with main waves that are VALU-issue bound, and as the younger waves they would
      // get the leftover issue slots (30 us for ~10 us of work): static priority, their demand is small
      __builtin_amdgcn_s_setprio(3);
      const int e = w - a.n_main;
      flash_extra_row2<256>(a, reinterpret_cast<float*>(&lds[0][0]), e / a.H, e % a.H, tid, TIMED ? tl : nullptr);
      if constexpr (TIMED) if ((tid & 63) == 0) tl[3] = __builtin_amdgcn_s_memrealtime();
      return;
    }
  }
  const int nqt = (a.S + 255) >> 8;
  const int hh = unit / nqt, b = hh / a.H, h = hh % a.H, row0 = (unit % nqt) * 256;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int S = a.S, S_pad = a.S_pad;
  const int64_t ld_qk = a.ld_qk;
  const bf16_t* qb_ = a.q + (int64_t)b * a.q_bs + h * 64;
  const bf16_t* kb_ = a.k + (int64_t)b * a.q_bs + h * 64;
  const bf16_t* vb_ = a.vt + ((int64_t)b * a.H + h) * 64 * S_pad;
  const int wrow0 = row0 + wv * 64;

  // the lane's two query rows (rows past S: the last one; their results are not stored): the block loads the fragments itself,
  // 4 x 16 bytes at 32-byte steps from qa0 / qa1
  const bf16_t* qa0 = qb_ + (int64_t)min(wrow0 + l31, S - 1) * ld_qk + hi * 8;
  const bf16_t* qa1 = qb_ + (int64_t)min(wrow0 + 32 + l31, S - 1) * ld_qk + hi * 8;
  // DMA pieces of this wave: MUBUF descriptors built by hand -- K rows past S read as zero
  const int prow = wv * 16 + (lane >> 3);
  const int pch0 = (lane & 7) ^ ((prow >> 1) & 7), pch1 = pch0 ^ 4;
  const int ko0 = (prow * (int)ld_qk + pch0 * 8) * 2, ko1 = ((prow + 8) * (int)ld_qk + pch1 * 8) * 2;
  const int vo0 = (prow * S_pad + pch0 * 8) * 2, vo1 = ((prow + 8) * S_pad + pch1 * 8) * 2;
  const int k_tile_bytes = 64 * (int)ld_qk * 2;
  const uint64_t kaddr = (uint64_t)(uintptr_t)kb_, vaddr = (uint64_t)(uintptr_t)vb_;
  i32x4_t rsk, rsv;
  rsk[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)kaddr);
  rsk[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(kaddr >> 32));
  rsk[2] = (int)((((int64_t)S - 1) * ld_qk + 64) * 2);
  rsk[3] = 0x00020000;
  rsv[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)vaddr);
  rsv[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(vaddr >> 32));
  rsv[2] = 64 * S_pad * 2;
  rsv[3] = 0x00020000;
  const uint32_t lds_u32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&lds[0][0];
  const uint32_t dma_base = lds_u32 + wv * 2048;
  const uint32_t ab0 = kt_off(l31, hi);
  const uint32_t dump = lds_u32 + wv * 16384 + lane * 16;
  const int hi4 = 4 * hi;
  const float scale_log2e = a.scale_log2e, rscale = 1.0f / a.scale_log2e;
  // the extra key (one per batch): the block loads the lane's share of it (k: 4 x 16 bytes at 32-byte steps from kxa, v: 8 x 8
  // bytes at 16-byte steps from vxa) and opens its running sums with it; without one the pointers only have to be readable
  const bf16_t* kxa = a.n_extra ? a.kx + (int64_t)b * a.x_bs + h * 64 + hi * 8 : a.q;
  const bf16_t* vxa = a.n_extra ? a.vx + (int64_t)b * a.x_bs + h * 64 + 4 * hi : a.q;
  const int xflag = a.n_extra;
  float mr0, lr0, mr1, lr1;
  int lane2;  // the lane id as the block returns it: keeps the epilogue's per-lane values from living across the block
  unsigned long long* dbg = g_flash_dbg + ((size_t)blockIdx.x * 4 + wv) * 8;  // TIMED: 5 section times, [7] = tiles
#define FDP2_OPERANDS                                                                                                  \
               : [mr0] "=&v"(mr0), [lr0] "=&v"(lr0), [mr1] "=&v"(mr1), [lr1] "=&v"(lr1), [lid] "=&v"(lane2)             \
               : [qa0] "v"(qa0), [qa1] "v"(qa1),                                                                        \
                 [ab0] "v"(ab0), [ko0] "v"(ko0), [ko1] "v"(ko1), [vo0] "v"(vo0), [vo1] "v"(vo1), [hi4] "v"(hi4),        \
                 [dump] "v"(dump), [rsk] "s"(rsk), [rsv] "s"(rsv), [lds] "s"(lds_u32), [dma_base] "s"(dma_base),        \
                 [ktile] "s"(k_tile_bytes), [seq] "s"(S), [ntile] "s"(ntile), [dbg] "v"(dbg),                           \
                 [scale] "s"(scale_log2e), [rscale] "s"(rscale), [kxa] "v"(kxa), [vxa] "v"(vxa), [xflag] "s"(xflag)
#define FDP2_RUN(SFX_) asm volatile(FLASH_DP2_ASM_TEXT##SFX_ FDP2_OPERANDS : FLASH_DP2_ASM_CLOBBERS##SFX_)
  if constexpr (TIMED) {
    if (lane == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
    if constexpr (QMODE == 0) FDP2_RUN(_X_TIMED); else FDP2_RUN(_TIMED);
    if (lane2 == 0) { dbg[7] = (unsigned long long)ntile; tl[2] = __builtin_amdgcn_s_memrealtime(); }
  } else {
    if constexpr (QMODE == 0) FDP2_RUN(_X); else FDP2_RUN();
  }
#undef FDP2_RUN
#undef FDP2_OPERANDS
  // the block left O^T in LDS: tuple T = 2 * block + nb, 16-byte quarter j at [wave][T * 4 + j][lane]
  const int hi2 = lane2 >> 5, l31b = lane2 & 31;
  const char* dp = &lds[0][0] + wv * 16384 + lane2 * 16;
#pragma unroll 1
  for (int blk = 0; blk < 2; ++blk) {  // one block at a time: 32 accumulator values live, not 64
    FdpBlock x;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 u = *reinterpret_cast<const float4*>(dp + ((2 * blk + nb) * 4 + j) * 1024);
        x.oacc[nb][4 * j] = u.x; x.oacc[nb][4 * j + 1] = u.y; x.oacc[nb][4 * j + 2] = u.z; x.oacc[nb][4 * j + 3] = u.w;
      }
    x.m_run = blk ? mr1 : mr0;
    x.l_run = blk ? lr1 : lr0;
    const bf16x8 no_q[4] = {};   // (the extra key was folded in by the block)
    fdp_finish(a, x, no_q, b, h, wrow0 + 32 * blk + l31b, hi2, QMODE != 0, true);
  }
  if constexpr (TIMED) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane2 == 0) tl[3] = __builtin_amdgcn_s_memrealtime();
  }
}

static bool g_flash_timed = false;
int flash_set_debug_buffer(void* p) {  // >= grid * 4 * 8 uint64, zeroed by the caller; null detaches
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  g_flash_timed = q != nullptr;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_flash_dbg), &q, sizeof(q)) == hipSuccess ? U2_OK : U2_ERR_LAUNCH;
}

int flash_attention_d64(const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* out, int nb, int S, int H,
                        int64_t ld_qk, int64_t q_bs, int64_t ld_out, int64_t out_bs, int S_pad, float scale,
                        const bf16_t* qx, const bf16_t* kx, const bf16_t* vx, bf16_t* outx, int64_t x_bs, int64_t ox_bs,
                        int n_extra, float* lse, int64_t lse_ld, hipStream_t stream, int q_prescaled) {
  if (!q || !k || !vt || !out || nb <= 0 || S <= 0 || H <= 0 || n_extra < 0 || n_extra > 1) return U2_ERR_ARG;
  if (lse && lse_ld < S + n_extra) return U2_ERR_ARG;
  if ((S_pad & 63) || S_pad < ((S + 63) & ~63)) return U2_ERR_ARG;
  if ((ld_qk & 7) || (q_bs & 7) || (ld_out & 3) || (out_bs & 3)) return U2_ERR_ARG;
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) || ((uintptr_t)out & 7)) return U2_ERR_ARG;
  if (n_extra) {
    if (!qx || !kx || !vx || !outx || (x_bs & 7) || (((uintptr_t)qx | (uintptr_t)kx | (uintptr_t)vx) & 15)) return U2_ERR_ARG;
    if ((size_t)S_pad * 4 + 128 > 32768) return U2_ERR_ARG;  // probabilities of the extra row live in LDS
  }
  FlashArgs a;
  a.q = q; a.k = k; a.vt = vt; a.out = out; a.qx = qx; a.kx = kx; a.vx = vx; a.outx = outx;
  a.S = S; a.H = H; a.nb = nb; a.S_pad = S_pad; a.n_extra = n_extra;
  a.ld_qk = ld_qk; a.q_bs = q_bs; a.ld_out = ld_out; a.out_bs = out_bs; a.x_bs = x_bs; a.ox_bs = ox_bs;
  a.scale_log2e = scale * 1.44269504088896340736f;
  a.lse = lse; a.lse_ld = lse_ld;
  a.wide_out = !((uintptr_t)out & 15) && !(ld_out & 7) && !(out_bs & 7);
  const int64_t nbh = (int64_t)nb * H;
  int mode = opts().flash_mode;
  if (mode != 1 && mode != 7) mode = S >= 512 ? 7 : 1;  // measured: the double pipeline wins from S = 513 up
  q_prescaled = q_prescaled || opts().flash_q_prescaled;
  if (q_prescaled) mode = 7;  // the only form that takes pre-scaled queries (the ViT launches it at S = 2048)
  a.q_prescaled = q_prescaled;
  const int64_t blocks = mode != 1 ? nbh * ((S + 255) / 256) : nbh * ((S + 127) / 128);
  a.mode = mode;
  a.n_main = (int)blocks;
  const int64_t grid = blocks + (n_extra ? nbh : 0);
  if (grid > 0x7fffffff) return U2_ERR_ARG;
  ProfScope ps(PROF_FLASH, 4.0 * nbh * (double)(S + n_extra) * (S + n_extra) * 64, stream,
               4.0 * nbh * (double)(S + n_extra) * 64 * 2.0);  // q, k, v^T read + o written, once
  if (mode == 7) {
    const dim3 g((unsigned)grid), t(256);
#define U2_FDP2_Q(T_)                                                                                            \
  do {                                                                                                           \
    if (q_prescaled) hipLaunchKernelGGL((flash_dp2_kernel<T_, 1>), g, t, 0, stream, a);                         \
    else hipLaunchKernelGGL((flash_dp2_kernel<T_, 0>), g, t, 0, stream, a);                                     \
  } while (0)
    if (g_flash_timed) U2_FDP2_Q(true); else U2_FDP2_Q(false);
#undef U2_FDP2_Q
  } else {
    hipLaunchKernelGGL(flash_d64_kernel, dim3((unsigned)grid), dim3(256), 0, stream, a);
  }
  return launch_status();
}

}  // namespace u2
