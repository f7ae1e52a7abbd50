// Fused attention core of the tokenizer's own attention modules:
//   RelativeMultiheadAttention (rma.py:60-75): softmax(Q K^T / sqrt(d) + relative_bias[j - i + L - 1][h]) V
//   RotaryMultiheadAttention   (rope.py:82-86), MultiHeadCrossAttention / LinearAggregation (tta.py:55-61): no bias
// with S_q <= 512 query rows, S_kv up to a few thousand keys and a WIDE head (d = E / 8 = 256 or 512; 64 / 128 for the small
// test configurations).  The reference materialises the (S_q x S_kv) scores and probabilities; the round-1/2 path did the
// same in three launches (batched GEMM -> fp32 scores in HBM -> softmax_rows -> batched GEMM on 64 x 64 tiles).  Here one
// flash-style kernel streams K and V tiles through LDS and keeps scores, probabilities and the running softmax state in
// registers.
//
// Work decomposition.  A workgroup = 4 waves = 64 query rows of one (batch, head) x one contiguous range of keys
// ("split"); a wave owns 16 query rows and the WHOLE head dim:
//   S^T (16 keys x 16 q) = K Q^T : v_mfma_f32_16x16x32_bf16, A = K fragment (LDS), B = Q fragment (registers, loaded once);
//                                  a lane (q = lane & 15, g = lane >> 4) then owns keys 4g..4g+3 of each 16-key block.
//   O^T (16 d x 16 q) += V^T P^T  : A = V^T fragment gathered from the ROW-MAJOR V tile by two ds_read_b64_tr_b16 (4 keys
//                                  x 1 d each), B = the lane's own exp'd scores of the tile's two key blocks packed to bf16
//                                  -- the MFMA k-slot -> key map is a free permutation as long as A and B agree, so
//                                  k-slot (g, j) carries key 16 (j >> 2) + 4 g + (j & 3) and P never leaves its lane.
//   O^T accumulators: d / 16 blocks x 4 registers = 128 VGPRs at d = 512 (a 32-row MFMA shape would need 256).
// With only nb * H * ceil(S_q / 64) workgroups for a whole call (32 for the query side of the TTA at batch 1) the key range
// is cut into `ns` splits that run as separate workgroups (flash-decoding style); each leaves un-normalised fp32 partial
// outputs + (running max, sum) per row, and tok_attn_combine_kernel merges them in a fixed order (bit-repeatable).
//
// LDS: two stages of one K tile and one V tile of 32 keys x d (row-major, 32 KB each at d = 512: 128 KB, one workgroup per
// CU with the whole register file -- O^T alone is 128 registers; 64 KB and two workgroups per CU at d = 256), filled by
// LDS-DMA (global_load_lds_dwordx4, the swizzle applied on the per-lane SOURCE address because the DMA destination is
// lane-linear).  Tile t + 1 is in flight under the whole of step t (K issued before Q K^T, V before P V); one barrier per step.
//   K tile (fragment rows are K-contiguous: ds_read_b128 of chunk 4 ks + g of row 16 kb + (lane & 15)):
//       16-byte chunk index XOR (row & 15)  [(row & 7) for 128-byte rows, as gemm.hip]  -> 16 distinct slots per lane group.
//   V tile (transpose reads; a 16-lane group reads a [4 keys][16 d] block, lane a supplies row a >> 2 / 8-byte piece a & 3):
//       chunks rotated inside each 256-byte segment by 2 (key & 3) + 8 ((key >> 2) & 1) so that the 8 rows x 32 bytes a
//       half wave touches fall into 16 different 16-byte slots (the layout family of gemm.hip's K-major operands).
#include "kernels.h"
#if U2_ELEM_IS_F16
#include "build_f16/tokattn_pv_asm.inc"  // derived at build time: tools/asm_elem_f16.py
#else
#include "tokattn_pv_asm.inc"
#endif

namespace u2 {

struct TokAttnArgs {
  const bf16_t *q, *k, *v;
  bf16_t* out;
  int64_t ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs;
  int nb, H, Sq, Skv, nqb;
  int ns, tps;  // key splits, 32-key tiles per split
  float scale_log2e;
  const bf16_t* rel_bias;
  int max_len;
  float* opart;  // ns > 1: [ns][nb][Sq][H * DH] fp32, un-normalised
  float* ml;     // ns > 1: [ns][nb * H][Sq][2] = (running max in log2 units, row sum)
  int kv_group;  // grouped-query attention: query head h reads key / value head h / kv_group (1: plain multi-head)
  int causal;    // 1: key j is visible to query i iff j <= i + (Skv - Sq)   (decoder prefill)
  unsigned long long* dbg;  // diagnostics (tok_attention_set_debug_buffer): s_memtime sums per (workgroup, wave), 8 slots
};

constexpr float TOKATTN_RESCALE_THR = 8.0f;
constexpr int TOKATTN_BIAS_SLOTS = 640;

template <int SEG>
__device__ __forceinline__ int tv_rot(int k) {
  if constexpr (SEG == 16) return 2 * (k & 3) + 8 * ((k >> 2) & 1);
  else return 2 * ((k >> 1) & 1) + 4 * ((k >> 2) & 1);
}

typedef short ta_v4s_t __attribute__((ext_vector_type(4)));

template <int DH, bool TIMED = false>
__global__ __launch_bounds__(256, DH >= 512 ? 1 : 2) void tok_attn_kernel(const TokAttnArgs a) {
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define U2_STAMP(i_)                                            \
  if constexpr (TIMED) {                                        \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    ts[i_] += t_ - tprev;                                       \
    tprev = t_;                                                 \
  }
  constexpr int BK = DH <= 128 ? 64 : 32;  // keys per tile (narrow heads: twice the keys per barrier / DMA wait / softmax step)
  constexpr int NKB = BK / 16;        // 16-key blocks of S^T per tile
  constexpr int NPF = BK / 32;        // P fragments (32 keys = one MFMA k step of P V) per tile
  constexpr int CPR = DH / 8;         // 16-byte chunks per tile row
  constexpr int ROWB = DH * 2;        // bytes per tile row
  constexpr int TILE = BK * ROWB;     // bytes per K (or V) tile
  constexpr int SEG = CPR >= 16 ? 16 : 8;
  constexpr int NP = BK * CPR / 256;  // DMA pieces per thread per tile
  constexpr int KS = DH / 32;         // k steps of Q K^T
  constexpr int DB = DH / 16;         // 16-wide d blocks of O^T
  static_assert(NP >= 1 && BK * CPR % 256 == 0, "tile");
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [stage][K tile | V tile] x 2, then the bias window
  float* const sbias = reinterpret_cast<float*>(lds + 4 * TILE);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;

  // ---- unit of this workgroup.  XCD-aware order: workgroup i runs on XCD i % 8 (observed dispatch rule); every XCD gets a
  // contiguous range of logical ids, and the query blocks of one (batch, head, split) are neighbours -> they stream the same
  // K / V rows through one L2.
  int bid;
  {
    const int nwg = gridDim.x, qn = nwg >> 3, rn = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  // causal: the LAST query block of a head has the longest key range -- hand those out first
  const int qblk = a.causal ? a.nqb - 1 - bid % a.nqb : bid % a.nqb;
  const int sp = (bid / a.nqb) % a.ns;
  const int bh = bid / (a.nqb * a.ns);
  const int b = bh / a.H, h = bh - b * a.H;
  const int q0 = qblk * 64;
  const int Sq = a.Sq, Skv = a.Skv;
  const int ntile_all = (Skv + BK - 1) / BK;
  const int c_off = Skv - Sq;  // causal: query i sees keys j <= i + c_off
  int kt1_ = min(ntile_all, sp * a.tps + a.tps);
  if (a.causal) kt1_ = min(kt1_, (q0 + 63 + c_off) / BK + 1);  // tiles past the block's last visible key are skipped
  const int kt0 = sp * a.tps, kt1 = kt1_;
  const int kbeg = kt0 * BK;

  const int hkv = h / a.kv_group;
  const bf16_t* kb_ = a.k + (int64_t)b * a.k_bs + hkv * DH;
  const bf16_t* vb_ = a.v + (int64_t)b * a.v_bs + hkv * DH;

  // ---- Q fragments (B operand): Q[q][32 ks + 8 g .. + 7]
  const int qrow = q0 + 16 * w + l15;
  bf16x8 qf[KS];
  {
    const bf16_t* qp = a.q + (int64_t)b * a.q_bs + (int64_t)min(qrow, Sq - 1) * a.ldq + h * DH + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
  }
  // ---- relative-bias window of this unit, pre-multiplied by log2 e: sbias[(j - kbeg) + (q0 + 63 - i)]
  const bool has_bias = a.rel_bias != nullptr;
  if (has_bias) {
    const int tmin = kbeg - (q0 + 63) + a.max_len - 1;  // table row of slot 0
    const int nslot = (kt1 - kt0) * BK + 63;
    for (int t = tid; t < nslot; t += 256) {
      const int row = tmin + t;
      float v = 0.f;
      if (row >= 0 && row < 2 * a.max_len - 1) v = bf16_to_f32(a.rel_bias[(int64_t)row * a.H + h]) * 1.44269504088896340736f;
      sbias[t] = v;
    }
  }

  // ---- LDS-DMA of one tile: piece i of this thread is LDS chunk c = i * 256 + tid = (row c / CPR, position c % CPR)
  auto dma_k = [&](int kt, int stage) {
    char* const sK = lds + stage * 2 * TILE;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      // (a wave instruction covers 64 consecutive chunks: one row when CPR == 64 -- its row index is then scalar)
      const int c = i * 256 + w * 64 + lane, row = CPR >= 64 ? (i * 256 + w * 64) / CPR : c / CPR, cp = c % CPR;
      const int src_chunk = cp ^ (row & (SEG - 1));
      const bf16_t* src = kb_ + (int64_t)min(kt * BK + row, Skv - 1) * a.ldk + src_chunk * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sK + (i * 256 + w * 64) * 16), 16, 0, 0);
    }
  };
  auto dma_v = [&](int kt, int stage) {
    char* const sV = lds + stage * 2 * TILE + TILE;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = i * 256 + w * 64 + lane, row = CPR >= 64 ? (i * 256 + w * 64) / CPR : c / CPR, cp = c % CPR;
      const int src_chunk = (cp & ~(SEG - 1)) | (((cp & (SEG - 1)) - tv_rot<SEG>(row)) & (SEG - 1));
      const bf16_t* src = vb_ + (int64_t)min(kt * BK + row, Skv - 1) * a.ldv + src_chunk * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sV + (i * 256 + w * 64) * 16), 16, 0, 0);
    }
  };

  // ---- per-lane fragment offsets
  // K: row 16 kb + l15, chunk (4 ks + g) ^ (row & (SEG - 1)); the XOR value does not depend on kb (16 kb keeps the low bits)
  const int k_row_off = l15 * ROWB;
  const int k_swz = l15 & (SEG - 1);
  // V: rows 4 g + (l15 >> 2) (+ 16 for the second read), chunk 2 db + ((l15 & 3) >> 1) rotated, 8-byte half l15 & 1
  const int v_row = 4 * g + (l15 >> 2);
  const int v_rot = tv_rot<SEG>(v_row);
  const int v_base_off = v_row * ROWB + (l15 & 1) * 8;
  const int v_cc = (l15 & 3) >> 1;

  f32x4 o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float c_scale = a.scale_log2e;
  const int bias_q = q0 + 63 - qrow;  // (q0 + 63 - i); rows past Sq compute on a clamped copy and are not stored

  typedef __attribute__((address_space(3))) ta_v4s_t* lds_v4;

  if (kt0 < kt1) {
    dma_k(kt0, 0);
    dma_v(kt0, 0);
  }
  if constexpr (TIMED) tprev = __builtin_amdgcn_s_memtime();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int stage = (kt - kt0) & 1;
    const char* const tK = lds + stage * 2 * TILE;
    const char* const tV = tK + TILE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    U2_STAMP(0)  // wait for the tile's DMA
    __syncthreads();  // tile kt has landed for every wave; every wave is done with tile kt - 1 (the other stage)
    U2_STAMP(1)  // barrier
    if (kt + 1 < kt1) dma_k(kt + 1, stage ^ 1);
    U2_STAMP(2)  // K DMA issue
    // ---- S^T = K Q^T, NKB 16-key blocks.  Independent accumulator chains (key blocks x even / odd k steps): a dependent
    // v_mfma_f32_16x16x32 chain issues at its ~8-pass latency, not at the pipe rate
    f32x4 sc[NKB];
    {
      // software pipeline: PD fragment reads ahead, then one read per MFMA (sched_group_barrier below: hipcc otherwise
      // either clusters all reads in front of all MFMAs or serialises read -> wait -> MFMA)
      constexpr int NQK = NKB * KS, PD = NQK < 8 ? NQK : 8;
      f32x4 acc[NKB][2];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) acc[kb][0] = acc[kb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 kf[NQK];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NQK; ++i)  // step i = (ks = i / NKB, kb = i % NKB)
        kf[i] = *reinterpret_cast<const bf16x8*>(tK + (i % NKB) * 16 * ROWB + k_row_off + ((((i / NKB) * 4 + g) ^ k_swz) << 4));
#pragma unroll
      for (int i = 0; i < NQK; ++i)
        acc[i % NKB][(i / NKB) & 1] =
            mfma16(kf[i], qf[i / NKB], acc[i % NKB][(i / NKB) & 1]);
      __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);  // 0x100 = DS read, 0x008 = MFMA
#pragma unroll
      for (int i = 0; i < NQK; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i + PD < NQK) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) sc[kb] = KS > 1 ? acc[kb][0] + acc[kb][1] : acc[kb][0];
    }
    if constexpr (TIMED) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) asm volatile("" : "+v"(sc[kb]));
    }
    U2_STAMP(3)  // Q K^T
    // ---- online softmax: lane owns keys kt * 32 + 16 kb + 4 g + r of query row qrow
    // (one wave per SIMD issues a VALU instruction every 5-9 cycles: the instruction count of this block is its cost.
    //  Without bias the scale is folded into the exponent's FMA and the row max is taken on the raw scores -- scale > 0
    //  commutes with max; key masking only in a partial last tile.)
    constexpr int NX = 4 * NKB;  // scores per lane: x[i] = key kt * BK + 16 (i >> 2) + 4 g + (i & 3)
    float x[NX];
    if (has_bias) {
      float bb[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) bb[i] = sbias[(kt * BK - kbeg) + (i >> 2) * 16 + 4 * g + (i & 3) + bias_q];
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = __builtin_fmaf(sc[i >> 2][i & 3], c_scale, bb[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = sc[i >> 2][i & 3];
    }
    if (kt == ntile_all - 1 && (Skv & (BK - 1))) {
#pragma unroll
      for (int i = 0; i < NX; ++i)
        if (kt * BK + (i >> 2) * 16 + 4 * g + (i & 3) >= Skv) x[i] = -INFINITY;
    }
    if (a.causal && kt * BK + BK - 1 > q0 + 16 * w + c_off) {  // (wave-uniform) the tile reaches past this wave's diagonal
#pragma unroll
      for (int i = 0; i < NX; ++i)
        if (kt * BK + (i >> 2) * 16 + 4 * g + (i & 3) > qrow + c_off) x[i] = -INFINITY;
    }
    float mt = x[0];
#pragma unroll
    for (int i = 1; i < NX; ++i) mt = fmaxf(mt, x[i]);
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float xs = has_bias ? 1.0f : c_scale;  // what is left to multiply into x
    mt *= xs;
    if (__any(mt > m_run + TOKATTN_RESCALE_THR)) {  // wave-uniform; always taken on the first tile
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
      }
    }
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], xs, -m_run));
      ps += x[i];
    }
    l_run += ps;
    // P fragment j = keys 32 j .. 32 j + 31 of the tile: k-slot (g, e) carries key 32 j + 16 (e >> 2) + 4 g + (e & 3)
    union { bf16x8 v; uint32_t u[4]; } pf[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      pf[j].u[0] = pack2_bf16(x[8 * j + 0], x[8 * j + 1]);
      pf[j].u[1] = pack2_bf16(x[8 * j + 2], x[8 * j + 3]);
      pf[j].u[2] = pack2_bf16(x[8 * j + 4], x[8 * j + 5]);
      pf[j].u[3] = pack2_bf16(x[8 * j + 6], x[8 * j + 7]);
    }

    if constexpr (TIMED) {
#pragma unroll
      for (int j = 0; j < NPF; ++j) asm volatile("" : "+v"(pf[j].v));
    }
    U2_STAMP(4)  // softmax
    if (kt + 1 < kt1) dma_v(kt + 1, stage ^ 1);
    U2_STAMP(5)  // V DMA issue
    // ---- O^T += V^T P^T (same software pipeline over the d blocks)
    {
      constexpr int NPV = NPF * DB, PD = NPV < 6 ? NPV : 6;  // step i = (P fragment j = i / DB, d block db = i % DB)
      bf16x8 vf[NPV];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NPV; ++i) {
        const int cc = 2 * (i % DB) + v_cc;
        const int cp = (cc & ~(SEG - 1)) | (((cc & (SEG - 1)) + v_rot) & (SEG - 1));
        const char* p = tV + (i / DB) * 32 * ROWB + v_base_off + cp * 16;
        const ta_v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p);
        const ta_v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 16 * ROWB));
        vf[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < NPV; ++i)
        o[i % DB] = mfma16(vf[i], pf[i / DB].v, o[i % DB]);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * PD, 0);
#pragma unroll
      for (int i = 0; i < NPV; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i + PD < NPV) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (TIMED) {
#pragma unroll
      for (int db = 0; db < DB; ++db) asm volatile("" : "+v"(o[db]));
    }
    U2_STAMP(6)  // P V
  }
  if constexpr (TIMED) {
    if (lane == 0 && a.dbg) {
      unsigned long long* dp = a.dbg + ((size_t)blockIdx.x * 4 + w) * 8;
#pragma unroll
      for (int i = 0; i < 7; ++i) dp[i] += ts[i];
      dp[7] += (unsigned long long)(kt1 - kt0);
    }
  }
#undef U2_STAMP

  // ---- epilogue: lane holds O^T[d = 16 db + 4 g + r][q = qrow]
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  if (qrow >= Sq) return;
  if (a.ns == 1) {
    const float inv = 1.f / l_tot;
    bf16_t* op = a.out + (int64_t)b * a.o_bs + (int64_t)qrow * a.ldo + h * DH + 4 * g;
#pragma unroll
    for (int db = 0; db < DB; ++db)
      *reinterpret_cast<uint2*>(op + db * 16) =
          uint2{pack2_bf16(o[db][0] * inv, o[db][1] * inv), pack2_bf16(o[db][2] * inv, o[db][3] * inv)};
  } else {
    const int E = a.H * DH;
    float* pp = a.opart + (((int64_t)sp * a.nb + b) * Sq + qrow) * E + h * DH + 4 * g;
#pragma unroll
    for (int db = 0; db < DB; ++db) *reinterpret_cast<float4*>(pp + db * 16) = float4{o[db][0], o[db][1], o[db][2], o[db][3]};
    if (g == 0) {
      float* mp = a.ml + ((((int64_t)sp * a.nb + b) * a.H + h) * Sq + qrow) * 2;
      mp[0] = m_run;
      mp[1] = l_tot;
    }
  }
}

// max over the four 16-lane groups of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48) on the VALU: v_permlane32_swap exchanges the
// wave's halves, v_permlane16_swap the odd / even 16-lane rows -- two instructions instead of two LDS round trips of a shuffle
__device__ __forceinline__ float row_max4(float v) {
  typedef unsigned u2x __attribute__((ext_vector_type(2)));
  const unsigned u = __float_as_uint(v);
  const u2x a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const unsigned um = __float_as_uint(m);
  const u2x b = __builtin_amdgcn_permlane16_swap(um, um, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float row_sum4(float v) {
  typedef unsigned u2x __attribute__((ext_vector_type(2)));
  const unsigned u = __float_as_uint(v);
  const u2x a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned um = __float_as_uint(m);
  const u2x b = __builtin_amdgcn_permlane16_swap(um, um, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide heads (d = 256 / 512), round 5: TWO WAVES PER SIMD.
//
// tok_attn_kernel above runs one wave per SIMD at d = 512 (O^T alone is 128 registers) and every wave walks the whole K
// and V tile: per 32-key tile and wave 32 + 32 MFMAs against 32 ds_read_b128 + 64 ds_read_b64_tr_b16 and a ~70-instruction
// softmax block, all in one in-order instruction stream -- 5600-6200 cycles per tile against 1088 of matrix-pipe issue
// (profiles/r03_tokattn_phase_cycles.log: a lone wave gets a fraction of the LDS read rate and issues a dependent VALU
// instruction every 5-9 cycles).  Here a workgroup is 8 waves; a PAIR of waves owns a 16-query block and splits
//   * the KEYS of the tile for S^T = K Q^T: wave `half` takes key block `half` (16 keys x 16 q, the whole head dim:
//     d / 32 MFMAs, half of the K tile read), and
//   * the HEAD DIM for O^T += V^T P^T: wave `half` owns d / 2 columns (d / 32 accumulator tiles = 64 registers at
//     d = 512, half of the V tile read).
// The raw scores cross the pair through 2 KB of LDS (each wave stores its 16 x 16 block as one float4 per lane and
// both read the pair's two blocks back), after which BOTH waves run the same online softmax on the same 8 scores per
// lane -- same running max, same row sums, same bf16 probabilities, bit for bit, with no further exchange -- and feed
// P from their own registers as above.  The loop is software-pipelined so that the exchange costs no extra barrier:
//
//   iteration t:  wait DMA | barrier | issue V(t+1), K(t+2) | read scores(t) | softmax(t) | S(t+1) -> exchange | P V(t)
//
// i.e. the K ring runs one tile ahead of the V ring, the scores of tile t+1 are written before the barrier of iteration
// t+1 and read after it, and every DMA has a whole iteration to land.  While one wave of a SIMD is in its softmax
// block the other one can hold the matrix pipe; every wave issues half the fragment reads per MFMA of the old loop.
// LDS at d = 512: 2 x 32 KB K stages + 2 x 32 KB V stages + 16 KB exchange (2 parities x 8 waves x 1 KB) + bias window
// = 146.5 KB, one workgroup per CU; key splits / bias / tails exactly as tok_attn_kernel (same TokAttnArgs, same merge).
// MUBUF (option tok_wide = 2, round 6): the tiles' pieces leave as `buffer_load_dwordx4 ... lds` (descriptor over the batch entry's K / V,
// 32-bit lane offset, tile and piece origin in the scalar offset) instead of the FLAT-encoded global_load_lds: beside waves that issue
// MFMAs the FLAT form stages a third of what the MUBUF form does (tools/ubench/stage_bw.hip, profiles/r06_stage_bw.log).
template <int DH, bool TIMED = false, bool MUBUF = false>
__global__ __launch_bounds__(512) void tok_attn2_kernel(const TokAttnArgs a) {
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define U2_STAMP(i_)                                            \
  if constexpr (TIMED) {                                        \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    ts[i_] += t_ - tprev;                                       \
    tprev = t_;                                                 \
  }
  unsigned long long rt[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // TIMED: s_memrealtime (100 MHz) at entry / loop start / loop end / exit; [4..7] prologue: requests issued / K(kt0) + Q here / barrier / (spare)
  if constexpr (TIMED) rt[0] = __builtin_amdgcn_s_memrealtime();
  constexpr int BK = 32;
  constexpr int CPR = DH / 8;          // 16-byte chunks per tile row
  constexpr int ROWB = DH * 2;         // bytes per tile row
  constexpr int TILE = BK * ROWB;      // bytes per K (or V) tile
  constexpr int SEG = 16;
  constexpr int NP = BK * CPR / 512;   // DMA pieces per thread per tile (8 waves)
  constexpr int KS = DH / 32;          // k steps of Q K^T
  constexpr int DBH = DH / 32;         // 16-wide d blocks of O^T owned by one wave (half the head dim)
  static_assert(DH == 256 || DH == 512, "wide heads only");
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [K stage 0 | K 1 | V 0 | V 1 | exchange | bias window]
  char* const sKb = lds;
  char* const sVb = lds + 2 * TILE;
  char* const xb = lds + 4 * TILE;
  float* const sbias = reinterpret_cast<float*>(lds + 4 * TILE + 16384);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = w >> 1, half = w & 1;
  const int l15 = lane & 15, g = lane >> 4;

  int bid;
  {
    const int nwg = gridDim.x, qn = nwg >> 3, rn = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  const int qblk = bid % a.nqb;
  const int sp = (bid / a.nqb) % a.ns;
  const int bh = bid / (a.nqb * a.ns);
  const int b = bh / a.H, h = bh - b * a.H;
  const int q0 = qblk * 64;
  const int Sq = a.Sq, Skv = a.Skv;
  const int ntile_all = (Skv + BK - 1) / BK;
  const int kt0 = sp * a.tps, kt1 = min(ntile_all, sp * a.tps + a.tps);
  const int kbeg = kt0 * BK;

  const int hkv = h / a.kv_group;
  const bf16_t* kb_ = a.k + (int64_t)b * a.k_bs + hkv * DH;
  const bf16_t* vb_ = a.v + (int64_t)b * a.v_bs + hkv * DH;

  // ---- LDS-DMA of one tile: piece i of this thread is LDS chunk c = i * 512 + tid = (row c / CPR, position c % CPR).
  // Full tiles: scalar tile origin + a lane offset computed once (rows never leave the tensor); the partial last tile clamps.
  // Piece i covers rows row0 + i * RSTEP: the V rotation does not depend on i, the K swizzle (row & 15) alternates when RSTEP = 8 --
  // so one lane offset for V and two for K serve all pieces, the rest of the address is scalar.
  constexpr int RSTEP = 512 / CPR;  // 8 (d = 512) / 16 (d = 256)
  const int row0 = CPR >= 64 ? w : w * 2 + (lane >> 5), cp0 = lane & (CPR - 1);
  uint32_t koff[2], vofs;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = row0 + j * RSTEP;
    koff[j] = (uint32_t)row0 * (uint32_t)(a.ldk * 2) + ((cp0 ^ (row & (SEG - 1))) << 4);
  }
  vofs = (uint32_t)row0 * (uint32_t)(a.ldv * 2) + (((cp0 & ~(SEG - 1)) | (((cp0 & (SEG - 1)) - tv_rot<SEG>(row0)) & (SEG - 1))) << 4);
  // (MUBUF: one descriptor per operand over this batch entry's rows; byte offsets of a tile stay far below 2^31 -- the launcher checks)
  auto dma_tile = [&](const bf16_t* base, int64_t ld, bool is_k, int kt, char* dst) {
    const char* const tb = reinterpret_cast<const char*>(base) + (int64_t)kt * BK * ld * 2;
    const int ld2 = (int)(ld * 2), tbo = kt * BK * ld2;   // (MUBUF: scalar byte offset of the tile)
    if (kt * BK + BK <= Skv) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if constexpr (MUBUF) {
          lds_dma_mubuf16(base, dst + (i * 512 + w * 64) * 16, (int)(is_k ? koff[i & 1] : vofs), tbo + i * RSTEP * ld2);
        } else {
          const char* const rb = tb + (int64_t)(i * RSTEP) * ld * 2;  // scalar
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rb + (is_k ? koff[i & 1] : vofs)),
                                           (__attribute__((address_space(3))) void*)(dst + (i * 512 + w * 64) * 16), 16, 0, 0);
        }
      }
    } else {
      const int last = Skv - 1 - kt * BK;  // rows past the last key read a copy of it (masked in the softmax)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int row = row0 + i * RSTEP;
        if constexpr (MUBUF) {
          const int back = (row - min(row, last)) * ld2;   // (<= the row's own offset inside the tile: the sum stays >= 0)
          lds_dma_mubuf16(base, dst + (i * 512 + w * 64) * 16, (int)(is_k ? koff[i & 1] : vofs) + i * RSTEP * ld2 - back, tbo);
        } else {
          const char* const rb = tb + (int64_t)(i * RSTEP) * ld * 2 - (int64_t)(row - min(row, last)) * ld * 2;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rb + (is_k ? koff[i & 1] : vofs)),
                                           (__attribute__((address_space(3))) void*)(dst + (i * 512 + w * 64) * 16), 16, 0, 0);
        }
      }
    }
  };
  auto dma_k = [&](int kt, int stage) { dma_tile(kb_, a.ldk, true, kt, sKb + stage * TILE); };
  auto dma_v = [&](int kt, int stage) { dma_tile(vb_, a.ldv, false, kt, sVb + stage * TILE); };

  // ---- prologue, in the order the memory system should see it: the bias window's table rows (oldest: their wait lets
  // everything younger fly), K(kt0), the Q fragments, V(kt0), K(kt0 + 1).  S(kt0) starts once K(kt0) and Q are here.
  const bool has_bias = a.rel_bias != nullptr;
  const int tmin = kbeg - (q0 + 63) + a.max_len - 1;  // table row of slot 0 of sbias[(j - kbeg) + (q0 + 63 - i)]
  const int nslot = (kt1 - kt0) * BK + 63;
  float bias_in[2] = {0.f, 0.f};
  if (has_bias) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int t = tid + 512 * r, row = tmin + t;
      if (t < nslot && row >= 0 && row < 2 * a.max_len - 1) bias_in[r] = bf16_to_f32(a.rel_bias[(int64_t)row * a.H + h]);
    }
  }
  dma_k(kt0, 0);
  // Q fragments (B operand): Q[q][32 ks + 8 g .. + 7]; both waves of a pair hold the same 16 rows
  const int qrow = q0 + 16 * qb + l15;
  bf16x8 qf[KS];
  {
    const bf16_t* qp = a.q + (int64_t)b * a.q_bs + (int64_t)min(qrow, Sq - 1) * a.ldq + h * DH + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
  }
  dma_v(kt0, 0);
  const bool two = kt0 + 1 < kt1;
  if (two) dma_k(kt0 + 1, 1);
  if (has_bias) {  // pre-multiplied by log2 e
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (tid + 512 * r < nslot) sbias[tid + 512 * r] = bias_in[r] * 1.44269504088896340736f;
  }

  // ---- per-lane fragment offsets (layouts of tok_attn_kernel)
  const int k_off = (16 * half + l15) * ROWB;  // this wave's key block
  const int k_swz = l15;
  const int v_row = 4 * g + (l15 >> 2);
  const int v_rot = tv_rot<SEG>(v_row);
  const int v_base_off = v_row * ROWB + (l15 & 1) * 8;
  const int v_cc = (l15 & 3) >> 1;
  // V^T fragment i of this wave (d block half * DBH + i) = chunk 2 (half * DBH + i) + v_cc rotated inside its 16-chunk segment:
  // lane offset voff[i & 7] + 256 (i >> 3) from the stage base (+ 16 key rows for the second half of the fragment)
  uint32_t voff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) voff[j] = v_base_off + half * (DBH * 32) + (((2 * j + v_cc + v_rot) & (SEG - 1)) << 4);
  const uint32_t v_lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)sVb;
  const int x_wr = w * 1024 + lane * 16;        // this wave's score block inside one parity of the exchange
  const int x_rd = (2 * qb) * 1024 + lane * 16;  // the pair's two blocks: + 0 (keys 0..15), + 1024 (keys 16..31)

  f32x4 o[DBH];
#pragma unroll
  for (int i = 0; i < DBH; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float c_scale = a.scale_log2e;
  const int bias_q = q0 + 63 - qrow;

  // S^T block of this wave for the tile in K stage `stage` -> exchange parity `par`, in two parts: the first K1 fragment reads
  // go out early (the loop puts the softmax of the previous tile between the two parts), the rest ride on the MFMAs.
  constexpr int K1 = DH >= 512 ? 4 : 8;  // (d = 512: 8 early fragments do not fit the 256 registers next to Q and O^T)
  bf16x8 kf[KS];
  auto scores_begin = [&](int stage) {
    const char* const tK = sKb + stage * TILE + k_off;
#pragma unroll
    for (int ks = 0; ks < K1; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(tK + (((ks * 4 + g) ^ k_swz) << 4));
  };
  auto scores_end = [&](int stage, int par) {
    const char* const tK = sKb + stage * TILE + k_off;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = K1; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(tK + (((ks * 4 + g) ^ k_swz) << 4));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc[ks & 3] = mfma16(kf[ks], qf[ks], acc[ks & 3]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 0x008 = MFMA, 0x100 = DS read
      if (ks + K1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 sc = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    *reinterpret_cast<f32x4*>(xb + par * 8192 + x_wr) = sc;
  };

  if constexpr (TIMED) rt[4] = __builtin_amdgcn_s_memrealtime();
  // K(kt0) and Q are the oldest requests after the bias rows: everything younger (V(kt0), K(kt0 + 1)) may still be in flight
  if (two) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
  if constexpr (TIMED) rt[5] = __builtin_amdgcn_s_memrealtime();
  __syncthreads();  // K(kt0) has landed for every wave, the bias window is written
  if constexpr (TIMED) rt[6] = __builtin_amdgcn_s_memrealtime();
  scores_begin(0);
  scores_end(0, 0);
  if constexpr (TIMED) {
    rt[1] = __builtin_amdgcn_s_memrealtime();
    tprev = __builtin_amdgcn_s_memtime();
  }
  for (int kt = kt0; kt < kt1; ++kt) {
    const int par = (kt - kt0) & 1;
    const bool more = kt + 1 < kt1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    U2_STAMP(0)  // wait for K(kt + 1), V(kt)
    __syncthreads();  // ... landed for every wave; scores(kt) of both halves are in the exchange; V stage par ^ 1 / K stage par are free
    U2_STAMP(1)  // barrier
    if (more) dma_v(kt + 1, par ^ 1);
    if (kt + 2 < kt1) dma_k(kt + 2, par);
    U2_STAMP(2)  // DMA issue
    // ---- first fragment reads of S(kt + 1): their latency passes under the softmax below
    if (more) scores_begin(par ^ 1);
    // ---- online softmax: lane owns keys kt * 32 + 16 kb + 4 g + r of query row qrow (both waves of the pair alike)
    constexpr int NX = 8;
    float x[NX];
    {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(xb + par * 8192 + x_rd);
      const f32x4 s1 = *reinterpret_cast<const f32x4*>(xb + par * 8192 + x_rd + 1024);
      if (has_bias) {
        float bb[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) bb[i] = sbias[(kt * BK - kbeg) + (i >> 2) * 16 + 4 * g + (i & 3) + bias_q];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[i] = __builtin_fmaf(s0[i], c_scale, bb[i]);
          x[4 + i] = __builtin_fmaf(s1[i], c_scale, bb[4 + i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[i] = s0[i];
          x[4 + i] = s1[i];
        }
      }
    }
    if (kt == ntile_all - 1 && (Skv & (BK - 1))) {
#pragma unroll
      for (int i = 0; i < NX; ++i)
        if (kt * BK + (i >> 2) * 16 + 4 * g + (i & 3) >= Skv) x[i] = -INFINITY;
    }
    float mt = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
    mt = row_max4(mt);  // over the four 16-lane groups (the other keys of this query row)
    const float xs = has_bias ? 1.0f : c_scale;  // what is left to multiply into x
    mt *= xs;
    if (__any(mt > m_run + TOKATTN_RESCALE_THR)) {  // wave-uniform, and the same decision in both waves of the pair
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < DBH; ++i) {
        o[i][0] *= alpha; o[i][1] *= alpha; o[i][2] *= alpha; o[i][3] *= alpha;
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], xs, -m_run));
    l_run += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    union { bf16x8 v; uint32_t u[4]; } pf;
    pf.u[0] = pack2_bf16(x[0], x[1]);
    pf.u[1] = pack2_bf16(x[2], x[3]);
    pf.u[2] = pack2_bf16(x[4], x[5]);
    pf.u[3] = pack2_bf16(x[6], x[7]);
    if constexpr (TIMED) asm volatile("" : "+v"(pf.v));
    U2_STAMP(4)  // softmax
    // ---- S^T of the next tile (its K tile landed with this iteration's wait)
    if (more) scores_end(par ^ 1, par ^ 1);
    U2_STAMP(3)  // Q K^T
    // ---- O^T += V^T P^T over this wave's half of the head dim: one asm block (tools/gen_tokattn_asm.py) -- hipcc would put
    // s_waitcnt vmcnt(0) between this iteration's DMA issue and the first transpose read
    {
      const uint32_t tv = v_lds0 + par * TILE;
#define U2_PV_ADDR [a0] "v"(tv + voff[0]), [a1] "v"(tv + voff[1]), [a2] "v"(tv + voff[2]), [a3] "v"(tv + voff[3]), \
                   [a4] "v"(tv + voff[4]), [a5] "v"(tv + voff[5]), [a6] "v"(tv + voff[6]), [a7] "v"(tv + voff[7])
      if constexpr (DBH == 16) {
        asm volatile(TOKATTN_PV_ASM_TEXT_16
                     : [o0] "+v"(o[0]), [o1] "+v"(o[1]), [o2] "+v"(o[2]), [o3] "+v"(o[3]), [o4] "+v"(o[4]), [o5] "+v"(o[5]),
                       [o6] "+v"(o[6]), [o7] "+v"(o[7]), [o8] "+v"(o[8]), [o9] "+v"(o[9]), [o10] "+v"(o[10]), [o11] "+v"(o[11]),
                       [o12] "+v"(o[12]), [o13] "+v"(o[13]), [o14] "+v"(o[14]), [o15] "+v"(o[15])
                     : [pf] "v"(pf.v), U2_PV_ADDR
                     : TOKATTN_PV_ASM_CLOBBERS, "memory");
      } else {
        asm volatile(TOKATTN_PV_ASM_TEXT_8
                     : [o0] "+v"(o[0]), [o1] "+v"(o[1]), [o2] "+v"(o[2]), [o3] "+v"(o[3]), [o4] "+v"(o[4]), [o5] "+v"(o[5]),
                       [o6] "+v"(o[6]), [o7] "+v"(o[7])
                     : [pf] "v"(pf.v), U2_PV_ADDR
                     : TOKATTN_PV_ASM_CLOBBERS, "memory");
      }
#undef U2_PV_ADDR
    }
    U2_STAMP(6)  // P V
  }
  if constexpr (TIMED) rt[2] = __builtin_amdgcn_s_memrealtime();
#undef U2_STAMP
  auto timed_out = [&]() {
    if constexpr (TIMED) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores of the epilogue
      rt[3] = __builtin_amdgcn_s_memrealtime();
      if (lane == 0 && a.dbg) {
        unsigned long long* dp = a.dbg + ((size_t)blockIdx.x * 8 + w) * 16;
#pragma unroll
        for (int i = 0; i < 7; ++i) dp[i] += ts[i];
        dp[7] += (unsigned long long)(kt1 - kt0);
#pragma unroll
        for (int i = 0; i < 8; ++i) dp[8 + i] = rt[i];
      }
    }
  };

  // ---- epilogue: lane holds O^T[d = 16 (half * DBH + i) + 4 g + r][q = qrow]
  const float l_tot = row_sum4(l_run);
  if (qrow >= Sq) {
    timed_out();
    return;
  }
  const int d0 = half * (DH / 2) + 4 * g;
  if (a.ns == 1) {
    const float inv = 1.f / l_tot;
    bf16_t* op = a.out + (int64_t)b * a.o_bs + (int64_t)qrow * a.ldo + h * DH + d0;
#pragma unroll
    for (int i = 0; i < DBH; ++i)
      *reinterpret_cast<uint2*>(op + i * 16) =
          uint2{pack2_bf16(o[i][0] * inv, o[i][1] * inv), pack2_bf16(o[i][2] * inv, o[i][3] * inv)};
  } else {
    const int E = a.H * DH;
    float* pp = a.opart + (((int64_t)sp * a.nb + b) * Sq + qrow) * E + h * DH + d0;
#pragma unroll
    for (int i = 0; i < DBH; ++i) *reinterpret_cast<float4*>(pp + i * 16) = float4{o[i][0], o[i][1], o[i][2], o[i][3]};
    if (g == 0 && half == 0) {
      float* mp = a.ml + ((((int64_t)sp * a.nb + b) * a.H + h) * Sq + qrow) * 2;
      mp[0] = m_run;
      mp[1] = l_tot;
    }
  }
  timed_out();
}

// out[b][q][e] = sum_s 2^(m_s - m) O_s[b][q][e] / sum_s 2^(m_s - m) l_s over the key splits, s in ascending order
__global__ __launch_bounds__(256) void tok_attn_combine_kernel(const TokAttnArgs a, int DH) {
  const int E = a.H * DH, e4 = E >> 2;
  const int64_t total = (int64_t)a.nb * a.Sq * e4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int e = (int)(idx % e4) * 4;
  const int64_t bq = idx / e4;
  const int qrow = (int)(bq % a.Sq), b = (int)(bq / a.Sq);
  const int h = e / DH;
  float m = -INFINITY;
  for (int s = 0; s < a.ns; ++s) m = fmaxf(m, a.ml[((((int64_t)s * a.nb + b) * a.H + h) * a.Sq + qrow) * 2]);
  float L = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.ns; ++s) {
    const float* mp = a.ml + ((((int64_t)s * a.nb + b) * a.H + h) * a.Sq + qrow) * 2;
    const float wgt = __builtin_amdgcn_exp2f(mp[0] - m);
    L += wgt * mp[1];
    const float4 t = *reinterpret_cast<const float4*>(a.opart + (((int64_t)s * a.nb + b) * a.Sq + qrow) * E + e);
    acc[0] += wgt * t.x; acc[1] += wgt * t.y; acc[2] += wgt * t.z; acc[3] += wgt * t.w;
  }
  const float inv = 1.f / L;
  *reinterpret_cast<uint2*>(a.out + (int64_t)b * a.o_bs + (int64_t)qrow * a.ldo + e) =
      uint2{pack2_bf16(acc[0] * inv, acc[1] * inv), pack2_bf16(acc[2] * inv, acc[3] * inv)};
}

// The same merge with the split count known at compile time: the NS (max, sum) pairs and the NS partial rows leave together instead of
// one dependent load after the other (round 6: 13 launches per volume, 8 us each for 6-18 MB).  Same operations in the same order.
template <int NS>
__global__ __launch_bounds__(256) void tok_attn_combine_ns_kernel(const TokAttnArgs a, int DH) {
  const int E = a.H * DH, e4 = E >> 2;
  const int64_t total = (int64_t)a.nb * a.Sq * e4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int e = (int)(idx % e4) * 4;
  const int64_t bq = idx / e4;
  const int qrow = (int)(bq % a.Sq), b = (int)(bq / a.Sq);
  const int h = e / DH;
  float2 ml[NS];
  float4 t[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    ml[s] = *reinterpret_cast<const float2*>(a.ml + ((((int64_t)s * a.nb + b) * a.H + h) * a.Sq + qrow) * 2);
    t[s] = *reinterpret_cast<const float4*>(a.opart + (((int64_t)s * a.nb + b) * a.Sq + qrow) * E + e);
  }
  float m = -INFINITY;
#pragma unroll
  for (int s = 0; s < NS; ++s) m = fmaxf(m, ml[s].x);
  float L = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const float wgt = __builtin_amdgcn_exp2f(ml[s].x - m);
    L += wgt * ml[s].y;
    acc[0] += wgt * t[s].x; acc[1] += wgt * t[s].y; acc[2] += wgt * t[s].z; acc[3] += wgt * t[s].w;
  }
  const float inv = 1.f / L;
  *reinterpret_cast<uint2*>(a.out + (int64_t)b * a.o_bs + (int64_t)qrow * a.ldo + e) =
      uint2{pack2_bf16(acc[0] * inv, acc[1] * inv), pack2_bf16(acc[2] * inv, acc[3] * inv)};
}

static inline int tok_attn_bk(int d) { return d <= 128 ? 64 : 32; }  // keys per tile (tok_attn_kernel: BK)

// ns for a call: enough workgroups to cover the 256 CUs, at least two tiles per split, partial sums within the scratch.
static int tok_attn_pick_splits(int nb, int H, int Sq, int Skv, int d, size_t ws_bytes) {
  const int64_t base = (int64_t)nb * H * cdiv(Sq, 64);
  const int ntile = (int)cdiv(Skv, tok_attn_bk(d));
  if (base >= 192 || ntile < 4) return 1;
  int ns = (int)std::min<int64_t>(cdiv(256, base), ntile / 2);
  const size_t per = (size_t)nb * Sq * ((size_t)H * d * 4 + (size_t)H * 8);
  if (per == 0 || ws_bytes / per < 2) return 1;
  ns = (int)std::min<size_t>((size_t)ns, ws_bytes / per);
  return std::max(ns, 1);
}

static unsigned long long* g_tokattn_dbg = nullptr;  // diagnostics only, process-wide (like flash_set_debug_buffer)
int tok_attention_set_debug_buffer(void* p) {
  g_tokattn_dbg = reinterpret_cast<unsigned long long*>(p);
  return U2_OK;
}

size_t tok_attention_workspace_bytes(int nb, int H, int Sq, int Skv, int d) {
  const int64_t base = (int64_t)nb * H * cdiv(Sq, 64);
  const int ntile = (int)cdiv(Skv, tok_attn_bk(d));
  if (base >= 192 || ntile < 4) return 0;
  const int ns = (int)std::min<int64_t>(cdiv(256, base), ntile / 2);
  return (size_t)ns * nb * Sq * ((size_t)H * d * 4 + (size_t)H * 8);
}

bool tok_attention_supported(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* out, int Sq, int Skv, int d,
                             int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                             int64_t o_bs, const bf16_t* rel_bias, int max_len) {
  if (d != 64 && d != 128 && d != 256 && d != 512) return false;
  if ((ldq | ldk | ldv | q_bs | k_bs | v_bs) & 7) return false;
  if ((ldo | o_bs) & 3) return false;
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) || ((uintptr_t)out & 7)) return false;
  if (rel_bias && (Sq > max_len || Skv > max_len || (int)cdiv(Skv, 64) * 64 + 63 > TOKATTN_BIAS_SLOTS)) return false;
  return true;
}

int tok_attention(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int nb, int Sq, int Skv, int H, int d,
                  int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                  float scale, const bf16_t* rel_bias, int max_len, int force_splits, void* ws, size_t ws_bytes,
                  hipStream_t stream) {
  return attention_ex(q, k, v, out, nb, Sq, Skv, H, H, d, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, scale, rel_bias, max_len, 0,
                      force_splits, ws, ws_bytes, stream);
}

int attention_ex(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int nb, int Sq, int Skv, int H, int Hkv, int d,
                 int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                 float scale, const bf16_t* rel_bias, int max_len, int causal, int force_splits, void* ws, size_t ws_bytes,
                 hipStream_t stream) {
  if (!q || !k || !v || !out || nb <= 0 || Sq <= 0 || Skv <= 0 || H <= 0 || Hkv <= 0 || H % Hkv) return U2_ERR_ARG;
  if (causal && (Skv < Sq || rel_bias)) return U2_ERR_ARG;
  if (!tok_attention_supported(q, k, v, out, Sq, Skv, d, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, rel_bias, max_len))
    return U2_ERR_ARG;
  TokAttnArgs a;
  a.q = q; a.k = k; a.v = v; a.out = out;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
  a.nb = nb; a.H = H; a.Sq = Sq; a.Skv = Skv; a.nqb = (int)cdiv(Sq, 64);
  a.scale_log2e = scale * 1.44269504088896340736f;
  a.rel_bias = rel_bias; a.max_len = max_len;
  const int ntile = (int)cdiv(Skv, tok_attn_bk(d));
  a.kv_group = H / Hkv;
  a.causal = causal ? 1 : 0;
  int ns = force_splits > 0 ? std::min(force_splits, ntile) : tok_attn_pick_splits(nb, H, Sq, Skv, d, ws ? ws_bytes : 0);
  if (causal) ns = 1;  // (a causal unit's key range depends on its query block: no key splits; prefill has enough units)
  const size_t per = (size_t)nb * Sq * ((size_t)H * d * 4 + (size_t)H * 8);
  if (ns > 1 && (!ws || ws_bytes < (size_t)ns * per || ((uintptr_t)ws & 15))) {
    if (force_splits > 0) return U2_ERR_WORKSPACE;
    ns = 1;
  }
  a.tps = (int)cdiv(ntile, ns);
  a.ns = (int)cdiv(ntile, a.tps);  // no empty splits
  a.opart = nullptr; a.ml = nullptr;
  if (a.ns > 1) {
    a.opart = reinterpret_cast<float*>(ws);
    a.ml = a.opart + (size_t)a.ns * nb * Sq * H * d;
  }
  const int64_t grid = (int64_t)nb * H * a.nqb * a.ns;
  if (grid > 0x7fffffff) return U2_ERR_ARG;
  ProfScope ps(PROF_TOKATTN, (causal ? 2.0 : 4.0) * nb * H * (double)Sq * Skv * d, stream,
               2.0 * nb * d * (2.0 * Sq * H + 2.0 * Skv * Hkv));  // q, k, v read + o written, once
  a.dbg = g_tokattn_dbg;
#define U2_TA(D_)                                                                                                      \
  do {                                                                                                                 \
    constexpr size_t smem_ = 4 * ((D_) <= 128 ? 64 : 32) * (D_) * 2 + TOKATTN_BIAS_SLOTS * 4;                                               \
    if (a.dbg) hipLaunchKernelGGL((tok_attn_kernel<D_, true>), dim3((unsigned)grid), dim3(256), smem_, stream, a);    \
    else hipLaunchKernelGGL((tok_attn_kernel<D_, false>), dim3((unsigned)grid), dim3(256), smem_, stream, a);         \
  } while (0)
#define U2_TA2(D_)                                                                                                     \
  do {                                                                                                                 \
    constexpr size_t smem_ = 4 * 32 * (D_) * 2 + 16384 + TOKATTN_BIAS_SLOTS * 4;                                       \
    if (a.dbg) hipLaunchKernelGGL((tok_attn2_kernel<D_, true>), dim3((unsigned)grid), dim3(512), smem_, stream, a);   \
    else if (mubuf) hipLaunchKernelGGL((tok_attn2_kernel<D_, false, true>), dim3((unsigned)grid), dim3(512), smem_, stream, a); \
    else hipLaunchKernelGGL((tok_attn2_kernel<D_, false>), dim3((unsigned)grid), dim3(512), smem_, stream, a);        \
  } while (0)
  const bool wide = d >= 256 && !causal && opts().tok_wide;  // two waves per SIMD (tok_attn2_kernel)
  const bool mubuf = opts().tok_wide == 2 && (int64_t)Skv * ldk < (1ll << 29) && (int64_t)Skv * ldv < (1ll << 29);
  if (d == 512) { if (wide) U2_TA2(512); else U2_TA(512); }
  else if (d == 256) { if (wide) U2_TA2(256); else U2_TA(256); }
  else if (d == 128) U2_TA(128);
  else U2_TA(64);
#undef U2_TA
#undef U2_TA2
  if (a.ns > 1) {
    const int64_t total = (int64_t)nb * Sq * (H * d / 4);
    const dim3 cg((unsigned)cdiv(total, 256));
#define U2_CMB(NS_) case NS_: hipLaunchKernelGGL(tok_attn_combine_ns_kernel<NS_>, cg, dim3(256), 0, stream, a, d); break
    switch (a.ns <= 8 ? a.ns : 0) {
      U2_CMB(2); U2_CMB(3); U2_CMB(4); U2_CMB(5); U2_CMB(6); U2_CMB(7); U2_CMB(8);
      default: hipLaunchKernelGGL(tok_attn_combine_kernel, cg, dim3(256), 0, stream, a, d);
    }
#undef U2_CMB

  }
  return launch_status();
}

}  // namespace u2
