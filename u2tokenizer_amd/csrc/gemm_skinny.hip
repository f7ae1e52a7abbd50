// M <= 256 rows against a wide COLD weight: the 22 query-side nn.Linear products of the tokenizer's TextConditionTokenAttMap chain
// (/root/reference/src/model/u2tokenizer/tta.py:93-103: self-attention out-projection, the q projections and dense layers of the visual
// and the text cross attention, four layers; 256 queries x 4096 x 4096 at the benchmark's size, 33.5 MB of weights each, streamed once).
//
// Round 5 ran them as 64 x 64 tiles x 4 K slices (1024 workgroups, fp32 partial sums, a reduce launch): 25.7 + 7.5 us.  Such a product is
// bound by what a CU's vector-memory path ingests (~25 B/clk through LDS-DMA: profiles/r01_stage_bw.log, r04_bt_kloop_ablations.log) and
// by the bytes it keeps in flight against the latency of a cold weight, not by the matrix pipe.  Ingest per workgroup = (rows + columns)
// of its tile x its K range, so the tile that minimises it under "one workgroup per CU" is ALL 256 rows x 64 columns x K / 4:
// 64 column strips x 4 K slices = 256 workgroups, 640 KB each (64 x 64 tiles: 1 MB per CU).  Structure (cdna_hip_programming.md, "projection
// GEMM at M = 256"): 8 waves = 8 (M) x 1 (N), wave w owns rows [32 w, 32 w + 32) and all 64 columns (two v_mfma_f32_32x32x16 accumulators);
// both operands through LDS in full 128-byte rows by LDS-DMA (32 + 8 pieces of 1 KB per K tile), XOR-swizzled rows as in gemm_bt.hip; the
// activations two tiles ahead in three stages, the weights SEVEN tiles ahead in a ring of eight, issued by different waves (below: why);
// one counted wait + one raw barrier per tile.
//
// The K slices are combined IN the launch (no reduce kernel, no launch boundary): every workgroup leaves its fp32 tile in the stream's
// scratch in the accumulators' own register order (1 KB per wave-instruction, fully coalesced), releases it (agent scope) and draws a
// ticket; the workgroup that draws the last ticket of its strip acquires, adds the slabs of ALL slices -- its own included, read back --
// in slice order (so the sum does not depend on which slice came last: bit-repeatable), applies the epilogue and resets the ticket.
// The tickets live in the zeroed header of the registered scratch (ctx.h: kScratchHeader).  A strip's slices sit on one XCD (its L2 then
// serves the reducer's reads).
#include <algorithm>
#include "kernels.h"

namespace u2 {
namespace {

// LDS: three stages of the activation tile (256 rows x 128 B) + a ring of eight weight tiles (64 rows x 128 B) = 160 KB
constexpr int SK_A = 32768, SK_W = 8192, SK_NA = 3, SK_NW = 8, SK_WBASE = SK_NA * SK_A, SK_LDS = SK_WBASE + SK_NW * SK_W;
typedef unsigned int sk_u32x4 __attribute__((ext_vector_type(4)));

struct SkinnyArgs {
  float* slabs;        // [slice][strip][wave][8 quads][64 lanes] float4: the accumulators' own order
  unsigned* tickets;   // [strip], zero between launches
  int S, nstrips, kt_per;
  int dbg;             // measurement only (option gemm_skinny = 2 .. 6: WRONG results): 2 no combine, 3 no MFMA, 4 no weight DMA, 5 no activation DMA, 6 no fragment reads
};

__global__ __launch_bounds__(512, 1) void gemm_skinny_kernel(GemmDesc d, SkinnyArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  // XCD-contiguous remap (workgroup b runs on XCD b & 7), then strip = id / S, slice = id % S: a strip's slices share an L2
  const int total = a.nstrips * a.S;
  int id = blockIdx.x;
  {
    const int q = total >> 3, r = total & 7, xcd = id & 7, idx = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int strip = id / a.S, slice = id - strip * a.S;
  const int n0 = strip * 64;
  const int nkt_all = d.K >> 6;
  const int kt0 = slice * a.kt_per, nk = min(a.kt_per, nkt_all - kt0);

  // ---- LDS-DMA: a piece = 8 rows x 128 B, lane -> (row lane >> 3, 16-byte position lane & 7); position p of row r holds global chunk
  // p ^ ((r >> 1) & 7).  The two operands have different masters: the activations (2 MB, L2-resident, re-read by every strip) arrive
  // in ~1 us, the weights come cold from HBM in 2-3 us -- and a wave's vector-memory counter retires IN ORDER, so a wave that issued both
  // could never wait for the fresh activation tile without also draining its older weight requests.  Waves 0-3 therefore issue the
  // activation pieces only (8 each per K tile, two tiles ahead), waves 4-7 the weight pieces only (2 each, SEVEN tiles ahead: 56 KB of
  // weights in flight per CU = 14 MB chip-wide against the latency); each waits on its own counter, the barrier publishes both.
  const int pr = lane >> 3, pc = lane & 7;
  const bool w_side = wave >= 4;
  const bf16_t* src[8];
  if (!w_side) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 8 * (8 * wave + i) + pr;              // rows past M read row M - 1: never stored
      src[i] = d.A + (int64_t)min(row, d.M - 1) * d.lda + (int64_t)kt0 * 64 + ((pc ^ ((row >> 1) & 7)) << 3);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 8 * (2 * (wave - 4) + (i & 1)) + pr;
      src[i] = d.B + (int64_t)min(n0 + row, d.N - 1) * d.ldb + (int64_t)kt0 * 64 + ((pc ^ ((row >> 1) & 7)) << 3);
    }
  }
  auto issue_a = [&](int t) {   // waves 0-3
    char* st = lds + (t % SK_NA) * SK_A + (8 * wave) * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + t * 64),
                                       (__attribute__((address_space(3))) void*)(st + i * 1024), 16, 0, 0);
  };
  auto issue_w = [&](int t) {   // waves 4-7
    char* st = lds + SK_WBASE + (t % SK_NW) * SK_W + (2 * (wave - 4)) * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + t * 64),
                                       (__attribute__((address_space(3))) void*)(st + i * 1024), 16, 0, 0);
  };

  // ---- fragments: lane (l31, hi) reads row l31 of a 32-row block, k chunk 2 kk + hi (swizzled)
  const int frow = l31 * 128;
  int foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) foff[kk] = ((2 * kk + hi) ^ ((l31 >> 1) & 7)) << 4;
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  if (!w_side) {
    if (a.dbg != 5) { issue_a(0); if (nk > 1) issue_a(1); }
  } else {
    if (a.dbg != 4) for (int t = 0; t < min(nk, SK_NW - 1); ++t) issue_w(t);
  }
  for (int t = 0; t < nk; ++t) {
    // K tile t has landed when only this wave's YOUNGER requests are outstanding (vector memory retires in order): one activation tile
    // (8 pieces), or the weight tiles t + 1 .. min(nk - 1, t + 6) (2 pieces each)
    if (!w_side) {
      if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      switch (min(nk - 1, t + SK_NW - 2) - t) {
        case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    }
    __builtin_amdgcn_s_barrier();                          // ... for every wave's pieces; and everybody is done with tile t - 1
    if (!w_side) {
      if (t + 2 < nk && a.dbg != 5) issue_a(t + 2);        // into the stage tile t - 1 just left
    } else {
      if (t + SK_NW - 1 < nk && a.dbg != 4) issue_w(t + SK_NW - 1);   // into the ring slot tile t - 1 just left
    }
    const char* sA = lds + (t % SK_NA) * SK_A + (32 * wave) * 128 + frow;
    const char* sW = lds + SK_WBASE + (t % SK_NW) * SK_W + frow;
    if (a.dbg == 6) continue;
    bf16x8 xf[4], w0[4], w1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      xf[kk] = *reinterpret_cast<const bf16x8*>(sA + foff[kk]);
      w0[kk] = *reinterpret_cast<const bf16x8*>(sW + foff[kk]);
      w1[kk] = *reinterpret_cast<const bf16x8*>(sW + 32 * 128 + foff[kk]);
    }
    if (a.dbg == 3) { asm volatile("" :: "v"(xf[0]), "v"(xf[1]), "v"(xf[2]), "v"(xf[3]), "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]), "v"(w1[0]), "v"(w1[1]), "v"(w1[2]), "v"(w1[3])); continue; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      acc[0] = mfma32(w0[kk], xf[kk], acc[0]);             // (weights as the first operand: lane = row m, registers = columns n)
      acc[1] = mfma32(w1[kk], xf[kk], acc[1]);
    }
  }

  // ---- combine the K slices in the launch
  if (a.S > 1) {
    sk_u32x4* mine = reinterpret_cast<sk_u32x4*>(a.slabs) + ((((int64_t)slice * a.nstrips + strip) * 8 + wave) * 8) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x16& c = acc[q >> 2];
      const int r = 4 * (q & 3);
      mine[q * 64] = sk_u32x4{__float_as_uint(c[r]), __float_as_uint(c[r + 1]), __float_as_uint(c[r + 2]), __float_as_uint(c[r + 3])};
    }
    // publish: every wave's stores retired -> barrier -> ONE agent-scope release -> ticket (this order: cdna_hip_programming.md,
    // "in-launch split-K reduction"); the ticket's value reaches the other waves through the (now idle) LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(lds);
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned tk = __hip_atomic_fetch_add(a.tickets + strip, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = tk == (unsigned)(a.S - 1);
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(a.tickets + strip, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
      }
      *flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (*flag == 0u || a.dbg == 2) return;
    // the last arriver: the slabs of all slices in slice order (its own read back: the sum is the same whoever comes last)
    const sk_u32x4* base = reinterpret_cast<const sk_u32x4*>(a.slabs) + (((int64_t)strip * 8 + wave) * 8) * 64 + lane;
    const int64_t sstride = (int64_t)a.nstrips * 8 * 8 * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    for (int s = 0; s < a.S; ++s) {
      sk_u32x4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load(base + s * sstride + q * 64);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        f32x16& c = acc[q >> 2];
        const int r = 4 * (q & 3);
        c[r] += __uint_as_float(v[q].x); c[r + 1] += __uint_as_float(v[q].y);
        c[r + 2] += __uint_as_float(v[q].z); c[r + 3] += __uint_as_float(v[q].w);
      }
    }
  }

  // ---- epilogue: v_permlane32_swap gives a lane 8 consecutive columns of its row (gemm_bt.hip: pp_epilogue)
  const int m = 32 * wave + l31;
  const bool out_f32 = d.flags & GEMM_OUT_F32;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][8 * t + e]), __float_as_uint(acc[j][8 * t + 4 + e]), false, false);
        v[e] = __uint_as_float(r[0]);
        v[4 + e] = __uint_as_float(r[1]);
      }
      const int n = n0 + 32 * j + 16 * t + 8 * hi;
      if (m >= d.M || n >= d.N) continue;                  // (N % 64 == 0: whole 8-column groups)
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
// gemm_bf16 (GEMM_VEC_OK resolved).  Taken: one batch entry, 64 < M <= 256 rows, row-major operands, K a multiple of 64 with >= 16 K
// tiles, N a multiple of 64 whose strips x slices make about one workgroup per CU, 16-byte epilogue accesses; the slices need the
// stream's scratch (tickets + slabs) -- without one the product stays where it was.
int gemm_skinny_try(const GemmDesc& d, hipStream_t stream) {
  const Options& o = opts();
  if (!o.gemm_skinny || o.gemm_tile != 0 || o.gemm_big != 0 || o.gemm_splitk != 0) return 0;   // (forced choices keep their kernels)
  if (d.nz != 1 || d.M <= 64 || d.M > 256 || (d.N & 63) || (d.K & 63) || d.ldbk) return 0;
  if (d.flags & (GEMM_BIAS_M | GEMM_A_KMAJOR | GEMM_B_KMAJOR | GEMM_SWIGLU) || !(d.flags & GEMM_VEC_OK) || d.vt) return 0;
  const bool f32 = d.flags & GEMM_OUT_F32;
  if (((uintptr_t)d.C & 15) || (d.ldc & (f32 ? 3 : 7))) return 0;
  if ((d.flags & GEMM_BIAS_N) && ((uintptr_t)d.bias & 15)) return 0;
  if ((d.flags & GEMM_RESIDUAL) && (((uintptr_t)d.R & 15) || (d.ldr & 7))) return 0;
  const int nstrips = d.N >> 6, nkt = d.K >> 6;
  if (nkt < 16 || nstrips < 32 || nstrips > 64) return 0;   // (N = 2048 .. 4096; wider products fill the chip with the big-tile kernel's slices)
  // slices: strips x S ~ one workgroup per CU, >= 8 K tiles each, equal slices
  int S = std::max(1, opts().gemm_big_grid / nstrips);
  while (S > 1 && (nkt % S || nkt / S < 8)) --S;
  SkinnyArgs a{nullptr, nullptr, S, nstrips, nkt / S, o.gemm_skinny};
  if (S > 1) {
    const Scratch sc = ctx().scratch_of(stream);
    const size_t need = (size_t)S * nstrips * 8 * 8 * 64 * 16;
    if (!sc.p || !sc.cnt || sc.bytes < need || nstrips > kScratchCounters) return 0;
    a.slabs = reinterpret_cast<float*>(sc.p);
    a.tickets = sc.cnt;
  }
  hipLaunchKernelGGL(gemm_skinny_kernel, dim3(nstrips * S), dim3(512), SK_LDS, stream, d, a);
  return launch_status() == U2_OK ? 1 : U2_ERR_LAUNCH;
}

}  // namespace u2
