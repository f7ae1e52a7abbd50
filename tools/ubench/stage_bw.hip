// Microbenchmark: how fast can one workgroup per CU move GEMM-shaped operand tiles from L2/HBM into LDS?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA)         mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: global_load_dwordx4 -> VGPR only (no LDS)
// Access pattern = gemm_bt.hip's: workgroup b streams K tiles of a [ROWS][BK] bf16 panel pair out of two
// [8192][8192] matrices, 16 bytes per lane, DEPTH K tiles in flight.  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// BG (round 6): what the OTHER four waves do meanwhile -- 0: nothing; 1: ds_read_b128 back to back (the LDS read port saturated);
//   2: the read : matrix-op mix of one k16 step of the deep 256 x 192 GEMM loop (7 ds_read_b128 : 12 v_mfma_f32_32x32x16_bf16), i.e.
//   the load the staging pieces of gemm_bt.hip compete with.  The background waves poll an LDS flag the last issuing wave sets.
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
// ROT (round 6): workgroup b walks the K tiles in the rotated order (kt + rot_b) mod nkt -- workgroups that share a panel are then at
// different K tiles instead of missing the same cache lines together
template <int MODE, int BK, int DEPTH, int NWAVES_ISSUE, int SWZ = 0, int BG = 0, int ROT = 0>
__global__ __launch_bounds__(512) void stage_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                    int K, int lda, int ntile, unsigned* sink, unsigned long long* bgcount) {
  constexpr int ROWS = BK > 64 ? 512 * 64 / BK : 512, HALF = ROWS / 2;   // (BK 128 / 256: fewer rows, the same 64 KB per stage -- round 6)
  constexpr int CPR = BK / 8, RPP = 64 / CPR, STAGE = ROWS * BK * 2, NP = STAGE / 1024;
  constexpr int PW = NP / NWAVES_ISSUE;  // pieces per issuing wave
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
  __shared__ volatile int done_flag;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned acc = 0;
  if (BG != 0) {
    if (threadIdx.x == 0) done_flag = 0;
    __syncthreads();
  }
  if (BG != 0 && wave >= 4) {
    // background: fragment-shaped reads (16 bytes per lane, 1 KB per instruction) walking over both stages
    f32x16_t c[4] = {};
    unsigned long long n = 0;
    const unsigned base = (unsigned)(size_t)(lds + lane * 16) + (wave - 4) * 7168;   // (LDS byte address: the low 32 bits of the pointer)
    bf16x8_t f[2][7];
#define ISSUE7(F, ADDR)                                                                                                               \
  asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:1024\n\tds_read_b128 %2, %7 offset:2048\n\tds_read_b128 %3, %7 offset:3072\n\t" \
               "ds_read_b128 %4, %7 offset:4096\n\tds_read_b128 %5, %7 offset:5120\n\tds_read_b128 %6, %7 offset:6144"                \
               : "=&v"(F[0]), "=&v"(F[1]), "=&v"(F[2]), "=&v"(F[3]), "=&v"(F[4]), "=&v"(F[5]), "=&v"(F[6]) : "v"(ADDR) : "memory")
#define LANDED7(F)                                                                                                                    \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6])::"memory")
    ISSUE7(f[0], base);
    unsigned step = 0;
    while (done_flag == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        step = (step + 28672) % (2 * STAGE - 32768);
        ISSUE7(f[(u + 1) & 1], base + step);            // the next k16 step's fragments are in flight under this step's matrix ops
        if constexpr (BG == 2) {
#pragma unroll
          for (int i = 0; i < 12; ++i) c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[u & 1][i % 4], f[u & 1][4 + i % 3], c[i & 3], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 7; ++i) acc += (unsigned)__builtin_bit_cast(unsigned short, f[u & 1][i][0]);
        }
        LANDED7(f[(u + 1) & 1]);
      }
      n += 28;
    }
    if constexpr (BG == 2) acc += (unsigned)c[0][0] + (unsigned)c[1][1] + (unsigned)c[2][2] + (unsigned)c[3][3];
    if (lane == 0) atomicAdd(bgcount, n);
  } else if (wave < NWAVES_ISSUE) {
    for (int t = 0; t < ntile; ++t) {
      const int tile = blockIdx.x + t * gridDim.x;
      const int bm0 = (tile % 32) * 256, bn0 = (tile / 32 % 32) * 256;
      const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
      int voff[PW];
      [[maybe_unused]] unsigned long long dummy64[4] = {1, 2, 3, 4};
#pragma unroll
      for (int i = 0; i < PW; ++i) {
        const int r = (i * NWAVES_ISSUE + wave) * RPP + lane / CPR;
        voff[i] = ((r < HALF ? (bm0 + r) : (bn0 + r - HALF)) * lda + (lane % CPR) * 8) * 2;
      }
      const unsigned short* src[PW];
#pragma unroll
      for (int i = 0; i < PW; ++i) {
        const int r = (i * NWAVES_ISSUE + wave) * RPP + lane / CPR;
        int gc = lane % CPR;
        if (SWZ == 1) gc ^= (r >> 1) & 7;            // full 16-B chunk swizzle (gemm_bt.hip)
        if (SWZ == 2) gc ^= ((r >> 1) & 3) << 1;     // 32-B pairs stay together
        if (SWZ == 3) gc ^= ((r >> 1) & 1) << 2;     // swap 64-B halves only
        if (SWZ == 4) gc = (gc + (r & 7)) & 7;       // rotation
        src[i] = (r < HALF ? A + (size_t)(bm0 + r) * lda : B + (size_t)(bn0 + r - HALF) * lda) + gc * 8;
      }
      const int nkt = K / BK;
      const int rot = ROT == 1 ? (int)((blockIdx.x / 8) * 37u % (unsigned)nkt) : ROT == 2 ? (int)((blockIdx.x / 32) * (nkt / 8)) : 0;
      for (int kt_ = 0; kt_ < nkt; ++kt_) {
        const int kt = ROT ? (kt_ + rot) % nkt : kt_;
        char* s = lds + (kt_ & 1) * STAGE + wave * 1024;
        if constexpr (MODE == 3 || MODE == 4 || MODE == 5) {
          // MUBUF form: one descriptor per matrix, 32-bit per-lane offsets, the K advance in the scalar offset
#pragma unroll
          for (int i = 0; i < PW; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (MODE == 4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(dummy64[i & 3]) : "s"((unsigned long long)kt));
            if ((i * NWAVES_ISSUE + wave) * RPP < HALF)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(s + i * NWAVES_ISSUE * 1024), 16,
                                                       voff[i], kt * BK * 2, 0, 0);
            else
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(s + i * NWAVES_ISSUE * 1024), 16,
                                                       voff[i], kt * BK * 2, 0, 0);
#endif
          }
          if constexpr (MODE == 5) {
            // the GEMM K loop's shape (round 6): all but the DEPTH youngest pieces of this wave must have landed, then the workgroup's barrier
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH > 63 ? 63 : DEPTH) : "memory");
            __builtin_amdgcn_s_barrier();
          } else if (kt_ >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * DEPTH > 63 ? 63 : PW * DEPTH) : "memory");
        } else if constexpr (MODE == 0) {
#pragma unroll
          for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(s + i * NWAVES_ISSUE * 1024), 16, 0, 0);
          if (kt_ >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * DEPTH > 63 ? 63 : PW * DEPTH) : "memory");
        } else {
          uint4 r[PW];
#pragma unroll
          for (int i = 0; i < PW; ++i) r[i] = *reinterpret_cast<const uint4*>(src[i] + kt * BK);
          if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < PW; ++i) *reinterpret_cast<uint4*>(s + i * NWAVES_ISSUE * 1024 + lane * 16) = r[i];
          } else {
#pragma unroll
            for (int i = 0; i < PW; ++i) acc += r[i].x ^ r[i].w;
          }
        }
      }
    }
  }
  if (BG != 0 && wave < NWAVES_ISSUE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == NWAVES_ISSUE - 1 && lane == 0) done_flag = 1;      // (the last issuing wave: the others are a piece ahead at most)
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int BK, int DEPTH, int NW, int SWZ = 0, int BG = 0, int ROT = 0>
static void run(const char* name, const unsigned short* A, const unsigned short* B, unsigned* sink) {
  const int K = 8192, ntile = 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  static unsigned long long* bgc = nullptr;
  if (!bgc) hipMalloc(&bgc, 8);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stage_kernel<MODE, BK, DEPTH, NW, SWZ, BG, ROT>), dim3(256), dim3(512), 0, 0, A, B, K, 8192, ntile, sink, bgc);
  hipMemset(bgc, 0, 8);
  hipEventRecord(e0);
  const int it = 5;
  for (int w = 0; w < it; ++w) hipLaunchKernelGGL((stage_kernel<MODE, BK, DEPTH, NW, SWZ, BG, ROT>), dim3(256), dim3(512), 0, 0, A, B, K, 8192, ntile, sink, bgc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  const double bytes = 256.0 * ntile * (K / BK) * (BK > 64 ? 512 * 64 / BK : 512) * BK * 2;
  printf("%-44s %8.1f us  %6.2f TB/s  %5.1f B/clk/CU@2.1GHz", name, ms * 1e3, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.1e9));
  if (BG) {
    unsigned long long n = 0;
    hipMemcpy(&n, bgc, 8, hipMemcpyDeviceToHost);
    const double rd = (double)n / it * 1024.0 / 256 / (ms * 1e-3 * 2.1e9);
    printf("   | background: %5.1f B/clk/CU of ds_read_b128%s", rd, BG == 2 ? " + 12 MFMA 32x32x16 per 7 reads" : "");
    if (BG == 2) printf(" = %4.1f %% of the matrix pipe", 100.0 * ((double)n / it / 7 * 12 * 32) / 256 / 4 / (ms * 1e-3 * 2.1e9));
  }
  printf("\n");
}

int main() {
  const size_t n = (size_t)8192 * 8192;
  unsigned short *A, *B; unsigned* sink;
  hipMalloc(&A, n * 2); hipMalloc(&B, n * 2); hipMalloc(&sink, 64);
  std::vector<unsigned short> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (unsigned short)(rand() & 0x3fff) | 0x3c00;
  hipMemcpy(A, h.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), n * 2, hipMemcpyHostToDevice);
  run<0, 32, 2, 8>("glds  BK32 depth2 8 waves", A, B, sink);
  run<0, 32, 4, 8>("glds  BK32 depth4 8 waves", A, B, sink);
  run<0, 64, 2, 8>("glds  BK64 depth2 8 waves", A, B, sink);
  run<0, 128, 2, 8>("glds  BK128 (256-B row pieces) 8 waves", A, B, sink);
  run<0, 128, 2, 4>("glds  BK128 (256-B row pieces) 4 waves", A, B, sink);
  run<0, 256, 2, 8>("glds  BK256 (512-B row pieces) 8 waves", A, B, sink);
  run<0, 256, 2, 4>("glds  BK256 (512-B row pieces) 4 waves", A, B, sink);
  run<0, 512, 2, 4>("glds  BK512 (1-KB row pieces) 4 waves", A, B, sink);
  run<0, 64, 2, 4, 1, 1>("glds  BK64 4 waves swz | 4 waves reading LDS", A, B, sink);
  run<0, 64, 2, 4, 1, 2>("glds  BK64 4 waves swz | 4 waves GEMM mix", A, B, sink);
  run<3, 64, 2, 4, 0, 2>("buffer_load..lds BK64 4 waves | GEMM mix", A, B, sink);
  run<4, 64, 2, 4, 0, 2>("buffer_load..lds + 1 VALU u64 add per piece | GEMM mix", A, B, sink);
  run<0, 64, 2, 4, 0, 2>("glds  BK64 4 waves no swizzle | GEMM mix", A, B, sink);
  run<4, 64, 2, 4, 0, 0>("buffer_load..lds + 1 VALU u64 add per piece", A, B, sink);
  run<3, 64, 2, 4, 0, 0, 1>("buffer_load..lds BK64 4 waves, K order rotated per workgroup", A, B, sink);
  run<3, 64, 2, 4, 0, 0, 2>("buffer_load..lds BK64 4 waves, rotated per A-panel sharer", A, B, sink);
  run<3, 64, 2, 4, 0, 2, 1>("buffer_load..lds BK64 4 waves, rotated | GEMM mix", A, B, sink);
  run<3, 64, 2, 1>("buffer_load..lds BK64 depth2 ONE issuing wave", A, B, sink);
  run<3, 64, 2, 2>("buffer_load..lds BK64 depth2 two issuing waves", A, B, sink);
  run<5, 64, 0, 4>("K-loop shape: 16 pieces / wave, vmcnt(0) + barrier per stage", A, B, sink);
  run<5, 64, 6, 4>("K-loop shape: 16 pieces / wave, vmcnt(6) + barrier", A, B, sink);
  run<5, 64, 12, 4>("K-loop shape: 16 pieces / wave, vmcnt(12) + barrier", A, B, sink);
  run<5, 64, 16, 4>("K-loop shape: 16 pieces / wave, vmcnt(16) + barrier", A, B, sink);
  run<5, 64, 24, 4>("K-loop shape: 16 pieces / wave, vmcnt(24) + barrier", A, B, sink);
  run<5, 64, 32, 4>("K-loop shape: 16 pieces / wave, vmcnt(32) + barrier", A, B, sink);
  run<5, 64, 48, 4>("K-loop shape: 16 pieces / wave, vmcnt(48) + barrier", A, B, sink);
  run<3, 64, 1, 4, 0, 0, 0>("buffer_load..lds BK64 4 waves depth 1", A, B, sink);
  run<3, 64, 1, 4, 0, 0, 1>("buffer_load..lds BK64 4 waves depth 1, rotated", A, B, sink);
  run<0, 64, 1, 8>("glds  BK64 depth1 8 waves", A, B, sink);
  run<0, 64, 2, 4>("glds  BK64 depth2 4 waves", A, B, sink);
  run<3, 64, 2, 8>("buffer_load..lds BK64 depth2 8 waves", A, B, sink);
  run<3, 64, 2, 4>("buffer_load..lds BK64 depth2 4 waves", A, B, sink);
  run<3, 32, 4, 8>("buffer_load..lds BK32 depth4 8 waves", A, B, sink);
  run<0, 64, 2, 8, 1>("glds  BK64 8 waves swz full-xor", A, B, sink);
  run<0, 64, 2, 8, 2>("glds  BK64 8 waves swz 32B-pairs", A, B, sink);
  run<0, 64, 2, 8, 3>("glds  BK64 8 waves swz 64B-halves", A, B, sink);
  run<0, 64, 2, 8, 4>("glds  BK64 8 waves swz rotate", A, B, sink);
  run<0, 64, 2, 4, 1>("glds  BK64 4 waves swz full-xor", A, B, sink);
  run<1, 64, 0, 4, 1>("load+ds_write BK64 4 waves swz full-xor", A, B, sink);
  run<0, 32, 4, 4>("glds  BK32 depth4 4 waves", A, B, sink);
  run<1, 32, 0, 8>("load+ds_write BK32 8 waves", A, B, sink);
  run<1, 64, 0, 8>("load+ds_write BK64 8 waves", A, B, sink);
  run<1, 64, 0, 4>("load+ds_write BK64 4 waves", A, B, sink);
  run<2, 32, 0, 8>("load only BK32 8 waves", A, B, sink);
  run<2, 64, 0, 8>("load only BK64 8 waves", A, B, sink);
  run<2, 64, 0, 4>("load only BK64 4 waves", A, B, sink);
  return 0;
}
