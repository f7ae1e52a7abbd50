// Probe: what ds_read_b64_tr_b16 returns.  LDS is filled with lds16[i] = i (16-bit elements); every lane issues the read
// with a per-lane byte address from one of several address patterns; the four 16-bit results of every lane are printed as
// LDS element indices.  build: hipcc --offload-arch=gfx950 -O3 tr_read.hip -o tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned pattern_addr(int pattern, int l) {
  switch (pattern) {
    case 0: return l * 8;                                    // lane-linear 8-byte pieces
    case 1: return (l & 15) * 256 + (l >> 4) * 8;            // 16 rows of 256 B per 16-lane group, group g at piece g
    case 2: return (l & 15) * 128 + (l >> 4) * 8;            // rows of 128 B (a [k][64] bf16 tile), 16 rows per group
    case 3: return (l & 3) * 8 + ((l >> 2) & 3) * 256 + (l >> 4) * 32;  // 4 rows x 4 pieces per group
    case 4: return 0;                                        // uniform
    default: return (l & 3) * 256 + ((l >> 2) & 3) * 8 + (l >> 4) * 32;  // 4 pieces x 4 rows per group
  }
}

__global__ void k(unsigned short* out, int pattern) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  // the LDS base goes through the asm operand: an array whose address never escapes may lose its stores
  const unsigned addr = (unsigned)(uintptr_t)(&lds[0]) + pattern_addr(pattern, l);
  unsigned long long r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(r >> (16 * j));
}

int main() {
  unsigned short* d;
  hipMalloc(&d, 64 * 4 * sizeof(unsigned short));
  std::vector<unsigned short> h(256);
  for (int p = 0; p < 6; ++p) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, p);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("pattern %d (lane: byte address -> the 4 elements it received, as LDS 16-bit element indices)\n", p);
    for (int l = 0; l < 64; ++l) {
      unsigned a = 0;
      switch (p) {
        case 0: a = l * 8; break;
        case 1: a = (l & 15) * 256 + (l >> 4) * 8; break;
        case 2: a = (l & 15) * 128 + (l >> 4) * 8; break;
        case 3: a = (l & 3) * 8 + ((l >> 2) & 3) * 256 + (l >> 4) * 32; break;
        case 4: a = 0; break;
        default: a = (l & 3) * 256 + ((l >> 2) & 3) * 8 + (l >> 4) * 32; break;
      }
      printf("  l%02d @%5u(el %4u): %4u %4u %4u %4u%s", l, a, a / 2, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3],
             (l & 3) == 3 ? "\n" : " |");
    }
  }
  return 0;
}
