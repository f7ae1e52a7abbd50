// Shared device/host helpers for the u2tok HIP library (gfx950 / CDNA4 only).
// Everything here is internal; the public surface is include/u2tok.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace u2 {

typedef unsigned short bf16_t;  // raw bfloat16 bits in HBM
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator fragment
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator fragment
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;

// Status codes returned through the C-ABI (0 = ok, negative = error).
enum : int {
  U2_OK = 0,
  U2_ERR_ARG = -1,       // bad dimension / null pointer / unsupported combination
  U2_ERR_LAUNCH = -2,    // hipGetLastError() after a launch was not hipSuccess
  U2_ERR_WORKSPACE = -3, // caller-provided workspace too small
  U2_ERR_DEVICE = -4,    // not a gfx950 device
};

// ---- the ELEMENT TYPE of the build.  Every kernel of this library stores activations and parameters as 16-bit elements and
// accumulates in fp32; which 16-bit format is a property of the build: bfloat16 (libu2tok_hip.so, the default) or IEEE half
// (libu2tok_hip_f16.so, compiled from the same sources with -DU2_ELEM_F16 -- evalscipt/ourmodel_amos.py:33,70 loads fp16 weights).
// The two differ in exactly these places: the conversions below, the MFMA opcode (same shapes, same rate, same fragment
// layouts on gfx950), the conversion / MFMA / dot2 opcodes inside the generated asm loops (U2_ELEM_ASM), and the flash loop's
// stale-max bound (U2_FLASH_THR_BITS: half has 5 exponent bits).  Names keep their historical "bf16": bf16_t = "the 16-bit
// element", pack2_bf16 = "round two floats to elements", ...
#if defined(U2_ELEM_F16)
#define U2_ELEM_ASM "f16"
#define U2_ELEM_IS_F16 1
typedef __attribute__((ext_vector_type(2))) _Float16 elem_x2_hw;
typedef __attribute__((ext_vector_type(8))) _Float16 elem_x8_hw;
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// Round-to-nearest-even (v_cvt_f16_f32); values beyond 65504 become +-inf, as in torch.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, elem_x2_hw));  // v_cvt_pk_f16_f32 (RNE)
}
__device__ __forceinline__ float bf16lo(uint32_t u) { return (float)__builtin_bit_cast(elem_x2_hw, u)[0]; }
__device__ __forceinline__ float bf16hi(uint32_t u) { return (float)__builtin_bit_cast(elem_x2_hw, u)[1]; }
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(elem_x8_hw, a), __builtin_bit_cast(elem_x8_hw, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(elem_x8_hw, a), __builtin_bit_cast(elem_x8_hw, b), c, 0, 0, 0);
}
__device__ __forceinline__ float dot2_elem(uint32_t a, uint32_t b, float c) {  // c + a.lo b.lo + a.hi b.hi
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(elem_x2_hw, a), __builtin_bit_cast(elem_x2_hw, b), c, false);
}
#else
#define U2_ELEM_ASM "bf16"
#define U2_ELEM_IS_F16 0
typedef __attribute__((ext_vector_type(2))) __bf16 elem_x2_hw;
__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// Round-to-nearest-even, lowered to v_cvt_pk_bf16_f32 on gfx950.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  f32x2 v = {lo, hi};
  bf16x2_hw b = __builtin_convertvector(v, bf16x2_hw);
  return *reinterpret_cast<uint32_t*>(&b);
}
__device__ __forceinline__ float bf16lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float dot2_elem(uint32_t a, uint32_t b, float c) {  // c + a.lo b.lo + a.hi b.hi
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(elem_x2_hw, a), __builtin_bit_cast(elem_x2_hw, b), c, false);
}
#endif

// VOXEL formats are data formats, not the build's element type: a bfloat16 volume is bfloat16 in either build.
__device__ __forceinline__ float voxel_bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f32_to_voxel_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// erf-GELU for the GEMM epilogues and u2tok_gelu_fwd: GELU(x) = x Phi(x) with the normal CDF as a logistic function of an odd
// polynomial,  Phi(x) = 1 / (1 + 2^(x (c0 + c1 x^2 + c2 x^4 + c3 x^6 + c4 x^8)))  (the c_k carry -log2 e; minimax fit of round 6, weight |x|,
// over [0, 9.5]: |dPhi| <= 6.2e-6, |d GELU| <= 3.3e-6 everywhere; the leading coefficient makes the exponent monotone beyond the fit, so
// 2^t -> 0 / inf and Phi -> 1 / 0 exactly for large |x|).  12 instructions per PAIR of values (4 packed FMAs, 3 packed multiplies, 1 packed
// add, 2 v_exp_f32, 2 v_rcp_f32) against 20 for the Abramowitz & Stegun 7.1.28 form of rounds 1-5 (|d GELU| <= 7.5e-7), and ~32 + two
// divergent branches for ocml's erff: the GELU of the ViT's MLP is 50 M evaluations per layer whose issue time ADDS to the K loop of a
// one-wave-per-SIMD GEMM wherever it is placed (profiles/r06_drain_probe_gelu_placement.log: four placements, 103.0-104.5 us).
// What the cheaper form costs, over ALL bf16 inputs (tests/test_host_modules.py re-derives this table in fp32 arithmetic): of the outputs with
// |y| >= 0.01, 0.09 % differ from the correctly rounded bf16 value (by one ulp; the 7.1.28 form: 0.00 %), 1.0 % of the fp16 outputs (0.27 %).
__device__ __forceinline__ float gelu_fast(float x) {
  const float x2 = x * x;
  float p = -3.2607881621515844e-06f;
  p = __builtin_fmaf(p, x2, 8.898991654859856e-05f);
  p = __builtin_fmaf(p, x2, 0.0003546576772350818f);
  p = __builtin_fmaf(p, x2, -0.10521142929792404f);
  p = __builtin_fmaf(p, x2, -2.3020575046539307f);
  const float e = __builtin_amdgcn_exp2f(p * x);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// Two values per instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: the same IEEE operations per half, so the results are
// those of gelu_fast bit for bit) -- the form the GEMM epilogues use, and the instruction sequence the drain form of the big-tile
// kernel carries in its K loop (tools/gen_gemm_bt_asm.py: gelu_chunk).
typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_fast2(float& x0, float& x1) {
  const gelu_f32x2 x = {x0, x1};
  const gelu_f32x2 x2 = x * x;
  gelu_f32x2 p = {-3.2607881621515844e-06f, -3.2607881621515844e-06f};
  p = __builtin_elementwise_fma(p, x2, (gelu_f32x2){8.898991654859856e-05f, 8.898991654859856e-05f});
  p = __builtin_elementwise_fma(p, x2, (gelu_f32x2){0.0003546576772350818f, 0.0003546576772350818f});
  p = __builtin_elementwise_fma(p, x2, (gelu_f32x2){-0.10521142929792404f, -0.10521142929792404f});
  p = __builtin_elementwise_fma(p, x2, (gelu_f32x2){-2.3020575046539307f, -2.3020575046539307f});
  const gelu_f32x2 t = p * x;
  const gelu_f32x2 d = (gelu_f32x2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
  const gelu_f32x2 r = x * (gelu_f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  x0 = r.x;
  x1 = r.y;
}

// The GELU of the GEMM epilogues (every form: big-tile, drain, small-tile, few-rows, split-K reduce): the pre-activation is ROUNDED TO THE
// ELEMENT TYPE FIRST, then gelu_fast -- "as if the Linear had stored its output and the GELU read it back", which is what the reference
// does (MONAI MLPBlock: nn.Linear returns a bf16 tensor, nn.GELU rounds again; /root/reference/src/model/multimodal_encoder/vit.py:100-105)
// and what this library's training path has always done (u2tok_gelu_fwd on the stored pre-activation).  Round 6: the drain form of the
// big-tile kernel holds a tile's pre-activations as packed elements while the next tile's K loop runs, so the rounding point is a
// property of that form; making it the rule keeps a row's bits independent of which form computed it (tests/test_gpu_path.py:
// chunk-count independence) and makes all forms agree bit for bit on equal accumulators.
__device__ __forceinline__ float gelu_epi(float x) { return gelu_fast(bf16_to_f32(f32_to_bf16(x))); }
__device__ __forceinline__ void gelu_epi2(float& x0, float& x1) {
  x0 = bf16_to_f32(f32_to_bf16(x0));
  x1 = bf16_to_f32(f32_to_bf16(x1));
  gelu_fast2(x0, x1);
}

static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? U2_OK : U2_ERR_LAUNCH;
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }


// 16 bytes per lane from a raw buffer (descriptor + 32-bit lane offset + scalar offset) straight into LDS: `buffer_load_dwordx4 ... lds`.
// Device pass only -- the host pass of hipcc does not know the builtin, and a template kernel that names it unguarded silently loses its
// host stub (undefined __device_stub__ symbol at load time).
__device__ __forceinline__ void lds_dma_mubuf16(const void* base_rsrc_ptr, void* lds_dst, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base_rsrc_ptr), 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
#else
  (void)base_rsrc_ptr; (void)lds_dst; (void)voffset; (void)soffset;
#endif
}
}  // namespace u2
