// Execution contexts of the u2tok HIP library.
//
// Everything the launchers used to keep in process-wide variables lives in a Context: the tuning / diagnostics options,
// the side streams + events of the tokenizer forward (one set per caller stream), the split-K scratch registrations and
// the profiling records.  A host thread works on its CURRENT context (thread-local binding, like a HIP device):
// u2tok_ctx_set_current(ctx), or the process default context when none is bound.  Two models, or two host threads, each
// with their own context never share mutable state; one context shared by several threads is safe for launches (its
// tables are mutex-protected) as long as nobody changes its options concurrently.
#pragma once
#include <mutex>
#include <vector>
#include "common.h"

namespace u2 {

struct Options {
  int ln_wide = 1;          // 1: LayerNorm rows of >= 2048 elements take a workgroup per row (layernorm_wide_kernel); 0: a wave per row (A/B)
  int gemm_tile = 0;        // 0 heuristic, 64 / 128 force the small-tile kernel's tile
  int gemm_mubuf = 1;       // 1: the small-tile kernel's LDS-DMA pieces leave as buffer_load ... lds where the operands span < 2 GB; 0: FLAT-encoded global_load_lds (A/B)
  int gemm_splitk = 0;      // -1 never, 0 heuristic, 2..16 force that many K slices where scratch allows
  int gemm_big = 0;         // -1 never, 0 heuristic, 20 / 21 / 22 force the 256x256 / 256x192 / 256x128 (ring) big-tile kernel, 24 / 26 the deep forms, 27 the drain form
  int gemm_big_grid = 256;  // persistent workgroups of the big-tile kernel
  int gemm_big_group_m = 0; // 0 heuristic; 1 .. 64: row tiles per group of the big-tile kernels' tile walk (A/B)
  int gemm_big_gelu = 1;    // 1: GELU products may take the big-tile kernel too (two-stage form); 0: always the 128^2 kernel
  int gemm_big_splitk = 0;  // K slices of a FORCED big-tile launch (gemm_big = 20 / 21): measurements, tests
  int gemm_big_deep = 1;    // 1: unsliced 256 x 192 / 256 x 256 products take the deep form (three LDS stages for B, two for A); 0: two stages (A/B)
  int gemm_big_drain = 1;   // 1: deep 256 x 192 products with >= 2 tiles per workgroup run the drain form (tile i's epilogue under tile i + 1's K loop; GELU products too); 2: the same without the GELU products (whose drain form rounds the pre-activation first); 0: never (A/B)
  int gemm_big_ring = 1;    // 1: products that make one round of 256 x 128 tiles take the ring form (bt_pick_ring); 0: never (A/B)
  int gemm_big_skinny = 1;  // 1: partial-round products may take the big-tile kernel with K slices (bt_pick_sliced); 0: never
  int gemm_skinny = 2;      // 2 (1: the same kernel with FLAT-encoded global_load_lds pieces instead of buffer_load ... lds -- A/B): 64 < M <= 256 rows against N = 2048 .. 4096 columns (the TTA query chain) take the unsplit 64 x 64 x 128 kernel (gemm_skinny.hip); 0: 64 x 64 tiles x 4 K slices + reduce launch (A/B)
  int gemm_tail_fused = 1;  // 1: <= 16 rows behind a multiple of 256 (the ViT's cls rows) are computed inside the big-tile launch; 0: few-rows launch
  int kmajor_b = 1;         // 1: P V / DiffTS aggregation read V / X in place as K-major B operands; 0: transposed copies
  int flash_mode = 0;       // 0 pick, 1 plain 128-row units, 7 double pipeline (generated asm KV loop)
  int flash_q_prescaled = 0;  // the q handed to u2tok_flash_attention_d64 already carries scale * log2 e (what the ViT's q|k|v product leaves)
  int vit_flash = 1;        // 0: unfused ViT attention (debug)
  int vit_vt_epilogue = 1;  // 1: the ViT's q|k|v product writes V^T from its own V tiles (transposed-tile form of the 256 x 192 kernel); 0: transpose launch
  int tok_flash = 1;        // 1: fused attention kernel for the tokenizer's attention cores (tokattn.hip); 0: GEMM chain
  int tok_wide = 2;         // 2 (1: the same kernel with FLAT-encoded global_load_lds pieces instead of buffer_load ... lds -- A/B): head dims 256 / 512 of the fused kernel run the 8-wave form (two waves per SIMD, tok_attn2_kernel); 0: the 4-wave form (A/B)
  int tta_overlap = 1;      // k | v projections of the TTA cross attentions on a side stream
  int profile = 0;          // bracket every launch with hipEvents (u2tok_profile_collect)
};

struct SideStream {  // created lazily, one per caller stream that ever ran a tokenizer forward on this context
  hipStream_t owner = nullptr;
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr;
  hipEvent_t done[16] = {};
};

struct Scratch {
  hipStream_t st;
  void* p;
  size_t bytes;
};

struct ProfRec {
  hipEvent_t a, b;
  int cat;
  double flops, bytes;
};

struct Context {
  Options opt;
  std::mutex mu;
  std::vector<SideStream*> sides;
  std::vector<Scratch> scratch;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;

  SideStream* side_for(hipStream_t owner);  // null if the stream / events cannot be created
  void set_scratch(hipStream_t st, void* p, size_t bytes);
  Scratch scratch_of(hipStream_t st);
  void release();  // destroys the side streams / events (no work may be in flight on them)
  ~Context() { release(); }
};

Context& ctx();                 // the calling thread's current context
void ctx_bind(Context* c);      // null = back to the process default context
Context* ctx_bound();           // null if the thread uses the default context
inline const Options& opts() { return ctx().opt; }

}  // namespace u2
