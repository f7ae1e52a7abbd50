// Internal declarations of the module-level forwards (pipeline.hip); configs are the public C structs.
#pragma once
#include <algorithm>
#include "../../include/u2tok.h"
#include "kernels.h"

namespace u2 {

typedef u2tok_vit_config VitConfig;
typedef u2tok_spp_config SppConfig;
typedef u2tok_tokenizer_config TokConfig;
typedef u2tok_tokenizer_taps TokTaps;

// dry == true: no launches, pointers may be null, *peak receives the workspace bytes required.
int vit_forward(const VitConfig& c, const void* const* W, const void* volume, bf16_t* out, void* ws, size_t ws_bytes,
                bool dry, size_t* peak, hipStream_t st);
int spp_forward(const SppConfig& c, const void* const* W, const bf16_t* x, bf16_t* out, void* ws, size_t ws_bytes,
                bool dry, size_t* peak, hipStream_t st);
int tokenizer_forward(const TokConfig& c, const void* const* W, const bf16_t* v_token, const bf16_t* t_token,
                      bf16_t* out, int64_t* topk_idx_out, bf16_t* svr_out, const TokTaps* taps, void* ws, size_t ws_bytes,
                      bool dry, size_t* peak, hipStream_t st);

}  // namespace u2
