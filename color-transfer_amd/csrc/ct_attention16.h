// ct_attention16.h -- launchers of the two-piece fp16 streaming attention (attention16.hip), called by the C entry points of
// gmflow.hip (ct_attention_tokens_f32, ct_attention_rows64_f32, ct_attention_colsum64_f32) unless CT_HIP_ATT16=0
#pragma once
#include <hip/hip_runtime.h>

namespace ct {

bool attention16_enabled();
void attention16_tokens128(const float *q, const float *k, const float *v, const int *region, const int *rowmap, float *out, int batch,
                           int len, int cv, float scale, int nsplit, float *ws, long long kv_shift, long long kv_total, hipStream_t s);
void attention16_rows64(const float *q, const float *k, const float *v, float *out, float *stats, int batch, int len, float scale,
                        hipStream_t s);
void attention16_colsum64(const float *q, const float *k, const float *stats, float *colsum, int batch, int len, float scale, hipStream_t s);

}  // namespace ct
