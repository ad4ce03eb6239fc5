// stubs.hip — entry points not implemented yet return HCIR_ERR_UNSUPPORTED (loud, never a fallback).
#include "common.h"
extern "C" {
size_t hcir_ntxent_workspace_bytes(int64_t, int32_t, int) { return 0; }
int hcir_ntxent_fwd(const void*, const void*, int64_t, int32_t, int, float, float*, float*, void*, size_t, void*) { return HCIR_ERR_UNSUPPORTED; }
int hcir_layernorm_f16(const float*, int64_t, int32_t, int64_t, const float*, const float*, float, void*, int64_t, void*) { return HCIR_ERR_UNSUPPORTED; }
int hcir_gemm_f16(const void*, int64_t, const void*, int64_t, const float*, const float*, int64_t, int32_t, int32_t, int, void*, int64_t, void*) { return HCIR_ERR_UNSUPPORTED; }
int hcir_patch_embed(const float*, int64_t, int32_t, int32_t, int32_t, int32_t, const void*, const float*, const float*, const float*, float, int32_t, float*, void*) { return HCIR_ERR_UNSUPPORTED; }
int hcir_attn_fwd(const void*, int64_t, int32_t, int32_t, int32_t, float, void*, void*) { return HCIR_ERR_UNSUPPORTED; }
int hcir_cls_head(const float*, int64_t, int32_t, int32_t, const float*, const float*, float, int, float*, void*, void*) { return HCIR_ERR_UNSUPPORTED; }
int hcir_patch_mean(const float*, int64_t, int32_t, int32_t, const float*, const float*, float, float*, void*) { return HCIR_ERR_UNSUPPORTED; }
}
