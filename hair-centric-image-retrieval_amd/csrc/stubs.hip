// stubs.hip — entry points not implemented yet return HCIR_ERR_UNSUPPORTED (loud, never a fallback).
#include "common.h"
extern "C" {
size_t hcir_ntxent_workspace_bytes(int64_t, int32_t, int) { return 0; }
int hcir_ntxent_fwd(const void*, const void*, int64_t, int32_t, int, float, float*, float*, void*, size_t, void*) { return HCIR_ERR_UNSUPPORTED; }
}
