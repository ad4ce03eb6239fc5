// api_misc.hip — version / status strings / build identity of the C ABI.
#include "common.h"

#ifndef HCIR_BUILD_ID
#define HCIR_BUILD_ID "unknown"
#endif
#ifndef HCIR_BUILD_FLAGS
#define HCIR_BUILD_FLAGS ""
#endif

extern "C" {
int hcir_version(void) {
  HCIR_ENTER(); return 100; }

const char* hcir_status_string(int status) {
  switch (status) {
    case HCIR_OK: return "ok";
    case HCIR_ERR_INVALID: return "invalid argument";
    case HCIR_ERR_UNSUPPORTED: return "unsupported configuration";
    case HCIR_ERR_LAUNCH: return "HIP launch error";
    case HCIR_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}

// "<sha256/16 of the sources this binary was compiled from>[ <-D flags>]": tools/build_variant.sh builds (A/B and
// ablation libraries) carry their flags, so a PMC summary can never be attributed to the wrong binary
const char* hcir_build_id(void) { return HCIR_BUILD_ID HCIR_BUILD_FLAGS; }
}
