// api_misc.hip — version / status strings of the C ABI.
#include "common.h"

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
}
