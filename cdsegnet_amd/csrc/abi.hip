// Library identification for the C ABI (include/cdseg.h).
#include "common.h"

extern "C" {

int cdseg_abi_version(void) { return 1; }

const char* cdseg_build_info(void) {
  return "libcdseg_hip: CDSegNet single-step inference kernels, gfx950 (CDNA4), HIP " __VERSION__;
}

}  // extern "C"
