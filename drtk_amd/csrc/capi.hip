// Status strings / version of the C ABI (include/drtk_amd.h).
#include "common.hpp"

extern "C" const char* drtk_amd_status_string(int status) {
  switch (status) {
    case DRTK_OK:
      return "ok";
    case DRTK_ERR_INVALID_ARGUMENT:
      return "invalid argument (size, pointer or dtype)";
    case DRTK_ERR_WORKSPACE_TOO_SMALL:
      return "workspace too small";
    case DRTK_ERR_LAUNCH:
      return "HIP launch failed";
    case DRTK_ERR_UNSUPPORTED:
      return "unsupported mode";
    case DRTK_ERR_TOO_MANY_VERTICES:
      return "expected second dim of v to be less than 268435456";
    default:
      return "unknown status";
  }
}

#define DRTK_STR2(x) #x
#define DRTK_STR(x) DRTK_STR2(x)
extern "C" const char* drtk_amd_version(void) {
  return DRTK_STR(DRTK_AMD_VERSION_MAJOR) "." DRTK_STR(DRTK_AMD_VERSION_MINOR) " (gfx950)";
}

namespace drtk_amd {
int num_compute_units() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

static int g_debug_flags = 0;
int debug_flags() {
  return g_debug_flags;
}
} // namespace drtk_amd

extern "C" void drtk_amd_debug_set_flags(int flags) {
  drtk_amd::g_debug_flags = flags;
}
