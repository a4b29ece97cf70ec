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

namespace {
// 16-byte words in a grid-stride loop; the (< 16) bytes before the first and after the last aligned word are
// written by the first threads of the grid.
__global__ __launch_bounds__(kBlock) void fill_bytes_kernel(
    unsigned char* __restrict__ p, size_t head, size_t n16, size_t tail, uint32_t word) {
  const size_t gid = size_t(blockIdx.x) * kBlock + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * kBlock;
  uint4* mid = reinterpret_cast<uint4*>(p + head);
  const uint4 w = make_uint4(word, word, word, word);
  for (size_t i = gid; i < n16; i += stride) mid[i] = w;
  if (gid < head) p[gid] = static_cast<unsigned char>(word);
  if (gid < tail) p[head + 16 * n16 + gid] = static_cast<unsigned char>(word);
}
} // namespace

int fill_bytes_async(void* p, int value, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return DRTK_OK;
  if (!p) return DRTK_ERR_LAUNCH;
  size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;
  if (head > bytes) head = bytes;
  const size_t n16 = (bytes - head) / 16;
  const size_t tail = bytes - head - 16 * n16;
  const uint32_t b = static_cast<uint32_t>(value) & 0xFFu;
  const size_t want = (n16 + kBlock - 1) / kBlock;
  const size_t cap = size_t(num_compute_units()) * 16;
  const unsigned blocks = static_cast<unsigned>(want < 1 ? 1 : (want > cap ? cap : want));
  hipLaunchKernelGGL(fill_bytes_kernel, dim3(blocks), dim3(kBlock), 0, stream, static_cast<unsigned char*>(p), head, n16, tail,
                     b * 0x01010101u);
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

#ifdef DRTK_AMD_ABLATION
static int g_debug_flags = 0;
int debug_flags() {
  return g_debug_flags;
}
#endif
} // namespace drtk_amd

#ifdef DRTK_AMD_ABLATION
// profiling build only (profiles/libdrtk_amd_ablate.so); not declared in include/drtk_amd.h, not in libdrtk_amd.so
extern "C" void drtk_amd_debug_set_flags(int flags) {
  drtk_amd::g_debug_flags = flags;
}
#endif
