// Status strings / version of the C ABI (include/drtk_amd.h).
#include "common.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

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
// The rasterizer's depth-order setting (include/drtk_amd.h): -1 = not read yet.
static std::atomic<int> g_depth_order{-1};
int depth_order_setting() {
  int o = g_depth_order.load(std::memory_order_relaxed);
  if (o < 0) {
    const char* e = std::getenv("DRTK_AMD_DEPTH_ORDER");
    const int from_env = (e && std::strcmp(e, "fastmath") == 0) ? DRTK_DEPTH_ORDER_FASTMATH : DRTK_DEPTH_ORDER_STRICT;
    int expected = -1;
    g_depth_order.compare_exchange_strong(expected, from_env, std::memory_order_relaxed);
    o = g_depth_order.load(std::memory_order_relaxed);
  }
  return o;
}
} // namespace drtk_amd

extern "C" int drtk_amd_set_depth_order(int order) {
  if (order != DRTK_DEPTH_ORDER_STRICT && order != DRTK_DEPTH_ORDER_FASTMATH) return DRTK_ERR_INVALID_ARGUMENT;
  drtk_amd::g_depth_order.store(order, std::memory_order_relaxed);
  return DRTK_OK;
}
extern "C" int drtk_amd_get_depth_order(void) {
  return drtk_amd::depth_order_setting();
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
  DRTK_LAUNCH(fill_bytes_kernel, dim3(blocks), dim3(kBlock), 0, stream, static_cast<unsigned char*>(p), head, n16, tail,
                     b * 0x01010101u);
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

// ---- per-kernel timing (benchmarks) ---------------------------------------------------------------------------
std::atomic<int> g_kernel_timing_on{0};
namespace {
struct TimingRec {
  const char* name; // string literal of the launch site
  hipEvent_t e0, e1;
};
std::mutex g_timing_mu;
std::vector<TimingRec> g_timing;
unsigned g_timing_generation = 0; // bumped by every begin / report: a mark that belongs to an older collection is ignored
struct OpenMark {
  unsigned generation;
  int index;
};
thread_local OpenMark t_timing_open = {0, -1}; // this thread's record between its two marks
void destroy_records(std::vector<TimingRec>& recs) {
  for (auto& r : recs) {
    if (r.e0) (void)hipEventDestroy(r.e0);
    if (r.e1) (void)hipEventDestroy(r.e1);
  }
  recs.clear();
}
} // namespace

void kernel_timing_mark(const char* name, hipStream_t stream, bool begin) {
  std::lock_guard<std::mutex> lock(g_timing_mu);
  if (begin) {
    t_timing_open.index = -1;
    if (g_kernel_timing_on.load(std::memory_order_relaxed) == 0) return; // the collection was closed after the launch site sampled the flag
    TimingRec r{name, nullptr, nullptr};
    if (hipEventCreate(&r.e0) != hipSuccess) return;
    if (hipEventCreate(&r.e1) != hipSuccess) {
      (void)hipEventDestroy(r.e0);
      return;
    }
    (void)hipEventRecord(r.e0, stream);
    g_timing.push_back(r);
    t_timing_open = {g_timing_generation, static_cast<int>(g_timing.size()) - 1};
  } else if (t_timing_open.index >= 0) {
    // only into the record this thread opened, and only while that collection is still the current one
    if (t_timing_open.generation == g_timing_generation && t_timing_open.index < static_cast<int>(g_timing.size()))
      (void)hipEventRecord(g_timing[t_timing_open.index].e1, stream);
    t_timing_open.index = -1;
  }
}

#ifdef DRTK_AMD_ABLATION
static int g_debug_flags = 0;
int debug_flags() {
  return g_debug_flags;
}
#endif
} // namespace drtk_amd

extern "C" int drtk_amd_kernel_timing_begin(void) {
  std::vector<drtk_amd::TimingRec> stale;
  {
    std::lock_guard<std::mutex> lock(drtk_amd::g_timing_mu);
    stale.swap(drtk_amd::g_timing);
    ++drtk_amd::g_timing_generation;
    drtk_amd::g_kernel_timing_on.store(1);
  }
  drtk_amd::destroy_records(stale);
  return DRTK_OK;
}

extern "C" int drtk_amd_kernel_timing_report(char* buf, size_t capacity, size_t* needed) {
  using namespace drtk_amd;
  // close the collection and take its records out under the lock; the events are synchronised OUTSIDE it, so launches
  // of other threads (timed or not) never wait behind this call's hipEventSynchronize
  std::vector<TimingRec> recs;
  {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    g_kernel_timing_on.store(0);
    ++g_timing_generation;
    recs.swap(g_timing);
  }
  // aggregate by launch site, in order of first appearance
  std::vector<const char*> names;
  std::vector<double> total;
  std::vector<long> count;
  int status = DRTK_OK;
  for (auto& r : recs) {
    float ms = 0.f;
    // (a record whose second mark was dropped -- its launch straddled the end of the collection -- has no end time)
    if (hipEventQuery(r.e1) == hipErrorInvalidResourceHandle || hipEventSynchronize(r.e1) != hipSuccess ||
        hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) {
      status = DRTK_ERR_LAUNCH;
      (void)hipGetLastError();
      continue;
    }
    size_t k = 0;
    while (k < names.size() && names[k] != r.name && std::strcmp(names[k], r.name) != 0) ++k;
    if (k == names.size()) {
      names.push_back(r.name);
      total.push_back(0.0);
      count.push_back(0);
    }
    total[k] += ms;
    count[k] += 1;
  }
  std::string out;
  char line[512];
  for (size_t k = 0; k < names.size(); ++k) {
    std::snprintf(line, sizeof(line), "%s\t%ld\t%.6f\n", names[k], count[k], total[k]);
    out += line;
  }
  destroy_records(recs);
  if (needed) *needed = out.size() + 1;
  if (buf && capacity > 0) {
    const size_t n = out.size() < capacity - 1 ? out.size() : capacity - 1;
    std::memcpy(buf, out.data(), n);
    buf[n] = 0;
    if (n < out.size() && status == DRTK_OK) status = DRTK_ERR_WORKSPACE_TOO_SMALL;
  }
  return status;
}

#ifdef DRTK_AMD_ABLATION
// profiling build only (profiles/libdrtk_amd_ablate.so); not declared in include/drtk_amd.h, not in libdrtk_amd.so
extern "C" __attribute__((visibility("default"))) void drtk_amd_debug_set_flags(int flags) {
  drtk_amd::g_debug_flags = flags;
}
#endif
