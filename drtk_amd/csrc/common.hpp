// Shared device/host helpers of the drtk_amd HIP kernels (gfx950 only).
//
// Everything here is compiled with -ffp-contract=off and without fast-math: coverage tests,
// depth ordering and the "was clamped" / pix_in_tri decisions of the reference are exact float
// comparisons (SURVEY.md §7 hard parts 1 and 3b), so what is written is what must be evaluated.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>

#include "drtk_amd.h"

namespace drtk_amd {

constexpr int kWave = 64;   // CDNA4 wavefront
constexpr int kBlock = 256; // 4 waves per workgroup, one per SIMD

template <typename T>
struct Eps;
template <>
struct Eps<float> {
  // cuda_math_helper.h:62-64
  static __host__ __device__ constexpr float value() { return 1e-8f; }
};
template <>
struct Eps<double> {
  // cuda_math_helper.h:67-69
  static __host__ __device__ constexpr double value() { return 1e-16; }
};

// cuda_math_helper.h:1036-1041 : v<0 ? min(v,-eps) : max(v,eps)   (epsclamp(+-0) = +eps)
template <typename T>
__device__ __forceinline__ T epsclamp(T v) {
  const T eps = Eps<T>::value();
  if (v < T(0)) {
    return v < -eps ? v : -eps;
  }
  return v > eps ? v : eps;
}

// Explicit float->int conversion of the bounding box: out-of-range -> INT_MIN, matching the x86
// reference build (cvttss2si); the CPU restatement used by the tests defines the same function.
template <typename T>
__device__ __forceinline__ int32_t trunc_i32(T x) {
  if (!(x > T(-2147483904.0) && x < T(2147483648.0))) return INT32_MIN;
  return static_cast<int32_t>(x);
}

template <typename T>
__device__ __forceinline__ T min3(T a, T b, T c) {
  T m = a;
  if (b < m) m = b;
  if (c < m) m = c;
  return m;
}
template <typename T>
__device__ __forceinline__ T max3(T a, T b, T c) {
  T m = a;
  if (m < b) m = b;
  if (m < c) m = c;
  return m;
}

__device__ __forceinline__ int lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Unordered float/double atomic add on global memory (hardware global_atomic_add_f32 / _f64).
template <typename T>
__device__ __forceinline__ void atomic_add_global(T* p, T v) {
  unsafeAtomicAdd(p, v);
}

// Wave-aggregated "+1" on counters[key] for the lanes with `on` set: lanes of a wave that hit the same
// counter (consecutive triangles of a mesh mostly land in the same few tiles) are merged into ONE
// atomic by their first lane.  With FETCH each lane gets its own slot (old value + rank).
template <bool FETCH>
__device__ __forceinline__ int wave_agg_inc(int32_t* __restrict__ counters, int key, bool on) {
  const int lane = lane_id();
  unsigned long long todo = __ballot(on);
  int pos = 0;
  while (todo) {
    const int leader = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
    const int k = __builtin_amdgcn_readlane(key, leader);
    const bool mine = on && key == k;
    const unsigned long long same = __ballot(mine);
    int base = 0;
    if (lane == leader) {
      const int cnt = __popcll(same);
      if (FETCH) {
        base = atomicAdd(counters + k, cnt);
      } else {
        atomicAdd(counters + k, cnt);
      }
    }
    if (FETCH) {
      base = __builtin_amdgcn_readlane(base, leader);
      if (mine) pos = base + __popcll(same & ((1ull << lane) - 1ull));
    }
    todo &= ~same;
  }
  return pos;
}


// Four consecutive elements as ONE non-temporal 16 / 32-byte load or store (`nt`: data this launch touches once).
// (__builtin_nontemporal_* take clang vector types, not HIP's struct wrappers.)
template <typename T>
using NtQuad = T __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ NtQuad<T> nt_load4(const T* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const NtQuad<T>*>(p));
}
template <typename T>
__device__ __forceinline__ void nt_store4(T* p, T a, T b, T c, T d) {
  __builtin_nontemporal_store(NtQuad<T>{a, b, c, d}, reinterpret_cast<NtQuad<T>*>(p));
}

// XCD-aware workgroup -> tile order.  The dispatcher deals workgroups round-robin to the 8 XCDs (block b
// runs on XCD b % 8), each with its own L2.  If block b simply takes tile b of an image whose channel
// planes are a power of two apart, every XCD only ever sees one residue class of 4 KB pieces, and all C
// planes of that piece alias onto the same slice of its L2 / fabric: measured with
// profiles/micro/plane_remap.hip, 16-plane stores run at 4.1 TB/s and loads at 4.7 TB/s that way, against
// 6.3 / 6.1 TB/s when each XCD owns strips of consecutive tiles (this mapping), with no change to the data
// layout; neighbouring tiles on one XCD also keep the shared vertices' gathers and atomics in one L2.
// `strip` = tiles per strip; bijective on [0, n) for any n and strip >= 1 (the last partial group is split
// evenly, leftovers map to themselves).
__device__ __forceinline__ int xcd_tile(int b, int n, int strip) {
  constexpr int kXcds = 8;
  const int group = kXcds * strip;
  const int base = b / group * group;
  const int r = b - base;
  const int left = n - base;
  if (left >= group) return base + (r % kXcds) * strip + r / kXcds;
  const int s = left / kXcds;
  if (r < s * kXcds) return base + (r % kXcds) * s + r / kXcds;
  return b;
}
// blockIdx.x of a 1-D-per-image grid mapped through xcd_tile.  `strip` comes from xcd_strip() on the host;
// 1 keeps the linear order.
__device__ __forceinline__ int tile_index(int strip) {
  const int b = static_cast<int>(blockIdx.x);
  return strip <= 1 ? b : xcd_tile(b, static_cast<int>(gridDim.x), strip);
}

// Ablation switches (profiles/kernel_bench.py --flags): a bit mask that switches single phases of a kernel off so
// that their time can be attributed.  They exist ONLY in the separate profiling build (-DDRTK_AMD_ABLATION ->
// profiles/libdrtk_amd_ablate.so, drtk_amd/build.py build_ablation()): in the product library the mask is the
// compile-time constant 0, every DRTK_DBG() test folds to `false`, and there is no symbol that could set it.
#ifdef DRTK_AMD_ABLATION
int debug_flags();
#define DRTK_DBG(mask, bit) (((mask) & (bit)) != 0)
#else
constexpr int debug_flags() { return 0; }
#define DRTK_DBG(mask, bit) false
#endif
// Strip length for tile_index(): the tiles of 16 consecutive image rows (128 KB of every float plane at
// W = 2048 -- strips of 64 to 256 KB measured best for every planar kernel, profiles/kernel_bench.py
// --flags).  Diagnostics: bits 10 ... 19 of the debug flags override it (1 = linear order).
inline int xcd_strip(int64_t tiles_per_16_rows) {
  const int forced = (debug_flags() >> 10) & 1023;
  if (forced) return forced;
  return tiles_per_16_rows < 2 ? 2 : static_cast<int>(tiles_per_16_rows);
}

// Compute units of the current device (cached per device id).
int num_compute_units();

// The rasterizer's depth-order setting (drtk_depth_order_t; include/drtk_amd.h: drtk_amd_set_depth_order, DRTK_AMD_DEPTH_ORDER).
int depth_order_setting();

// memset(p, value, bytes) in stream order, as an ordinary KERNEL launch (any alignment, any size; DRTK_OK or
// DRTK_ERR_LAUNCH).  The library does not use hipMemsetAsync: captured into a graph (torch.cuda.graph around a
// training step) its memset node stops doing its job on replays that follow other work on the device -- measured
// on MI355X / ROCm 7.2 with profiles-style probes: replay 1 of rasterize found pointer-like garbage instead of
// zeros in its counter block and scattered bins out of bounds (GPU memory fault), while every kernel node of the
// same graph replayed correctly.  A fill kernel is a kernel node like all the others.
int fill_bytes_async(void* p, int value, size_t bytes, hipStream_t stream);

inline int64_t ceil_div(int64_t a, int64_t b) {
  return (a + b - 1) / b;
}

// The view is blockIdx.y of every image kernel (65 535 at most per launch): an entry point that is handed more views --
// the reference's grid-stride kernels take any N (render_kernel.cu:349-377) -- calls itself on consecutive slices of
// at most kMaxViewsPerLaunch views (views are independent: same results, a few more launches).
constexpr int64_t kMaxViewsPerLaunch = 65535;
inline size_t dtype_size(drtk_dtype_t d) { return d == DRTK_F32 ? 4 : 8; }
inline const void* advance(const void* p, int64_t elements, size_t elem_size) {
  return p ? static_cast<const char*>(p) + elements * int64_t(elem_size) : nullptr;
}
inline void* advance(void* p, int64_t elements, size_t elem_size) {
  return p ? static_cast<char*>(p) + elements * int64_t(elem_size) : nullptr;
}
template <typename P>
inline P* advance_typed(P* p, int64_t elements) {
  return p ? p + elements : nullptr;
}
#define DRTK_FOR_VIEW_SLICES(N, n0, n, CALL)                                        \
  if ((N) > kMaxViewsPerLaunch) {                                                    \
    for (int64_t n0 = 0; n0 < (N); n0 += kMaxViewsPerLaunch) {                       \
      const int64_t n = (N) - n0 < kMaxViewsPerLaunch ? (N) - n0 : kMaxViewsPerLaunch; \
      const int rc_slice_ = (CALL);                                                  \
      if (rc_slice_ != DRTK_OK) return rc_slice_;                                    \
    }                                                                                \
    return DRTK_OK;                                                                  \
  }

// Per-kernel timing for benchmarks (include/drtk_amd.h: drtk_amd_kernel_timing_begin / _report).  Every launch of
// the library goes through DRTK_LAUNCH; while a collection is open the launch is bracketed by two HIP events on its
// own stream (what is computed is unaffected), otherwise the scope object costs one relaxed atomic load.
extern std::atomic<int> g_kernel_timing_on;
void kernel_timing_mark(const char* name, hipStream_t stream, bool begin);
struct KernelTimingScope {
  hipStream_t stream;
  bool on;
  KernelTimingScope(const char* name, hipStream_t s) : stream(s), on(g_kernel_timing_on.load(std::memory_order_relaxed) != 0) {
    if (on) kernel_timing_mark(name, stream, true);
  }
  ~KernelTimingScope() {
    if (on) kernel_timing_mark(nullptr, stream, false);
  }
};
#define DRTK_LAUNCH(kernel, grid, block, shmem, stream, ...)             \
  do {                                                                   \
    ::drtk_amd::KernelTimingScope drtk_timing_scope_(#kernel, stream);   \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__); \
  } while (0)

#define DRTK_RETURN_IF_LAUNCH_FAILED()                      \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return DRTK_ERR_LAUNCH; \
  } while (0)

} // namespace drtk_amd
