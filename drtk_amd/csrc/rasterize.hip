// rasterize -- tile-binned z-buffer rasterizer for CDNA4 (gfx950).
//
// Reference behaviour: src/rasterize/rasterize_kernel.cu:42-168 (one thread per triangle, 64-bit
// global atomicMin per fragment into a packed [N,H,W] int64 buffer that is memset to 0xFF before
// and unpacked by a second kernel after, :402-415,:484-548).  Same arithmetic, different machine
// mapping:
//
//   1. bin_count   one thread per (view, triangle): cull exactly like the reference, compute the
//                  clamped bounding box, derive the range of screen tiles it touches.  Triangles
//                  touching <= 4 tiles bump per-tile counters; larger ones go to a per-view "big"
//                  list (this bounds bin memory at 4 entries per triangle for any input).
//   2. bin_scan    exclusive scan of the N*tiles counters (one workgroup).
//   3. bin_fill    second pass over the triangles writes their id into each touched tile's list.
//   4. tile_raster persistent workgroups pull work items (a tile, or a sub-rectangle of a heavy tile) from a sharded
//                  queue; the item's packed (depth_bits<<32 | id) z-buffer lives in LDS (64x64x8 B = 32 KiB).  Each
//                  wave screens its share of the tile's list on 16-byte records, compacts the survivors, sets 64 of
//                  them up one per lane (exact arithmetic) and shades them FOUR AT A TIME, one per 16-lane DPP row:
//                  the plane equations reach the row by `row_newbcast`, the row walks the clipped bbox 16 pixels a
//                  pass, resolved with LDS 64-bit atomicMin (ds_min_u64); the finished tile is unpacked, stored and
//                  cleared in one sweep.
//                  Counters in passes 1/3 are bumped with wave-aggregated atomics.
//
// HBM traffic is therefore the 8 B/px of real output plus the bins (~28 B per triangle): no
// memset, no global atomics, no unpack pass (the reference moves >= 24 B/px).
// The min-reduction over (depth bits, id) is order independent, so the result is deterministic
// and bit-identical to the reference regardless of bin order.
#include <algorithm>

#include "common.hpp"
#include "segscatter.hpp" // wave_lds_sync

#ifdef DRTK_AMD_ABLATION
// Phase clocks of the profiling build (this translation unit only: device globals do not link across units without
// -fgpu-rdc): DRTK_PHASE(i) adds the shader clocks since the calling thread's previous mark to g_phase_clocks[i];
// drtk_amd_debug_read_phases() copies them out and zeroes them.  Call from ONE thread per workgroup.
namespace drtk_amd {
__device__ unsigned long long g_phase_clocks[16];
}
extern "C" __attribute__((visibility("default"))) int drtk_amd_debug_read_phases(unsigned long long* out) {
  if (hipDeviceSynchronize() != hipSuccess) return DRTK_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(drtk_amd::g_phase_clocks), 16 * sizeof(unsigned long long)) != hipSuccess) return DRTK_ERR_LAUNCH;
  const unsigned long long zero[16] = {};
  if (hipMemcpyToSymbol(HIP_SYMBOL(drtk_amd::g_phase_clocks), zero, sizeof(zero)) != hipSuccess) return DRTK_ERR_LAUNCH;
  return DRTK_OK;
}
#define DRTK_PHASE_INIT() unsigned long long drtk_phase_t_ = __builtin_readcyclecounter()
#define DRTK_PHASE(i)                                                     \
  do {                                                                    \
    if (DRTK_DBG(dbg, 256)) { /* only when asked for: the counter updates themselves slow the kernel 2x */ \
      const unsigned long long now_ = __builtin_readcyclecounter();       \
      atomicAdd(&::drtk_amd::g_phase_clocks[i], now_ - drtk_phase_t_);    \
      drtk_phase_t_ = __builtin_readcyclecounter();                       \
    }                                                                     \
  } while (0)
#else
#define DRTK_PHASE_INIT() do { } while (0)
#define DRTK_PHASE(i) do { } while (0)
#endif

namespace drtk_amd {
namespace {

constexpr uint32_t kCulled = 0xFFFFFFFFu;
constexpr uint32_t kFacingBit = 0x80000000u; // in tri_range.y: triangle has positive orientation (den > 0)
constexpr int kMaxSmallTiles = 4; // bins hold at most this many entries per triangle

// Exact per-triangle setup (rasterize_kernel.cu:73-113,133-141).
template <typename T>
struct TriSetup {
  T p0x, p0y, p1x, p1y, p2x, p2y;
  T dinv0, dinv1, dinv2;
  T sign_denom, abs_denom, rdenom; // rdenom = RN(1 / abs_denom)
  int bb_min_x, bb_min_y, bb_max_x, bb_max_y;
  bool c0, c1, c2;    // canonical edge orientation flags (vi_a <= vi_b)
  bool tl0, tl1, tl2; // top-left classification
  uint32_t z_lo_bits; // f32 bits of a rigorous lower bound of every depth this triangle can produce (0: none)
};

// Correctly rounded n / d for 0 <= n, d > 0 from a precomputed r = RN(1/d) (IEEE division, once per
// triangle): q0 = RN(n r); two Markstein corrections q <- fma(fma(-d, q, n), r, q).  After the first
// correction q is a faithful rounding of n/d, and with r the correctly rounded reciprocal the second
// one yields RN(n/d) exactly (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic,
// Newton-Raphson division, final-rounding theorem) as long as nothing under/overflows -- which the
// range guard ensures; outside it the IEEE division is used.  5 full-rate instructions instead of
// the 11-instruction / 5 quarter-rate IEEE expansion, three times per fragment: the divisions were
// ~60 % of the raster loop's issue cycles.  drtk_amd_selftest_exact_div() compares it with `/` on
// the device over arbitrary many random operand pairs.
template <typename T>
struct DivRange;
template <>
struct DivRange<float> {
  static __device__ constexpr float lo() { return 0x1p-60f; }
  static __device__ constexpr float hi() { return 0x1p60f; }
  static __device__ constexpr float tiny() { return 0x1p-100f; }
};
template <>
struct DivRange<double> {
  static __device__ constexpr double lo() { return 0x1p-500; }
  static __device__ constexpr double hi() { return 0x1p500; }
  static __device__ constexpr double tiny() { return 0x1p-900; }
};
__device__ __forceinline__ float fma_t(float a, float b, float c) {
  return __builtin_fmaf(a, b, c);
}
__device__ __forceinline__ double fma_t(double a, double b, double c) {
  return __builtin_fma(a, b, c);
}
template <typename T>
__device__ __forceinline__ bool exact_div_ok(T d) {
  return d >= DivRange<T>::lo() && d <= DivRange<T>::hi();
}
template <typename T>
__device__ __forceinline__ T exact_div(T n, T d, T r, bool range_ok) {
  const T q0 = n * r;
  if (range_ok && (q0 >= DivRange<T>::tiny() || n == T(0))) {
    const T q1 = fma_t(fma_t(-d, q0, n), r, q0);
    return fma_t(fma_t(-d, q1, n), r, q1);
  }
  return n / d;
}

// The two Markstein corrections of exact_div on their own (valid under exact_div's guard, which the caller tests).
template <typename T>
__device__ __forceinline__ T markstein2(T n, T d, T r, T q0) {
  const T q1 = fma_t(fma_t(-d, q0, n), r, q0);
  return fma_t(fma_t(-d, q1, n), r, q1);
}

// Correctly rounded 1 / x (float, x normal, result normal): hardware v_rcp_f32 (<= 1 ulp) plus one
// Markstein correction y' = fma(y, fma(-x, y, 1), y).  The correction is exact-rounding for every
// faithful y except when x's significand is all ones (classical exception of the reciprocal
// theorem), which -- like operands outside the safe exponent range -- takes the IEEE division.
// Reciprocals depend on the significand only, so drtk_amd_selftest_exact_div() checks this routine
// EXHAUSTIVELY over all 2^23 significands.
__device__ __forceinline__ float exact_rcp(float x) {
  const uint32_t bits = __float_as_uint(x);
  const float ax = __uint_as_float(bits & 0x7FFFFFFFu);
  if ((bits & 0x7FFFFFu) != 0x7FFFFFu && ax >= 0x1p-100f && ax <= 0x1p100f) {
    const float y = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(y, __builtin_fmaf(-x, y, 1.0f), y);
  }
  return 1.0f / x;
}
__device__ __forceinline__ double exact_rcp(double x) {
  return 1.0 / x;
}
// exact_rcp's guard and its fast form, separately (the raster loop tests the guard once per pass for the whole wave)
__device__ __forceinline__ bool fast_rcp_ok(float x) {
  const uint32_t bits = __float_as_uint(x);
  const float ax = __uint_as_float(bits & 0x7FFFFFFFu);
  return ((bits & 0x7FFFFFu) != 0x7FFFFFu) & (ax >= 0x1p-100f) & (ax <= 0x1p100f);
}
__device__ __forceinline__ float fast_rcp(float x) {
  const float y = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(y, __builtin_fmaf(-x, y, 1.0f), y);
}
__device__ __forceinline__ bool fast_rcp_ok(double) {
  return true;
}
// fast_rcp_ok for an argument that went through epsclamp (|x| >= 1e-8 > 2^-100, NaN replaced by eps): only the upper
// end of the range and the all-ones significand remain to be tested
__device__ __forceinline__ bool fast_rcp_ok_clamped(float x) {
  const uint32_t bits = __float_as_uint(x);
  return ((bits & 0x7FFFFFu) != 0x7FFFFFu) & (__uint_as_float(bits & 0x7FFFFFFFu) <= 0x1p100f);
}
__device__ __forceinline__ bool fast_rcp_ok_clamped(double) {
  return true;
}
// +0, or the smallest positive (subnormal) number of T
template <typename T>
__device__ __forceinline__ T smallest_positive_if(bool on);
template <>
__device__ __forceinline__ float smallest_positive_if<float>(bool on) {
  return __uint_as_float(on ? 1u : 0u);
}
template <>
__device__ __forceinline__ double smallest_positive_if<double>(bool on) {
  return __longlong_as_double(on ? 1ll : 0ll);
}
__device__ __forceinline__ double fast_rcp(double x) {
  return 1.0 / x;
}

// Hierarchical-z bound.  Exactly, depth = 1 / sum_k(b_k / z_k) with b_k >= 0, sum b_k = 1, so depth >= min z.
// As evaluated (:148-153) the b_k are rounded quotients of rounded edge functions and need not sum
// to one: each edge function and the denominator carry an absolute error <= 8 eps L^2 (L = extent of
// the triangle in pixels, +2 for the pixel itself), hence sum b_k <= 1 + 40 eps L^2 / |den|, and the
// three products, two additions, the reciprocal and the cast to float add < 1e-6 relative.
// Returns the f32 bits of a rigorous lower bound of every depth the triangle can produce, 0 if there is none.
template <typename T>
__device__ __forceinline__ uint32_t z_lower_bound_bits(T min_x, T max_x, T min_y, T max_y, T abs_denom, T zmin) {
  const T ext = (max_x - min_x > max_y - min_y ? max_x - min_x : max_y - min_y) + T(2);
  const T eps = sizeof(T) == 4 ? T(5.97e-8) : T(1.12e-16);
  const T delta = T(40) * eps * ext * ext / abs_denom + T(2e-6);
  if (!(delta < T(0.25))) return 0u;
  // ... and depth = 1 / epsclamp(depth_inverse) never exceeds 1 / eps (1e8 in f32, 1e16 in f64): geometry
  // beyond that comes out AT 1 / eps, below its own min z, so the bound is the smaller of the two.
  const T cap = (T(1) / Eps<T>::value()) * (T(1) - T(2e-6));
  const T zb = zmin * (T(1) - delta);
  const float lo = static_cast<float>(zb < cap ? zb : cap);
  const uint32_t bits = __float_as_uint(lo);
  return (lo > 0.0f && bits > 8u) ? bits - 8u : 0u; // 8 ulps below, against the rounding of `lo` itself
}

// Culling + bounding box only (used by the binning passes).  Returns false if the triangle is
// dropped by the reference (:81 degenerate indices, :96 near plane, :97-98 off canvas, :107 zero
// area).  On success the pixel bbox (already clamped to the canvas) is returned.
template <typename T>
__device__ __forceinline__ bool tri_bbox(
    const T* __restrict__ v_n, const int32_t* __restrict__ vi_face, int H, int W, int& bx0,
    int& by0, int& bx1, int& by1, bool& positive, float& z_mean, uint32_t& z_lo_bits) {
  const int32_t vi_0 = static_cast<int32_t>(static_cast<uint32_t>(vi_face[0]) & 0x0FFFFFFFu);
  const int32_t vi_1 = vi_face[1];
  const int32_t vi_2 = vi_face[2];
  if (vi_0 == vi_1 && vi_1 == vi_2) return false;
  const T p0x = v_n[3 * (int64_t)vi_0 + 0], p0y = v_n[3 * (int64_t)vi_0 + 1], p0z = v_n[3 * (int64_t)vi_0 + 2];
  const T p1x = v_n[3 * (int64_t)vi_1 + 0], p1y = v_n[3 * (int64_t)vi_1 + 1], p1z = v_n[3 * (int64_t)vi_1 + 2];
  const T p2x = v_n[3 * (int64_t)vi_2 + 0], p2y = v_n[3 * (int64_t)vi_2 + 1], p2z = v_n[3 * (int64_t)vi_2 + 2];
  if (!(p0z > T(1e-8f) && p1z > T(1e-8f) && p2z > T(1e-8f))) return false;
  const T min_x = min3(p0x, p1x, p2x), min_y = min3(p0y, p1y, p2y);
  const T max_x = max3(p0x, p1x, p2x), max_y = max3(p0y, p1y, p2y);
  if (!(min_x <= T(W - 1) && min_y <= T(H - 1) && max_x > T(0) && max_y > T(0))) return false;
  const T v01x = p1x - p0x, v01y = p1y - p0y, v02x = p2x - p0x, v02y = p2y - p0y;
  const T den = v01x * v02y - v01y * v02x;
  if (den == T(0)) return false;
  positive = den > T(0);
  z_lo_bits = z_lower_bound_bits<T>(min_x, max_x, min_y, max_y, den > T(0) ? den : -den, min3(p0z, p1z, p2z));
  z_mean = static_cast<float>((p0z + p1z + p2z) * T(1.0 / 3.0));
  bx0 = max(0, trunc_i32(min_x));
  by0 = max(0, trunc_i32(min_y));
  bx1 = min(W - 1, static_cast<int32_t>(static_cast<uint32_t>(trunc_i32(max_x)) + 1u));
  by1 = min(H - 1, static_cast<int32_t>(static_cast<uint32_t>(trunc_i32(max_y)) + 1u));
  return bx0 <= bx1 && by0 <= by1;
}

template <typename T>
__device__ __forceinline__ bool tri_setup(
    const T* __restrict__ v_n, const int32_t* __restrict__ vi_face, int H, int W, TriSetup<T>& s) {
  const int32_t vi_0 = static_cast<int32_t>(static_cast<uint32_t>(vi_face[0]) & 0x0FFFFFFFu);
  const int32_t vi_1 = vi_face[1];
  const int32_t vi_2 = vi_face[2];
  if (vi_0 == vi_1 && vi_1 == vi_2) return false;
  s.p0x = v_n[3 * (int64_t)vi_0 + 0];
  s.p0y = v_n[3 * (int64_t)vi_0 + 1];
  s.p1x = v_n[3 * (int64_t)vi_1 + 0];
  s.p1y = v_n[3 * (int64_t)vi_1 + 1];
  s.p2x = v_n[3 * (int64_t)vi_2 + 0];
  s.p2y = v_n[3 * (int64_t)vi_2 + 1];
  const T p0z = v_n[3 * (int64_t)vi_0 + 2];
  const T p1z = v_n[3 * (int64_t)vi_1 + 2];
  const T p2z = v_n[3 * (int64_t)vi_2 + 2];
  if (!(p0z > T(1e-8f) && p1z > T(1e-8f) && p2z > T(1e-8f))) return false;
  const T min_x = min3(s.p0x, s.p1x, s.p2x), min_y = min3(s.p0y, s.p1y, s.p2y);
  const T max_x = max3(s.p0x, s.p1x, s.p2x), max_y = max3(s.p0y, s.p1y, s.p2y);
  if (!(min_x <= T(W - 1) && min_y <= T(H - 1) && max_x > T(0) && max_y > T(0))) return false;
  const T v01x = s.p1x - s.p0x, v01y = s.p1y - s.p0y;
  const T v02x = s.p2x - s.p0x, v02y = s.p2y - s.p0y;
  const T v12x = s.p2x - s.p1x, v12y = s.p2y - s.p1y;
  const T den = v01x * v02y - v01y * v02x;
  if (den == T(0)) return false;
  s.sign_denom = den > T(0) ? T(1) : T(-1);
  s.abs_denom = den > T(0) ? den : -den;
  s.rdenom = T(1) / s.abs_denom;
  s.bb_min_x = max(0, trunc_i32(min_x));
  s.bb_min_y = max(0, trunc_i32(min_y));
  s.bb_max_x = min(W - 1, static_cast<int32_t>(static_cast<uint32_t>(trunc_i32(max_x)) + 1u));
  s.bb_max_y = min(H - 1, static_cast<int32_t>(static_cast<uint32_t>(trunc_i32(max_y)) + 1u));
  s.dinv0 = T(1) / epsclamp(p0z);
  s.dinv1 = T(1) / epsclamp(p1z);
  s.dinv2 = T(1) / epsclamp(p2z);
  s.c0 = vi_1 <= vi_2;
  s.c1 = vi_2 <= vi_0;
  s.c2 = vi_0 <= vi_1;
  const bool pos = den > T(0);
  s.tl0 = pos ? (v12y < T(0) || (v12y == T(0) && v12x > T(0))) : (v12y > T(0) || (v12y == T(0) && v12x < T(0)));
  s.tl1 = pos ? (v02y > T(0) || (v02y == T(0) && v02x < T(0))) : (v02y < T(0) || (v02y == T(0) && v02x > T(0)));
  s.tl2 = pos ? (v01y < T(0) || (v01y == T(0) && v01x > T(0))) : (v01y > T(0) || (v01y == T(0) && v01x < T(0)));
  s.z_lo_bits = 0; // the bound comes from the binning pass's record (tri_pre), see tile_raster_kernel
  return true;
}

// rasterize_kernel.cu:19-40
template <typename T>
__device__ __forceinline__ T edge_fn(T ax, T ay, T bx, T by, T px, T py) {
  return (py - ay) * (bx - ax) - (px - ax) * (by - ay);
}
template <typename T>
__device__ __forceinline__ T canon_edge(bool a_le_b, T ax, T ay, T bx, T by, T px, T py) {
  return a_le_b ? edge_fn(ax, ay, bx, by, px, py) : -edge_fn(bx, by, ax, ay, px, py);
}

// Coverage + depth of one pixel centre (rasterize_kernel.cu:117-156).  Returns true and the f32
// depth bits if the fragment is kept.
template <typename T>
__device__ __forceinline__ bool fragment(const TriSetup<T>& s, int x, int y, uint32_t& depth_bits) {
  const T px = static_cast<T>(x), py = static_cast<T>(y);
  T b0 = canon_edge(s.c0, s.p1x, s.p1y, s.p2x, s.p2y, px, py) * s.sign_denom;
  T b1 = canon_edge(s.c1, s.p2x, s.p2y, s.p0x, s.p0y, px, py) * s.sign_denom;
  T b2 = canon_edge(s.c2, s.p0x, s.p0y, s.p1x, s.p1y, px, py) * s.sign_denom;
  if (!((b0 >= T(0)) && (b1 >= T(0)) && (b2 >= T(0)))) return false;
  if ((!s.tl0 && b0 == T(0)) || (!s.tl1 && b1 == T(0)) || (!s.tl2 && b2 == T(0))) return false;
  b0 /= s.abs_denom;
  b1 /= s.abs_denom;
  b2 /= s.abs_denom;
  const T depth_inverse = s.dinv0 * b0 + s.dinv1 * b1 + s.dinv2 * b2;
  const float depth = static_cast<float>(T(1) / epsclamp(depth_inverse));
  depth_bits = __float_as_uint(depth);
  return true;
}

// The work queue is sharded: a single head word serves ~88 pops per microsecond (MI355X_MICROARCH.md, "dequeue"), and
// the ~9000 items of the bench shape, popped by 1024 workgroups, sat on it for most of the ~0.15 ms the pass takes
// with no triangle work at all.  Item i belongs to shard i % 8; a workgroup starts on shard blockIdx % 8 (its XCD, for
// what that is worth) and moves on to the next one when its shard runs dry.
#ifndef DRTK_RASTER_QUEUE_SHARDS
#define DRTK_RASTER_QUEUE_SHARDS 8
#endif
constexpr int kQueueShards = DRTK_RASTER_QUEUE_SHARDS;
constexpr int kQueueStride = 32; // int32 words between the heads (128 B: one head per memory-side line)
constexpr uint32_t kItemEmpty = 0x80000000u; // work item flag: nothing was binned to this tile

struct BinLayout {
  int tile_shift; // log2(tile size in pixels)
  int tiles_x, tiles_y;
  int64_t tiles_per_view, num_tiles; // per view / total
  size_t off_count, off_cursor, off_big_count, off_view_stats, off_queue, zero_bytes; // zero-filled prefix
  size_t off_offset, off_range, off_pre, off_big_list, off_pairs, off_items, total_bytes;
  int64_t max_items;
};

// Work items of the raster pass: a tile, or one of 4 / 16 sub-rectangles of a heavy tile.
// Triangles in a tile's list above which the tile is split 2x2 (four times that: 4x4).  A split buys parallelism and a
// shorter tail at the price of every sub-rectangle re-scanning the tile's list, so the threshold follows the size of the
// job: kSplit4Min where there are few tiles to go round (4 views of 10k triangles at 512^2: 0.085 ms with 384, 0.096
// with 512), a multiple of the mean list length per workgroup slot where there are many (split_threshold(): 8 x 100k at
// 2048^2 0.428 -> 0.410 ms with 512; 8 x 250k 0.686 -> 0.577 with 1280; 2 x 1M at 4096^2 0.716 -> 0.628; 8 x 1M 2.5 ->
// 1.9 with 2048; lower thresholds only lose: 256 0.451, 192 0.487, 128 0.577 ms on the bench shape).
constexpr int kSplit4Min = 384, kSplit4Max = 2048;
inline int split_threshold(int64_t N, int64_t F) {
#ifndef DRTK_RASTER_SPLIT_SLOTS
#define DRTK_RASTER_SPLIT_SLOTS 4
#endif
  const int64_t slots = int64_t(num_compute_units()) * DRTK_RASTER_SPLIT_SLOTS; // resident raster workgroups
  const int64_t t = (N * F * 13) / (20 * (slots > 0 ? slots : 1)); // 0.65 x triangles per slot
  return static_cast<int>(t < kSplit4Min ? kSplit4Min : (t > kSplit4Max ? kSplit4Max : t));
}
__host__ __device__ inline uint32_t make_item(uint32_t tile, uint32_t sub, uint32_t split_log) {
  return tile | (sub << 24) | (split_log << 28);
}

inline size_t align_up(size_t x, size_t a) {
  return (x + a - 1) / a * a;
}

inline BinLayout make_layout(int64_t N, int64_t F, int64_t H, int64_t W) {
  BinLayout L;
  const int64_t t64 = N * ceil_div(W, 64) * ceil_div(H, 64);
  L.tile_shift = (t64 >= 2048) ? 6 : 5; // fewer than ~8 workgroups per CU -> use 32x32 tiles
  const int64_t ts = int64_t(1) << L.tile_shift;
  L.tiles_x = static_cast<int>(ceil_div(W, ts));
  L.tiles_y = static_cast<int>(ceil_div(H, ts));
  L.tiles_per_view = int64_t(L.tiles_x) * L.tiles_y;
  L.num_tiles = N * L.tiles_per_view;
  size_t o = 0;
  L.off_count = o; // per tile, packed: low word = list entries, high word = entries with positive orientation
  o += align_up(sizeof(unsigned long long) * L.num_tiles, 256);
  L.off_cursor = o; // per tile, packed fill cursors: low word = positive entries, high word = negative entries
  o += align_up(sizeof(unsigned long long) * L.num_tiles, 256);
  L.off_big_count = o;
  o += align_up(sizeof(int32_t) * (N > 0 ? N : 1), 256);
  L.off_view_stats = o; // per view: {sum z, count} of positively / negatively oriented triangles
  o += align_up(sizeof(float) * 4 * (N > 0 ? N : 1), 256);
  L.off_queue = o; // [1] number of work items; [kQueueStride * (1 + s)] next position of shard s
  o += align_up(sizeof(int32_t) * kQueueStride * (1 + kQueueShards), 256);
  L.zero_bytes = o;
  L.off_offset = o;
  o += align_up(sizeof(int32_t) * (L.num_tiles + 1), 256);
  L.off_range = o;
  o += align_up(sizeof(uint2) * N * F, 256);
  L.off_pre = o;
  o += align_up(sizeof(uint4) * N * F, 256);
  L.off_big_list = o;
  o += align_up(sizeof(int32_t) * N * F, 256);
  L.off_pairs = o;
  o += align_up(sizeof(int32_t) * kMaxSmallTiles * N * F, 256);
  // every split tile holds > kSplit4Min of the <= 4*N*F list entries and yields <= 16 items
  L.max_items = L.num_tiles + 16 * ((kMaxSmallTiles * N * F) / kSplit4Min);
  L.off_items = o;
  o += align_up(sizeof(uint32_t) * L.max_items, 256);
  L.total_bytes = o > 0 ? o : 256;
  return L;
}

// (wave_agg_inc -- the wave-aggregated "+1" on counters[key] -- lives in common.hpp: the wireframe binner uses it too)
// Same aggregation on a pair of 32-bit counters packed in one 64-bit word: lanes with `hi` set bump the
// high word, the others (or, with BOTH, every lane) the low word -- one atomic per distinct key and wave.
//   count pass (BOTH):  low += lanes, high += lanes with hi        -> {entries, positive entries}
//   fill pass (!BOTH):  low += lanes without hi, high += lanes with hi; returns this lane's slot in its
//                       own word's sequence                        -> {positive cursor, negative cursor}
template <bool BOTH, bool FETCH>
__device__ __forceinline__ int wave_agg_inc2(unsigned long long* __restrict__ counters, int key, bool on, bool hi) {
  const int lane = lane_id();
  unsigned long long todo = __ballot(on);
  int pos = 0;
  while (todo) {
    const int leader = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
    const int k = __builtin_amdgcn_readlane(key, leader);
    const bool mine = on && key == k;
    const unsigned long long same = __ballot(mine);
    const unsigned long long same_hi = __ballot(mine && hi);
    const unsigned long long same_lo = BOTH ? same : (same & ~same_hi);
    unsigned long long base = 0;
    if (lane == leader) {
      const unsigned long long add =
          static_cast<unsigned long long>(__popcll(same_lo)) | (static_cast<unsigned long long>(__popcll(same_hi)) << 32);
      if (FETCH) {
        base = atomicAdd(counters + k, add);
      } else {
        atomicAdd(counters + k, add);
      }
    }
    if (FETCH) {
      const unsigned lo = __builtin_amdgcn_readlane(static_cast<int>(base & 0xFFFFFFFFull), leader);
      const unsigned hh = __builtin_amdgcn_readlane(static_cast<int>(base >> 32), leader);
      const unsigned long long below = (1ull << lane) - 1ull;
      if (mine) pos = hi ? static_cast<int>(hh) + __popcll(same_hi & below) : static_cast<int>(lo) + __popcll(same_lo & below);
    }
    todo &= ~same;
  }
  return pos;
}

// ---- pass 1: cull, bbox -> tile range, count -------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void bin_count_kernel(
    const T* __restrict__ v, const int32_t* __restrict__ vi, int F, int64_t V, int64_t vi_sN, int H,
    int W, int tile_shift, int tiles_x, int tiles_per_view, unsigned long long* __restrict__ tile_count,
    int32_t* __restrict__ big_count, int32_t* __restrict__ big_list,
    uint2* __restrict__ tri_range, uint4* __restrict__ tri_pre, float* __restrict__ view_stats) {
  const int n = blockIdx.y;
  const int f = blockIdx.x * kBlock + threadIdx.x;
  const bool in_range = f < F;
  int bx0, by0, bx1, by1;
  uint2 r = make_uint2(kCulled, kCulled);
  int tx0 = 0, ty0 = 0, tw = 0, ntiles = 0;
  bool positive = false;
  float z_mean = 0.0f;
  uint32_t z_lo_bits = 0;
  if (in_range &&
      tri_bbox<T>(v + int64_t(n) * V * 3, vi + int64_t(n) * vi_sN + int64_t(f) * 3, H, W, bx0, by0, bx1, by1, positive, z_mean, z_lo_bits)) {
    // The raster pass's pre-reject record: clamped pixel bbox (16 bits per coordinate) + depth lower bound.  With it
    // a hidden triangle is dropped after ONE 16-byte load -- no index / vertex gathers, no set-up.  Canvases beyond
    // 65535 pixels a side: .w = 0, "no record" (the raster pass then sets the triangle up and tests it as before).
    const bool fits = bx1 < 65536 && by1 < 65536;
    tri_pre[int64_t(n) * F + f] = make_uint4(
        static_cast<uint32_t>(bx0) | (static_cast<uint32_t>(bx1) << 16), static_cast<uint32_t>(by0) | (static_cast<uint32_t>(by1) << 16),
        fits ? z_lo_bits : 0u, fits ? 1u : 0u);
    tx0 = bx0 >> tile_shift;
    ty0 = by0 >> tile_shift;
    const int tx1 = bx1 >> tile_shift, ty1 = by1 >> tile_shift;
    r.x = static_cast<uint32_t>(tx0) | (static_cast<uint32_t>(tx1) << 16);
    // tile coordinates are < 2^15: bit 31 of .y carries the orientation for the fill pass
    r.y = static_cast<uint32_t>(ty0) | (static_cast<uint32_t>(ty1) << 16) | (positive ? kFacingBit : 0u);
    tw = tx1 - tx0 + 1;
    ntiles = tw * (ty1 - ty0 + 1);
  }
  // per-view mean depth of either orientation (decides which group the raster pass draws first); a
  // 1-in-64 sample of the workgroups is plenty for that decision and keeps the four same-address
  // atomics per view off the critical path
  if ((blockIdx.x & 63) == 0) {
    const bool live = ntiles >= 1;
    float zp = (live && positive) ? z_mean : 0.0f, zn = (live && !positive) ? z_mean : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      zp += __shfl_xor(zp, o);
      zn += __shfl_xor(zn, o);
    }
    const int cp = __popcll(__ballot(live && positive)), cn = __popcll(__ballot(live && !positive));
    if (lane_id() == 0 && (cp | cn)) {
      float* st = view_stats + 4 * n;
      if (cp) {
        atomic_add_global(st + 0, zp);
        atomic_add_global(st + 1, static_cast<float>(cp));
      }
      if (cn) {
        atomic_add_global(st + 2, zn);
        atomic_add_global(st + 3, static_cast<float>(cn));
      }
    }
  }
  if (in_range) tri_range[int64_t(n) * F + f] = r;
  const bool small = ntiles >= 1 && ntiles <= kMaxSmallTiles;
  const int base = n * tiles_per_view;
  for (int s = 0; s < kMaxSmallTiles; ++s) {
    const bool on = small && s < ntiles;
    const int dy = on ? s / tw : 0;
    const int dx = on ? s - dy * tw : 0;
    const int t = base + (ty0 + dy) * tiles_x + tx0 + dx;
    wave_agg_inc2<true, false>(tile_count, t, on, positive);
  }
  const bool big = ntiles > kMaxSmallTiles;
  if (__ballot(big)) {
    const int pos = wave_agg_inc<true>(big_count, n, big);
    if (big) big_list[int64_t(n) * F + pos] = f;
  }
}

// ---- pass 2: exclusive scan over all tile counters + raster work list (single workgroup) -------
// The work list orders the raster pass heavy-first: tiles whose lists are long are split into 4x4 or
// 2x2 sub-rectangles (each sub-rectangle re-scans the tile's list but rasterizes only its part), then
// ordinary tiles, then tiles with empty lists.  The raster kernel pulls items from this list through
// an atomic counter, so the pole / limb tiles of a mesh (10x the mean triangle count) no longer form
// a serial tail.
__global__ __launch_bounds__(1024) void bin_scan_kernel(
    const unsigned long long* __restrict__ tile_count64, int32_t* __restrict__ tile_offset, int num_tiles,
    int tile_shift, int split4, uint32_t* __restrict__ items, int32_t* __restrict__ queue) {
  // One block-wide scan of FIVE running sums per thread -- list entries, and work items of each of the four
  // classes -- gives every thread the list offset of its first tile and its first slot in every class of the
  // work list: no atomics (1024 threads bumping four LDS counters are served one lane at a time), two
  // barriers, and the work list comes out ordered by tile within a class.
  constexpr int kThreads = 1024, kWaves = kThreads / kWave, kVals = 5, kReg = 8;
  __shared__ int32_t s_wave[kWaves][kVals];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int chunk = (num_tiles + kThreads - 1) / kThreads;
  const int begin = tid * chunk;
  const int end = min(begin + chunk, num_tiles);
  const int max_split_log = max(0, min(2, tile_shift - 4)); // sub-rectangles are at least 16 px
  auto split_log_of = [&](int c) {
    int sl = c > 4 * split4 ? 2 : (c > split4 ? 1 : 0);
    return min(sl, max_split_log);
  };
  auto class_of = [&](int c, int sl) { return c == 0 ? 3 : 2 - sl; }; // 0: 4x4 split, 1: 2x2, 2: whole tile, 3: empty list
  // the thread's counters: in registers when they fit (one batch of independent loads), re-read otherwise
  const bool in_regs = chunk <= kReg;
  int32_t cnt[kReg];
#pragma unroll
  for (int j = 0; j < kReg; ++j) {
    cnt[j] = (in_regs && begin + j < end) ? static_cast<int32_t>(tile_count64[begin + j] & 0xFFFFFFFFull) : 0;
  }
  auto count_at = [&](int j) -> int32_t { // j-th tile of this thread's chunk
    return static_cast<int32_t>(tile_count64[begin + j] & 0xFFFFFFFFull);
  };
  int32_t v[kVals] = {0, 0, 0, 0, 0};
  auto tally = [&](int c) {
    const int sl = split_log_of(c);
    v[0] += c;
    v[1 + class_of(c, sl)] += 1 << (2 * sl);
  };
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < kReg; ++j) {
      if (begin + j < end) tally(cnt[j]);
    }
  } else {
    for (int j = 0; begin + j < end; ++j) tally(count_at(j));
  }
  // inclusive scan within the wave, wave totals through LDS
  int32_t inc[kVals];
#pragma unroll
  for (int k = 0; k < kVals; ++k) {
    int32_t x = v[k];
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int32_t up = __shfl_up(x, o);
      if (lane >= o) x += up;
    }
    inc[k] = x;
    if (lane == kWave - 1) s_wave[wave][k] = x;
  }
  __syncthreads();
  int32_t before[kVals], total[kVals];
#pragma unroll
  for (int k = 0; k < kVals; ++k) {
    int32_t lo = 0, all = 0;
    for (int w = 0; w < kWaves; ++w) {
      const int32_t t = s_wave[w][k];
      all += t;
      if (w < wave) lo += t;
    }
    before[k] = lo + inc[k] - v[k]; // exclusive prefix of this thread
    total[k] = all;
  }
  // class k's slots start after the classes before it
  int32_t slot[4];
  {
    int32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      slot[k] = acc + before[1 + k];
      acc += total[1 + k];
    }
    if (tid == 0) {
      queue[1] = acc; // number of work items
      tile_offset[num_tiles] = total[0];
    }
  }
  int32_t run = before[0];
  auto emit = [&](int i, int c) {
    tile_offset[i] = run;
    run += c;
    const int sl = split_log_of(c);
    const int cls = class_of(c, sl);
    const int n_sub = 1 << (2 * sl);
    for (int j = 0; j < n_sub; ++j) items[slot[cls] + j] = make_item(static_cast<uint32_t>(i), j, sl) | (c == 0 ? kItemEmpty : 0u);
    slot[cls] += n_sub;
  };
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < kReg; ++j) {
      if (begin + j < end) emit(begin + j, cnt[j]);
    }
  } else {
    for (int j = 0; begin + j < end; ++j) emit(begin + j, count_at(j));
  }
}

// ---- pass 3: write triangle ids into the tile lists -----------------------------------------
__global__ __launch_bounds__(kBlock) void bin_fill_kernel(
    const uint2* __restrict__ tri_range, int F, int tiles_x, int tiles_per_view,
    const int32_t* __restrict__ tile_offset, unsigned long long* __restrict__ tile_cursor,
    int32_t* __restrict__ pairs) {
  const int n = blockIdx.y;
  const int f = blockIdx.x * kBlock + threadIdx.x;
  int tx0 = 0, ty0 = 0, tw = 0, ntiles = 0;
  bool positive = false;
  if (f < F) {
    const uint2 r = tri_range[int64_t(n) * F + f];
    if (r.x != kCulled) {
      tx0 = r.x & 0xFFFF;
      ty0 = r.y & 0xFFFF;
      tw = static_cast<int>(r.x >> 16) - tx0 + 1;
      ntiles = tw * (static_cast<int>((r.y & ~kFacingBit) >> 16) - ty0 + 1);
      positive = (r.y & kFacingBit) != 0;
    }
  }
  const bool small = ntiles >= 1 && ntiles <= kMaxSmallTiles;
  const int base = n * tiles_per_view;
  // Slots come from returning atomics, whose latency would be paid once per distinct tile and wave if
  // they were issued inside the grouping loop.  So: group first (no memory traffic), then let every
  // group's first lane issue its atomic -- all groups of all four tile slots in flight together -- and
  // only then hand the results to the group members.
  const int lane = lane_id();
  const unsigned long long below = (1ull << lane) - 1ull;
  int t_[kMaxSmallTiles], leader_[kMaxSmallTiles], rank_[kMaxSmallTiles];
  bool on_[kMaxSmallTiles];
  unsigned long long base_[kMaxSmallTiles];
#pragma unroll
  for (int s = 0; s < kMaxSmallTiles; ++s) {
    const bool on = small && s < ntiles;
    const int dy = on ? s / tw : 0;
    const int dx = on ? s - dy * tw : 0;
    const int t = base + (ty0 + dy) * tiles_x + tx0 + dx;
    int leader = lane, rank = 0;
    unsigned long long add = 0;
    unsigned long long todo = __ballot(on);
    while (todo) {
      const int ld = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
      const int k = __builtin_amdgcn_readlane(t, ld);
      const bool mine = on && t == k;
      const unsigned long long same = __ballot(mine);
      const unsigned long long same_neg = __ballot(mine && !positive);
      const unsigned long long same_pos = same & ~same_neg;
      if (mine) {
        leader = ld;
        rank = positive ? __popcll(same_pos & below) : __popcll(same_neg & below);
        add = static_cast<unsigned long long>(__popcll(same_pos)) | (static_cast<unsigned long long>(__popcll(same_neg)) << 32);
      }
      todo &= ~same;
    }
    on_[s] = on, t_[s] = t, leader_[s] = leader, rank_[s] = rank;
    base_[s] = 0;
    if (on && lane == leader) base_[s] = atomicAdd(tile_cursor + t, add); // low word: positive cursor, high: negative
  }
#pragma unroll
  for (int s = 0; s < kMaxSmallTiles; ++s) {
    const unsigned lo = __shfl(static_cast<unsigned>(base_[s] & 0xFFFFFFFFull), leader_[s]);
    const unsigned hi = __shfl(static_cast<unsigned>(base_[s] >> 32), leader_[s]);
    if (on_[s]) {
      const int pos = static_cast<int>(positive ? lo : hi) + rank_[s];
      pairs[positive ? tile_offset[t_[s]] + pos : tile_offset[t_[s] + 1] - 1 - pos] = f;
    }
  }
}


// ---- pass 4: per-tile rasterization with the z-buffer in LDS -----------------------------------
// Four triangles per pass, one per 16-lane DPP row.  A wave sets 64 triangles up one per lane (gathers + exact set-up)
// and then walks them in 16 steps: in step k row r (lanes 16r .. 16r+15) shades the triangle held by ITS lane k -- the
// plane equations reach the row's lanes with row-local DPP broadcasts (`row_newbcast:k`, which the compiler folds into
// the consuming instruction where a value is used once), stay in VGPRs, and the row covers the triangle's clipped bbox
// 16 pixels at a time in row-major order of the bbox (pixel p of the bbox = p % bw, p / bw), resolved with ds_min_u64
// into the tile.  Against the first design (one triangle per pass, its equations broadcast into SGPRs with 28
// v_readlane, 64-pixel stamps) the serial per-TRIANGLE chain is paid once per four triangles and runs on the vector
// unit, and a 16-pixel stamp wastes fewer lanes on a ~100-pixel bbox than a 64-pixel one.
__device__ __forceinline__ int mbcnt(unsigned long long m) { // set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}
template <int K>
__device__ __forceinline__ int row_bcast(int x) { // lane l <- lane K of l's 16-lane row (all lanes active at the call)
  return __builtin_amdgcn_mov_dpp(x, 0x150 + K, 0xF, 0xF, true);
}
template <int K>
__device__ __forceinline__ float row_bcast(float x) {
  return __builtin_bit_cast(float, row_bcast<K>(__builtin_bit_cast(int, x)));
}
template <int K>
__device__ __forceinline__ double row_bcast(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = row_bcast<K>(static_cast<int>(u & 0xFFFFFFFFull));
  const unsigned hi = row_bcast<K>(static_cast<int>(u >> 32));
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}

// What a row needs to shade one triangle.  Edge k: b_k = (py - ay) * dx - (px - ax) * dy with the orientation sign
// (rasterize_kernel.cu:133-141: canonical direction x sign of the denominator) folded into dx, dy -- negating both
// factors' partner negates the rounded products and their rounded difference exactly, so b_k is the reference's value
// up to the sign of a zero, which no later operation can see (b_k == 0, b_k >= 0, +-0 / d added to a non-zero sum).
template <typename T>
struct TriRow {
  T ax[3], ay[3], dx[3], dy[3];
  T abs_denom, rdenom, dinv0, dinv1, dinv2;
  int id_tl; // triangle id | top-left bits of the three edges << 29 (ids are < 2^29: N F < 2^31 / 4)
  int box;   // clipped bbox relative to the item's rectangle, 6 bits each: x, y of the first pixel, width - 1, height - 1;
             // bit 24: there is something to draw
  uint32_t z_lo; // f32 bits of the rigorous lower bound of the triangle's depths (bin_count's record; 0: none)
};
constexpr int kTlShift = 29;
constexpr int kBoxDraw = 1 << 24;

template <int K, typename T>
__device__ __forceinline__ TriRow<T> row_bcast_tri(const TriRow<T>& o) {
  TriRow<T> u;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    u.ax[k] = row_bcast<K>(o.ax[k]);
    u.ay[k] = row_bcast<K>(o.ay[k]);
    u.dx[k] = row_bcast<K>(o.dx[k]);
    u.dy[k] = row_bcast<K>(o.dy[k]);
  }
  u.abs_denom = row_bcast<K>(o.abs_denom);
  u.rdenom = row_bcast<K>(o.rdenom);
  u.dinv0 = row_bcast<K>(o.dinv0);
  u.dinv1 = row_bcast<K>(o.dinv1);
  u.dinv2 = row_bcast<K>(o.dinv2);
  u.id_tl = row_bcast<K>(o.id_tl);
  u.box = row_bcast<K>(o.box);
  u.z_lo = static_cast<uint32_t>(row_bcast<K>(static_cast<int>(o.z_lo)));
  return u;
}

// Own-lane state from the exact set-up: oriented edges, bbox clipped to the item's rectangle [x0, x1] x [y0, y1].
template <typename T>
__device__ __forceinline__ TriRow<T> make_row_state(bool valid, const TriSetup<T>& s, int f, uint32_t z_lo, int x0, int y0, int x1, int y1) {
  TriRow<T> o;
  o.z_lo = z_lo;
  const T px[3] = {s.p1x, s.p2x, s.p0x}, py[3] = {s.p1y, s.p2y, s.p0y}; // edge k starts at p_{k+1}
  const T qx[3] = {s.p2x, s.p0x, s.p1x}, qy[3] = {s.p2y, s.p0y, s.p1y}; // ... and ends at p_{k+2}
  const bool c[3] = {s.c0, s.c1, s.c2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o.ax[k] = c[k] ? px[k] : qx[k];
    o.ay[k] = c[k] ? py[k] : qy[k];
    const T bx = c[k] ? qx[k] : px[k], by = c[k] ? qy[k] : py[k];
    const bool neg = c[k] ? (s.sign_denom < T(0)) : !(s.sign_denom < T(0));
    const T ex = bx - o.ax[k], ey = by - o.ay[k];
    o.dx[k] = neg ? -ex : ex;
    o.dy[k] = neg ? -ey : ey;
  }
  o.abs_denom = s.abs_denom;
  o.rdenom = s.rdenom;
  o.dinv0 = s.dinv0;
  o.dinv1 = s.dinv1;
  o.dinv2 = s.dinv2;
  const int bx0 = max(s.bb_min_x, x0), bx1 = min(s.bb_max_x, x1);
  const int by0 = max(s.bb_min_y, y0), by1 = min(s.bb_max_y, y1);
  const bool draw = valid && bx0 <= bx1 && by0 <= by1;
  o.box = draw ? ((bx0 - x0) | ((by0 - y0) << 6) | ((bx1 - bx0) << 12) | ((by1 - by0) << 18) | kBoxDraw) : 0;
  o.id_tl = f | ((s.tl0 ? 1 : 0) | (s.tl1 ? 2 : 0) | (s.tl2 ? 4 : 0)) << kTlShift;
  return o;
}

template <typename T>
using TriOwn = TriRow<T>;

// One step: every row shades the triangle `u` it was handed.  Coverage, depth and the packed atomicMin are the
// reference's (rasterize_kernel.cu:117-161), evaluated per lane = per pixel.
// STRIDE = lanes that share the triangle: 16 (one DPP row per triangle, `lane16_half` = lane % 16 + 1/2) or the whole
// workgroup (cooperative pass over a large triangle, `lane16_half` = thread id + 1/2).
// FM (round 6; before: a second library built with -DDRTK_DEPTH_FASTMATH_ORDER): the order of the depth sum, chosen per
// LAUNCH from the library's depth-order setting (drtk_amd_set_depth_order / DRTK_AMD_DEPTH_ORDER, include/drtk_amd.h).
template <typename T, int TILE_SHIFT, bool EARLY_Z, bool FM, int STRIDE = 16>
__device__ __forceinline__ void shade_rows(
    const TriRow<T>& u, float lane16_half, float x0f, float y0f, unsigned long long* __restrict__ zbuf, int dbg) {
  constexpr float kTileF = static_cast<float>(1 << TILE_SHIFT);
  const float rx0f = static_cast<float>(u.box & 63), ry0f = static_cast<float>((u.box >> 6) & 63);
  const float bwf = static_cast<float>(((u.box >> 12) & 63) + 1), bhf = static_cast<float>(((u.box >> 18) & 63) + 1);
  const float npxf = (u.box & kBoxDraw) ? bwf * bhf : 0.0f;
  const float bx0f = x0f + rx0f, by0f = y0f + ry0f; // absolute pixel coordinates: integers < 2^22, exact
  const float c0f = __builtin_fmaf(ry0f, kTileF, rx0f); // index of the bbox's first pixel in the LDS tile
  // pixel p of the bbox -> (p % bw, p / bw) in float: (p + 1/2) / bw is never within 1 / (2 bw) >= 1/128 of an integer,
  // the product with the 1-ulp reciprocal is off by < 2e-5; p - ly * bw is exact (integers below 2^13)
  float rbw = __builtin_amdgcn_rcpf(bwf);
  asm volatile("" : "+v"(rbw)); // keeps the quarter-rate reciprocal OUT of the pixel loop (the compiler rematerialises it there)
  const bool div_ok = exact_div_ok(u.abs_denom);
  // Coverage + top-left rule (:133-145) as three comparisons: a pixel is kept iff b_k >= 0 on top-left edges and b_k > 0
  // on the others, i.e. b_k >= thr_k with thr_k = +0 or the smallest positive (subnormal) number -- `b >= denorm_min`
  // IS `b > 0`; comparisons never flush subnormals.  (-0 >= +0 holds, -0 >= denorm_min does not: same as the two-sided form.)
  const T thr0 = smallest_positive_if<T>(!(u.id_tl & (1 << kTlShift)));
  const T thr1 = smallest_positive_if<T>(!(u.id_tl & (2 << kTlShift)));
  const T thr2 = smallest_positive_if<T>(!(u.id_tl & (4 << kTlShift)));
  // The fast quotients are valid when b_k * rdenom >= DivRange::tiny() or b_k == 0 (exact_div's guard).  ONE cheaper,
  // stricter test decides for the wave: every b_k >= 2^-40 abs_denom (then the quotient is >= ~2^-41); a pixel exactly on
  // an edge (b_k == 0) therefore also sends its wave through exact_div, which is correct for it -- only slower.
  const T b_min = u.abs_denom * T(0x1p-40);
  const unsigned long long id = static_cast<uint32_t>(u.id_tl) & ((1u << kTlShift) - 1u);
#ifdef DRTK_AMD_ABLATION
  if (STRIDE == 16 && DRTK_DBG(dbg, 512)) { // row balance of the steps: [10] += passes the wave runs (the longest row's), [11] += passes the four rows need together
    const int need = static_cast<int>((npxf + 15.0f) * 0.0625f);
    int mx = need;
#pragma unroll
    for (int o = 32; o >= 16; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    int sum = need;
#pragma unroll
    for (int o = 32; o >= 16; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane_id() == 0) {
      atomicAdd(&::drtk_amd::g_phase_clocks[10], static_cast<unsigned long long>(mx));
      atomicAdd(&::drtk_amd::g_phase_clocks[11], static_cast<unsigned long long>(sum));
      atomicAdd(&::drtk_amd::g_phase_clocks[12], 1ull);
    }
  }
#endif
  for (float ph = lane16_half; ph < npxf; ph += static_cast<float>(STRIDE)) {
    const float fly = __builtin_truncf(ph * rbw);
    const float flx = __builtin_fmaf(-fly, bwf, ph - 0.5f);
    const T px = static_cast<T>(bx0f + flx), py = static_cast<T>(by0f + fly);
    const T b0 = (py - u.ay[0]) * u.dx[0] - (px - u.ax[0]) * u.dy[0];
    const T b1 = (py - u.ay[1]) * u.dx[1] - (px - u.ax[1]) * u.dy[1];
    const T b2 = (py - u.ay[2]) * u.dx[2] - (px - u.ax[2]) * u.dy[2];
    if (!((b0 >= thr0) & (b1 >= thr1) & (b2 >= thr2))) continue;
    const int zi = static_cast<int>(__builtin_fmaf(fly, kTileF, c0f) + flx);
    if (EARLY_Z) {
      // second group only: a fragment whose triangle cannot be nearer than what the pixel already holds -- the
      // rigorous bound against the stored depth, strictly, as in the block test -- skips the depth arithmetic (two
      // thirds of the loop body); most of that group's fragments are hidden back faces that got through the 8x8-block
      // test because their blocks contain background
      if (u.z_lo > static_cast<uint32_t>(zbuf[zi] >> 32)) continue;
    }
    if (DRTK_DBG(dbg, 4)) {
      atomicMin(&zbuf[zi], id);
      continue;
    }
    // b_k / abs_denom (:148), 1 / epsclamp(depth_inverse) (:153): the correctly rounded fast forms are evaluated
    // unconditionally; whether any lane needs the IEEE fallback (operands outside the guarded range: denormal
    // quotients, an all-ones significand) is ONE wave-uniform test per pass instead of a divergent branch per
    // division -- the fallback itself is exact_div / exact_rcp
    T d0 = T(0), d1 = T(0), d2 = T(0);
    if constexpr (!FM) {
      const T q0 = b0 * u.rdenom, q1 = b1 * u.rdenom, q2 = b2 * u.rdenom;
      const bool fast_ok = div_ok & (min3(b0, b1, b2) >= b_min);
      d0 = markstein2(b0, u.abs_denom, u.rdenom, q0);
      d1 = markstein2(b1, u.abs_denom, u.rdenom, q1);
      d2 = markstein2(b2, u.abs_denom, u.rdenom, q2);
      if (__ballot(!fast_ok) != 0) {
        d0 = exact_div(b0, u.abs_denom, u.rdenom, div_ok);
        d1 = exact_div(b1, u.abs_denom, u.rdenom, div_ok);
        d2 = exact_div(b2, u.abs_denom, u.rdenom, div_ok);
      }
    }
    // FM: the depth as the reference's host path evaluates it WHEN BUILT WITH ITS OWN FLAGS (setup.py:22-24: -O3
    // --fast-math, GCC 11.4 on x86-64; SURVEY App. A.1 step 8, from the disassembly): one IEEE reciprocal r = 1 / |den|
    // per triangle and  s = ((e1 dinv1 + e0 dinv0) + e2 dinv2) r  instead of three quotients e_k / |den| -- the last bits
    // of 40 % of the depths move, and with them the owner of a handful of near-tie pixels per view
    // (tests/golden/fastmath_owner_changes_*.npz).  No FMA there either.  !FM (the default): the source order, strict.
    const T depth_inverse = FM ? ((b1 * u.dinv1 + b0 * u.dinv0) + b2 * u.dinv2) * u.rdenom : u.dinv0 * d0 + u.dinv1 * d1 + u.dinv2 * d2;
    // epsclamp (:153) of a value that cannot be negative: every d_k is a quotient of b_k >= 0 (the fragment passed the
    // coverage test) and abs_denom > 0, every dinv_k is 1 / z_k with z_k > 1e-8 (the near-plane cull, :96) -- so only the
    // `v > eps ? v : eps` branch of the clamp exists here (a NaN sum takes eps there as well)
    const T di = depth_inverse > Eps<T>::value() ? depth_inverse : Eps<T>::value();
    T rd = fast_rcp(di);
    if (__ballot(!fast_rcp_ok_clamped(di)) != 0) rd = exact_rcp(di);
    const float depth = static_cast<float>(rd);
    const unsigned long long packed = (static_cast<unsigned long long>(__float_as_uint(depth)) << 32) | id;
    atomicMin(&zbuf[zi], packed);
  }
}

// Rasterize the triangles held one per lane (own-lane state `o`): `nsteps` steps, step k = lane k of every row.
template <typename T, int TILE_SHIFT, bool EARLY_Z, bool FM>
__device__ __forceinline__ void raster_rows(
    const TriOwn<T>& o, int nsteps, int x0, int y0, unsigned long long* __restrict__ zbuf, int dbg) {
  const float x0f = static_cast<float>(x0), y0f = static_cast<float>(y0);
  const float lane16_half = static_cast<float>(lane_id() & 15) + 0.5f;
  for (int k = 0; k < nsteps; ++k) {
    TriRow<T> u;
    switch (k) {
#define DRTK_ROW_STEP(K) case K: u = row_bcast_tri<K>(o); break;
      DRTK_ROW_STEP(0) DRTK_ROW_STEP(1) DRTK_ROW_STEP(2) DRTK_ROW_STEP(3) DRTK_ROW_STEP(4) DRTK_ROW_STEP(5)
      DRTK_ROW_STEP(6) DRTK_ROW_STEP(7) DRTK_ROW_STEP(8) DRTK_ROW_STEP(9) DRTK_ROW_STEP(10) DRTK_ROW_STEP(11)
      DRTK_ROW_STEP(12) DRTK_ROW_STEP(13) DRTK_ROW_STEP(14)
      default: u = row_bcast_tri<15>(o); break;
#undef DRTK_ROW_STEP
    }
    shade_rows<T, TILE_SHIFT, EARLY_Z, FM>(u, lane16_half, x0f, y0f, zbuf, dbg);
  }
}

// 8 waves per workgroup share one 32 KiB tile.  Registers: the lane's own triangle (19) + the row's triangle (19) + the
// pixel loop; bounded to 80 VGPRs = 3 workgroups = 24 waves per CU, WITHOUT spills (the build fails on one: item-level
// constants that the compiler hoists out of the item loop are rematerialised behind `asm volatile("" : "+v")`).
#ifndef DRTK_RASTER_WAVES_PER_SIMD
#define DRTK_RASTER_WAVES_PER_SIMD 6
#endif
#ifndef DRTK_RASTER_BLOCK
#define DRTK_RASTER_BLOCK 512 // 256 (4 waves per item, 4 items per CU by LDS; DRTK_RASTER_BLOCKS_PER_CU=4): 0.305 / 0.375 ms against
#endif                        // 0.305 / 0.36 at 100k / 250k triangles -- a wave-group holds ~3.4 triangles either way
constexpr int kRasterBlock = DRTK_RASTER_BLOCK;
constexpr int kRasterWaves = kRasterBlock / kWave;
template <typename T>
constexpr int raster_waves_per_simd() { // double: twice the registers per value (134 VGPRs) -> ONE workgroup (2 waves per
  return sizeof(T) == 4 ? DRTK_RASTER_WAVES_PER_SIMD : 2; // SIMD) per CU; bounded to 128 for two it spills
}
constexpr int kIdRing = 128; // accepted triangle ids waiting for set-up, per wave (power of two, >= 2 * kWave - 1)
// Cooperative pass.  A 16-lane row walks its triangle's clipped bbox 16 pixels a pass, whatever the size: a triangle that
// fills the 64 x 64 tile is 256 passes on ONE row while -- in a tile that holds two or four such triangles and nothing
// else (close-ups, low-poly meshes at high resolution, screen-filling quads) -- the other rows and the other seven waves
// have nothing to do (round 4, profiles/r04/raster_regimes.json: 1.4-3.4x the bench scene's time per covered pixel).  A
// triangle whose clipped bbox has at least kCoopMin pixels is therefore not queued in the wave's ring but in a list of
// the workgroup, and after the rounds ALL waves shade it together, 512 pixels a pass, its row state -- set up once, by
// one lane -- read from LDS.  Same fragments, same packed minimum: the image cannot change.
#ifndef DRTK_RASTER_COOP_MIN
#define DRTK_RASTER_COOP_MIN 256
#endif
#ifndef DRTK_RASTER_COOP_MIN_DENSE
#define DRTK_RASTER_COOP_MIN_DENSE 1024
#endif
#ifndef DRTK_RASTER_COOP_DENSE
#define DRTK_RASTER_COOP_DENSE 64
#endif
// ... where the rows would idle, that is: in a tile whose list is long (kCoopDense entries, both groups) the ordinary
// path already keeps all 32 rows of the workgroup busy, and moving its larger triangles (the 600-pixel bounding boxes
// at the centre of the benchmark's sphere) to the cooperative pass only adds the pass's own costs (measured: +12 % on
// the bench scene with one threshold of 512 for every tile) -- there the threshold is kCoopMinDense.
constexpr int kCoopMin = DRTK_RASTER_COOP_MIN, kCoopMinDense = DRTK_RASTER_COOP_MIN_DENSE, kCoopDense = DRTK_RASTER_COOP_DENSE;
constexpr int kCoopBatch = 64; // triangles set up together, one per lane of the waves' first lanes, and parked in LDS
constexpr int kCoopMax = 256; // entries per item and phase; what does not fit takes the ordinary path

template <typename T, int TILE_SHIFT, bool FM>
__global__ __launch_bounds__(kRasterBlock, raster_waves_per_simd<T>()) void tile_raster_kernel(
    const T* __restrict__ v, const int32_t* __restrict__ vi, int F, int64_t V, int64_t vi_sN,
    int H, int W, int tiles_x, int tiles_per_view, const int32_t* __restrict__ tile_offset,
    const unsigned long long* __restrict__ tile_count, const float* __restrict__ view_stats,
    const int32_t* __restrict__ pairs, const int32_t* __restrict__ big_count,
    const int32_t* __restrict__ big_list, const uint2* __restrict__ tri_range, const uint4* __restrict__ tri_pre,
    const uint32_t* __restrict__ items, int32_t* __restrict__ queue, float* __restrict__ depth_img,
    int32_t* __restrict__ index_img, int dbg) {
  constexpr int TILE = 1 << TILE_SHIFT;
  constexpr int NPIX = TILE * TILE;
  __shared__ unsigned long long zbuf[NPIX];
  __shared__ uint32_t s_zmax[(TILE / 8) * (TILE / 8)];
  __shared__ int32_t s_idq[kRasterWaves][kIdRing];
  __shared__ int32_t s_idk[kRasterWaves][kIdRing]; // ... and the size of what each will have to shade (sort key)
  __shared__ int s_item;
  __shared__ int s_queue[3];
  __shared__ int32_t s_coop[kCoopMax];
  __shared__ int s_ncoop;
  constexpr int kTriWords = sizeof(TriRow<T>) / 4;
  static_assert(sizeof(TriRow<T>) % 4 == 0, "TriRow is parked in LDS word by word");
  __shared__ uint32_t s_tri[kCoopMin > 0 ? kCoopBatch * kTriWords : 1];
  // the batch's set-up deals kCoopBatch / kRasterWaves entries to the first lanes of every wave (any other
  // DRTK_RASTER_BLOCK would leave entries of s_tri unset), and the launcher's resident-workgroup count assumes
  // raster_waves_per_simd / 2 workgroups per CU also fit by LDS (160 KiB per CU)
  static_assert(kCoopBatch % kRasterWaves == 0 && kCoopBatch / kRasterWaves <= kWave, "cooperative batch set-up needs kRasterWaves | kCoopBatch");
  static_assert(
      (sizeof(unsigned long long) * NPIX + sizeof(uint32_t) * (TILE / 8) * (TILE / 8) + 2 * sizeof(int32_t) * kRasterWaves * kIdRing +
       sizeof(int32_t) * kCoopMax + sizeof(uint32_t) * kCoopBatch * kTriWords + 64) * (raster_waves_per_simd<T>() * 4 / kRasterWaves > 0 ? raster_waves_per_simd<T>() * 4 / kRasterWaves : 1) <= 160 * 1024,
      "tile_raster's static LDS no longer fits the workgroups per CU its launch bounds ask for");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid & (kWave - 1);
  const int n_items = queue[1];
  // The queue state of the workgroup -- the shard it draws from, how many shards it has seen run dry, the item it has
  // reserved -- lives in LDS, read and written by thread 0 alone: as loop-carried registers of one lane they were kept
  // in spill slots across the item loop, and one such spill was placed where EXEC is empty (straight after a divergent
  // loop's exit, before the mask is restored), so the update was lost and almost half of the tiles were never drawn.
  auto pop = [&]() { // thread 0 only; leaves the reserved item (or -1) in s_queue[2]
    int shard = s_queue[0], dry = s_queue[1], idx = -1;
    while (dry < kQueueShards) {
      idx = atomicAdd(&queue[kQueueStride * (1 + shard)], 1) * kQueueShards + shard;
      if (idx < n_items) break;
      idx = -1;
      ++dry;
      shard = (shard + 1) % kQueueShards;
    }
    s_queue[0] = shard, s_queue[1] = dry, s_queue[2] = idx;
  };
  if (tid == 0) {
    s_queue[0] = blockIdx.x % kQueueShards, s_queue[1] = 0;
    s_ncoop = 0;
    pop();
  }
  for (int i = tid; i < NPIX; i += kRasterBlock) zbuf[i] = ~0ull; // rasterize_kernel.cu:484-488; every item leaves the tile cleared
  DRTK_PHASE_INIT();
  for (;;) {
    if (tid == 0) s_item = s_queue[2];
    __syncthreads();
    if (tid == 0) DRTK_PHASE(0); // queue pop
    const int item_index = s_item;
    if (item_index < 0) break;
    const uint32_t item = items[item_index];
    const int tile = item & 0xFFFFFF;
    const int sub = (item >> 24) & 0xF, split_log = (item >> 28) & 3;
    const int n = tile / tiles_per_view;
    const int t_in_view = tile - n * tiles_per_view;
    const int ty = t_in_view / tiles_x, tx = t_in_view - ty * tiles_x;
    const int ss = TILE >> split_log; // side of this item's rectangle
    const int x0 = (tx << TILE_SHIFT) + (sub & ((1 << split_log) - 1)) * ss;
    const int y0 = (ty << TILE_SHIFT) + (sub >> split_log) * ss;
    const int x1 = min(x0 + ss - 1, W - 1), y1 = min(y0 + ss - 1, H - 1);
    if (!(x0 < W && y0 < H)) { // a sub-rectangle of a border tile that lies off the canvas
      if (tid == 0) pop();
      __syncthreads();
      continue;
    }
    {
      const int rows = y1 - y0 + 1;
      const int nbig = big_count[n];
      const int64_t img_base = int64_t(n) * H * W;
      const bool vec_ok = (W & 3) == 0;
      // 16-byte stores that need the element's alignment only (rows of any width, output views at any element offset)
      typedef int32_t IQuad __attribute__((ext_vector_type(4), aligned(4)));
      typedef float FQuad __attribute__((ext_vector_type(4), aligned(4)));
      const int quads_per_row = ss >> 2;
      if ((item & kItemEmpty) && nbig == 0) {
        // nothing can touch this tile: background straight to the images, no LDS tile, no barrier
        for (int q = tid; q < rows * quads_per_row; q += kRasterBlock) {
          const int row = q / quads_per_row;
          const int col = (q - row * quads_per_row) << 2;
          const int y = y0 + row, x = x0 + col;
          if (x > x1) continue;
          const int64_t o = img_base + int64_t(y) * W + x;
          int none = -1;
          float zero = 0.0f;
          asm volatile("" : "+v"(none), "+v"(zero)); // constants made here (see `cleared` below)
          if (vec_ok || x + 3 <= x1) { // whole quad inside the row: one 16-byte store, element-aligned when W % 4 != 0
            *reinterpret_cast<IQuad*>(index_img + o) = IQuad{none, none, none, none};
            *reinterpret_cast<FQuad*>(depth_img + o) = FQuad{zero, zero, zero, zero};
          } else {
            for (int j = 0; j < 4 && x + j <= x1; ++j) {
              index_img[o + j] = none;
              depth_img[o + j] = zero;
            }
          }
        }
        if (tid == 0) pop();
        __syncthreads(); // s_item is reused by the next pop
        continue;
      }

      const T* v_n = v + int64_t(n) * V * 3;
      const int32_t* vi_n = vi + int64_t(n) * vi_sN;
      const int nb = ss >> 3; // 8x8-pixel blocks per side of this item's rectangle
      int32_t* idq = s_idq[wave];
      int32_t* idk = s_idk[wave];

      // The tile's list is partitioned by orientation (bin_fill): [positive ... | ... negative].  The
      // group that is nearer on average in this view (on a closed mesh: the visible one) is drawn first;
      // then the farthest depth of every 8x8 block is known (empty pixels count as infinitely far), and
      // a triangle of the second group whose depth lower bound lies beyond the farthest depth of all
      // blocks its clipped bbox touches cannot win a single pixel -- (depth, id) only ever decreases --
      // so it is dropped before any fragment work.  On a closed mesh that removes the hidden half of
      // the triangles; on any input the image is unchanged (strict comparison: ties still go by id).
      const int begin = tile_offset[tile], end_all = tile_offset[tile + 1];
      const int n_pos = static_cast<int>(tile_count[tile] >> 32);
      const float* st = view_stats + 4 * n;
      const bool pos_first = st[0] * st[3] <= st[2] * st[1]; // mean z of positive <= mean z of negative
      const uint4* pre_n = tri_pre + int64_t(n) * F;
      const int32_t* big_n = big_list + int64_t(n) * F;
      const uint2* range_n = tri_range + int64_t(n) * F;
      for (int phase = 0; phase < 2; ++phase) {
        if (phase == 1 && DRTK_DBG(dbg, 64)) break; // timing only: the second (mostly hidden) group is not drawn at all
        const bool take_pos = (phase == 0) == pos_first;
        const int g_begin = take_pos ? begin : begin + n_pos, g_end = take_pos ? begin + n_pos : end_all;
        // the test on a triangle's pre-reject record (bin_count: clamped pixel bbox + depth lower bound): does its bbox
        // touch this item's rectangle, and (second group) can it still win a pixel of the 8x8 blocks it touches
        auto accept = [&](int bx_min, int by_min, int bx_max, int by_max, uint32_t z_lo_bits) -> bool {
          if (!(bx_min <= x1 && bx_max >= x0 && by_min <= y1 && by_max >= y0)) return false;
          if (phase == 0 || z_lo_bits == 0 || DRTK_DBG(dbg, 16)) return true;
          const int cx0 = (max(bx_min, x0) - x0) >> 3, cx1 = (min(bx_max, x1) - x0) >> 3;
          const int cy0 = (max(by_min, y0) - y0) >> 3, cy1 = (min(by_max, y1) - y0) >> 3;
          if ((cx1 - cx0 + 1) * (cy1 - cy0 + 1) > 16) return true;
          uint32_t far = 0;
          for (int by = cy0; by <= cy1; ++by)
            for (int bx = cx0; bx <= cx1; ++bx) far = max(far, s_zmax[by * nb + bx]);
          return !(z_lo_bits > far);
        };
        // triangle f of this view on its record alone (one 16-byte load); canvases beyond 65535 pixels a side have no
        // record (.w == 0): those are tested after the set-up
        // `size` = pixels of the bbox clipped to the item's rectangle, the number the row that draws it will walk
        auto pre_accept = [&](int f, int& size) -> bool {
          const uint4 pre = pre_n[f];
          size = NPIX + 1;
          if (pre.w == 0u) return true;
          const int bx_min = static_cast<int>(pre.x & 0xFFFFu), by_min = static_cast<int>(pre.y & 0xFFFFu);
          const int bx_max = static_cast<int>(pre.x >> 16), by_max = static_cast<int>(pre.y >> 16);
          size = (min(bx_max, x1) - max(bx_min, x0) + 1) * (min(by_max, y1) - max(by_min, y0) + 1);
          return accept(bx_min, by_min, bx_max, by_max, pre.z);
        };
        // The wave's share of the group (equal parts, so that the waves reach the barrier together), then -- with the
        // first group, whatever their orientation -- its share of the view's big triangles (more than kMaxSmallTiles
        // tiles; filtered by tile range).  Candidates are screened 64 at a time on their records; the ids that pass
        // queue up in a wave-private ring until 64 are there (or the input is exhausted), so that set-up and the raster
        // steps always run on full rows: a split tile's sub-rectangle or the hidden second group pass a few per cent.
        const int per_wave = (g_end - g_begin + kRasterWaves - 1) / kRasterWaves;
        int cursor = g_begin + wave * per_wave;
        const int end = min(cursor + per_wave, g_end);
        const int big_per_wave = phase == 0 ? (nbig + kRasterWaves - 1) / kRasterWaves : 0;
        int big_cursor = wave * big_per_wave;
        const int big_end = min(big_cursor + big_per_wave, phase == 0 ? nbig : 0);
        int head = 0, tail = 0; // ring positions (wave-uniform)
        const int coop_min = end_all - begin >= kCoopDense ? kCoopMinDense : kCoopMin; // (the tile's whole list, both groups)
        if (DRTK_DBG(dbg, 8)) cursor = end, big_cursor = big_end;
        for (;;) {
          while (tail - head < kWave && (cursor < end || big_cursor < big_end)) {
            bool ok = false;
            int f = 0, size = 0;
            if (cursor < end) {
              const int i = cursor + lane;
              if (i < end) {
                f = pairs[i];
                ok = pre_accept(f, size);
              }
              cursor += kWave;
            } else {
              const int i = big_cursor + lane;
              if (i < big_end) {
                f = big_n[i];
                const uint2 r = range_n[f];
                const int rtx0 = r.x & 0xFFFF, rtx1 = r.x >> 16, rty0 = r.y & 0xFFFF, rty1 = (r.y & ~kFacingBit) >> 16;
                ok = tx >= rtx0 && tx <= rtx1 && ty >= rty0 && ty <= rty1 && pre_accept(f, size);
              }
              big_cursor += kWave;
            }
            if (kCoopMin > 0) { // large in this item's rectangle: to the workgroup's list (size == NPIX + 1: no record)
              const bool coop = ok && size >= coop_min && size <= NPIX;
              const unsigned long long mc = __ballot(coop);
              if (mc != 0) {
                const int leader = __builtin_ctzll(mc);
                int base = 0;
                if (lane == leader) base = atomicAdd(&s_ncoop, __popcll(mc));
                base = __shfl(base, leader);
                const int slot = base + mbcnt(mc);
                if (coop && slot < kCoopMax) {
                  s_coop[slot] = f;
                  ok = false;
                }
              }
            }
            const unsigned long long m = __ballot(ok);
            if (ok) {
              const int slot = (tail + mbcnt(m)) & (kIdRing - 1);
              idq[slot] = f;
              idk[slot] = size;
            }
            tail += __popcll(m);
          }
          const int cnt = min(kWave, tail - head);
          if (cnt == 0) break;
          wave_lds_sync();
          // Entry e of the round goes to row e % 4, lane e / 4 of the row: every step has (up to) four triangles, and
          // the step lasts as long as its LARGEST one -- in list order the rows of a step were busy 68 % of its passes
          // (100k triangles; 77 % at 250k).  So the round's entries are first sorted by the size of their clipped bbox
          // (bitonic network over the wave, key and source lane in one word), descending: the four triangles of a step
          // are then neighbours in size, and the invalid tail of a partial round stays at the end.
          const int e = ((lane & 15) << 2) | (lane >> 4);
          bool valid = e < cnt;
          int f;
#ifndef DRTK_RASTER_NO_SORT
          if (cnt > 4) {
            const bool has = lane < cnt;
            const int slot = (head + lane) & (kIdRing - 1);
            const int f_nat = has ? idq[slot] : 0;
            uint32_t p = has ? ((static_cast<uint32_t>(idk[slot]) << 6) | static_cast<uint32_t>(lane)) : static_cast<uint32_t>(lane);
#pragma unroll
            for (int k = 2; k <= kWave; k <<= 1) {
#pragma unroll
              for (int j = k >> 1; j > 0; j >>= 1) {
                const uint32_t other = static_cast<uint32_t>(__shfl_xor(static_cast<int>(p), j));
                const bool take_max = ((lane & j) == 0) == ((lane & k) == 0); // descending over the whole wave
                p = take_max ? max(p, other) : min(p, other);
              }
            }
            // lane i now holds the i-th largest entry's source lane; entry e's triangle comes from there
            const int src = __shfl(static_cast<int>(p & 63u), e);
            f = __shfl(f_nat, src);
            if (!valid) f = 0;
          } else
#endif
          {
            f = valid ? idq[(head + e) & (kIdRing - 1)] : 0;
          }
          head += cnt;
          wave_lds_sync(); // the ring slots just read may be overwritten by the next screening round
          TriSetup<T> s = {};
          uint32_t z_lo = 0;
          if (valid) {
            const uint4 pre = pre_n[f];
            z_lo = pre.z;
            valid = tri_setup<T>(v_n, vi_n + int64_t(f) * 3, H, W, s);
            if (valid && pre.w == 0u) valid = accept(s.bb_min_x, s.bb_min_y, s.bb_max_x, s.bb_max_y, 0u);
          }
          const TriOwn<T> own = make_row_state<T>(valid, s, f, z_lo, x0, y0, x1, y1);
          if (!DRTK_DBG(dbg, 1)) {
            if (phase == 0) {
              raster_rows<T, TILE_SHIFT, false, FM>(own, (cnt + 3) >> 2, x0, y0, zbuf, dbg);
            } else {
              raster_rows<T, TILE_SHIFT, true, FM>(own, (cnt + 3) >> 2, x0, y0, zbuf, dbg);
            }
          }
        }
        if (tid == 0) DRTK_PHASE(2 + 4 * phase); // wave 0's share of the group: 2 = first group, 6 = second
        if (kCoopMin > 0) {
          __syncthreads(); // every wave has screened its share: the list is complete
          const int ncoop = min(s_ncoop, kCoopMax);
          int t_here = tid;
          asm volatile("" : "+v"(t_here)); // made here: hoisted out of the item loop, `first` holds a register for the whole kernel
          const float x0f = static_cast<float>(x0), y0f = static_cast<float>(y0), first = static_cast<float>(t_here) + 0.5f;
          for (int b0 = 0; b0 < ncoop; b0 += kCoopBatch) {
            // set-up of a batch: wave w takes entries b0 + 8 w .. + 7, one per lane (every wave runs the set-up code once
            // per batch instead of once per triangle); the row states are parked in LDS, from where every lane of the
            // workgroup reads the triangle it is about to shade (one address per instruction: a broadcast)
            if (b0 > 0) __syncthreads(); // the previous batch has been read
            const int nb_here = min(kCoopBatch, ncoop - b0);
            constexpr int kPerWave = kCoopBatch / kRasterWaves;
            const int e = wave * kPerWave + lane;
            if (lane < kPerWave && e < nb_here) {
              const int f = s_coop[b0 + e];
              const uint4 pre = pre_n[f];
              TriSetup<T> s = {};
              const bool valid = tri_setup<T>(v_n, vi_n + int64_t(f) * 3, H, W, s);
              const TriRow<T> o = make_row_state<T>(valid, s, f, pre.z, x0, y0, x1, y1);
              uint32_t w[kTriWords];
              __builtin_memcpy(w, &o, sizeof(o));
#pragma unroll
              for (int j = 0; j < kTriWords; ++j) s_tri[e * kTriWords + j] = w[j];
            }
            __syncthreads();
            for (int i = 0; i < nb_here; ++i) {
              uint32_t w[kTriWords];
#pragma unroll
              for (int j = 0; j < kTriWords; ++j) w[j] = s_tri[i * kTriWords + j];
              TriRow<T> u;
              __builtin_memcpy(&u, w, sizeof(u));
              if (!DRTK_DBG(dbg, 1)) {
                if (phase == 0) {
                  shade_rows<T, TILE_SHIFT, false, FM, kRasterBlock>(u, first, x0f, y0f, zbuf, dbg);
                } else {
                  shade_rows<T, TILE_SHIFT, true, FM, kRasterBlock>(u, first, x0f, y0f, zbuf, dbg);
                }
              }
            }
          }
          if (ncoop > 0) {
            __syncthreads(); // all reads of the list done before it is reset (the barriers below order the reset before the next pushes)
            if (tid == 0) s_ncoop = 0;
          }
        }
        if (phase == 1) break;
        __syncthreads();
        if (tid == 0) DRTK_PHASE(4); // wave 0 waiting for the other waves' first group
        // farthest depth of every 8x8 block: a wave takes a row of blocks, lane = pixel column, eight rows per lane, then
        // the maximum over each group of 8 lanes (xor 1, xor 2 within quads, mirror of the 8-lane half row: DPP, no LDS)
        for (int by = wave; by < nb; by += kRasterWaves) {
          uint32_t m = 0;
          int l_here = lane;
          asm volatile("" : "+v"(l_here)); // the LDS addresses below are made here (hoisted out of the item loop they are spilled)
          if (lane < ss) {
#pragma unroll
            for (int r = 0; r < 8; ++r) m = max(m, static_cast<uint32_t>(zbuf[(((by << 3) + r) << TILE_SHIFT) + l_here] >> 32));
          }
          m = max(m, static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(m), 0xB1, 0xF, 0xF, true)));  // quad_perm [1,0,3,2]
          m = max(m, static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(m), 0x4E, 0xF, 0xF, true)));  // quad_perm [2,3,0,1]
          m = max(m, static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(m), 0x141, 0xF, 0xF, true))); // row_half_mirror
          if ((lane & 7) == 0 && lane < ss) s_zmax[by * nb + (l_here >> 3)] = m;
        }
        __syncthreads();
        if (tid == 0) DRTK_PHASE(5); // block-farthest reduction
      }
      __syncthreads();
      if (tid == 0) DRTK_PHASE(7); // wave 0 waiting for the other waves' second group

      // unpack + store (rasterize_kernel.cu:402-415); the tile is cleared for the next item as it is read, and the next
      // item is requested now, so that the queue's round trip runs under the stores
      if (tid == 0) pop();
      for (int q = tid; q < rows * quads_per_row; q += kRasterBlock) {
        const int row = q / quads_per_row;
        const int col = (q - row * quads_per_row) << 2;
        const int y = y0 + row, x = x0 + col;
        if (x > x1) continue;
        int32_t idx4[4];
        float dep4[4];
        unsigned long long cleared = ~0ull;
        asm volatile("" : "+v"(cleared)); // a constant made here, per quad: hoisted out of the item loop it is spilled across it
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned long long pv = zbuf[(row << TILE_SHIFT) + col + j];
          zbuf[(row << TILE_SHIFT) + col + j] = cleared;
          const uint32_t hi = static_cast<uint32_t>(pv >> 32);
          dep4[j] = (hi == 0xFFFFFFFFu) ? 0.0f : __uint_as_float(hi);
          idx4[j] = static_cast<int32_t>(static_cast<uint32_t>(pv & 0xFFFFFFFFu));
        }
        const int64_t o = img_base + int64_t(y) * W + x;
        if (vec_ok || x + 3 <= x1) { // W % 4 == 0 (x % 4 == 0 -> x + 3 < W), or a quad that ends inside the row
          *reinterpret_cast<IQuad*>(index_img + o) = IQuad{idx4[0], idx4[1], idx4[2], idx4[3]};
          *reinterpret_cast<FQuad*>(depth_img + o) = FQuad{dep4[0], dep4[1], dep4[2], dep4[3]};
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (x + j <= x1) {
              index_img[o + j] = idx4[j];
              depth_img[o + j] = dep4[j];
            }
          }
        }
      }
    }
    if (tid == 0) DRTK_PHASE(8); // unpack + store (issue)
    __syncthreads(); // zbuf and s_item are reused by the next item
    if (tid == 0) DRTK_PHASE(9); // waiting for the other waves' stores
  }
}

// Diagnostics: exact_div against the IEEE division on pseudo-random operands.
template <typename T>
__global__ __launch_bounds__(kBlock) void exact_div_selftest_kernel(
    unsigned long long seed, long long count, unsigned long long* __restrict__ mismatches) {
  const long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x;
  unsigned long long local = 0;
  for (long long k = i; k < count; k += static_cast<long long>(gridDim.x) * kBlock) {
    // splitmix64
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * static_cast<unsigned long long>(k + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    unsigned long long w = z * 0xD6E8FEB86659FD93ull;
    w ^= w >> 32;
    T n, d;
    if constexpr (sizeof(T) == 4) {
      // random mantissas (every 8th: all-ones / single-bit patterns), exponents spread over ~2^+-40
      uint32_t md = static_cast<uint32_t>(z) & 0x7FFFFFu, mn = static_cast<uint32_t>(z >> 23) & 0x7FFFFFu;
      if ((k & 7) == 0) md = 0x7FFFFFu >> ((z >> 50) & 7);
      if ((k & 7) == 1) mn = 0x7FFFFFu << ((z >> 53) & 7) & 0x7FFFFFu;
      const int ed = 127 + static_cast<int>((w >> 8) % 81) - 40, en = ed - static_cast<int>((w >> 20) % 30);
      d = __uint_as_float((static_cast<uint32_t>(ed) << 23) | md);
      n = __uint_as_float((static_cast<uint32_t>(en > 1 ? en : 1) << 23) | mn);
    } else {
      unsigned long long md = z & 0xFFFFFFFFFFFFFull, mn = w & 0xFFFFFFFFFFFFFull;
      if ((k & 7) == 0) md = 0xFFFFFFFFFFFFFull >> ((z >> 55) & 15);
      const int ed = 1023 + static_cast<int>((w >> 52) % 81) - 40, en = ed - static_cast<int>((z >> 52) % 60);
      d = __longlong_as_double((static_cast<unsigned long long>(ed) << 52) | md);
      n = __longlong_as_double((static_cast<unsigned long long>(en) << 52) | mn);
    }
    const T r = T(1) / d;
    const T fast = exact_div(n, d, r, exact_div_ok(d));
    const T ref = n / d;
    if (!(fast == ref)) ++local;
    if (!(exact_rcp(d) == T(1) / d)) ++local;
    if constexpr (sizeof(T) == 4) {
      // exhaustive over significands: pair k < 2^23 x 8 exponents
      if (k < (1ll << 26)) {
        const int ex[8] = {127, 126, 128, 100, 160, 60, 190, 127 - 90};
        const float x = __uint_as_float((static_cast<uint32_t>(ex[k >> 23]) << 23) | static_cast<uint32_t>(k & 0x7FFFFF));
        if (!(exact_rcp(x) == 1.0f / x)) ++local;
        if (!(exact_rcp(-x) == 1.0f / -x)) ++local;
      }
    }
  }
  if (local) atomicAdd(mismatches, local);
}

template <typename T>
int rasterize_impl(
    const T* v, const int32_t* vi, int64_t N, int64_t V, int64_t F, int64_t vi_sN, int64_t H,
    int64_t W, float* depth_img, int32_t* index_img, void* workspace, size_t workspace_bytes,
    hipStream_t stream) {
  const BinLayout L = make_layout(N, F, H, W);
  if (workspace_bytes < L.total_bytes) return DRTK_ERR_WORKSPACE_TOO_SMALL;
  if (N * H * W == 0) return DRTK_OK;
  char* ws = static_cast<char*>(workspace);
  auto* tile_count = reinterpret_cast<unsigned long long*>(ws + L.off_count);
  auto* tile_cursor = reinterpret_cast<unsigned long long*>(ws + L.off_cursor);
  auto* view_stats = reinterpret_cast<float*>(ws + L.off_view_stats);
  auto* big_count = reinterpret_cast<int32_t*>(ws + L.off_big_count);
  auto* tile_offset = reinterpret_cast<int32_t*>(ws + L.off_offset);
  auto* tri_range = reinterpret_cast<uint2*>(ws + L.off_range);
  auto* tri_pre = reinterpret_cast<uint4*>(ws + L.off_pre);
  auto* big_list = reinterpret_cast<int32_t*>(ws + L.off_big_list);
  auto* pairs = reinterpret_cast<int32_t*>(ws + L.off_pairs);
  auto* items = reinterpret_cast<uint32_t*>(ws + L.off_items);
  auto* queue = reinterpret_cast<int32_t*>(ws + L.off_queue);

  if (fill_bytes_async(ws, 0, L.zero_bytes, stream) != DRTK_OK) return DRTK_ERR_LAUNCH;
  const int64_t total = N * F;
  const dim3 tri_grid(static_cast<unsigned>(ceil_div(F > 0 ? F : 1, kBlock)), static_cast<unsigned>(N));
  if (total > 0) {
    DRTK_LAUNCH(
        bin_count_kernel<T>, tri_grid, dim3(kBlock), 0, stream, v, vi, (int)F, V, vi_sN, (int)H,
        (int)W, L.tile_shift, L.tiles_x, (int)L.tiles_per_view, tile_count, big_count, big_list,
        tri_range, tri_pre, view_stats);
    DRTK_RETURN_IF_LAUNCH_FAILED();
  }
  DRTK_LAUNCH(
      bin_scan_kernel, dim3(1), dim3(1024), 0, stream, tile_count, tile_offset, (int)L.num_tiles,
      L.tile_shift, split_threshold(N, F), items, queue);
  DRTK_RETURN_IF_LAUNCH_FAILED();
  if (total > 0) {
    DRTK_LAUNCH(
        bin_fill_kernel, tri_grid, dim3(kBlock), 0, stream, tri_range, (int)F, L.tiles_x,
        (int)L.tiles_per_view, tile_offset, tile_cursor, pairs);
    DRTK_RETURN_IF_LAUNCH_FAILED();
  }
  // persistent workgroups pulling work items: as many as can be resident, never more than items
#ifdef DRTK_RASTER_BLOCKS_PER_CU
  const int64_t resident = int64_t(num_compute_units()) * DRTK_RASTER_BLOCKS_PER_CU;
#else
  const int64_t resident = int64_t(num_compute_units()) * (raster_waves_per_simd<T>() / 2);
#endif
  const unsigned blocks = static_cast<unsigned>(std::min<int64_t>(L.max_items, resident));
#define DRTK_RASTER_LAUNCH(SHIFT, FM)                                                                              \
  DRTK_LAUNCH(                                                                                                    \
      (tile_raster_kernel<T, SHIFT, FM>), dim3(blocks), dim3(kRasterBlock), 0, stream, v, vi, (int)F, V, vi_sN,   \
      (int)H, (int)W, L.tiles_x, (int)L.tiles_per_view, tile_offset, tile_count, view_stats, pairs, big_count,    \
      big_list, tri_range, tri_pre, items, queue, depth_img, index_img, debug_flags())
  const bool fm = depth_order_setting() == DRTK_DEPTH_ORDER_FASTMATH;
  if (L.tile_shift == 6) {
    if (fm) DRTK_RASTER_LAUNCH(6, true); else DRTK_RASTER_LAUNCH(6, false);
  } else {
    if (fm) DRTK_RASTER_LAUNCH(5, true); else DRTK_RASTER_LAUNCH(5, false);
  }
#undef DRTK_RASTER_LAUNCH
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

} // namespace
} // namespace drtk_amd

namespace drtk_amd {
int rasterize_lines_dispatch( // rasterize_lines.hip
    drtk_dtype_t dtype, const void* v, const int32_t* vi, int64_t N, int64_t V, int64_t F, int64_t vi_sN, int64_t H,
    int64_t W, float* depth_img, int32_t* index_img, void* workspace, size_t workspace_bytes, hipStream_t stream);
}
using namespace drtk_amd;

extern "C" int drtk_amd_rasterize_workspace_bytes(
    int64_t N, int64_t F, int64_t H, int64_t W, size_t* bytes) {
  if (!bytes || N < 0 || F < 0 || H <= 0 || W <= 0) return DRTK_ERR_INVALID_ARGUMENT;
  *bytes = make_layout(N, F, H, W).total_bytes;
  return DRTK_OK;
}

extern "C" int drtk_amd_rasterize(
    drtk_dtype_t dtype, const void* v, const int32_t* vi, int64_t N, int64_t V, int64_t F,
    int64_t vi_sN, int64_t H, int64_t W, int wireframe, float* depth_img, int32_t* index_img,
    void* workspace, size_t workspace_bytes, drtk_stream_t stream) {
  if (N < 0 || V < 0 || F < 0 || H <= 0 || W <= 0) return DRTK_ERR_INVALID_ARGUMENT; // :464-468
  // 64-bit counters / packed pixels updated with 64-bit atomics live in the workspace (include/drtk_amd.h, alignment)
  if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return DRTK_ERR_INVALID_ARGUMENT;
  if (V >= 0x10000000LL) return DRTK_ERR_TOO_MANY_VERTICES;                           // :459-462
  if (dtype != DRTK_F32 && dtype != DRTK_F64) return DRTK_ERR_INVALID_ARGUMENT;
  if (vi_sN != 0 && vi_sN != F * 3) return DRTK_ERR_INVALID_ARGUMENT;
  // more views than one launch takes: slices, each reusing the workspace (sized for all N: enough for any slice)
  DRTK_FOR_VIEW_SLICES(N, n0, n, drtk_amd_rasterize(
      dtype, advance(v, n0 * V * 3, dtype_size(dtype)), advance_typed(vi, n0 * vi_sN), n, V, F, vi_sN, H, W, wireframe,
      advance_typed(depth_img, n0 * H * W), advance_typed(index_img, n0 * H * W), workspace, workspace_bytes, stream))
  if (wireframe) {
    if (N * H * W >= (int64_t(1) << 40)) return DRTK_ERR_INVALID_ARGUMENT;
    if (N * H * W > 0 && (!depth_img || !index_img)) return DRTK_ERR_INVALID_ARGUMENT;
    // Triangles need vertices: with F > 0 an empty `v` (V == 0, no storage) is rejected here on purpose -- every
  // non-degenerate index would be read off a null base (like the reference, the kernels do not bounds-check vi).
  if (N * F > 0 && (!v || !vi)) return DRTK_ERR_INVALID_ARGUMENT;
    if (vi_sN != 0 && vi_sN != F * 3) return DRTK_ERR_INVALID_ARGUMENT;
    return rasterize_lines_dispatch(dtype, v, vi, N, V, F, vi_sN, H, W, depth_img, index_img, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
  }
  if (H > 65535LL * 32 || W > 65535LL * 32 || N * F >= (int64_t(1) << 31) / kMaxSmallTiles ||
      N * H * W >= (int64_t(1) << 40))
    return DRTK_ERR_INVALID_ARGUMENT;
  if (N * H * W > 0 && (!depth_img || !index_img || !workspace)) return DRTK_ERR_INVALID_ARGUMENT;
  // Triangles need vertices: with F > 0 an empty `v` (V == 0, no storage) is rejected here on purpose -- every
  // non-degenerate index would be read off a null base (like the reference, the kernels do not bounds-check vi).
  if (N * F > 0 && (!v || !vi)) return DRTK_ERR_INVALID_ARGUMENT;
  if (vi_sN != 0 && vi_sN != F * 3) return DRTK_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case DRTK_F32:
      return rasterize_impl<float>(static_cast<const float*>(v), vi, N, V, F, vi_sN, H, W, depth_img, index_img, workspace, workspace_bytes, s);
    case DRTK_F64:
      return rasterize_impl<double>(static_cast<const double*>(v), vi, N, V, F, vi_sN, H, W, depth_img, index_img, workspace, workspace_bytes, s);
    default:
      return DRTK_ERR_INVALID_ARGUMENT;
  }
}

extern "C" int drtk_amd_selftest_exact_div(
    drtk_dtype_t dtype, uint64_t seed, int64_t count, uint64_t* d_mismatches, drtk_stream_t stream) {
  if (!d_mismatches || count < 0) return DRTK_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (fill_bytes_async(d_mismatches, 0, sizeof(uint64_t), s) != DRTK_OK) return DRTK_ERR_LAUNCH;
  const unsigned blocks = 4096;
  if (dtype == DRTK_F32) {
    DRTK_LAUNCH((exact_div_selftest_kernel<float>), dim3(blocks), dim3(kBlock), 0, s, seed, (long long)count, (unsigned long long*)d_mismatches);
  } else if (dtype == DRTK_F64) {
    DRTK_LAUNCH((exact_div_selftest_kernel<double>), dim3(blocks), dim3(kBlock), 0, s, seed, (long long)count, (unsigned long long*)d_mismatches);
  } else {
    return DRTK_ERR_INVALID_ARGUMENT;
  }
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}
