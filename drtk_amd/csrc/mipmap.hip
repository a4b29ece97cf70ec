// mipmap_grid_sampler_2d -- grid_sample with trilinear mip selection and anisotropic taps, forward
// and backward.
//
// Reference: src/mipmap_grid_sampler/mipmap_grid_sampler_kernel.cu:20-897 (device code), built on
// PyTorch's grid-sampler primitives (ATen/native/cuda/GridSampler.cuh, UpSample.cuh -- unnormalize,
// clip / reflect, safe_downgrade_to_int_range, cubic convolution with A = -0.75).
//
// Reference shape: one thread per output pixel; for every tap and mip level it walks the channels
// and read-modify-writes the *global* output once per corner (4 RMW per tap, level and channel).
// Here: lane = output pixel as well (the uv field is pixel-ordered, so grid / Jacobian loads are one
// 8- and one 16-byte vector per lane and the channel planes of the output are written coalesced),
// but
//   * the tap geometry (source index, padding, corner offsets and weights -- or the 4+4 bicubic
//     coordinates) is computed once per (tap, level) and reused by all channels,
//   * channels accumulate in registers in blocks of four and every output element is stored once,
//   * the level table lives in LDS (levels are selected per lane, so a kernel-argument array would
//     be indexed through scratch memory).
// Accumulation order per output element is the reference's (taps, then level d1 / d1+1, then
// corners), so results match the CPU restatement to rounding of the transcendental log2 only.
// The forward pass ignores align_corners exactly like the reference (:423 forces it to false);
// the backward pass honours it (:641 ff).
#include <type_traits>

#include "common.hpp"
#include "segscatter.hpp" // wave_lds_sync

#ifdef DRTK_AMD_ABLATION
// Profiling build only: what the tiled backward's rounds are made of (profiles/mipmap_bench.py --rounds-stats).
// [0] tiles with upstream gradient, [r] tiles that enter round r (1..7), [8] (tap, level) pairs walked in the first pass,
// [9] pairs left pending by it, [10] pairs that end in global memory after the last round
namespace drtk_amd {
__device__ unsigned long long g_mip_stats[16];
}
extern "C" __attribute__((visibility("default"))) int drtk_amd_debug_read_mip_stats(unsigned long long* out) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(drtk_amd::g_mip_stats), sizeof(unsigned long long) * 16) != hipSuccess) return -3;
  unsigned long long z[16] = {};
  if (hipMemcpyToSymbol(HIP_SYMBOL(drtk_amd::g_mip_stats), z, sizeof(z)) != hipSuccess) return -3;
  return 0;
}
#define DRTK_MIP_STAT(i, v) do { if (DRTK_DBG(dbg, 64)) atomicAdd(&::drtk_amd::g_mip_stats[i], static_cast<unsigned long long>(v)); } while (0)
// ... and WHERE the pairs are that end in global memory (flag 64 as well): {view << 24 | tile, thread << 16 | level, x, y} of
// the pair's north-west texel, the first 2^20 of them (profiles/mipmap_bench.py --leftover-dump)
namespace drtk_amd {
constexpr unsigned kMipDumpMax = 1u << 20;
__device__ unsigned int g_mip_dump_n;
__device__ uint4 g_mip_dump[kMipDumpMax];
}
extern "C" __attribute__((visibility("default"))) int drtk_amd_debug_read_mip_dump(unsigned int* out, unsigned int* n) {
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(drtk_amd::g_mip_dump_n), sizeof(unsigned int)) != hipSuccess) return -3;
  if (*n > drtk_amd::kMipDumpMax) *n = drtk_amd::kMipDumpMax;
  if (*n && hipMemcpyFromSymbol(out, HIP_SYMBOL(drtk_amd::g_mip_dump), sizeof(uint4) * *n) != hipSuccess) return -3;
  const unsigned int z = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(drtk_amd::g_mip_dump_n), &z, sizeof(z)) != hipSuccess) return -3;
  return 0;
}
#define DRTK_MIP_DUMP_IF(flag, a, b, c, d) do { if (DRTK_DBG(dbg, flag)) { const unsigned int k_ = atomicAdd(&::drtk_amd::g_mip_dump_n, 1u); \
  if (k_ < ::drtk_amd::kMipDumpMax) ::drtk_amd::g_mip_dump[k_] = make_uint4(a, b, c, d); } } while (0)
#define DRTK_MIP_DUMP(a, b, c, d) DRTK_MIP_DUMP_IF(64, a, b, c, d)
// flag 1 << 20: a tile's timeline instead -- {view << 24 | tile, further rounds, start, end} in 10 ns ticks (--tile-times)
#define DRTK_MIP_TILE_T0() const unsigned int tile_t0_ = static_cast<unsigned int>(wall_clock64()); unsigned int tile_ph_[3] = {0u, 0u, 0u}
#define DRTK_MIP_TILE_PHASE(k) tile_ph_[k] = static_cast<unsigned int>(wall_clock64()) - tile_t0_
// flag 1 << 21: the phases of a tile instead -- {.., rounds | inputs there << 16, windows placed | taps done << 16, end}, ticks since the start
#define DRTK_MIP_TILE_DONE(rounds) do { if (tid == 0) { \
  DRTK_MIP_DUMP_IF(1 << 20, static_cast<unsigned>(n) << 24 | static_cast<unsigned>(tile), rounds, tile_t0_, static_cast<unsigned int>(wall_clock64())); \
  DRTK_MIP_DUMP_IF(1 << 21, static_cast<unsigned>(n) << 24 | static_cast<unsigned>(tile), static_cast<unsigned>(rounds) | tile_ph_[0] << 16, tile_ph_[1] | tile_ph_[2] << 16, static_cast<unsigned int>(wall_clock64()) - tile_t0_); } } while (0)
#else
#define DRTK_MIP_STAT(i, v) do { } while (0)
#define DRTK_MIP_DUMP(a, b, c, d) do { } while (0)
#define DRTK_MIP_TILE_T0() do { } while (0)
#define DRTK_MIP_TILE_PHASE(k) do { } while (0)
#define DRTK_MIP_TILE_DONE(rounds) do { } while (0)
#endif

namespace drtk_amd {
namespace {

constexpr int kMaxLevels = 11; // mipmap_grid_sampler_kernel.cu:16

struct LevelTable {
  const void* ptr[kMaxLevels];
  void* grad[kMaxLevels];
  long long sn[kMaxLevels]; // elements between the views of a level (0: one texture shared by all views)
  int h[kMaxLevels];
  int w[kMaxLevels];
};

// Correctly rounded (the footprint lengths feed a floor() that picks the mip level).  Not `__fsqrt_rn`, which this
// toolchain defines as the native 1-ulp square root -- see edge_grad.hip: sqrt_t.
__device__ __forceinline__ float sqrt_rn(float x) {
  return __builtin_sqrtf(x);
}
__device__ __forceinline__ double sqrt_rn(double x) {
  return __builtin_sqrt(x);
}

// GridSampler.cuh primitives -------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T unnormalize(T coord, int size, bool align_corners, T* grad_in) {
  if (align_corners) {
    *grad_in = static_cast<T>(size - 1) / 2;
    return ((coord + 1.f) / 2) * (size - 1);
  }
  *grad_in = static_cast<T>(size) / 2;
  return ((coord + 1.f) * size - 1) / 2;
}
template <typename T>
__device__ __forceinline__ T clip_coord(T in, int limit, T* grad_in) {
  if (in <= T(0)) {
    *grad_in = T(0);
    return T(0);
  }
  const T mx = static_cast<T>(limit - 1);
  if (in >= mx) {
    *grad_in = T(0);
    return mx;
  }
  *grad_in = T(1);
  return in;
}
template <typename T>
__device__ __forceinline__ T clip_plain(T in, int limit) { // ::min(limit-1, ::max(in, 0))
  const T hi = static_cast<T>(limit - 1);
  const T lo = in > T(0) ? in : T(0);
  return hi < lo ? hi : lo;
}
template <typename T>
__device__ __forceinline__ T reflect_coord(T in, int twice_low, int twice_high, T* grad_in) {
  if (twice_low == twice_high) {
    *grad_in = T(0);
    return T(0);
  }
  int mult = 1;
  const T mn = static_cast<T>(twice_low) / 2;
  const T span = static_cast<T>(twice_high - twice_low) / 2;
  in = in - mn;
  if (in < T(0)) {
    mult = -1;
    in = -in;
  }
  const T extra = fmod(in, span);
  const int flips = static_cast<int>(floor(in / span));
  if (flips % 2 == 0) {
    *grad_in = static_cast<T>(mult);
    return extra + mn;
  }
  *grad_in = static_cast<T>(-mult);
  return span - extra + mn;
}
__device__ __forceinline__ float tmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double tmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float tmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double tmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float tfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double tfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float tabs(float a) { return fabsf(a); }
__device__ __forceinline__ double tabs(double a) { return fabs(a); }
// reflect_coord for the lean kernels: the same value and multiplier, the long form only when a lane of the wave needs it.
// With a = |in - low| < span no flip occurs: fmod(a, span) == a exactly and floor(a / span) == 0, so the reference's
// `extra + low` is `a + low` -- two additions instead of an fmod, a division and a floor.  (Taps further than one texture
// width beyond the border, or non-finite: the wave takes reflect_coord itself.)
template <typename T>
__device__ __forceinline__ T reflect_coord_lean(T in, int twice_low, int twice_high, T* grad_in) {
  if (twice_low == twice_high) {
    *grad_in = T(0);
    return T(0);
  }
  const T mn = static_cast<T>(twice_low) / 2;
  const T span = static_cast<T>(twice_high - twice_low) / 2;
  T a = in - mn;
  T m = T(1);
  if (a < T(0)) m = T(-1), a = -a;
  if (__ballot(!(a < span)) != 0) return reflect_coord(in, twice_low, twice_high, grad_in);
  *grad_in = m;
  return a + mn;
}
// ... followed by the clamp to the level (clip_coord: a clamped coordinate has a zero gradient), both axes of a tap
template <typename T>
__device__ __forceinline__ void reflect_clip_lean(T ixu, T iyu, int w, int h, bool align_corners, T& ix, T& iy, T& mx, T& my) {
  T gx, gy;
  const T rx = align_corners ? reflect_coord_lean(ixu, 0, 2 * (w - 1), &gx) : reflect_coord_lean(ixu, -1, 2 * w - 1, &gx);
  const T ry = align_corners ? reflect_coord_lean(iyu, 0, 2 * (h - 1), &gy) : reflect_coord_lean(iyu, -1, 2 * h - 1, &gy);
  const T wm1 = static_cast<T>(w - 1), hm1 = static_cast<T>(h - 1);
  mx = ((rx > T(0)) & (rx < wm1)) ? mx * gx : T(0), my = ((ry > T(0)) & (ry < hm1)) ? my * gy : T(0);
  ix = tmin(tmax(rx, T(0)), wm1), iy = tmin(tmax(ry, T(0)), hm1);
}
template <typename T>
__device__ __forceinline__ T safe_int_range(T x) {
  if (x > static_cast<T>(INT32_MAX - 1) || x < static_cast<T>(INT32_MIN) || !isfinite(static_cast<double>(x)))
    return T(-100.0);
  return x;
}
template <typename T>
__device__ __forceinline__ T compute_coordinates(T coord, int size, int padding, bool align_corners) {
  T unused;
  if (padding == 1) {
    coord = clip_plain(coord, size);
  } else if (padding == 2) {
    coord = align_corners ? reflect_coord(coord, 0, 2 * (size - 1), &unused) : reflect_coord(coord, -1, 2 * size - 1, &unused);
    coord = clip_plain(coord, size);
  }
  return safe_int_range(coord);
}
template <typename T>
__device__ __forceinline__ T source_index(T coord, int size, int padding, bool align_corners, T* grad_in) {
  T g_un, g_clip = T(1), g_refl = T(1);
  coord = unnormalize(coord, size, align_corners, &g_un);
  if (padding == 1) {
    coord = clip_coord(coord, size, &g_clip);
    g_un = g_un * g_clip;
  } else if (padding == 2) {
    coord = align_corners ? reflect_coord(coord, 0, 2 * (size - 1), &g_refl) : reflect_coord(coord, -1, 2 * size - 1, &g_refl);
    coord = clip_coord(coord, size, &g_clip);
    g_un = g_un * g_refl * g_clip;
  }
  *grad_in = g_un;
  return safe_int_range(coord);
}
template <typename T>
__device__ __forceinline__ void cubic_coeffs(T co[4], T t) { // UpSample.cuh get_cubic_upsampling_coefficients
  const T A = T(-0.75);
  T x = t + T(1.0);
  co[0] = ((A * x - 5 * A) * x + 8 * A) * x - 4 * A;
  x = t;
  co[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  x = T(1.0) - t;
  co[2] = ((A + 2) * x - (A + 3)) * x * x + 1;
  x = x + T(1.0);
  co[3] = ((A * x - 5 * A) * x + 8 * A) * x - 4 * A;
}
template <typename T>
__device__ __forceinline__ void cubic_coeffs_grad(T co[4], T t) { // grid_utils.h:130-144
  const T A = T(-0.75);
  T x = -1 - t;
  co[0] = (-3 * A * x - 10 * A) * x - 8 * A;
  x = -t;
  co[1] = (-3 * (A + 2) * x - 2 * (A + 3)) * x;
  x = 1 - t;
  co[2] = (3 * (A + 2) * x - 2 * (A + 3)) * x;
  x = 2 - t;
  co[3] = (3 * A * x - 10 * A) * x + 8 * A;
}

// Per-pixel tap setup: mipmap_grid_sampler_kernel.cu:441-508 (== :679-746 in the backward kernel).
template <typename T>
struct Taps {
  T u, v, du, dv, a;
  int d1, n;
};
// the pixel's uv and uv Jacobian: one 8- and one 16-byte load (float)
template <typename T>
struct PixelUV {
  T u, v, dudx, dvdx, dudy, dvdy;
};
// Where element (n, pixel, c) of a uv field [N,H,W,2] lives: n * sN + pixel * sP + c * sC elements from the base (pixel =
// y * W + x).  Contiguous: (2HW, 2, 1), read and written as one 8- / 16-byte pair per pixel (`pair`); the channel-first
// image `interpolate` produces, seen through permute(0, 2, 3, 1): (2HW, 1, HW) -- two coalesced loads, no copy.
struct GridLayout {
  long long sN, sP, sC;
  bool pair;
};
template <typename T>
__device__ __forceinline__ PixelUV<T> load_pixel_uv(
    const T* __restrict__ grid, const GridLayout& gl, const T* __restrict__ vt, int64_t n, int64_t pix, int64_t index) {
  PixelUV<T> p;
  const T* g = grid + n * gl.sN + pix * gl.sP;
  if (gl.pair) {
    using V2 = typename std::conditional<sizeof(T) == 4, float2, double2>::type;
    const V2 q = *reinterpret_cast<const V2*>(g);
    p.u = q.x, p.v = q.y;
  } else {
    p.u = g[0], p.v = g[gl.sC];
  }
  if constexpr (sizeof(T) == 4) {
    const float4 j = *reinterpret_cast<const float4*>(vt + index * 4);
    p.dudx = j.x, p.dvdx = j.y, p.dudy = j.z, p.dvdy = j.w;
  } else {
    const double2 j0 = *reinterpret_cast<const double2*>(vt + index * 4);
    const double2 j1 = *reinterpret_cast<const double2*>(vt + index * 4 + 2);
    p.dudx = j0.x, p.dvdx = j0.y, p.dudy = j1.x, p.dvdy = j1.y;
  }
  return p;
}
template <typename T>
__device__ __forceinline__ void store_grid_grad(T* __restrict__ gg, const GridLayout& gl, int64_t n, int64_t pix, T gx, T gy) {
  T* g = gg + n * gl.sN + pix * gl.sP;
  if (gl.pair) {
    using V2 = typename std::conditional<sizeof(T) == 4, float2, double2>::type;
    *reinterpret_cast<V2*>(g) = V2{gx, gy};
  } else {
    g[0] = gx, g[gl.sC] = gy;
  }
}
template <typename T>
__device__ __forceinline__ Taps<T> setup_taps(
    const PixelUV<T>& uv, int inp_H, int inp_W, int mipmaps, int max_aniso, bool force_max_aniso, bool clip_grad) {
  Taps<T> t;
  t.u = uv.u, t.v = uv.v;
  T dudx = uv.dudx, dvdx = uv.dvdx, dudy = uv.dudy, dvdy = uv.dvdy;
  // footprint lengths (:455-456, written with pow there): sqrt(a*a + b*b + 1e-12) with IEEE multiply and
  // sqrt -- the tap count below is a discontinuous function of them, so they are evaluated in the one
  // form that is bit-reproducible everywhere (profiles/NOTES.md §3.7)
  const T ax = fabs(dudx * inp_W), bx = fabs(dvdx * inp_H);
  const T ay = fabs(dudy * inp_W), by = fabs(dvdy * inp_H);
  const T px = sqrt_rn(ax * ax + bx * bx + 1e-12f);
  const T py = sqrt_rn(ay * ay + by * by + 1e-12f);
  const T p_max = px > py ? px : py;
  const T p_min = px < py ? px : py;
  T Nf = ceil(p_max / p_min);
  if (static_cast<T>(max_aniso) < Nf) Nf = static_cast<T>(max_aniso);
  if (p_min == 0.0 || Nf == 0) Nf = 1;
  T lambda_ = log2(p_max / Nf);
  if (isnan(lambda_) || isinf(lambda_)) lambda_ = 0.0f;
  const double lim = static_cast<double>(mipmaps - 1) - 1e-6;
  T l = static_cast<T>(static_cast<double>(lambda_) < lim ? static_cast<double>(lambda_) : lim);
  if (clip_grad && lambda_ > static_cast<T>(mipmaps - 1)) {
    const T p_max_corrected = exp2(l) * Nf;
    const T scaling = p_max_corrected / p_max;
    dudx *= scaling;
    dvdx *= scaling;
    dudy *= scaling;
    dvdy *= scaling;
  }
  l = static_cast<T>(static_cast<double>(l) > 0.0 ? static_cast<double>(l) : 0.0);
  t.d1 = static_cast<int>(floor(l));
  t.a = l - static_cast<T>(t.d1);
  t.n = force_max_aniso ? max_aniso : static_cast<int>(Nf);
  if (px > py) {
    t.du = dudx, t.dv = dvdx;
  } else {
    t.du = dudy, t.dv = dvdy;
  }
  return t;
}

// Bilinear corner geometry of one (tap, level): offsets (or -1 when out of bounds) and weights.
template <typename T>
struct Quad {
  int o_nw, o_ne, o_sw, o_se;
  T nw, ne, sw, se;
  T ix, iy;
  int ix_nw, iy_nw;
  T mx, my;
};
template <typename T>
__device__ __forceinline__ Quad<T> bilinear_quad(T x, T y, int H, int W, int padding, bool align_corners) {
  Quad<T> q;
  q.ix = source_index(x, W, padding, align_corners, &q.mx);
  q.iy = source_index(y, H, padding, align_corners, &q.my);
  q.ix_nw = static_cast<int>(floor(q.ix));
  q.iy_nw = static_cast<int>(floor(q.iy));
  const int ix_se = q.ix_nw + 1, iy_se = q.iy_nw + 1;
  q.nw = (ix_se - q.ix) * (iy_se - q.iy);
  q.ne = (q.ix - q.ix_nw) * (iy_se - q.iy);
  q.sw = (ix_se - q.ix) * (q.iy - q.iy_nw);
  q.se = (q.ix - q.ix_nw) * (q.iy - q.iy_nw);
  const bool x0 = q.ix_nw >= 0 && q.ix_nw < W, x1 = ix_se >= 0 && ix_se < W;
  const bool y0 = q.iy_nw >= 0 && q.iy_nw < H, y1 = iy_se >= 0 && iy_se < H;
  q.o_nw = (x0 && y0) ? q.iy_nw * W + q.ix_nw : -1;
  q.o_ne = (x1 && y0) ? q.iy_nw * W + ix_se : -1;
  q.o_sw = (x0 && y1) ? iy_se * W + q.ix_nw : -1;
  q.o_se = (x1 && y1) ? iy_se * W + ix_se : -1;
  return q;
}

// Bicubic footprint of one (tap, level): 4 column and 4 row indices after padding (-1: zero tap).
template <typename T>
struct Cubic {
  int xi[4], yi[4];
  T tx, ty, mx, my;
};
template <typename T>
__device__ __forceinline__ Cubic<T> bicubic_footprint(T x, T y, int H, int W, int padding, bool align_corners) {
  Cubic<T> c;
  const T ix = unnormalize(x, W, align_corners, &c.mx);
  const T iy = unnormalize(y, H, align_corners, &c.my);
  const T ix_nw = floor(ix), iy_nw = floor(iy);
  c.tx = ix - ix_nw;
  c.ty = iy - iy_nw;
  // Round 6: a footprint whose sixteen texels lie inside the level -- nearly all -- needs none of the eight padding
  // transforms below: clip, reflect and the integer-range guard map an in-range INTEGER coordinate to itself exactly
  // (reflection: (k + 1/2) - 1/2 with fmod(k + 1/2, size) == k + 1/2 and no flip), so the indices are nw - 1 ... nw + 2.
  // Each transform is ~20 instructions (branches on the padding mode, an isfinite through double): 160 per (tap, level).
  // (the forward kernel under reflection padding goes from four to three waves per SIMD with it -- 138 registers -- and is
  // still 17 % faster, 1.66 -> 1.38 ms: a reflected index is an fmod, a division and a floor)
  if (W >= 4 && H >= 4 && fabs(ix) < T(1e9) && fabs(iy) < T(1e9)) { // (the comparisons are false for NaN)
    const int kx = static_cast<int>(ix_nw), ky = static_cast<int>(iy_nw);
    if (static_cast<unsigned>(kx - 1) < static_cast<unsigned>(W - 3) && static_cast<unsigned>(ky - 1) < static_cast<unsigned>(H - 3)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) c.xi[i] = kx - 1 + i, c.yi[i] = ky - 1 + i;
      return c;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int xv = static_cast<int>(compute_coordinates<T>(ix_nw - 1 + i, W, padding, align_corners));
    const int yv = static_cast<int>(compute_coordinates<T>(iy_nw - 1 + i, H, padding, align_corners));
    c.xi[i] = (xv >= 0 && xv < W) ? xv : -1;
    c.yi[i] = (yv >= 0 && yv < H) ? yv : -1;
  }
  return c;
}

__device__ __forceinline__ void stage_levels(
    const LevelTable& lv, int mipmaps, const void** s_ptr, void** s_grad, int* s_h, int* s_w, long long* s_sn) {
  if (threadIdx.x < kMaxLevels) {
    const int i = threadIdx.x < mipmaps ? threadIdx.x : 0;
    s_ptr[threadIdx.x] = lv.ptr[i];
    s_grad[threadIdx.x] = lv.grad[i];
    s_sn[threadIdx.x] = lv.sn[i];
    s_h[threadIdx.x] = lv.h[i];
    s_w[threadIdx.x] = lv.w[i];
  }
  __syncthreads();
}

// Tap i of n sits at u + du * f, f = (i + 1.0) / (n + 1.0) * 2.0 - 1.0 in double (mipmap_grid_sampler_kernel.cu:497-499):
// a double division per tap and pixel.  The workgroup computes the table of f for n <= 8 once -- with the same device
// arithmetic, so the values are the same -- and the taps read it from LDS; larger n (max_aniso > 8) divide as before.
#ifndef DRTK_MIP_TAPTAB
#define DRTK_MIP_TAPTAB 8
#endif
constexpr int kTapTab = DRTK_MIP_TAPTAB;
static_assert(kTapTab * kTapTab <= kBlock, "one table entry per thread");
__device__ __forceinline__ void stage_tap_table(double* s_f) { // call before a __syncthreads()
  if (threadIdx.x < kTapTab * kTapTab) {
    const int i = threadIdx.x % kTapTab, n = threadIdx.x / kTapTab + 1;
    s_f[threadIdx.x] = (i + 1.0) / (n + 1.0) * 2.0 - 1.0;
  }
}
__device__ __forceinline__ double tap_f(const double* s_f, int i, int n) {
  return n <= kTapTab ? s_f[(n - 1) * kTapTab + i] : (i + 1.0) / (n + 1.0) * 2.0 - 1.0;
}

template <typename T>
using GlobalPtr = __attribute__((address_space(1))) T*;

// float/double atomic add through an explicitly global pointer (global_atomic_add_f32 / _f64)
template <typename T>
__device__ __forceinline__ void atomic_add_g1(GlobalPtr<T> p, T v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// float/double add on an explicitly LDS-typed pointer (ds_add_f32 / ds_add_f64, no return)
template <typename T>
__device__ __forceinline__ void lds_add(T* p, T v) {
  using LdsPtr = __attribute__((address_space(3))) T*;
  __hip_atomic_fetch_add((LdsPtr)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// two horizontally adjacent texels, loaded with one element-aligned 8/16-byte access
template <typename T>
struct PairOf;
template <>
struct PairOf<float> {
  typedef float type __attribute__((ext_vector_type(2), aligned(4)));
};
template <>
struct PairOf<double> {
  typedef double type __attribute__((ext_vector_type(2), aligned(8)));
};
template <typename T>
using Pair = typename PairOf<T>::type;

constexpr int kChBlock = 4; // channels accumulated in registers per sweep over the taps
// (Round 3, measured and not kept: the bilinear forward's taps in groups of 2 or 4 -- geometry of the group first, the
// texel loads of its interior (tap, level) pairs in flight together, products in tap order, bit-identical -- to cut the
// per-pixel chain quad -> texels -> products -> next tap: 2.01 / 2.09 ms against 0.76 on the textured benchmark.  The
// group's quads and texels take the kernel from 110 to 258 / 434 registers, and this kernel lives on occupancy.)

// PAD: the padding mode as a compile-time constant (see the tile kernels of the backward pass).
#ifndef DRTK_MIP_BICUBIC_FWD_OCC
#define DRTK_MIP_BICUBIC_FWD_OCC 6 // float bicubic: 76-78 registers (asked for seven it spills 2-3)
#endif
#ifndef DRTK_MIP_BICUBIC_FWD_LDS_LEVELS
#define DRTK_MIP_BICUBIC_FWD_LDS_LEVELS 1
#endif
template <typename T, int MODE, int PAD>
__global__ __launch_bounds__(kBlock, (MODE == 2 && sizeof(T) == 4 && (PAD != 2 || DRTK_MIP_BICUBIC_FWD_LDS_LEVELS)) ? DRTK_MIP_BICUBIC_FWD_OCC : 1) void mipmap_forward_kernel(
    LevelTable lv, int mipmaps, const T* __restrict__ grid, GridLayout gl, const T* __restrict__ vt, int64_t count, int C,
    int64_t HW, int max_aniso, bool force_max_aniso, bool clip_grad, T* __restrict__ out, int strip) {
  constexpr int padding = PAD;
  __shared__ const void* s_ptr[kMaxLevels];
  __shared__ void* s_grad[kMaxLevels];
  __shared__ int s_h[kMaxLevels], s_w[kMaxLevels];
  __shared__ long long s_sn[kMaxLevels];
  __shared__ double s_f[kTapTab * kTapTab];
  stage_tap_table(s_f);
  stage_levels(lv, mipmaps, s_ptr, s_grad, s_h, s_w, s_sn);
  const int64_t index = int64_t(tile_index(strip)) * kBlock + threadIdx.x;
  if (index >= count) return;
  const int64_t n = index / HW;
  const bool align_corners = false; // mipmap_grid_sampler_kernel.cu:423
  const Taps<T> t = setup_taps<T>(load_pixel_uv<T>(grid, gl, vt, n, index - n * HW, index), s_h[0], s_w[0], mipmaps, max_aniso, force_max_aniso, clip_grad);
  const int n_lv = mipmaps > 1 ? 2 : 1;
  T* out_px = out + n * C * HW + (index - n * HW);
  const T alpha_1 = t.a / t.n;
  const T alpha_2 = static_cast<T>((1.0 - t.a) / t.n);
  // the pixel's two levels: sizes and base pointers once, not per tap (the level pointers come back from LDS as generic
  // pointers: pin them to the global address space, otherwise every texel access is a flat_load)
  int lv_h[2], lv_w[2];
  GlobalPtr<const T> lv_base[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int d = s < n_lv ? t.d1 + s : 0;
    lv_h[s] = s_h[d], lv_w[s] = s_w[d];
    lv_base[s] = (GlobalPtr<const T>)(static_cast<const T*>(s_ptr[d]) + n * s_sn[d]);
  }

  for (int c0 = 0; c0 < C; c0 += kChBlock) {
    T acc[kChBlock];
#pragma unroll
    for (int cc = 0; cc < kChBlock; ++cc) acc[cc] = T(0);
    for (int i = 0; i < t.n; ++i) {
      const double f = tap_f(s_f, i, t.n);
      const T x = t.u + static_cast<T>(t.du * f), y = t.v + static_cast<T>(t.dv * f);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s >= n_lv) break;
        const T alpha = s == 0 ? alpha_2 : alpha_1;
        // A level whose weight is exactly zero -- the second level of every magnified pixel (a == 0) -- contributes
        // +-0 * texel to every channel: skipped, texels are taken to be finite (as in the backward pass).
        if (alpha == T(0)) continue;
#if DRTK_MIP_BICUBIC_FWD_LDS_LEVELS
        // (bicubic, round 6: the level's size and the view's base pointer re-read from the LDS table per (tap, level) instead
        // of living in registers across the tap loop, as in the lean forward: float 104-128 -> 76-78 registers = SIX waves per
        // SIMD instead of four, 1.15 -> 1.00 ms on the textured benchmark; double 144-184 -> 108-110 = four instead of 2-3)
        const bool reread = MODE == 2;
        const int d_s = t.d1 + s;
        const int h = reread ? s_h[d_s] : lv_h[s], w = reread ? s_w[d_s] : lv_w[s];
        const int64_t plane = int64_t(h) * w;
        const GlobalPtr<const T> base =
            (reread ? (GlobalPtr<const T>)(static_cast<const T*>(s_ptr[d_s]) + n * s_sn[d_s]) : lv_base[s]) + c0 * plane;
#else
        const int h = lv_h[s], w = lv_w[s];
        const int64_t plane = int64_t(h) * w;
        const GlobalPtr<const T> base = lv_base[s] + c0 * plane;
#endif
        if constexpr (MODE == 0) {
          const Quad<T> q = bilinear_quad<T>(x, y, h, w, padding, align_corners);
          if ((q.o_nw | q.o_ne | q.o_sw | q.o_se) >= 0) {
            // interior tap: two 8-byte loads and four products per channel, no per-corner branch
            Pair<T> top[kChBlock], bot[kChBlock];
#pragma unroll
            for (int cc = 0; cc < kChBlock; ++cc) {
              if (c0 + cc < C) {
                top[cc] = *(GlobalPtr<const Pair<T>>)(base + cc * plane + q.o_nw);
                bot[cc] = *(GlobalPtr<const Pair<T>>)(base + cc * plane + q.o_sw);
              }
            }
#pragma unroll
            for (int cc = 0; cc < kChBlock; ++cc) {
              if (c0 + cc < C) {
                acc[cc] += top[cc].x * q.nw * alpha;
                acc[cc] += top[cc].y * q.ne * alpha;
                acc[cc] += bot[cc].x * q.sw * alpha;
                acc[cc] += bot[cc].y * q.se * alpha;
              }
            }
            continue;
          }
#pragma unroll
          for (int cc = 0; cc < kChBlock; ++cc) {
            if (c0 + cc < C) {
              const GlobalPtr<const T> p = base + cc * plane;
              // the two texels of a row are adjacent in memory: one 8-byte (4-byte aligned) load per row
              T v_nw = T(0), v_ne = T(0), v_sw = T(0), v_se = T(0);
              if (q.o_nw >= 0 && q.o_ne >= 0) {
                const Pair<T> t2 = *(GlobalPtr<const Pair<T>>)(p + q.o_nw);
                v_nw = t2.x, v_ne = t2.y;
              } else {
                if (q.o_nw >= 0) v_nw = p[q.o_nw];
                if (q.o_ne >= 0) v_ne = p[q.o_ne];
              }
              if (q.o_sw >= 0 && q.o_se >= 0) {
                const Pair<T> t2 = *(GlobalPtr<const Pair<T>>)(p + q.o_sw);
                v_sw = t2.x, v_se = t2.y;
              } else {
                if (q.o_sw >= 0) v_sw = p[q.o_sw];
                if (q.o_se >= 0) v_se = p[q.o_se];
              }
              if (q.o_nw >= 0) acc[cc] += v_nw * q.nw * alpha;
              if (q.o_ne >= 0) acc[cc] += v_ne * q.ne * alpha;
              if (q.o_sw >= 0) acc[cc] += v_sw * q.sw * alpha;
              if (q.o_se >= 0) acc[cc] += v_se * q.se * alpha;
            }
          }
        } else {
          const Cubic<T> cb = bicubic_footprint<T>(x, y, h, w, padding, align_corners);
          T cx[4], cy[4];
          cubic_coeffs(cx, cb.tx);
          cubic_coeffs(cy, cb.ty);
          // interior footprint (sixteen consecutive texels inside the level, nearly all): a row is ONE 16-byte load
          // (element-aligned) instead of four predicated 4-byte ones; same products, same order
          const int bx = cb.xi[0];
          const bool interior = bx >= 0 && cb.xi[1] == bx + 1 && cb.xi[2] == bx + 2 && cb.xi[3] == bx + 3 &&
              (cb.yi[0] | cb.yi[1] | cb.yi[2] | cb.yi[3]) >= 0;
#pragma unroll
          for (int cc = 0; cc < kChBlock; ++cc) {
            if (c0 + cc < C) {
              const GlobalPtr<const T> p = base + cc * plane;
              T co[4];
              if (interior) {
                typedef T Quad4 __attribute__((ext_vector_type(4), aligned(sizeof(T))));
                Quad4 row[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) row[r] = *(GlobalPtr<const Quad4>)(p + cb.yi[r] * w + bx);
#pragma unroll
                for (int r = 0; r < 4; ++r) co[r] = row[r].x * cx[0] + row[r].y * cx[1] + row[r].z * cx[2] + row[r].w * cx[3];
              } else {
                // (a footprint on the border of its level, rare: row by row in a loop that is NOT unrolled -- the four rows'
                // sixteen predicated loads in flight at once were part of what this kernel's registers were for; same sums)
                T sum = T(0);
#pragma unroll 1
                for (int r = 0; r < 4; ++r) {
                  const int yr = r == 0 ? cb.yi[0] : r == 1 ? cb.yi[1] : r == 2 ? cb.yi[2] : cb.yi[3];
                  const T cyr = r == 0 ? cy[0] : r == 1 ? cy[1] : r == 2 ? cy[2] : cy[3];
                  T xv[4];
#pragma unroll
                  for (int k = 0; k < 4; ++k) xv[k] = (yr >= 0 && cb.xi[k] >= 0) ? p[yr * w + cb.xi[k]] : T(0);
                  const T cor = xv[0] * cx[0] + xv[1] * cx[1] + xv[2] * cx[2] + xv[3] * cx[3];
                  sum = r == 0 ? cor * cyr : sum + cor * cyr;
                }
                acc[cc] += sum * alpha;
                continue;
              }
              acc[cc] += (co[0] * cy[0] + co[1] * cy[1] + co[2] * cy[2] + co[3] * cy[3]) * alpha;
            }
          }
        }
      }
    }
#pragma unroll
    for (int cc = 0; cc < kChBlock; ++cc) {
      if (c0 + cc < C) out_px[int64_t(c0 + cc) * HW] = acc[cc];
    }
  }
}

// Wave-wide min / max of an int, every lane active: four DPP steps leave each 16-lane row's result in all of its lanes,
// four v_readlane + scalar min / max join the rows (a __shfl_xor ladder is six dependent ds_bpermute round trips).
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, dpp_i32<0xB1>(v));  // quad_perm [1,0,3,2]
  v = min(v, dpp_i32<0x4E>(v));  // quad_perm [2,3,0,1]
  v = min(v, dpp_i32<0x141>(v)); // row_half_mirror
  v = min(v, dpp_i32<0x140>(v)); // row_mirror
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, dpp_i32<0xB1>(v));
  v = max(v, dpp_i32<0x4E>(v));
  v = max(v, dpp_i32<0x141>(v));
  v = max(v, dpp_i32<0x140>(v));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ---- Forward, bilinear, any padding, float and double: the LEAN kernel (round 5; reflection and double: round 6) ---------
// What the counters of rounds 3-4 meant, read with the right cost model (profiles/r05/micro_valu_issue.txt: a wave64 VALU
// instruction occupies its SIMD for ~2.3 cycles, f64 / conversions / 64-bit adds / DPP 4.2, transcendentals 8, and scalar
// instructions take issue slots too): SQ_ACTIVE_INST_ANY of mipmap_forward_kernel is 420 M quad-cycles per launch = 1.64 M
// cycles per SIMD = 0.73 of its 0.76 ms -- the kernel is INSTRUCTION-bound, not latency-bound (which is why keeping K items in
// flight bought nothing: 0.79 vs 0.77 ms, note further down).  Per (tap, level) it issues ~170 vector and
// ~60 scalar instructions: per-corner validity logic in 87 exec-mask regions, 64-bit address arithmetic per texel load, three
// dependent operations per product, the backward pass's gradient multipliers.  Here the common case -- every corner of the
// tap inside its level -- is ONE straight line for the whole wave:
//   * cell decisions are the reference's operations in the reference's order (tap position through the double product,
//     unnormalise, clip, floor: a tap must land in the same texel cell as in the oracle, the grid gradient's terms are
//     discontinuous there); what is CONTINUOUS is computed the cheap way: corner weights times the level's weight once per
//     tap, then one fma per texel and channel (12 instead of 36 operations for RGB; differs from the reference's
//     (texel * weight) * alpha by an ulp of each term);
//   * lanes whose tap is not interior (or that have no tap i) get offset 0 and weight 0 and run the same loads and fmas -- no
//     exec-mask regions in the loop; the border taps themselves (rare) are added by a branch the wave only takes if it has one;
//   * 32-bit texel offsets from per-pixel level bases.
// Bicubic stays with the kernels above.  Double (round 6): the same kernel on `T` -- 62-102 registers, 4-8 waves.
// Waves per SIMD (late round 5).  The kernel is a chain of gathers per tap and lives on waves in flight, like the backward on
// tiles in flight: the compiler's own choice was 100 registers = FOUR waves per SIMD (the allocation granule puts five at
// <= 96).  Asked for five it fits 84-94 without a spill: 0.70 -> 0.65 ms on the textured benchmark.  With the per-level sizes
// and the view's base pointers re-read from the LDS table inside the tap loop (three LDS reads per (tap, level)) instead of
// living in twelve registers per lane: 62-64 registers = EIGHT waves, 0.59 ms (kernel_bench's minified scenes 0.97 -> 0.80 and
// 1.23 -> 1.10); four channels per sweep: 70-72 = seven.
#ifndef DRTK_MIP_FWD_OCC_F64
#define DRTK_MIP_FWD_OCC_F64 4 // double: waves per SIMD the lean forward is compiled for
#endif
#ifndef DRTK_MIP_FWD_OCC
#define DRTK_MIP_FWD_OCC 8
#endif
template <typename T, int PAD, int CB>
__global__ __launch_bounds__(kBlock, sizeof(T) == 8 ? DRTK_MIP_FWD_OCC_F64 : (CB <= 3 ? DRTK_MIP_FWD_OCC : 7)) void mipmap_forward_lean_kernel(
    LevelTable lv, int mipmaps, const T* __restrict__ grid, GridLayout gl, const T* __restrict__ vt, int C,
    int64_t HW, int max_aniso, bool force_max_aniso, bool clip_grad, T* __restrict__ out, int strip) {
  static_assert(PAD >= 0 && PAD <= 2, "zeros, border or reflection padding");
  static_assert(CB >= 1 && CB <= 4, "channels per sweep over the taps (C is a multiple of CB)");
  constexpr int padding = PAD;
  constexpr bool align_corners = false; // mipmap_grid_sampler_kernel.cu:423
  __shared__ const void* s_ptr[kMaxLevels];
  __shared__ void* s_grad[kMaxLevels];
  __shared__ int s_h[kMaxLevels], s_w[kMaxLevels];
  __shared__ long long s_sn[kMaxLevels];
  __shared__ double s_f[kTapTab * kTapTab];
  stage_tap_table(s_f);
  stage_levels(lv, mipmaps, s_ptr, s_grad, s_h, s_w, s_sn);
  // (view = blockIdx.y: `index / HW` on a flat grid is a 64-bit division per pixel, ~80 instructions)
  const int64_t n = blockIdx.y;
  const int64_t pix_raw = int64_t(tile_index(strip)) * kBlock + threadIdx.x;
  const bool valid = pix_raw < HW;
  const int64_t pix = valid ? pix_raw : 0;
  const int64_t index = n * HW + pix;
  Taps<T> t = {};
  if (valid) t = setup_taps<T>(load_pixel_uv<T>(grid, gl, vt, n, pix, index), s_h[0], s_w[0], mipmaps, max_aniso, force_max_aniso, clip_grad);
  const int n_lv = mipmaps > 1 ? 2 : 1;
  // the levels' weights a / n and (1 - a) / n: continuous quantities, computed in float from one reciprocal (the reference
  // divides the second one in double, :486: a 16-instruction sequence at a quarter of the float rate, for the last ulp)
  // (double: the reference's two divisions, :485-486 -- the once-per-pixel cost is nothing beside the taps)
  const T rn = T(1) / static_cast<T>(max(t.n, 1));
  const T alpha_1 = !valid ? T(0) : sizeof(T) == 8 ? t.a / static_cast<T>(max(t.n, 1)) : t.a * rn;
  const T alpha_2 = !valid ? T(0) : sizeof(T) == 8 ? (T(1) - t.a) / static_cast<T>(max(t.n, 1)) : (T(1) - t.a) * rn;
  // level slot s = 0: level d1 with weight alpha_2; s = 1: level d1 + 1 with alpha_1.  A slot whose weight is exactly
  // zero -- the coarser level of every magnified pixel -- is dead (as in the kernels above).
  const bool live[2] = {bool(valid & (alpha_2 != T(0))), bool(valid & (n_lv == 2) & (alpha_1 != T(0)))};
  // (per-level sizes and the view's base pointer are re-read from the LDS table inside the tap loop instead of living in
  // twelve registers per lane)
  if (threadIdx.x < kMaxLevels) s_ptr[threadIdx.x] = static_cast<const T*>(s_ptr[threadIdx.x]) + n * s_sn[threadIdx.x];
  __syncthreads();
  const int ld[2] = {live[0] ? t.d1 : 0, live[1] ? t.d1 + 1 : 0};
  const double du_d = t.du, dv_d = t.dv;
  const int n_max = wave_max_i32(valid ? t.n : 0);
  const bool table = max_aniso <= kTapTab; // kernel-uniform: the taps' positions come from the LDS table
  const int tab_row = (max(t.n, 1) - 1) * kTapTab;
  T* out_px = out + n * C * HW + pix;

  for (int c0 = 0; c0 < C; c0 += CB) {
    T acc[CB];
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) acc[cc] = T(0);
    for (int i = 0; i < n_max; ++i) {
      const bool has_tap = i < t.n;
      // tap i of t.n: f = (i + 1.0) / (n + 1.0) * 2.0 - 1.0 (:497-499); a lane that has no tap i reads its last one
      // (its weights are zeroed below)
      const int ic = min(i, max(t.n, 1) - 1);
      const double f = table ? s_f[tab_row + ic] : (ic + 1.0) / (t.n + 1.0) * 2.0 - 1.0;
      const T x = t.u + static_cast<T>(du_d * f), y = t.v + static_cast<T>(dv_d * f);
      const bool ordered = (x == x) & (y == y); // a NaN coordinate samples nothing (safe_downgrade_to_int_range: -100)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bool on = has_tap & live[s];
        if (__ballot(on) == 0) continue; // wave-uniform
        int lw[2], lh[2];
        T lwf[2], lhf[2];
        GlobalPtr<const T> lbase[2];
        lw[s] = s_w[ld[s]], lh[s] = s_h[ld[s]];
        lwf[s] = static_cast<T>(lw[s]), lhf[s] = static_cast<T>(lh[s]);
        lbase[s] = (GlobalPtr<const T>)static_cast<const T*>(s_ptr[ld[s]]);
        // the reference's coordinate pipeline for an INTERIOR tap: unnormalize, clip (border padding), floor.  Values are
        // made finite first (a clamp that cannot move a tap whose cell lies inside the level), so that the weights of the
        // lanes that do not count -- no tap, a tap on or beyond the border, a NaN -- are finite and their products with
        // alpha = 0 vanish; those lanes' real contribution, if any, comes from the branch below.
        T ix = ((x + T(1)) * lwf[s] - 1) / 2, iy = ((y + T(1)) * lhf[s] - 1) / 2;
        bool sane = true; // (reflection: a coordinate the short pipeline must not touch goes to the corner-by-corner branch)
        if (padding == 1) {
          ix = tmin(tmax(ix, T(0)), lwf[s] - T(1)), iy = tmin(tmax(iy, T(0)), lhf[s] - T(1)); // == clip_coord but NaN -> 0
        } else if (padding == 2) {
          sane = (tabs(ix) < T(1e9f)) & (tabs(iy) < T(1e9f)); // (false for NaN)
          T unused_x = T(1), unused_y = T(1);
          reflect_clip_lean<T>(sane ? ix : T(0), sane ? iy : T(0), lw[s], lh[s], align_corners, ix, iy, unused_x, unused_y);
        } else {
          ix = tmin(tmax(ix, T(-4)), T(2e9f)), iy = tmin(tmax(iy, T(-4)), T(2e9f));
        }
        const T fx_floor = floor(ix), fy_floor = floor(iy);
        const int ix_nw = static_cast<int>(fx_floor), iy_nw = static_cast<int>(fy_floor);
        const bool interior = on & ordered & sane & (static_cast<unsigned>(ix_nw) < static_cast<unsigned>(lw[s] - 1)) &
            (static_cast<unsigned>(iy_nw) < static_cast<unsigned>(lh[s] - 1));
        // corner weights (bilinear_quad: (ix_se - ix) with ix_se = ix_nw + 1, an exact float) times the level's weight
        const T wx1 = (fx_floor + T(1)) - ix, wx0 = ix - fx_floor;
        const int plane = lw[s] * lh[s]; // <= kLeanMaxPlane (the dispatch): cc * plane + offset fits 32 bits
        const GlobalPtr<const T> b = lbase[s] + int64_t(c0) * plane;
        // Round 6: the tap's loads and fmas run under `interior` -- ONE exec-mask region per (tap, level).  Round 5 let the
        // other lanes (no tap i, a border tap, a NaN coordinate, a dead level) run the same loads at offset 0 with weight 0:
        // that read texels (0,0) / (1,0) of the level for pixels that never sample them -- a NaN or Inf there turned
        // `texel * 0` into NaN where the reference stays finite -- and on a 1 x 1 level the pair at offset 0 of the last
        // channel of the last view ended one element beyond the tensor.  The lanes that are switched off touch nothing now.
        if (interior) {
          const T al = s == 0 ? alpha_2 : alpha_1;
          const T wy1 = ((fy_floor + T(1)) - iy) * al, wy0 = (iy - fy_floor) * al;
          const T w_nw = wx1 * wy1, w_ne = wx0 * wy1, w_sw = wx1 * wy0, w_se = wx0 * wy0;
          const int o_top = iy_nw * lw[s] + ix_nw;
          const int o_bot = o_top + lw[s];
          Pair<T> top[CB], bot[CB];
#pragma unroll
          for (int cc = 0; cc < CB; ++cc) {
            top[cc] = *(GlobalPtr<const Pair<T>>)(b + (cc * plane + o_top));
            bot[cc] = *(GlobalPtr<const Pair<T>>)(b + (cc * plane + o_bot));
          }
#pragma unroll
          for (int cc = 0; cc < CB; ++cc) {
            acc[cc] = tfma(top[cc].x, w_nw, acc[cc]);
            acc[cc] = tfma(top[cc].y, w_ne, acc[cc]);
            acc[cc] = tfma(bot[cc].x, w_sw, acc[cc]);
            acc[cc] = tfma(bot[cc].y, w_se, acc[cc]);
          }
        }
        if (__ballot(on & !interior) != 0) {
          // a tap on the border of its level (or a non-finite coordinate) somewhere in the wave: those lanes alone, corner by
          // corner (the reference's form)
          if (on & !interior) {
            const Quad<T> q = bilinear_quad<T>(x, y, lh[s], lw[s], padding, align_corners);
            const T aq = s == 0 ? alpha_2 : alpha_1;
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) {
              const GlobalPtr<const T> pch = b + cc * plane; // (cc <= 3, plane <= kLeanMaxPlane)
              if (q.o_nw >= 0) acc[cc] += pch[q.o_nw] * q.nw * aq;
              if (q.o_ne >= 0) acc[cc] += pch[q.o_ne] * q.ne * aq;
              if (q.o_sw >= 0) acc[cc] += pch[q.o_sw] * q.sw * aq;
              if (q.o_se >= 0) acc[cc] += pch[q.o_se] * q.se * aq;
            }
          }
        }
      }
    }
    if (valid) {
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) out_px[int64_t(c0 + cc) * HW] = acc[cc];
    }
  }
}

// (Round 5 also built the forward with K (tap, level) items of a pixel in flight -- lean per-item state, the texel loads of all
// K issued together, 108 / 118 / 158 VGPRs at K = 2 / 3 / 4: 0.79 / 0.80 / 0.93 ms against 0.77 for the kernel at the top of this
// file.  The forward is not a chain of exposed round trips; removed, profiles/NOTES.md R5.2.)

template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void mipmap_backward_kernel(
    LevelTable lv, int mipmaps, const T* __restrict__ grad_out, const T* __restrict__ grid, GridLayout gl,
    const T* __restrict__ vt, int64_t count, int C, int64_t HW, int max_aniso, int padding, bool align_corners,
    bool force_max_aniso, bool clip_grad, T* __restrict__ grad_grid, GridLayout ggl, int strip) {
  __shared__ const void* s_ptr[kMaxLevels];
  __shared__ void* s_grad[kMaxLevels];
  __shared__ int s_h[kMaxLevels], s_w[kMaxLevels];
  __shared__ long long s_sn[kMaxLevels];
  stage_levels(lv, mipmaps, s_ptr, s_grad, s_h, s_w, s_sn);
  const int64_t index = int64_t(tile_index(strip)) * kBlock + threadIdx.x;
  if (index >= count) return;
  const int64_t n = index / HW;
  const Taps<T> t = setup_taps<T>(load_pixel_uv<T>(grid, gl, vt, n, index - n * HW, index), s_h[0], s_w[0], mipmaps, max_aniso, force_max_aniso, clip_grad);
  const int n_lv = mipmaps > 1 ? 2 : 1;
  const T* gout_px = grad_out + n * C * HW + (index - n * HW);
  const T alpha_1 = t.a / t.n;
  const T alpha_2 = static_cast<T>((1.0 - t.a) / t.n);
  T acc_x = T(0), acc_y = T(0);
  for (int i = 0; i < t.n; ++i) {
    const double f = (i + 1.0) / (t.n + 1.0) * 2.0 - 1.0;
    const T x = t.u + static_cast<T>(t.du * f), y = t.v + static_cast<T>(t.dv * f);
    for (int s = 0; s < n_lv; ++s) {
      const int d = t.d1 + s;
      const int h = s_h[d], w = s_w[d];
      const int64_t plane = int64_t(h) * w;
      const GlobalPtr<const T> inp = (GlobalPtr<const T>)(static_cast<const T*>(s_ptr[d]) + n * s_sn[d]);
      const GlobalPtr<T> ginp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + n * C * plane);
      const T alpha = s == 0 ? alpha_2 : alpha_1;
      T gix = T(0), giy = T(0);
      if constexpr (MODE == 0) {
        const Quad<T> q = bilinear_quad<T>(x, y, h, w, padding, align_corners);
        const int ix_se = q.ix_nw + 1, iy_se = q.iy_nw + 1;
        for (int c = 0; c < C; ++c) {
          const GlobalPtr<const T> p = inp + c * plane;
          const GlobalPtr<T> gp = ginp + c * plane;
          const T gOut = gout_px[int64_t(c) * HW] * alpha;
          // a zero upstream gradient (masked background) adds nothing: skip its four atomics
          if (gOut != T(0)) {
            if (q.o_nw >= 0) atomic_add_g1(gp + q.o_nw, q.nw * gOut);
            if (q.o_ne >= 0) atomic_add_g1(gp + q.o_ne, q.ne * gOut);
            if (q.o_sw >= 0) atomic_add_g1(gp + q.o_sw, q.sw * gOut);
            if (q.o_se >= 0) atomic_add_g1(gp + q.o_se, q.se * gOut);
          }
          // texel values for the grid gradient: the two texels of a row in one 8-byte load when both exist
          T v_nw = T(0), v_ne = T(0), v_sw = T(0), v_se = T(0);
          if (gOut != T(0)) { // with a zero upstream gradient every term below is +-0 * finite
            if (q.o_nw >= 0 && q.o_ne >= 0) {
              const Pair<T> t2 = *(GlobalPtr<const Pair<T>>)(p + q.o_nw);
              v_nw = t2.x, v_ne = t2.y;
            } else {
              if (q.o_nw >= 0) v_nw = p[q.o_nw];
              if (q.o_ne >= 0) v_ne = p[q.o_ne];
            }
            if (q.o_sw >= 0 && q.o_se >= 0) {
              const Pair<T> t2 = *(GlobalPtr<const Pair<T>>)(p + q.o_sw);
              v_sw = t2.x, v_se = t2.y;
            } else {
              if (q.o_sw >= 0) v_sw = p[q.o_sw];
              if (q.o_se >= 0) v_se = p[q.o_se];
            }
          }
          if (q.o_nw >= 0) {
            gix -= v_nw * (iy_se - q.iy) * gOut;
            giy -= v_nw * (ix_se - q.ix) * gOut;
          }
          if (q.o_ne >= 0) {
            gix += v_ne * (iy_se - q.iy) * gOut;
            giy -= v_ne * (q.ix - q.ix_nw) * gOut;
          }
          if (q.o_sw >= 0) {
            gix -= v_sw * (q.iy - q.iy_nw) * gOut;
            giy += v_sw * (ix_se - q.ix) * gOut;
          }
          if (q.o_se >= 0) {
            gix += v_se * (q.iy - q.iy_nw) * gOut;
            giy += v_se * (q.ix - q.ix_nw) * gOut;
          }
        }
        acc_x += q.mx * gix;
        acc_y += q.my * giy;
      } else {
        const Cubic<T> cb = bicubic_footprint<T>(x, y, h, w, padding, align_corners);
        T xc[4], yc[4], xg[4], yg[4];
        cubic_coeffs(xc, cb.tx);
        cubic_coeffs(yc, cb.ty);
        cubic_coeffs_grad(xg, cb.tx);
        cubic_coeffs_grad(yg, cb.ty);
        for (int c = 0; c < C; ++c) {
          const GlobalPtr<const T> p = inp + c * plane;
          const GlobalPtr<T> gp = ginp + c * plane;
          const T gOut = gout_px[int64_t(c) * HW] * alpha;
#pragma unroll
          for (int i2 = 0; i2 < 4; ++i2) {
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
              const bool ok = cb.xi[i2] >= 0 && cb.yi[j2] >= 0;
              const int o = ok ? cb.yi[j2] * w + cb.xi[i2] : 0;
              if (ok && gOut != T(0)) atomic_add_g1(gp + o, gOut * xc[i2] * yc[j2]);
              const T val = ok ? p[o] : T(0);
              gix -= gOut * val * (xg[i2] * yc[j2]);
              giy -= gOut * val * (yg[j2] * xc[i2]);
            }
          }
        }
        acc_x += cb.mx * gix;
        acc_y += cb.my * giy;
      }
    }
  }
  store_grid_grad<T>(grad_grid, ggl, n, index - n * HW, acc_x, acc_y);
}

// Backward, bilinear, C <= 4: LDS-windowed accumulation of the texture gradient.
// The direct kernel above is bound by the float-atomic REQUEST rate (~20 G/s): every (tap, level, corner,
// channel) is one atomic instruction touching ~5 64-byte segments.  Here a workgroup owns a 16 x 16 PIXEL
// tile, whose taps land in a compact texel window on the two or three mip levels its pixels select; the
// window (32 x 32 texels per level, 3 levels from the tile's finest level) is accumulated in LDS with
// ds_add_f32 and flushed once, row-major, so the global atomics are coalesced and each touched texel costs
// one request per tile instead of one per tap.  Corners outside the window, or on other levels, go to
// global memory directly, so any uv field is handled.
// Pixel tile: kTileW x kTileH pixels, one thread each.  16 x 16 is a sweet spot, same-box A/B on the textured benchmark /
// kernel_bench at 1 and 4 texels per pixel: 16 x 32 (8 waves per workgroup over the same 48 KB of windows, twice the
// waves per CU) 3.11 / 4.61 / 10.8 ms against 1.94 / 2.99 / 8.13 -- the taller tile's taps spread over more texels than a
// window holds, and its misses cost rounds; 16 x 8: 2.9 ms -- the per-tile costs (window init + flush scan, five
// barriers) are paid twice as often.
constexpr int kTileW = 16;
#ifndef DRTK_MIP_TILE_H
#define DRTK_MIP_TILE_H 16
#endif
constexpr int kTileH = DRTK_MIP_TILE_H;
constexpr int kMipBlock = kTileW * kTileH;
#ifndef DRTK_MIP_WIN
#define DRTK_MIP_WIN 32
#endif
#ifndef DRTK_MIP_WINB
#define DRTK_MIP_WINB DRTK_MIP_WIN
#endif
constexpr int kWin = DRTK_MIP_WIN;   // texel window side on the tile's finest level
constexpr int kWinB = DRTK_MIP_WINB; // ... and on the next one (even: the flush reads cells in pairs)
constexpr int kWinLevels = 2;
#ifndef DRTK_MIP_WINPAD
#define DRTK_MIP_WINPAD 0
#endif
// Extra cells per window row and per channel plane (even: the flush reads cells in 16-byte pairs).  A 32-cell row is 256
// bytes, one sweep of the 64 LDS banks, so the south corners of a tap share the banks of its north corners and so do the
// pixel rows of a wave -- SQ_LDS_BANK_CONFLICT + SQ_LDS_ADDR_CONFLICT are more than half of this kernel's LDS-active
// cycles -- but the LDS pipe is only ~11 % busy and padding the strides to 34 cells bought nothing: 2.23 vs 2.17 ms on the
// textured benchmark, 4.65 vs 4.58 and 11.9 vs 11.8 ms on the minified scenes of profiles/kernel_bench.py.  Neither did
// walking the 4C cells of a tap in a per-lane rotated order, which removes the same-address meetings of neighbouring
// pixels on a magnified texture (2.16 vs 2.23 ms).  Kept as a switch for the next look at this kernel.
constexpr int kWinPad = DRTK_MIP_WINPAD;
__host__ __device__ constexpr int win_side(int l) { return l == 0 ? kWin : kWinB; }
__host__ __device__ constexpr int win_stride(int l) { return win_side(l) + kWinPad; }
__host__ __device__ constexpr int win_cells(int l) { return win_side(l) * win_stride(l) + kWinPad; } // per channel (the channels of a cell in different banks, too)
__host__ __device__ constexpr int win_cells_before(int l) { return l == 0 ? 0 : win_cells(0) + (l - 1) * win_cells(1); } // per channel
constexpr int kWinCells = win_cells_before(kWinLevels);
// `dbg` (diagnostics, profiles/kernel_bench.py --flags): 1 = no texture-gradient accumulation, 2 = no texel reads,
// 4 = no flush.  Where the 6.0 ms of the bench shape go (finer timing-only variants, r01): the global-atomic fallback for
// corners outside the windows ~1.5 ms, the four LDS adds ~0.8 ms, the flush 0.5 ms, texel reads 0.2 ms, everything else
// (tap set-up, window bookkeeping, barriers) the rest.  Flag 1 alone overstates the atomics: it also drops the fallback
// and lets the compiler delete the bookkeeping (2.2 ms remain).  On that scene 91.5 % of the (tap, level) pairs hit the
// windows, 7.0 % miss by POSITION (anisotropic taps: a 16-pixel tile spans up to 16 * N * 2 texels along the major
// axis), 1.5 % by level; 16 % of the tiles have a miss and the worst 1 % hold a third of them (atlas seam, silhouette).
// Measured dead end: the same LDS split 2048 / 768 / 256 cells over the levels with every window shaped like its tile's
// bounding box -- 1-4 % (6.06 -> 6.02 ms at 1 texel/px, 11.5 -> 11.0 at 4), one outlier pixel stretches the box.

// Round 4, where the time is WITHOUT the scatter (ablation flags 16 = no grid-gradient products, 32 = no tap loop at all, on
// top of 1 / 2 / 4 / 8; textured benchmark, 1.96 ms with the 0.14 ms zero-fill of the pyramid): no accumulation, texel reads,
// flush or rounds 1.02; also no grid-gradient products 0.95; no tap loop at all 0.75 -- of which 0.14 the fill and 0.3 the
// 1.5 GB of upstream gradient, uv, Jacobian and grid gradient at stream speed.  Two structural variants were measured
// against that and NOT kept: wave-private windows without any workgroup barrier (mipmap_backward_wave_kernel below; same
// 3 waves per SIMD, same 1.0 ms floor, 2.52 vs 1.95 ms at C = 3: the floor is per-(tap, level) arithmetic and the stream,
// not barriers), and persistent workgroups that request the next tile's upstream gradient before working on the current
// one (2.20 vs 1.96 ms; the floor with no tap loop rose from 0.75 to 0.98: the hardware's own workgroup dispatch balances
// the mix of background and silhouette tiles better than a static stride, and the job loop costs registers).  Also
// nothing: the window products in double (7 conversions + 12 double multiplications instead of 12 + 12) and the grid
// gradient's eight products regrouped into 10 operations per channel with written-out fmas (1.967 vs 1.955 ms).
// PAD / ALIGN: padding mode and align_corners as compile-time constants (the coordinate pipeline of every tap branches on
// them: SQ counters of the runtime-parameter version showed 490 scalar and 950 vector instructions per wave).
// (mipmap_backward_tiled_kernel, the kernel these notes are about -- square 32 x 32 windows, the general tap loop, five
// barriers per tile -- was replaced in round 5 by the two kernels below: git history, profiles/NOTES.md R5.2.)

// ---- round 5: the tile kernel with a shorter chain per tile (reflection padding, double; everything else: the lean kernel
// further down) -----------------------------------------------------------------------------------------------------------
// The ablation of round 4 had left 0.75 of the tile kernel's 1.95 ms outside the tap loop and ~0.95 in the scatter (LDS
// accumulation, rounds, flush), with the waves parked half of their life at 3 per SIMD.  Changes against round 4's
// mipmap_backward_tiled_kernel:
//   * the level tables, the placement cells and the windows' zero-fill are issued UNDER the loads of the upstream
//     gradient / uv / Jacobian, in front of the first barrier (the level-0 size that the tap set-up needs comes from the
//     kernel argument) -- one barrier and one exposed phase less;
//   * reference level and window origins come out of ONE phase: every wave reduces the extremes of its taps per
//     ABSOLUTE level (one to three levels per wave) into s_lox / s_loy / s_hiy[d]; after one barrier ref = s_ref and the
//     origins are s_lox[ref + l], read into scalar registers -- two barriers less;
//   * taps of a pixel that land in the same 2 x 2 cell are merged in registers before they touch LDS (run_cell / run_w);
//   * the flush reads only the window rows the first round can have written (s_hiy).
// Sums regrouped, never dropped: same results to rounding (fixtures, known answers, fuzz_mipmap, fuzz_mipmap_snapped).
#ifndef DRTK_MIP_T2_OCC
#define DRTK_MIP_T2_OCC 3
#endif
#ifndef DRTK_MIP_HOPELESS
#define DRTK_MIP_HOPELESS 0 // measured and left OFF (below): -1.5 % on the textured benchmark at its best threshold, +3 ... +10 % on minified scenes
#endif
#ifndef DRTK_MIP_HOPELESS_PAIRS
#define DRTK_MIP_HOPELESS_PAIRS 96 // (a round costs about as much as sending this many pairs' 12 atomics each to global memory)
#endif
template <typename T, int PAD, bool ALIGN>
// (3 workgroups per CU by registers as by LDS; reflection padding needs ~200: 2)
__global__ __launch_bounds__(kMipBlock, sizeof(T) == 8 ? 1 : (PAD == 2 ? 2 : DRTK_MIP_T2_OCC)) void mipmap_backward_tiled2_kernel(
    LevelTable lv, int mipmaps, const T* __restrict__ grad_out, const T* __restrict__ grid, GridLayout gl,
    const T* __restrict__ vt, int H, int W, int C, int tiles_x, int max_aniso,
    bool force_max_aniso, bool clip_grad, T* __restrict__ grad_grid, GridLayout ggl, int strip, int dbg) {
  constexpr int padding = PAD;
  constexpr bool align_corners = ALIGN;
  __shared__ double s_f[kTapTab * kTapTab];
  __shared__ const void* s_ptr[kMaxLevels];
  __shared__ void* s_grad[kMaxLevels];
  __shared__ int s_h[kMaxLevels], s_w[kMaxLevels];
  __shared__ long long s_sn[kMaxLevels];
  __shared__ int s_ref, s_npend, s_ox[kWinLevels], s_oy[kWinLevels], s_hx[kWinLevels], s_hy[kWinLevels];
  // first-round placement per ABSOLUTE level (so that the reference level and the origins come out of ONE phase):
  // bounding box of the north-west texels of the tile's taps on level d
  __shared__ int s_lox[kMaxLevels + 1], s_loy[kMaxLevels + 1], s_hix[kMaxLevels + 1], s_hiy[kMaxLevels + 1];
  extern __shared__ __attribute__((aligned(16))) unsigned char s_win_raw[];
  double* const s_win = reinterpret_cast<double*>(s_win_raw);
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int tile = tile_index(strip);
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int px = tx * kTileW + (tid & (kTileW - 1)), py = ty * kTileH + tid / kTileW;
  const bool valid = px < W && py < H;
  const int64_t HW = int64_t(H) * W;
  const int64_t index = int64_t(n) * HW + int64_t(py) * W + px;
  // the pixel's upstream gradient, uv and Jacobian: one batch of loads ...
  T go[4] = {T(0), T(0), T(0), T(0)};
  if (valid) {
    const T* gout_px = grad_out + int64_t(n) * C * HW + (int64_t(py) * W + px);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < C) go[c] = gout_px[int64_t(c) * HW];
    }
  }
  PixelUV<T> uv = {};
  const int64_t pix = int64_t(py) * W + px;
  if (valid) uv = load_pixel_uv<T>(grid, gl, vt, n, pix, index);
  // ... and under them everything that needs no data: the tables, the placement cells, the windows' zero-fill (a tile
  // that turns out to have no upstream gradient has zeroed its windows for nothing; the stores wait for nobody)
  stage_tap_table(s_f);
  if (tid < kMaxLevels) {
    const int i = tid < mipmaps ? tid : 0;
    s_ptr[tid] = lv.ptr[i], s_grad[tid] = lv.grad[i], s_sn[tid] = lv.sn[i], s_h[tid] = lv.h[i], s_w[tid] = lv.w[i];
  }
  if (tid <= kMaxLevels) s_lox[tid] = s_loy[tid] = INT32_MAX, s_hix[tid] = s_hiy[tid] = INT32_MIN;
  if (tid == 0) s_ref = kMaxLevels;
  {
    double2* w2 = reinterpret_cast<double2*>(s_win);
    const double2 z = {0.0, 0.0};
    for (int i = tid; i < C * kWinCells / 2; i += kMipBlock) w2[i] = z;
  }
  const bool has_go = go[0] != T(0) || go[1] != T(0) || go[2] != T(0) || go[3] != T(0);
  if (!__syncthreads_or(has_go)) { // (also publishes the tables and the zero-fill)
    if (valid) store_grid_grad<T>(grad_grid, ggl, n, pix, T(0), T(0));
    return;
  }

  Taps<T> t = {};
  if (has_go) t = setup_taps<T>(uv, lv.h[0], lv.w[0], mipmaps, max_aniso, force_max_aniso, clip_grad);
  const int n_lv = mipmaps > 1 ? 2 : 1;
  const T alpha_1 = has_go ? t.a / t.n : T(0);
  const T alpha_2 = has_go ? static_cast<T>((1.0 - t.a) / t.n) : T(0);
  // A (pixel, level) whose weighted upstream gradient is zero in every channel adds nothing anywhere (every term is
  // +-0 * finite): the masked background of a silhouette tile, and the second level of a magnified pixel (a == 0).
  // Such pairs neither place the windows nor run their taps.
  bool live[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const T alpha = s == 0 ? alpha_2 : alpha_1;
    live[s] = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) live[s] = live[s] || (go[c] * alpha != T(0));
    live[s] = live[s] && s < n_lv;
  }
  auto tap_xy = [&](int i, T& x, T& y) {
    const double f = tap_f(s_f, i, t.n);
    x = t.u + static_cast<T>(t.du * f);
    y = t.v + static_cast<T>(t.dv * f);
  };
  // window origins: the taps of a pixel are collinear, so their extreme texels are those of the first and the last
  // tap.  ANY origin is correct -- a corner is windowed iff its exact cell lies inside, tested below -- so the origin
  // comes from a short form of the coordinate pipeline: unnormalise, clamp to the level, floor (the exact north-west
  // texel for zeros / border padding; under reflection padding taps beyond the border miss the window).
  auto texel_floor = [&](T coord, int size) -> int {
    T unused;
    const T c = unnormalize(coord, size, align_corners, &unused);
    const T lo = padding == 0 ? T(-1) : T(0);
    return static_cast<int>(floor(fmin(fmax(c, lo), static_cast<T>(size - 1)))); // fmax(NaN, lo) = lo
  };
  {
    // ONE phase for the reference level and the origins: the extremes go into cells of their ABSOLUTE level, wave by
    // wave over the (one to three) levels a wave's pixels use
    int lo_x[2] = {INT32_MAX, INT32_MAX}, lo_y[2] = {INT32_MAX, INT32_MAX}, hi_x[2] = {INT32_MIN, INT32_MIN}, hi_y[2] = {INT32_MIN, INT32_MIN};
    if (live[0] || live[1]) {
      for (int e = 0; e < 2; ++e) {
        T x, y;
        tap_xy(e == 0 ? 0 : t.n - 1, x, y);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s < n_lv && live[s]) {
            const int ox = texel_floor(x, s_w[t.d1 + s]), oy = texel_floor(y, s_h[t.d1 + s]);
            lo_x[s] = min(lo_x[s], ox), lo_y[s] = min(lo_y[s], oy), hi_x[s] = max(hi_x[s], ox), hi_y[s] = max(hi_y[s], oy);
          }
        }
      }
    }
    const int d_lo = wave_min_i32(live[0] ? t.d1 : live[1] ? t.d1 + 1 : kMaxLevels);
    const int d_hi = wave_max_i32(live[1] ? t.d1 + 1 : live[0] ? t.d1 : -1);
    for (int d = d_lo; d <= d_hi; ++d) { // wave-uniform
      const bool m0 = live[0] && t.d1 == d, m1 = live[1] && t.d1 + 1 == d;
      const int a = wave_min_i32(min(m0 ? lo_x[0] : INT32_MAX, m1 ? lo_x[1] : INT32_MAX));
      const int b = wave_min_i32(min(m0 ? lo_y[0] : INT32_MAX, m1 ? lo_y[1] : INT32_MAX));
      const int e = wave_max_i32(max(m0 ? hi_y[0] : INT32_MIN, m1 ? hi_y[1] : INT32_MIN));
      const int g = wave_max_i32(max(m0 ? hi_x[0] : INT32_MIN, m1 ? hi_x[1] : INT32_MIN));
      if ((tid & (kWave - 1)) == 0 && a != INT32_MAX) {
        atomicMin(&s_lox[d], a);
        atomicMin(&s_loy[d], b);
        atomicMax(&s_hix[d], g);
        atomicMax(&s_hiy[d], e);
      }
    }
    if ((tid & (kWave - 1)) == 0 && d_lo < kMaxLevels) atomicMin(&s_ref, d_lo);
  }
  __syncthreads();
  const int ref = s_ref;
  // ---- the two window SLOTS (round 5: shaped, not square).  Slot l holds level ref + l.  The windows' memory is
  // kWinLevels x kWinSlotCells accumulators per channel; a tile with no live tap on level ref + 1 -- every tile of a magnified
  // texture -- gives it all to slot 0; a slot is 2^sx cells wide, sx chosen by the bounding box of the taps it has to
  // hold (16 ... 128 wide: the elongated footprints of a limb tile fit where a 32 x 32 square needed four rounds).
  // Wave-uniform values: scalar registers.
  constexpr int kWinSlotCells = kWin * kWin;
  static_assert(kWinPad == 0 && kWinB == kWin && kWinCells == kWinLevels * kWinSlotCells, "the shaped slots re-partition the square windows' memory");
  int wox[kWinLevels], woy[kWinLevels], wsx[kWinLevels], wny[kWinLevels], wcells[kWinLevels], win_rows[kWinLevels];
  auto shape_slots = [&](const int (&lox)[kWinLevels], const int (&loy)[kWinLevels], const int (&hix)[kWinLevels], const int (&hiy)[kWinLevels], bool all_rows) {
    const bool two = lox[1] != INT32_MAX;
#pragma unroll
    for (int l = 0; l < kWinLevels; ++l) {
      wox[l] = lox[l], woy[l] = loy[l];
      wcells[l] = l == 0 ? (two ? kWinSlotCells : kWinLevels * kWinSlotCells) : (two ? kWinSlotCells : 0);
      const long long need_w = static_cast<long long>(hix[l]) - lox[l] + 2, need_h = static_cast<long long>(hiy[l]) - loy[l] + 2;
      int sx = need_w <= 16 ? 4 : need_w <= 32 ? 5 : (need_w <= 64 && need_h <= (wcells[l] >> 6)) ? 6 : (need_w > 64 && need_h <= (wcells[l] >> 7)) ? 7 : 5;
      if (lox[l] == INT32_MAX) sx = 5;
      wsx[l] = sx;
      wny[l] = wcells[l] > 0 ? (wcells[l] >> sx) : 1; // (1: no cell passes the row test of an empty slot)
      win_rows[l] = (all_rows || need_h > wny[l]) ? wny[l] : static_cast<int>(need_h < 0 ? 0 : need_h);
      if (wcells[l] == 0) win_rows[l] = 0;
    }
  };
  {
    int lox[kWinLevels], loy[kWinLevels], hix[kWinLevels], hiy[kWinLevels];
#pragma unroll
    for (int l = 0; l < kWinLevels; ++l) {
      const int d = min(ref + l, kMaxLevels);
      lox[l] = __builtin_amdgcn_readfirstlane(s_lox[d]), loy[l] = __builtin_amdgcn_readfirstlane(s_loy[d]);
      hix[l] = __builtin_amdgcn_readfirstlane(s_hix[d]), hiy[l] = __builtin_amdgcn_readfirstlane(s_hiy[d]);
    }
    // rows the first round can touch: exact for zeros / border padding, where the extremes are the true north-west texels;
    // under reflection padding every row
    shape_slots(lox, loy, hix, hiy, padding == 2);
  }
  // cell of a north-west texel in slot l (the other three corners are +1, +stride, +stride+1), or -1
  auto slot_cell = [&](int l, int ix_nw, int iy_nw) -> int {
    if (l < 0 || l >= kWinLevels) return -1;
    const int wx = ix_nw - (l == 0 ? wox[0] : wox[1]), wy = iy_nw - (l == 0 ? woy[0] : woy[1]);
    const int sx = l == 0 ? wsx[0] : wsx[1], ny = l == 0 ? wny[0] : wny[1];
    return (static_cast<unsigned>(wx) < (1u << sx) - 1u && static_cast<unsigned>(wy) < static_cast<unsigned>(ny - 1)) ? (wy << sx) + wx : -1;
  };
  auto slot_stride = [&](int l) -> int { return 1 << (l == 0 ? wsx[0] : wsx[1]); };
  auto slot_chan = [&](int l) -> int { return l == 0 ? wcells[0] : wcells[1]; };  // cells per channel
  auto slot_base = [&](int l) -> int { return l == 0 ? 0 : C * wcells[0]; };        // first cell of the slot's channel 0

  // (tap, level) pairs that find no window cell in this round: not sent to global memory one corner and channel at a
  // time -- scattered float atomics of single lanes, 0.72 of this kernel's 2.2 ms on the textured benchmark although
  // only a few per cent of the taps miss -- but remembered (per level of the pixel: did any tap miss, and where) for a
  // further round with the windows moved onto them (below, up to DRTK_MIP_ROUNDS rounds).
  // (where they missed is NOT carried through the tap loop -- six registers at its bound: a round recomputes the
  // north-west texels of its pending taps, which it walks anyway)
  uint32_t pending = 0; // bit 2 i + s: tap i on the pixel's level s found no window cell yet (taps >= 16 are never deferred)
  if (valid) {
    T acc_x = T(0), acc_y = T(0);
    // the pixel's two levels: sizes and base pointers once, not per tap
    // (the gradient planes' base, which only the rare fallbacks to global memory need, is rebuilt from the LDS table there)
    int lv_h[2], lv_w[2];
    GlobalPtr<const T> lv_inp[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int d = live[s] ? t.d1 + s : 0;
      lv_h[s] = s_h[d], lv_w[s] = s_w[d];
      lv_inp[s] = (GlobalPtr<const T>)(static_cast<const T*>(s_ptr[d]) + int64_t(n) * s_sn[d]);
    }
    auto grad_base = [&](int d, int64_t plane) -> GlobalPtr<T> { return (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + int64_t(n) * C * plane); };
    // (one run slot, for the pixel's FINER level: a magnified pixel has no other live level, and the taps of a minified
    // one are a texel apart on the finer level -- half a texel on the coarser, whose adds stay as they were)
    int run_cell = -1;
    T run_w[4] = {T(0), T(0), T(0), T(0)};
    auto emit_run = [&](int l, const T (&g)[4]) {
      if (run_cell < 0) return;
      double* wp = s_win + slot_base(l) + run_cell;
      const int stride = slot_stride(l), chan = slot_chan(l);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c >= C) break;
        lds_add(wp + c * chan, static_cast<double>(run_w[0] * g[c]));
        lds_add(wp + c * chan + 1, static_cast<double>(run_w[1] * g[c]));
        lds_add(wp + c * chan + stride, static_cast<double>(run_w[2] * g[c]));
        lds_add(wp + c * chan + stride + 1, static_cast<double>(run_w[3] * g[c]));
      }
      run_cell = -1;
    };
    for (int i = 0; (live[0] || live[1]) && i < t.n && !DRTK_DBG(dbg, 32); ++i) {
      T x, y;
      tap_xy(i, x, y);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (!live[s]) continue;
        const int d = t.d1 + s;
        const int h = lv_h[s], w = lv_w[s];
        const int64_t plane = lv_h[s] * lv_w[s]; // < 2^31, checked by fill_table()
        const GlobalPtr<const T> inp = lv_inp[s];
        const T alpha = s == 0 ? alpha_2 : alpha_1;
        const Quad<T> q = bilinear_quad<T>(x, y, h, w, padding, align_corners);
        const int ix_se = q.ix_nw + 1, iy_se = q.iy_nw + 1;
        // window cell of the north-west corner (the other three are +1 in x / y), or -1 if not windowed
        const int l = d - ref;
        const int cell = slot_cell(l, q.ix_nw, q.iy_nw);
        const int stride = slot_stride(l), chan = slot_chan(l);
        T g[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) g[c] = c < C ? go[c] * alpha : T(0);
        T gix = T(0), giy = T(0);
        // Interior taps -- all four corners inside the level, i.e. nearly all of them -- take a straight-line path: the
        // per-corner validity tests of the general form below each cost a divergent branch (the general form compiled to
        // 234 exec-mask regions), while here a tap is 2C 8-byte texel loads, 4C adds and the grid-gradient products with no
        // branch but window / fallback.  A channel whose weighted gradient is zero adds +-0 (the general form skips it).
        if ((q.o_nw | q.o_ne | q.o_sw | q.o_se) >= 0) {
          Pair<T> top[4], bot[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            top[c] = bot[c] = Pair<T>{T(0), T(0)};
            if (c < C && !DRTK_DBG(dbg, 2)) {
              top[c] = *(GlobalPtr<const Pair<T>>)(inp + c * plane + q.o_nw);
              bot[c] = *(GlobalPtr<const Pair<T>>)(inp + c * plane + q.o_sw);
            }
          }
          if (DRTK_DBG(dbg, 1)) {
          } else if (cell >= 0) {
            // the taps of a pixel that land in the SAME 2 x 2 cell of a level -- on a magnified texture nearly all of
            // them -- are merged before they touch LDS: their corner weights add up in registers and the cell gets one
            // set of adds when the run ends (a different cell, or the last tap)
            if (s == 1) {
              double* wp = s_win + slot_base(l) + cell;
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                if (c >= C) break;
                lds_add(wp + c * chan, static_cast<double>(q.nw * g[c]));
                lds_add(wp + c * chan + 1, static_cast<double>(q.ne * g[c]));
                lds_add(wp + c * chan + stride, static_cast<double>(q.sw * g[c]));
                lds_add(wp + c * chan + stride + 1, static_cast<double>(q.se * g[c]));
              }
            } else if (cell != run_cell) {
              emit_run(l, g);
              run_cell = cell;
              run_w[0] = q.nw, run_w[1] = q.ne, run_w[2] = q.sw, run_w[3] = q.se;
            } else {
              run_w[0] += q.nw, run_w[1] += q.ne, run_w[2] += q.sw, run_w[3] += q.se;
            }
          } else if (i < 16) {
            pending |= 1u << (2 * i + s);
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (c >= C) break;
              const GlobalPtr<T> gp = grad_base(d, plane) + c * plane;
              atomic_add_g1(gp + q.o_nw, q.nw * g[c]);
              atomic_add_g1(gp + q.o_ne, q.ne * g[c]);
              atomic_add_g1(gp + q.o_sw, q.sw * g[c]);
              atomic_add_g1(gp + q.o_se, q.se * g[c]);
            }
          }
          const T fy1 = iy_se - q.iy, fy0 = q.iy - q.iy_nw, fx1 = ix_se - q.ix, fx0 = q.ix - q.ix_nw;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (c >= C || DRTK_DBG(dbg, 16)) break;
            const T gOut = g[c];
            // with a zero upstream gradient every term is +-0 * finite: the texels count as 0
            const T v_nw = gOut != T(0) ? top[c].x : T(0), v_ne = gOut != T(0) ? top[c].y : T(0);
            const T v_sw = gOut != T(0) ? bot[c].x : T(0), v_se = gOut != T(0) ? bot[c].y : T(0);
            gix -= v_nw * fy1 * gOut;
            giy -= v_nw * fx1 * gOut;
            gix += v_ne * fy1 * gOut;
            giy -= v_ne * fx0 * gOut;
            gix -= v_sw * fy0 * gOut;
            giy += v_sw * fx1 * gOut;
            gix += v_se * fy0 * gOut;
            giy += v_se * fx0 * gOut;
          }
        } else {
        // General form (a tap on the border of its level: rare), one channel at a time -- nothing of it is live across the
        // interior path above
        const bool defer = !DRTK_DBG(dbg, 1) && cell < 0 && i < 16 && (q.o_nw & q.o_ne & q.o_sw & q.o_se) != -1; // (all four corners outside the level: nothing to add anywhere)
        if (defer) pending |= 1u << (2 * i + s);
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
          const T gOut = c == 0 ? g[0] : c == 1 ? g[1] : c == 2 ? g[2] : g[3];
          if (gOut == T(0)) continue; // a zero upstream gradient (masked channel) adds nothing; its texels count as 0
          const GlobalPtr<const T> p = inp + c * plane;
          T v_nw = T(0), v_ne = T(0), v_sw = T(0), v_se = T(0);
          if (!DRTK_DBG(dbg, 2)) {
            if (q.o_nw >= 0) v_nw = p[q.o_nw];
            if (q.o_ne >= 0) v_ne = p[q.o_ne];
            if (q.o_sw >= 0) v_sw = p[q.o_sw];
            if (q.o_se >= 0) v_se = p[q.o_se];
          }
          if (!DRTK_DBG(dbg, 1)) {
            if (cell >= 0) {
              double* wp = s_win + slot_base(l) + c * chan + cell;
              if (q.o_nw >= 0) lds_add(wp, static_cast<double>(q.nw * gOut));
              if (q.o_ne >= 0) lds_add(wp + 1, static_cast<double>(q.ne * gOut));
              if (q.o_sw >= 0) lds_add(wp + stride, static_cast<double>(q.sw * gOut));
              if (q.o_se >= 0) lds_add(wp + stride + 1, static_cast<double>(q.se * gOut));
            } else if (!defer) {
              const GlobalPtr<T> gp = grad_base(d, plane) + c * plane;
              if (q.o_nw >= 0) atomic_add_g1(gp + q.o_nw, q.nw * gOut);
              if (q.o_ne >= 0) atomic_add_g1(gp + q.o_ne, q.ne * gOut);
              if (q.o_sw >= 0) atomic_add_g1(gp + q.o_sw, q.sw * gOut);
              if (q.o_se >= 0) atomic_add_g1(gp + q.o_se, q.se * gOut);
            }
          }
          if (q.o_nw >= 0) {
            gix -= v_nw * (iy_se - q.iy) * gOut;
            giy -= v_nw * (ix_se - q.ix) * gOut;
          }
          if (q.o_ne >= 0) {
            gix += v_ne * (iy_se - q.iy) * gOut;
            giy -= v_ne * (q.ix - q.ix_nw) * gOut;
          }
          if (q.o_sw >= 0) {
            gix -= v_sw * (q.iy - q.iy_nw) * gOut;
            giy += v_sw * (ix_se - q.ix) * gOut;
          }
          if (q.o_se >= 0) {
            gix += v_se * (q.iy - q.iy_nw) * gOut;
            giy += v_se * (q.ix - q.ix_nw) * gOut;
          }
        }
        } // general form
        acc_x += q.mx * gix;
        acc_y += q.my * giy;
      }
    }
    // the last run
    if (live[0]) {
      T g[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) g[c] = c < C ? go[c] * alpha_2 : T(0);
      emit_run(t.d1 - ref, g);
    }
    store_grid_grad<T>(grad_grid, ggl, n, pix, acc_x, acc_y);
    DRTK_MIP_STAT(8, (live[0] ? t.n : 0) + (live[1] ? t.n : 0));
    DRTK_MIP_STAT(9, __popc(pending));
  }
  if (tid == 0) DRTK_MIP_STAT(0, 1);
  __syncthreads();
  // flush the windows: consecutive threads = consecutive texels of a row; cells that stayed 0 cost nothing,
  // cells outside the level were never written (only in-bounds corners are accumulated).  With `rearm` the cells are
  // zeroed as they are read: the windows serve a second round.
  auto flush = [&](int ref_level, bool rearm) {
#pragma unroll
    for (int l = 0; l < kWinLevels; ++l) {
      const int d = ref_level + l;
      if (d >= mipmaps || wox[l] == INT32_MAX || DRTK_DBG(dbg, 4)) continue;
      const int h = s_h[d], w = s_w[d];
      const int64_t plane = int64_t(h) * w;
      const GlobalPtr<T> ginp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + int64_t(n) * C * plane);
      for (int c = 0; c < C; ++c) {
        // ONE cell per lane: an atomic instruction's lanes are consecutive texels of a row, one request per 64-byte line
        // (two cells per lane asked for every line twice -- see mipmap_backward_lean_kernel's flush)
        const int stride = 1 << wsx[l], chan = wcells[l];
        double* win1 = s_win + (l == 0 ? 0 : C * wcells[0]) + c * chan;
        for (int i = tid; i < win_rows[l] * stride; i += kMipBlock) { // rows beyond win_rows were never written
          const double q = win1[i];
          if (q != 0.0) {
            if (rearm) win1[i] = 0.0;
            const int gx = wox[l] + (i & (stride - 1)), gy = woy[l] + (i >> wsx[l]);
            atomic_add_g1(ginp + c * plane + int64_t(gy) * w + gx, static_cast<T>(q));
          }
        }
      }
    }
  };
  // Does any pixel of the tile have taps the windows did not hold?  (16 % of the tiles of the minified benchmark scenes;
  // on the textured benchmark the atlas seam -- neighbouring pixels sample opposite ends of the texture -- and the limb,
  // where eight anisotropic taps spread over more texels than a window is wide.)  `pending` says which.
#ifndef DRTK_MIP_ROUNDS
#define DRTK_MIP_ROUNDS 6 // same-box A/B (textured benchmark / kernel_bench at 1 texel per pixel / at 4): 1 round (all misses to
#endif                    // global memory, rounds 1-2) 2.15 / 4.58 / 11.7 ms; 2: 2.00 / 3.80 / 9.85; 3: 1.95 / 3.41 / 9.10; 4: 1.94 / 3.11 / 8.63; 6: 1.97 / 2.97 / 8.09
  // ---- further rounds: the windows are moved onto the taps that are still pending and those taps alone are accumulated
  // (texture gradient only: the grid gradient is complete).  What is still pending after the last round -- a region of
  // the texture or a level too many -- goes to global memory corner by corner, as all misses did before.
  // HOPELESS tiles (round 5): where a round catches only a handful of (tap, level) pairs, the taps are scattered beyond what
  // windows can hold -- the poles of an atlas, where neighbouring pixels sample texels thousands of columns apart: on the
  // textured benchmark 1.7 % of the tiles ran all five further rounds and still sent nearly all of their taps to global
  // memory afterwards (profiles/mipmap_bench.py --rounds-stats), 39 % of all tile-rounds.  The round after such a round is
  // the tile's last (what is pending goes to global memory at once).  The test is on PAIRS caught, not on a share: a tile
  // of a minified texture needs many windows and every one of its rounds catches hundreds of pairs -- cutting those short
  // (a first version tested the share of pixels that still had pending taps) cost 11 % on the minified scenes.
  // Same-box A/B of the threshold (textured benchmark / kernel_bench at 1 and 4 texels per pixel, ms): off 1.565 / 2.755 /
  // 7.74; 32 pairs 1.565 / 2.80 / 7.94; 96 1.56 / 2.98 / 8.23; 256 1.538 / 3.03 / 8.50 -- what the pole tiles save, the
  // tiles of a minified texture lose several times over: the switch stays off.
  int ref_now = ref, npend_before = 0;
  for (int round = 1;; ++round) {
    const bool again = __syncthreads_or(pending != 0) && !DRTK_DBG(dbg, 8);
    flush(ref_now, again);
    if (!again) return;
    if (tid == 0) DRTK_MIP_STAT(round < 7 ? round : 7, 1);
    bool last = round >= DRTK_MIP_ROUNDS - 1; // (or hopeless, below)
    __syncthreads(); // everybody has finished its flush (it reads the origins)
    if (tid == 0) s_ref = kMaxLevels, s_npend = 0;
    if (tid < kWinLevels) s_ox[tid] = s_oy[tid] = INT32_MAX, s_hx[tid] = s_hy[tid] = INT32_MIN;
    // where this pixel's pending taps are: bounding box of their north-west texels per level
    bool miss[2] = {false, false};
    int miss_x[2] = {INT32_MAX, INT32_MAX}, miss_y[2] = {INT32_MAX, INT32_MAX}, miss_hx[2] = {INT32_MIN, INT32_MIN}, miss_hy[2] = {INT32_MIN, INT32_MIN};
    for (uint32_t todo = pending; todo;) {
      const int bit = __builtin_ctz(todo);
      todo &= todo - 1;
      const int i = bit >> 1, s2 = bit & 1;
      T x, y;
      tap_xy(i, x, y);
      const int d = t.d1 + s2;
      const Quad<T> q = bilinear_quad<T>(x, y, s_h[d], s_w[d], padding, align_corners);
      if (s2 == 0) {
        miss[0] = true, miss_x[0] = min(miss_x[0], q.ix_nw), miss_y[0] = min(miss_y[0], q.iy_nw);
        miss_hx[0] = max(miss_hx[0], q.ix_nw), miss_hy[0] = max(miss_hy[0], q.iy_nw);
      } else {
        miss[1] = true, miss_x[1] = min(miss_x[1], q.ix_nw), miss_y[1] = min(miss_y[1], q.iy_nw);
        miss_hx[1] = max(miss_hx[1], q.ix_nw), miss_hy[1] = max(miss_hy[1], q.iy_nw);
      }
    }
    __syncthreads();
    {
      const int d_min = wave_min_i32(miss[0] ? t.d1 : miss[1] ? t.d1 + 1 : kMaxLevels);
      // the tile's pending (tap, level) pairs: a wave's count from five ballots over the bits of its lanes' counts (<= 32)
      const int pc = __popc(pending);
      int wave_pc = 0;
#pragma unroll
      for (int b = 0; b < 6; ++b) wave_pc += __popcll(__ballot((pc >> b) & 1)) << b;
      if ((tid & (kWave - 1)) == 0) {
        atomicMin(&s_ref, d_min);
        atomicAdd(&s_npend, wave_pc);
      }
    }
    __syncthreads();
    ref_now = s_ref;
    {
      // HOPELESS: the round before this one caught fewer than kHopelessPairs pairs -- scattered taps; this round is the last
      const int npend = s_npend;
      const bool hopeless = DRTK_MIP_HOPELESS && round >= 2 && npend_before - npend < DRTK_MIP_HOPELESS_PAIRS;
      if (tid == 0 && hopeless) DRTK_MIP_STAT(11, 1);
      npend_before = npend;
      last = last || hopeless;
    }
    {
      int lo_x[kWinLevels], lo_y[kWinLevels], hi_x[kWinLevels], hi_y[kWinLevels];
#pragma unroll
      for (int l = 0; l < kWinLevels; ++l) lo_x[l] = lo_y[l] = INT32_MAX, hi_x[l] = hi_y[l] = INT32_MIN;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int l = t.d1 + s2 - ref_now;
#pragma unroll
        for (int k = 0; k < kWinLevels; ++k) {
          if (miss[s2] && k == l) {
            lo_x[k] = min(lo_x[k], miss_x[s2]), lo_y[k] = min(lo_y[k], miss_y[s2]);
            hi_x[k] = max(hi_x[k], miss_hx[s2]), hi_y[k] = max(hi_y[k], miss_hy[s2]);
          }
        }
      }
#pragma unroll
      for (int l = 0; l < kWinLevels; ++l) {
        const int a = wave_min_i32(lo_x[l]), b = wave_min_i32(lo_y[l]);
        const int g = wave_max_i32(hi_x[l]), e = wave_max_i32(hi_y[l]);
        if ((tid & (kWave - 1)) == 0 && a != INT32_MAX) {
          atomicMin(&s_ox[l], a);
          atomicMin(&s_oy[l], b);
          atomicMax(&s_hx[l], g);
          atomicMax(&s_hy[l], e);
        }
      }
    }
    __syncthreads();
    {
      int lox[kWinLevels], loy[kWinLevels], hix[kWinLevels], hiy[kWinLevels];
#pragma unroll
      for (int l = 0; l < kWinLevels; ++l) {
        lox[l] = __builtin_amdgcn_readfirstlane(s_ox[l]), loy[l] = __builtin_amdgcn_readfirstlane(s_oy[l]);
        hix[l] = __builtin_amdgcn_readfirstlane(s_hx[l]), hiy[l] = __builtin_amdgcn_readfirstlane(s_hy[l]);
      }
      // (slot 0 always has pending taps: ref_now is the finest level that has any.  The boxes are those of the pending
      // taps' true north-west texels -- whatever the padding mode -- and only pending taps are accumulated in a round, so
      // the flush may stop at the box's last row)
      shape_slots(lox, loy, hix, hiy, false);
    }
    if (pending != 0) {
      uint32_t todo = pending;
      while (todo) {
        const int bit = __builtin_ctz(todo);
        todo &= todo - 1;
        const int i = bit >> 1, s2 = bit & 1;
        T x, y;
        tap_xy(i, x, y);
        const int d = t.d1 + s2;
        const int h = s_h[d], w = s_w[d];
        const int64_t plane = int64_t(h) * w;
        const T alpha = s2 == 0 ? alpha_2 : alpha_1;
        const Quad<T> q = bilinear_quad<T>(x, y, h, w, padding, align_corners);
        const int l = d - ref_now;
        const int cell = slot_cell(l, q.ix_nw, q.iy_nw);
        const int stride = slot_stride(l), chan = slot_chan(l);
        if (cell < 0 && !last) continue; // stays pending: the next round's windows
        pending &= ~(1u << bit);
        if (cell < 0) DRTK_MIP_STAT(10, 1);
        const GlobalPtr<T> ginp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + int64_t(n) * C * plane);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c >= C) break;
          const T gc = go[c] * alpha;
          if (gc == T(0)) continue;
          if (cell >= 0) {
            double* wp = s_win + slot_base(l) + c * chan + cell;
            if (q.o_nw >= 0) lds_add(wp, static_cast<double>(q.nw * gc));
            if (q.o_ne >= 0) lds_add(wp + 1, static_cast<double>(q.ne * gc));
            if (q.o_sw >= 0) lds_add(wp + stride, static_cast<double>(q.sw * gc));
            if (q.o_se >= 0) lds_add(wp + stride + 1, static_cast<double>(q.se * gc));
          } else {
            const GlobalPtr<T> gp = ginp + c * plane;
            if (q.o_nw >= 0) atomic_add_g1(gp + q.o_nw, q.nw * gc);
            if (q.o_ne >= 0) atomic_add_g1(gp + q.o_ne, q.ne * gc);
            if (q.o_sw >= 0) atomic_add_g1(gp + q.o_sw, q.sw * gc);
            if (q.o_se >= 0) atomic_add_g1(gp + q.o_se, q.se * gc);
          }
        }
      }
    }
  }
}

// ---- Backward, bilinear, any padding, any C, float and double: the tile kernel with the LEAN tap loop (round 5; reflection
// and double: round 6) ----------------------------------------------------------------------------------------------------
// mipmap_backward_tiled2_kernel's placement, shaped window slots, rounds and flush, with the tap loop rewritten the way
// the lean forward was (one straight line per (tap, level) for the whole wave: no Quad, no per-corner logic, regrouped
// products, texture channels a template parameter) -- for the instructions, and for the REGISTERS: the tile kernels sat at
// 151-168 VGPRs = 3 waves per SIMD with every tile's chain (loads -> barrier -> placement -> barrier -> taps -> barrier ->
// flush) exposed; at <= 128 a CU holds four tiles instead of three.
// Wider textures (C > 4: neural textures) run it once per block of four channels (`C_total`, `c0`; the grid gradient
// accumulates over the blocks, the tap geometry is recomputed per block) instead of the wave-private kernel below.
// Tiles per CU (= waves per SIMD), up to three channels: FIVE since late round 5 -- 94-96 registers (the pixel's index is
// rebuilt where it is needed instead of being carried: it was what spilled) and windows of 2 x 512 accumulators per channel
// (24.5 KB for RGB).  Same-box A/B, ms (textured benchmark 2 x 4096^2 / 8 x 2048^2 / kernel_bench's minified scenes at 1 and 4
// texels per pixel): four tiles with 2 x 768 cells 1.383 / 1.99 / 2.13 / 4.35; four with 2 x 512 1.42 / 2.12 / 2.27 / 4.85; FIVE
// with 2 x 512 1.28 / 1.95 / 2.11 / 4.53 -- the fifth tile is worth more than the larger windows except where a tile's taps
// cover four texels per pixel.  (2 x 640 cells = 32 144 B per tile is five tiles by the occupancy query, profiles/micro/
// lds_occupancy.hip, and runs like four: 1.41 / - / 2.23 / 4.74.)  Four channels per pass: DRTK_MIP_T3_OCC4.
#ifndef DRTK_MIP_T3_OCC
#define DRTK_MIP_T3_OCC 5
#endif
#ifndef DRTK_MIP_T3_OCC_F64
#define DRTK_MIP_T3_OCC_F64 3 // double: 126-152 registers = three tiles per CU (asked for four it spills 4-17 of them)
#endif
#ifndef DRTK_MIP_T3_OCC4
#define DRTK_MIP_T3_OCC4 5 // four channels per pass: 96 registers, windows of 2 x 384 accumulators (24.6 KB): C = 4 / 8 / 16 1.62 / 3.17 / 6.33 -> 1.51 / 2.92 / 5.90 ms
#endif
#ifndef DRTK_MIP_ROWS_OUTSIDE_IN
#define DRTK_MIP_ROWS_OUTSIDE_IN 3
#endif
#ifndef DRTK_MIP_ROUND_BY_TAP
#define DRTK_MIP_ROUND_BY_TAP 1
#endif
// Same-box A/B (textured benchmark incl. the pyramid's zero-fill / kernel_bench at 1 and 4 texels per pixel, ms): placement by
// everything pending, 6 rounds (the first half of round 5) 1.53 / 2.30 / 5.90; by tap: 24 rounds, hopeless below 24 pairs
// 1.44 / 2.18 / 4.48; 32 rounds, 16 pairs 1.44 / 2.16 / 4.35; 24 rounds, no cut-off 1.70 / 2.43 / 4.71; 12 rounds 1.58 / 2.32 / 4.59
#ifndef DRTK_MIP_LEAN_ROUNDS
#define DRTK_MIP_LEAN_ROUNDS 32
#endif
#ifndef DRTK_MIP_LEAN_HOPELESS_PAIRS
#define DRTK_MIP_LEAN_HOPELESS_PAIRS 16 // a round that catches fewer (tap, level) pairs than this is the tile's last but one
#endif
// window accumulators per channel and slot (two slots; one slot of twice the size where a tile has one live level).  1024 =
// the square windows' memory of round 4 (48 KB for RGB: 3 tiles per CU); 768 -> 36 KB: 4 tiles (the first half of round 5);
// 512 -> 24.5 KB: 5 tiles, with the registers that go with them (DRTK_MIP_T3_OCC above).
#ifndef DRTK_MIP_T3_SLOT_CELLS
#define DRTK_MIP_T3_SLOT_CELLS 512 // (768 until late round 5: see DRTK_MIP_T3_OCC)
#endif
#ifndef DRTK_MIP_T3_SLOT_CELLS12
#define DRTK_MIP_T3_SLOT_CELLS12 768 // one or two channels: 12-24 KB, five tiles either way (C = 1 / 2: 0.91 / 1.08 ms; with 512: 0.94 / 1.09)
#endif
#ifndef DRTK_MIP_T3_SLOT_CELLS4
#define DRTK_MIP_T3_SLOT_CELLS4 384 // four channels: 2 x 384 x 4 x 8 B = 24.6 KB: five tiles per CU (512: 32 KB, four)
#endif
template <int CN>
constexpr int lean_slot_cells() { return CN == 4 ? DRTK_MIP_T3_SLOT_CELLS4 : CN == 3 ? DRTK_MIP_T3_SLOT_CELLS : DRTK_MIP_T3_SLOT_CELLS12; }
static_assert(DRTK_MIP_T3_SLOT_CELLS % 128 == 0 && DRTK_MIP_T3_SLOT_CELLS4 % 128 == 0 && DRTK_MIP_T3_SLOT_CELLS12 % 128 == 0, "whole rows at every slot width (16 ... 128 cells), cells in pairs");
template <typename T, int PAD, bool ALIGN, int CN>
// (reflection padding, round 6: the reflect + clip of both axes takes the kernel 2-4 registers over the 96 of five tiles per CU;
// compiled for four -- 128 registers, no spill; the windows keep the five-tile size)
__global__ __launch_bounds__(kMipBlock, sizeof(T) == 8 ? DRTK_MIP_T3_OCC_F64 : PAD == 2 ? 4 : (CN <= 3 ? DRTK_MIP_T3_OCC : DRTK_MIP_T3_OCC4)) void mipmap_backward_lean_kernel(
    LevelTable lv, int mipmaps, const T* __restrict__ grad_out, const T* __restrict__ grid, GridLayout gl,
    const T* __restrict__ vt, int H, int W, int tiles_x, int max_aniso,
    bool force_max_aniso, bool clip_grad, T* __restrict__ grad_grid, GridLayout ggl, int strip, int dbg, int C_total, int c0) {
  static_assert(PAD >= 0 && PAD <= 2, "zeros, border or reflection padding");
  static_assert(CN >= 1 && CN <= 4, "texture channels");
  constexpr int C = CN;
  constexpr int padding = PAD;
  constexpr bool align_corners = ALIGN;
  __shared__ double s_f[kTapTab * kTapTab];
  __shared__ const void* s_ptr[kMaxLevels];
  __shared__ void* s_grad[kMaxLevels];
  __shared__ int s_h[kMaxLevels], s_w[kMaxLevels];
  __shared__ long long s_sn[kMaxLevels];
  __shared__ int s_ref, s_npend, s_ox[kWinLevels], s_oy[kWinLevels], s_hx[kWinLevels], s_hy[kWinLevels];
  // rounds: the lowest pending (tap, level) bit of the tile (two cells, by round parity), and the placement by THAT tap alone
  __shared__ int s_bit[2], s_refb, s_bx[kWinLevels], s_by[kWinLevels], s_bhx[kWinLevels], s_bhy[kWinLevels];
  __shared__ unsigned long long s_seed[kWinLevels];
  // first-round placement per ABSOLUTE level (so that the reference level and the origins come out of ONE phase):
  // bounding box of the north-west texels of the tile's taps on level d
  __shared__ int s_lox[kMaxLevels + 1], s_loy[kMaxLevels + 1], s_hix[kMaxLevels + 1], s_hiy[kMaxLevels + 1];
  extern __shared__ __attribute__((aligned(16))) unsigned char s_win_raw[];
  double* const s_win = reinterpret_cast<double*>(s_win_raw);
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  // The tile rows of the launch's LAST view are dispatched from the outside in (0, last, 1, last - 1, ...).  A tile's life varies
  // 20x (11 us; 140+ where a limb tile needs nine further rounds), and tiles that are dispatched last and live longest are a
  // tail every other CU waits for: on BASELINE configs[4]'s shape (2 x 4096^2) the last 1 % of the tiles -- the bottom limb of
  // the last view -- finished 0.17 ms after the others (per-tile timeline, profiles/r05/mipmap_tile_times.txt).  Silhouettes
  // and grazing angles, the expensive tiles, lie around an object; background and the magnified interior are cheap: the
  // interior goes last.  Same-box A/B, ms (2 x 4096^2 / 8 x 2048^2 / kernel_bench's minified scenes at 1 and 4 texels per
  // pixel): in order 1.46 / 1.98 / 2.14 / 4.35; every view outside-in 1.35 / 2.00 / 2.18 / 4.53 (two bands of tiles in
  // flight: less reuse between neighbouring tile rows); groups of eight rows outside-in 1.41 / 2.03 / 2.21 / 4.45; the last
  // view only 1.37 / 1.98 / 2.12 / 4.33.
  const int tile_disp = tile_index(strip);
  DRTK_MIP_TILE_T0();
  const int ty_disp = tile_disp / tiles_x, tx = tile_disp - ty_disp * tiles_x;
  const int tiles_y = (H + kTileH - 1) / kTileH;
  int ty = ty_disp;
  if (DRTK_MIP_ROWS_OUTSIDE_IN == 1 || (DRTK_MIP_ROWS_OUTSIDE_IN == 3 && blockIdx.y + 1 == gridDim.y)) { // 3: the launch's last view only
    ty = (ty_disp & 1) ? tiles_y - 1 - (ty_disp >> 1) : (ty_disp >> 1);
  }
  const int tile = ty * tiles_x + tx;
  (void)tile;
  const int px = tx * kTileW + (tid & (kTileW - 1)), py = ty * kTileH + tid / kTileW;
  const bool valid = px < W && py < H;
  const int64_t HW = int64_t(H) * W;
  const int64_t index = int64_t(n) * HW + int64_t(py) * W + px;
  // the pixel's upstream gradient, uv and Jacobian: one batch of loads ...
  T go[CN];
#pragma unroll
  for (int c = 0; c < CN; ++c) go[c] = T(0);
  if (valid) {
    const T* gout_px = grad_out + (int64_t(n) * C_total + c0) * HW + (int64_t(py) * W + px);
#pragma unroll
    for (int c = 0; c < CN; ++c) go[c] = gout_px[int64_t(c) * HW]; // (as non-temporal loads, with the Jacobian image: 1.283 vs 1.290 ms, round 6 -- nothing)
  }
  PixelUV<T> uv = {};
  // (the pixel's index is rebuilt where it is needed -- here, at an early exit, after the tap loop -- from a laundered thread id:
  // carried through the kernel its two registers are the ones that spill at five waves per SIMD)
  auto pixel_index = [&]() -> int64_t {
    int tid_l = tid;
    asm volatile("" : "+v"(tid_l));
    return int64_t(ty * kTileH + tid_l / kTileW) * W + (tx * kTileW + (tid_l & (kTileW - 1)));
  };
  if (valid) uv = load_pixel_uv<T>(grid, gl, vt, n, int64_t(py) * W + px, index);
  // ... and under them everything that needs no data: the tables, the placement cells, the windows' zero-fill (a tile
  // that turns out to have no upstream gradient has zeroed its windows for nothing; the stores wait for nobody)
  stage_tap_table(s_f);
  if (tid < kMaxLevels) {
    const int i = tid < mipmaps ? tid : 0;
    s_ptr[tid] = lv.ptr[i], s_grad[tid] = lv.grad[i], s_sn[tid] = lv.sn[i], s_h[tid] = lv.h[i], s_w[tid] = lv.w[i];
  }
  if (tid <= kMaxLevels) s_lox[tid] = s_loy[tid] = INT32_MAX, s_hix[tid] = s_hiy[tid] = INT32_MIN;
  if (tid == 0) s_ref = kMaxLevels, s_bit[0] = s_bit[1] = 32;
  {
    double2* w2 = reinterpret_cast<double2*>(s_win);
    const double2 z = {0.0, 0.0};
    for (int i = tid; i < C * (kWinLevels * lean_slot_cells<CN>()) / 2; i += kMipBlock) w2[i] = z;
  }
  bool has_go = false;
#pragma unroll
  for (int c = 0; c < CN; ++c) has_go = has_go | (go[c] != T(0));
  if (!__syncthreads_or(has_go)) { // (also publishes the tables and the zero-fill)
    if (valid && c0 == 0) store_grid_grad<T>(grad_grid, ggl, n, pixel_index(), T(0), T(0)); // (a later channel block adds nothing)
    return;
  }
  DRTK_MIP_TILE_PHASE(0);

  Taps<T> t = {};
  if (has_go) t = setup_taps<T>(uv, lv.h[0], lv.w[0], mipmaps, max_aniso, force_max_aniso, clip_grad);
  const int n_lv = mipmaps > 1 ? 2 : 1;
  // the levels' weights a / n and (1 - a) / n: continuous quantities, in float from one reciprocal (as in the lean forward)
  const T rn = T(1) / static_cast<T>(max(t.n, 1));
  const T alpha_1 = has_go ? t.a * rn : T(0);
  const T alpha_2 = has_go ? (T(1) - t.a) * rn : T(0);
  // A (pixel, level) whose weighted upstream gradient is zero in every channel adds nothing anywhere (every term is
  // +-0 * finite): the masked background of a silhouette tile, and the second level of a magnified pixel (a == 0).
  // Such pairs neither place the windows nor run their taps.
  bool live[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const T alpha = s == 0 ? alpha_2 : alpha_1;
    live[s] = false;
#pragma unroll
    for (int c = 0; c < CN; ++c) live[s] = live[s] || (go[c] * alpha != T(0));
    live[s] = live[s] && s < n_lv;
  }
  auto tap_xy = [&](int i, T& x, T& y) {
    const double f = tap_f(s_f, i, t.n);
    x = t.u + static_cast<T>(t.du * f);
    y = t.v + static_cast<T>(t.dv * f);
  };
  // window origins: the taps of a pixel are collinear, so their extreme texels are those of the first and the last
  // tap.  ANY origin is correct -- a corner is windowed iff its exact cell lies inside, tested below -- so the origin
  // comes from a short form of the coordinate pipeline: unnormalise, clamp to the level, floor (the exact north-west
  // texel for zeros / border padding; under reflection padding taps beyond the border miss the window).
  auto texel_floor = [&](T coord, int size) -> int {
    T unused;
    const T c = unnormalize(coord, size, align_corners, &unused);
    const T lo = padding == 0 ? T(-1) : T(0);
    return static_cast<int>(floor(fmin(fmax(c, lo), static_cast<T>(size - 1)))); // fmax(NaN, lo) = lo
  };
  {
    // ONE phase for the reference level and the origins: the extremes go into cells of their ABSOLUTE level, wave by
    // wave over the (one to three) levels a wave's pixels use
    int lo_x[2] = {INT32_MAX, INT32_MAX}, lo_y[2] = {INT32_MAX, INT32_MAX}, hi_x[2] = {INT32_MIN, INT32_MIN}, hi_y[2] = {INT32_MIN, INT32_MIN};
    if (live[0] || live[1]) {
      for (int e = 0; e < 2; ++e) {
        T x, y;
        tap_xy(e == 0 ? 0 : t.n - 1, x, y);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s < n_lv && live[s]) {
            const int ox = texel_floor(x, s_w[t.d1 + s]), oy = texel_floor(y, s_h[t.d1 + s]);
            lo_x[s] = min(lo_x[s], ox), lo_y[s] = min(lo_y[s], oy), hi_x[s] = max(hi_x[s], ox), hi_y[s] = max(hi_y[s], oy);
          }
        }
      }
    }
    const int d_lo = wave_min_i32(live[0] ? t.d1 : live[1] ? t.d1 + 1 : kMaxLevels);
    const int d_hi = wave_max_i32(live[1] ? t.d1 + 1 : live[0] ? t.d1 : -1);
    for (int d = d_lo; d <= d_hi; ++d) { // wave-uniform
      const bool m0 = live[0] && t.d1 == d, m1 = live[1] && t.d1 + 1 == d;
      const int a = wave_min_i32(min(m0 ? lo_x[0] : INT32_MAX, m1 ? lo_x[1] : INT32_MAX));
      const int b = wave_min_i32(min(m0 ? lo_y[0] : INT32_MAX, m1 ? lo_y[1] : INT32_MAX));
      const int e = wave_max_i32(max(m0 ? hi_y[0] : INT32_MIN, m1 ? hi_y[1] : INT32_MIN));
      const int g = wave_max_i32(max(m0 ? hi_x[0] : INT32_MIN, m1 ? hi_x[1] : INT32_MIN));
      if ((tid & (kWave - 1)) == 0 && a != INT32_MAX) {
        atomicMin(&s_lox[d], a);
        atomicMin(&s_loy[d], b);
        atomicMax(&s_hix[d], g);
        atomicMax(&s_hiy[d], e);
      }
    }
    if ((tid & (kWave - 1)) == 0 && d_lo < kMaxLevels) atomicMin(&s_ref, d_lo);
  }
  __syncthreads();
  DRTK_MIP_TILE_PHASE(1);
  const int ref = s_ref;
  // ---- the two window SLOTS (round 5: shaped, not square).  Slot l holds level ref + l.  The windows' memory is
  // kWinLevels x kWinSlotCells accumulators per channel; a tile with no live tap on level ref + 1 -- every tile of a magnified
  // texture -- gives it all to slot 0; a slot is 2^sx cells wide, sx chosen by the bounding box of the taps it has to
  // hold (16 ... 128 wide: the elongated footprints of a limb tile fit where a 32 x 32 square needed four rounds).
  // Wave-uniform values: scalar registers.
  constexpr int kWinSlotCells = lean_slot_cells<CN>();
  int wox[kWinLevels], woy[kWinLevels], wsx[kWinLevels], wny[kWinLevels], wcells[kWinLevels], win_rows[kWinLevels];
  auto shape_slots = [&](const int (&lox)[kWinLevels], const int (&loy)[kWinLevels], const int (&hix)[kWinLevels], const int (&hiy)[kWinLevels], bool all_rows) {
    const bool two = lox[1] != INT32_MAX;
#pragma unroll
    for (int l = 0; l < kWinLevels; ++l) {
      wox[l] = lox[l], woy[l] = loy[l];
      wcells[l] = l == 0 ? (two ? kWinSlotCells : kWinLevels * kWinSlotCells) : (two ? kWinSlotCells : 0);
      const long long need_w = static_cast<long long>(hix[l]) - lox[l] + 2, need_h = static_cast<long long>(hiy[l]) - loy[l] + 2;
      int sx = need_w <= 16 ? 4 : need_w <= 32 ? 5 : (need_w <= 64 && need_h <= (wcells[l] >> 6)) ? 6 : (need_w > 64 && need_h <= (wcells[l] >> 7)) ? 7 : 5;
      if (lox[l] == INT32_MAX) sx = 5;
      wsx[l] = sx;
      wny[l] = wcells[l] > 0 ? (wcells[l] >> sx) : 1; // (1: no cell passes the row test of an empty slot)
      win_rows[l] = (all_rows || need_h > wny[l]) ? wny[l] : static_cast<int>(need_h < 0 ? 0 : need_h);
      if (wcells[l] == 0) win_rows[l] = 0;
    }
  };
  {
    int lox[kWinLevels], loy[kWinLevels], hix[kWinLevels], hiy[kWinLevels];
#pragma unroll
    for (int l = 0; l < kWinLevels; ++l) {
      const int d = min(ref + l, kMaxLevels);
      lox[l] = __builtin_amdgcn_readfirstlane(s_lox[d]), loy[l] = __builtin_amdgcn_readfirstlane(s_loy[d]);
      hix[l] = __builtin_amdgcn_readfirstlane(s_hix[d]), hiy[l] = __builtin_amdgcn_readfirstlane(s_hiy[d]);
    }
    // rows the first round can touch: exact for zeros / border padding, where the extremes are the true north-west texels;
    // under reflection padding every row
    shape_slots(lox, loy, hix, hiy, padding == 2);
  }
  // cell of a north-west texel in slot l (the other three corners are +1, +stride, +stride+1), or -1
  auto slot_cell = [&](int l, int ix_nw, int iy_nw) -> int {
    if (l < 0 || l >= kWinLevels) return -1;
    const int wx = ix_nw - (l == 0 ? wox[0] : wox[1]), wy = iy_nw - (l == 0 ? woy[0] : woy[1]);
    // (rows: those the flush walks -- the bounding box the slot was shaped for, where that is less than the slot)
    const int sx = l == 0 ? wsx[0] : wsx[1], ny = l == 0 ? win_rows[0] : win_rows[1];
    return (static_cast<unsigned>(wx) < (1u << sx) - 1u && wy >= 0 && wy < ny - 1) ? (wy << sx) + wx : -1;
  };
  auto slot_stride = [&](int l) -> int { return 1 << (l == 0 ? wsx[0] : wsx[1]); };
  auto slot_chan = [&](int l) -> int { return l == 0 ? wcells[0] : wcells[1]; };  // cells per channel
  auto slot_base = [&](int l) -> int { return l == 0 ? 0 : C * wcells[0]; };        // first cell of the slot's channel 0

  // (tap, level) pairs that find no window cell in this round: not sent to global memory one corner and channel at a
  // time -- scattered float atomics of single lanes, 0.72 of this kernel's 2.2 ms on the textured benchmark although
  // only a few per cent of the taps miss -- but remembered (per level of the pixel: did any tap miss, and where) for a
  // further round with the windows moved onto them (below, up to DRTK_MIP_ROUNDS rounds).
  // (where they missed is NOT carried through the tap loop -- six registers at its bound: a round recomputes the
  // north-west texels of its pending taps, which it walks anyway)
  uint32_t pending = 0; // bit 2 i + s: tap i on the pixel's level s found no window cell yet (taps >= 16 are never deferred)

  // ---- the tap loop, lean (see mipmap_forward_lean_kernel): ONE straight line per (tap, level) for the whole wave.  Cell
  // decisions are the reference's operations in the reference's order; the continuous quantities are regrouped (corner
  // weights as products of the two axis weights, the grid gradient as differences of texels: 10 operations per channel
  // instead of 24).  A lane whose tap is not interior runs the same loads and products with offset 0 and weight 0; border
  // taps and the (rare) taps beyond the deferral range are added by a branch the wave only takes if it has one.
  T acc_x = T(0), acc_y = T(0);
  {
    const int n_max = wave_max_i32((live[0] || live[1]) ? t.n : 0);
    const bool table = max_aniso <= kTapTab; // kernel-uniform
    const int tn1 = max(t.n, 1);
    const double du_d = t.du, dv_d = t.dv;
    auto grad_base = [&](int d, int64_t plane) -> GlobalPtr<T> { return (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + (int64_t(n) * C_total + c0) * plane); };
    // (one run slot, for the pixel's FINER level: a magnified pixel has no other live level, and the taps of a minified
    // one are a texel apart on the finer level -- half a texel on the coarser, whose adds go to LDS tap by tap)
    int run_cell = -1;
    T run_w[4] = {T(0), T(0), T(0), T(0)};
    auto emit_run = [&](int l, const T (&g)[CN]) {
      if (run_cell < 0) return;
      double* wp = s_win + slot_base(l) + run_cell;
      const int stride = slot_stride(l), chan = slot_chan(l);
#pragma unroll
      for (int c = 0; c < CN; ++c) {
        lds_add(wp + c * chan, static_cast<double>(run_w[0] * g[c]));
        lds_add(wp + c * chan + 1, static_cast<double>(run_w[1] * g[c]));
        lds_add(wp + c * chan + stride, static_cast<double>(run_w[2] * g[c]));
        lds_add(wp + c * chan + stride + 1, static_cast<double>(run_w[3] * g[c]));
      }
      run_cell = -1;
    };
    for (int i = 0; i < n_max && !DRTK_DBG(dbg, 32); ++i) {
      const bool has_tap = i < t.n;
      const int ic = min(i, tn1 - 1); // a lane that has no tap i recomputes its last one (its weights are zeroed)
      const double f = table ? s_f[(tn1 - 1) * kTapTab + ic] : (ic + 1.0) / (t.n + 1.0) * 2.0 - 1.0;
      const T x = t.u + static_cast<T>(du_d * f), y = t.v + static_cast<T>(dv_d * f);
      const bool ordered = (x == x) & (y == y); // a NaN coordinate samples nothing (safe_downgrade_to_int_range: -100)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bool on = has_tap & live[s];
        if (__ballot(on) == 0) continue; // wave-uniform
        const int d = on ? t.d1 + s : 0;
        const int w = s_w[d], h = s_h[d];
        const T wm1 = static_cast<T>(w - 1), hm1 = static_cast<T>(h - 1);
        // unnormalize (GridSampler.cuh): the position and the gradient multiplier of the coordinate transform
        T ixu, iyu, mx, my;
        if (align_corners) {
          ixu = ((x + 1.f) / 2) * wm1, iyu = ((y + 1.f) / 2) * hm1;
          mx = wm1 / 2, my = hm1 / 2;
        } else {
          ixu = ((x + 1.f) * static_cast<T>(w) - 1) / 2, iyu = ((y + 1.f) * static_cast<T>(h) - 1) / 2;
          mx = static_cast<T>(w) / 2, my = static_cast<T>(h) / 2;
        }
        T ix, iy;
        bool sane = true; // (reflection: a coordinate the short pipeline must not touch goes to the corner-by-corner branch)
        if (padding == 1) { // clip_coord: clamped coordinates have a zero gradient
          mx = ((ixu > T(0)) & (ixu < wm1)) ? mx : T(0), my = ((iyu > T(0)) & (iyu < hm1)) ? my : T(0);
          ix = tmin(tmax(ixu, T(0)), wm1), iy = tmin(tmax(iyu, T(0)), hm1);
        } else if (padding == 2) { // reflect_coord, then clip_coord (source_index's order)
          sane = (tabs(ixu) < T(1e9f)) & (tabs(iyu) < T(1e9f));
          reflect_clip_lean<T>(sane ? ixu : T(0), sane ? iyu : T(0), w, h, align_corners, ix, iy, mx, my);
        } else { // (made finite; a tap whose cell lies inside the level is not moved)
          ix = tmin(tmax(ixu, T(-4)), T(2e9f)), iy = tmin(tmax(iyu, T(-4)), T(2e9f));
        }
        const T fx_floor = floor(ix), fy_floor = floor(iy);
        const int ix_nw = static_cast<int>(fx_floor), iy_nw = static_cast<int>(fy_floor);
        const bool interior = on & ordered & sane & (static_cast<unsigned>(ix_nw) < static_cast<unsigned>(w - 1)) &
            (static_cast<unsigned>(iy_nw) < static_cast<unsigned>(h - 1));
        const T wx1 = (fx_floor + T(1)) - ix, wx0 = ix - fx_floor, wy1 = (fy_floor + T(1)) - iy, wy0 = iy - fy_floor;
        const int l = d - ref;
        const int cell = interior ? slot_cell(l, ix_nw, iy_nw) : -1;
        const T al = interior ? (s == 0 ? alpha_2 : alpha_1) : T(0);
        T g[CN];
#pragma unroll
        for (int c = 0; c < CN; ++c) g[c] = go[c] * al;
        // ---- texture gradient: into the window (merged per cell on the finer level), or pending
        if (!DRTK_DBG(dbg, 1)) {
          if (cell >= 0) {
            const T q_nw = wx1 * wy1, q_ne = wx0 * wy1, q_sw = wx1 * wy0, q_se = wx0 * wy0;
            if (s == 1) {
              double* wp = s_win + slot_base(l) + cell;
              const int stride = slot_stride(l), chan = slot_chan(l);
#pragma unroll
              for (int c = 0; c < CN; ++c) {
                lds_add(wp + c * chan, static_cast<double>(q_nw * g[c]));
                lds_add(wp + c * chan + 1, static_cast<double>(q_ne * g[c]));
                lds_add(wp + c * chan + stride, static_cast<double>(q_sw * g[c]));
                lds_add(wp + c * chan + stride + 1, static_cast<double>(q_se * g[c]));
              }
            } else if (cell != run_cell) {
              emit_run(l, g);
              run_cell = cell;
              run_w[0] = q_nw, run_w[1] = q_ne, run_w[2] = q_sw, run_w[3] = q_se;
            } else {
              run_w[0] += q_nw, run_w[1] += q_ne, run_w[2] += q_sw, run_w[3] += q_se;
            }
          } else if (interior & (i < 16)) {
            pending |= 1u << (2 * i + s);
          }
        }
        // ---- grid gradient: texels of the (interior) tap; d/dx of the bilinear form = differences of texels
        const int plane = w * h; // <= kLeanMaxPlane (the dispatch): c * plane + offset fits 32 bits
        const GlobalPtr<const T> inp = (GlobalPtr<const T>)(static_cast<const T*>(s_ptr[d]) + int64_t(n) * s_sn[d]) + int64_t(c0) * plane;
        // (round 6: the loads run under `interior`, one exec-mask region -- the lanes without an interior tap used to load the
        // pair at offset 0 of their level: on a 1 x 1 level, one element beyond the last channel of the last view.  What
        // their registers hold instead is never used: g[c] is zero on those lanes)
        const int o_top = iy_nw * w + ix_nw;
        const int o_bot = o_top + w;
        Pair<T> top[CN], bot[CN];
#pragma unroll
        for (int c = 0; c < CN; ++c) {
          top[c] = __builtin_nondeterministic_value(top[c]);
          bot[c] = __builtin_nondeterministic_value(bot[c]);
        }
        if (interior && !DRTK_DBG(dbg, 2)) {
#pragma unroll
          for (int c = 0; c < CN; ++c) {
            top[c] = *(GlobalPtr<const Pair<T>>)(inp + (c * plane + o_top));
            bot[c] = *(GlobalPtr<const Pair<T>>)(inp + (c * plane + o_bot));
          }
        }
        T gix = T(0), giy = T(0);
#pragma unroll
        for (int c = 0; c < CN; ++c) {
          if (DRTK_DBG(dbg, 16)) break;
          // (with a zero upstream gradient every term is +-0 * finite in the reference: the texels count as 0)
          const T tx = (top[c].y - top[c].x) * wy1 + (bot[c].y - bot[c].x) * wy0;
          const T ty = (bot[c].x - top[c].x) * wx1 + (bot[c].y - top[c].y) * wx0;
          gix += g[c] != T(0) ? g[c] * tx : T(0);
          giy += g[c] != T(0) ? g[c] * ty : T(0);
        }
        acc_x += mx * gix;
        acc_y += my * giy;
        // ---- the lanes the straight line does not serve: a tap on the border of its level (or with a non-finite coordinate),
        // and interior taps beyond the deferral range that found no window (texture gradient straight to global memory)
        const bool slow = on & (!interior | ((cell < 0) & (i >= 16)));
        if (__ballot(slow) != 0) {
          if (slow) {
            const T alpha = s == 0 ? alpha_2 : alpha_1;
            const Quad<T> q = bilinear_quad<T>(x, y, h, w, padding, align_corners);
            const int ix_se = q.ix_nw + 1, iy_se = q.iy_nw + 1;
            const int qcell = slot_cell(l, q.ix_nw, q.iy_nw);
            const int stride = slot_stride(l), chan = slot_chan(l);
            const bool defer = !interior && !DRTK_DBG(dbg, 1) && qcell < 0 && i < 16 && (q.o_nw & q.o_ne & q.o_sw & q.o_se) != -1; // (all four corners outside the level: nothing to add anywhere)
            if (defer) pending |= 1u << (2 * i + s);
            T sgx = T(0), sgy = T(0);
#pragma unroll 1
            for (int c = 0; c < CN; ++c) {
              const T gOut = (c == 0 ? go[0] : c == 1 ? go[CN > 1 ? 1 : 0] : c == 2 ? go[CN > 2 ? 2 : 0] : go[CN > 3 ? 3 : 0]) * alpha;
              if (gOut == T(0)) continue; // a zero upstream gradient (masked channel) adds nothing; its texels count as 0
              if (!DRTK_DBG(dbg, 1)) {
                if (qcell >= 0 && !interior) {
                  double* wp = s_win + slot_base(l) + c * chan + qcell;
                  if (q.o_nw >= 0) lds_add(wp, static_cast<double>(q.nw * gOut));
                  if (q.o_ne >= 0) lds_add(wp + 1, static_cast<double>(q.ne * gOut));
                  if (q.o_sw >= 0) lds_add(wp + stride, static_cast<double>(q.sw * gOut));
                  if (q.o_se >= 0) lds_add(wp + stride + 1, static_cast<double>(q.se * gOut));
                } else if (!defer) {
                  const GlobalPtr<T> gp = grad_base(d, plane) + c * plane;
                  if (q.o_nw >= 0) atomic_add_g1(gp + q.o_nw, q.nw * gOut);
                  if (q.o_ne >= 0) atomic_add_g1(gp + q.o_ne, q.ne * gOut);
                  if (q.o_sw >= 0) atomic_add_g1(gp + q.o_sw, q.sw * gOut);
                  if (q.o_se >= 0) atomic_add_g1(gp + q.o_se, q.se * gOut);
                }
              }
              if (interior) continue; // (its grid gradient is in the straight line above)
              const GlobalPtr<const T> pch = inp + c * plane; // (c <= 3, plane <= kLeanMaxPlane)
              T v_nw = T(0), v_ne = T(0), v_sw = T(0), v_se = T(0);
              if (!DRTK_DBG(dbg, 2)) {
                if (q.o_nw >= 0) v_nw = pch[q.o_nw];
                if (q.o_ne >= 0) v_ne = pch[q.o_ne];
                if (q.o_sw >= 0) v_sw = pch[q.o_sw];
                if (q.o_se >= 0) v_se = pch[q.o_se];
              }
              if (q.o_nw >= 0) {
                sgx -= v_nw * (iy_se - q.iy) * gOut;
                sgy -= v_nw * (ix_se - q.ix) * gOut;
              }
              if (q.o_ne >= 0) {
                sgx += v_ne * (iy_se - q.iy) * gOut;
                sgy -= v_ne * (q.ix - q.ix_nw) * gOut;
              }
              if (q.o_sw >= 0) {
                sgx -= v_sw * (q.iy - q.iy_nw) * gOut;
                sgy += v_sw * (ix_se - q.ix) * gOut;
              }
              if (q.o_se >= 0) {
                sgx += v_se * (q.iy - q.iy_nw) * gOut;
                sgy += v_se * (q.ix - q.ix_nw) * gOut;
              }
            }
            acc_x += q.mx * sgx;
            acc_y += q.my * sgy;
          }
        }
      }
    }
    // the last run
    if (live[0]) {
      T g[CN];
#pragma unroll
      for (int c = 0; c < CN; ++c) g[c] = go[c] * alpha_2;
      emit_run(t.d1 - ref, g);
    }
  }
  if (valid) {
    const int64_t pix = pixel_index();
    if (c0 != 0) { // a further block of channels of a wide texture: the grid gradient accumulates (one thread per pixel, launches in stream order)
      const T* gq = grad_grid + int64_t(n) * ggl.sN + pix * ggl.sP;
      acc_x += gq[0], acc_y += gq[ggl.sC];
    }
    store_grid_grad<T>(grad_grid, ggl, n, pix, acc_x, acc_y);
    DRTK_MIP_STAT(8, (live[0] ? t.n : 0) + (live[1] ? t.n : 0));
    DRTK_MIP_STAT(9, __popc(pending));
  }
  if (tid == 0) DRTK_MIP_STAT(0, 1);
  __syncthreads();
  DRTK_MIP_TILE_PHASE(2);
  // flush the windows: consecutive threads = consecutive texels of a row; cells that stayed 0 cost nothing,
  // cells outside the level were never written (only in-bounds corners are accumulated).  With `rearm` the cells are
  // zeroed as they are read: the windows serve a second round.
  auto flush = [&](int ref_level, bool rearm) {
#pragma unroll
    for (int l = 0; l < kWinLevels; ++l) {
      const int d = ref_level + l;
      if (d >= mipmaps || wox[l] == INT32_MAX || DRTK_DBG(dbg, 4)) continue;
      const int h = s_h[d], w = s_w[d];
      const int64_t plane = int64_t(h) * w;
      const GlobalPtr<T> ginp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + (int64_t(n) * C_total + c0) * plane);
      for (int c = 0; c < C; ++c) {
        const int stride = 1 << wsx[l], chan = wcells[l];
        // ONE cell per lane: the lanes of an atomic instruction are consecutive texels of a row, so a row's adds travel as
        // one request per 64-byte line -- with two cells per lane (one 16-byte LDS read, round 4) every line was asked for
        // twice, by the even and by the odd texels' instruction: 10.2 M of the kernel's 16.4 M atomic requests on the
        // textured benchmark were this flush (TCP_TCC_ATOMIC_WITHOUT_RET_REQ), at what a request costs at the memory side.
        double* win1 = s_win + (l == 0 ? 0 : C * wcells[0]) + c * chan;
        for (int i = tid; i < win_rows[l] * stride; i += kMipBlock) { // rows beyond win_rows were never written
          const double q = win1[i];
          if (q != 0.0) {
            if (rearm) win1[i] = 0.0;
            const int gx = wox[l] + (i & (stride - 1)), gy = woy[l] + (i >> wsx[l]);
            atomic_add_g1(ginp + c * plane + int64_t(gy) * w + gx, static_cast<T>(q));
          }
        }
      }
    }
  };
  // Does any pixel of the tile have taps the windows did not hold?  (16 % of the tiles of the minified benchmark scenes;
  // on the textured benchmark the atlas seam and the limb of the sphere, where a pixel spreads eight anisotropic taps on
  // two levels over thousands of texels.)  `pending` says which.
  // ---- further rounds: the windows are moved onto taps that are still pending and those taps alone are accumulated
  // (texture gradient only: the grid gradient is complete).  Where the windows go: onto everything pending if that fits
  // (A), else onto ONE tap of the tile's pixels (B) -- see the placement below.  What is still pending after the last
  // round goes to global memory corner by corner, as all misses did before round 3.
  // A round that catches fewer than DRTK_MIP_LEAN_HOPELESS_PAIRS pairs makes the next one the tile's last: its taps are
  // scattered beyond what windows hold (random uv), and a round costs what some tens of pairs' atomics do.  (With the
  // windows in the corner of everything pending -- the first half of round 5 -- this cut-off cost the minified scenes
  // 3-10 %: their rounds' windows then often held little although the tile had plenty to give; by tap they do not.)
  // What a round costs (profiles/mipmap_bench.py --tile-times / --tile-phases, 10 ns clock per tile): a tile of the textured
  // benchmark without further rounds lives 11.6 us (inputs 2.4, tap setup + placement 2.1, taps 5.2, flush 1.9) with three
  // others on its CU; every further round adds 8-12 us -- ~1500 instructions per wave at the ~16 cycles per instruction a
  // wave gets when four share a SIMD.
  // the lean form of a tap's cell and axis weights on a level of (h, w) texels -- the operations of the first pass's straight
  // line; returns whether the tap is interior (all four corners inside the level: nearly every pending tap; the others go
  // through bilinear_quad below)
  auto lean_tap = [&](T x, T y, int w, int h, int& ix_nw, int& iy_nw, T& wx0, T& wx1, T& wy0, T& wy1) -> bool {
    const T wm1 = static_cast<T>(w - 1), hm1 = static_cast<T>(h - 1);
    T ixu, iyu;
    if (align_corners) {
      ixu = ((x + 1.f) / 2) * wm1, iyu = ((y + 1.f) / 2) * hm1;
    } else {
      ixu = ((x + 1.f) * static_cast<T>(w) - 1) / 2, iyu = ((y + 1.f) * static_cast<T>(h) - 1) / 2;
    }
    T ix, iy;
    bool sane = true;
    if (padding == 1) {
      ix = tmin(tmax(ixu, T(0)), wm1), iy = tmin(tmax(iyu, T(0)), hm1);
    } else if (padding == 2) {
      sane = (tabs(ixu) < T(1e9f)) & (tabs(iyu) < T(1e9f));
      T unused_x = T(1), unused_y = T(1);
      reflect_clip_lean<T>(sane ? ixu : T(0), sane ? iyu : T(0), w, h, align_corners, ix, iy, unused_x, unused_y);
    } else {
      ix = tmin(tmax(ixu, T(-4)), T(2e9f)), iy = tmin(tmax(iyu, T(-4)), T(2e9f));
    }
    const T fx_floor = floor(ix), fy_floor = floor(iy);
    ix_nw = static_cast<int>(fx_floor), iy_nw = static_cast<int>(fy_floor);
    wx1 = (fx_floor + T(1)) - ix, wx0 = ix - fx_floor, wy1 = (fy_floor + T(1)) - iy, wy0 = iy - fy_floor;
    return sane & ((x == x) & (y == y)) & (static_cast<unsigned>(ix_nw) < static_cast<unsigned>(w - 1)) & (static_cast<unsigned>(iy_nw) < static_cast<unsigned>(h - 1));
  };
  // how many taps away from one of its taps a pixel can have another one inside the same window (<= 128 cells wide and
  // high): the taps are equally spaced, 2 / (n + 1) of (du, dv) apart -- on the coarser of the pixel's levels half as many
  // texels as on the finer.  A filter for the rounds below, not a decision: a tap it wrongly leaves out stays pending.
  // (evaluated where a round needs it: a register less across the tap loop)
  auto tap_span = [&]() -> int {
    const T step = T(2) / static_cast<T>(t.n + 1);
    const T sp = tmax(tabs(static_cast<T>(t.du)) * static_cast<T>(s_w[t.d1]), tabs(static_cast<T>(t.dv)) * static_cast<T>(s_h[t.d1])) * step * T(0.5);
    return static_cast<int>(tmin(T(16), T(258) / tmax(sp, T(1)))) + 1;
  };
  int ref_now = ref, npend_before = 0;
  for (int round = 1;; ++round) {
    {
      const int wb = wave_min_i32(pending ? __builtin_ctz(pending) : 32);
      if ((tid & (kWave - 1)) == 0 && wb < 32) atomicMin(&s_bit[round & 1], wb);
    }
    const bool again = __syncthreads_or(pending != 0) && !DRTK_DBG(dbg, 8);
    flush(ref_now, again);
    if (!again) {
      DRTK_MIP_TILE_DONE(round - 1);
      return;
    }
    if (tid == 0) DRTK_MIP_STAT(round < 7 ? round : 7, 1);
    bool last = round >= DRTK_MIP_LEAN_ROUNDS - 1; // (or hopeless, below)
    const int i_star = s_bit[round & 1] >> 1; // the tile's first pending tap
    __syncthreads(); // everybody has finished its flush (it reads the origins)
    if (tid == 0) s_ref = s_refb = kMaxLevels, s_npend = 0, s_bit[(round + 1) & 1] = 32;
    if (tid < kWinLevels) {
      s_ox[tid] = s_oy[tid] = s_bx[tid] = s_by[tid] = INT32_MAX, s_hx[tid] = s_hy[tid] = s_bhx[tid] = s_bhy[tid] = INT32_MIN;
      s_seed[tid] = ~0ull;
    }
    // where this pixel's pending taps are: bounding box of their north-west texels per level (A: all of them), and
    // where its tap i_star is, if pending (B)
    bool miss[2] = {false, false}, star[2] = {false, false};
    int miss_x[2] = {INT32_MAX, INT32_MAX}, miss_y[2] = {INT32_MAX, INT32_MAX}, miss_hx[2] = {INT32_MIN, INT32_MIN}, miss_hy[2] = {INT32_MIN, INT32_MIN};
    int star_x[2] = {0, 0}, star_y[2] = {0, 0};
    // (the taps of a pixel are collinear and ordered: the extremes of its pending taps on a level are the first and the last)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const uint32_t m = pending & (0x55555555u << s2);
      if (m == 0) continue;
      const int i_lo = __builtin_ctz(m) >> 1, i_hi = (31 - __builtin_clz(m)) >> 1;
      const bool star_pending = ((m >> (2 * i_star + s2)) & 1u) != 0;
      const int d = t.d1 + s2;
      const int w = s_w[d], h = s_h[d];
      miss[s2] = true;
      for (int e = 0; e < 3; ++e) {
        const int i = e == 0 ? i_lo : e == 1 ? i_hi : i_star;
        if ((e == 1 && i_hi == i_lo) || (e == 2 && (!star_pending || i_star == i_lo || i_star == i_hi))) continue;
        T x, y;
        tap_xy(i, x, y);
        int qx, qy; // (of a tap on the border of its level: its clamped position -- any origin is correct)
        T u0, u1, u2, u3;
        lean_tap(x, y, w, h, qx, qy, u0, u1, u2, u3);
        if (e < 2) {
          miss_x[s2] = min(miss_x[s2], qx), miss_y[s2] = min(miss_y[s2], qy);
          miss_hx[s2] = max(miss_hx[s2], qx), miss_hy[s2] = max(miss_hy[s2], qy);
        }
        if (i == i_star) star[s2] = true, star_x[s2] = qx, star_y[s2] = qy;
      }
    }
    __syncthreads();
    {
      const int d_min = wave_min_i32(miss[0] ? t.d1 : miss[1] ? t.d1 + 1 : kMaxLevels);
      const int d_minb = wave_min_i32(star[0] ? t.d1 : star[1] ? t.d1 + 1 : kMaxLevels);
      // the tile's pending (tap, level) pairs: a wave's count from five ballots over the bits of its lanes' counts (<= 32)
      const int pc = __popc(pending);
      int wave_pc = 0;
#pragma unroll
      for (int b = 0; b < 6; ++b) wave_pc += __popcll(__ballot((pc >> b) & 1)) << b;
      if ((tid & (kWave - 1)) == 0) {
        atomicMin(&s_ref, d_min);
        atomicMin(&s_refb, d_minb);
        atomicAdd(&s_npend, wave_pc);
      }
    }
    __syncthreads();
    const int ref_a = s_ref, ref_b = s_refb;
    {
      // HOPELESS: the round before this one caught fewer than kHopelessPairs pairs -- scattered taps; this round is the last
      const int npend = s_npend;
      const bool hopeless = DRTK_MIP_LEAN_HOPELESS_PAIRS > 0 && round >= 2 && npend_before - npend < DRTK_MIP_LEAN_HOPELESS_PAIRS;
      if (tid == 0 && hopeless) DRTK_MIP_STAT(11, 1);
      npend_before = npend;
      last = last || hopeless;
    }
    {
      int lo_x[kWinLevels], lo_y[kWinLevels], hi_x[kWinLevels], hi_y[kWinLevels];
      int b_x[kWinLevels], b_y[kWinLevels], b_hx[kWinLevels], b_hy[kWinLevels];
      unsigned long long seed[kWinLevels];
#pragma unroll
      for (int l = 0; l < kWinLevels; ++l) lo_x[l] = lo_y[l] = b_x[l] = b_y[l] = INT32_MAX, hi_x[l] = hi_y[l] = b_hx[l] = b_hy[l] = INT32_MIN, seed[l] = ~0ull;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int la = t.d1 + s2 - ref_a, lb = t.d1 + s2 - ref_b;
#pragma unroll
        for (int k = 0; k < kWinLevels; ++k) {
          if (miss[s2] && k == la) {
            lo_x[k] = min(lo_x[k], miss_x[s2]), lo_y[k] = min(lo_y[k], miss_y[s2]);
            hi_x[k] = max(hi_x[k], miss_hx[s2]), hi_y[k] = max(hi_y[k], miss_hy[s2]);
          }
          if (star[s2] && k == lb) {
            b_x[k] = b_hx[k] = star_x[s2], b_y[k] = b_hy[k] = star_y[s2];
            seed[k] = static_cast<unsigned long long>(static_cast<unsigned>(star_y[s2] + 1)) << 32 | static_cast<unsigned>(star_x[s2] + 1);
          }
        }
      }
#pragma unroll
      for (int l = 0; l < kWinLevels; ++l) {
        const int a = wave_min_i32(lo_x[l]), b = wave_min_i32(lo_y[l]);
        const int g = wave_max_i32(hi_x[l]), e = wave_max_i32(hi_y[l]);
        if ((tid & (kWave - 1)) == 0 && a != INT32_MAX) {
          atomicMin(&s_ox[l], a);
          atomicMin(&s_oy[l], b);
          atomicMax(&s_hx[l], g);
          atomicMax(&s_hy[l], e);
        }
        if (__ballot(b_x[l] != INT32_MAX) != 0) { // (wave-uniform)
          const int ba = wave_min_i32(b_x[l]), bb = wave_min_i32(b_y[l]);
          const int bg = wave_max_i32(b_hx[l]), be = wave_max_i32(b_hy[l]);
          if ((tid & (kWave - 1)) == 0) {
            atomicMin(&s_bx[l], ba);
            atomicMin(&s_by[l], bb);
            atomicMax(&s_bhx[l], bg);
            atomicMax(&s_bhy[l], be);
          }
          // the seed: the topmost (then leftmost) of the candidates -- a window that starts on its row and is centred on its
          // column holds at least this tap, whatever the shape of the rest
          if (b_y[l] == bb && b_x[l] != INT32_MAX) atomicMin(&s_seed[l], seed[l]);
        }
      }
    }
    __syncthreads();
    bool fits;
    {
      int lox[kWinLevels], loy[kWinLevels], hix[kWinLevels], hiy[kWinLevels];
#pragma unroll
      for (int l = 0; l < kWinLevels; ++l) {
        lox[l] = __builtin_amdgcn_readfirstlane(s_ox[l]), loy[l] = __builtin_amdgcn_readfirstlane(s_oy[l]);
        hix[l] = __builtin_amdgcn_readfirstlane(s_hx[l]), hiy[l] = __builtin_amdgcn_readfirstlane(s_hy[l]);
      }
      // A: do ALL pending taps of the two finest pending levels fit the slots?  Then the windows take them (the usual case:
      // a tile whose taps overflowed the first windows a little).  Otherwise B: the windows go onto ONE tap of the tile's
      // pixels, the first that is pending anywhere (and its two finest levels).  The pixels of a tile are neighbours, so their
      // taps number i are too, however far the taps of one pixel lie apart -- the limb of the textured benchmark: 600 tiles
      // whose pixels spread eight taps on two levels over thousands of texels; the bounding box of everything pending
      // is then the whole texture, a window in its corner holds nothing, and 0.8 M pairs x 12 scattered atomics (0.2 ms
      // of 1.55) went to global memory after five empty rounds.
      fits = true;
      {
        const bool two = lox[1] != INT32_MAX;
#pragma unroll
        for (int l = 0; l < kWinLevels; ++l) {
          if (lox[l] == INT32_MAX) continue;
          const int cells = two ? kWinSlotCells : kWinLevels * kWinSlotCells;
          const long long need_w = static_cast<long long>(hix[l]) - lox[l] + 2, need_h = static_cast<long long>(hiy[l]) - loy[l] + 2;
          const int sx = need_w <= 16 ? 4 : need_w <= 32 ? 5 : need_w <= 64 ? 6 : 7;
          fits = fits && need_w <= 128 && need_h <= (cells >> sx);
        }
      }
      ref_now = ref_a;
      if (!fits && DRTK_MIP_ROUND_BY_TAP) {
        ref_now = ref_b;
#pragma unroll
        for (int l = 0; l < kWinLevels; ++l) {
          lox[l] = __builtin_amdgcn_readfirstlane(s_bx[l]), loy[l] = __builtin_amdgcn_readfirstlane(s_by[l]);
          hix[l] = __builtin_amdgcn_readfirstlane(s_bhx[l]), hiy[l] = __builtin_amdgcn_readfirstlane(s_bhy[l]);
        }
      }
      // a window whose slot has no pending tap on level ref_now must still exist as slot 0: ref_now IS a level with pending taps
      shape_slots(lox, loy, hix, hiy, false); // (rows: the bounding box's -- slot_cell() admits no others -- so the flush walks no more)
      if (!fits && DRTK_MIP_ROUND_BY_TAP) {
#pragma unroll
        for (int l = 0; l < kWinLevels; ++l) {
          if (lox[l] == INT32_MAX) continue;
          const int width = 1 << wsx[l];
          if (static_cast<long long>(hix[l]) - lox[l] + 2 > width) { // wider than the slot: centred on the seed's column
            const unsigned long long sd = s_seed[l];
            const int seed_x = __builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(sd))) - 1;
            wox[l] = max(lox[l], seed_x - (width / 2 - 1));
          }
        }
      }
    }
    if (pending != 0) {
      uint32_t todo = pending;
      if (!fits && DRTK_MIP_ROUND_BY_TAP && !last) {
        // windows placed by tap i_star: a pixel whose own tap i_star is inside them walks its taps within k_span of it only
        bool star_in = false;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) star_in = star_in || (star[s2] && slot_cell(t.d1 + s2 - ref_now, star_x[s2], star_y[s2]) >= 0);
        if (star_in) {
          const int k_span = tap_span();
          const int b_lo = 2 * max(i_star - k_span, 0), b_hi = 2 * min(i_star + k_span, 15) + 1;
          todo &= (0xFFFFFFFFu << b_lo) & (0xFFFFFFFFu >> (31 - b_hi));
        }
      }
      while (todo) {
        const int bit = __builtin_ctz(todo);
        todo &= todo - 1;
        const int i = bit >> 1, s2 = bit & 1;
        T x, y;
        tap_xy(i, x, y);
        const int d = t.d1 + s2;
        const int h = s_h[d], w = s_w[d];
        const int64_t plane = int64_t(h) * w;
        const T alpha = s2 == 0 ? alpha_2 : alpha_1;
        const int l = d - ref_now;
        {
          int ix_nw, iy_nw;
          T wx0, wx1, wy0, wy1;
          if (lean_tap(x, y, w, h, ix_nw, iy_nw, wx0, wx1, wy0, wy1)) {
            const int cell = slot_cell(l, ix_nw, iy_nw);
            if (cell < 0 && !last) continue; // stays pending: the next round's windows
            pending &= ~(1u << bit);
            if (cell < 0) DRTK_MIP_STAT(10, 1);
            if (cell < 0) DRTK_MIP_STAT(12 + min(max(d - ref, 0), 3), 1);
            if (cell < 0) DRTK_MIP_DUMP(static_cast<unsigned>(n) << 24 | static_cast<unsigned>(tile), static_cast<unsigned>(tid) << 16 | static_cast<unsigned>(d), static_cast<unsigned>(ix_nw), static_cast<unsigned>(iy_nw));
            if (cell < 0 && DRTK_DBG(dbg, 256)) continue; // (ablation: what the left-over global atomics cost)
            const T q_nw = wx1 * wy1, q_ne = wx0 * wy1, q_sw = wx1 * wy0, q_se = wx0 * wy0;
            if (cell >= 0) {
              double* wp = s_win + slot_base(l) + cell;
              const int stride = slot_stride(l), chan = slot_chan(l);
#pragma unroll
              for (int c = 0; c < CN; ++c) {
                const T gc = go[c] * alpha;
                lds_add(wp + c * chan, static_cast<double>(q_nw * gc));
                lds_add(wp + c * chan + 1, static_cast<double>(q_ne * gc));
                lds_add(wp + c * chan + stride, static_cast<double>(q_sw * gc));
                lds_add(wp + c * chan + stride + 1, static_cast<double>(q_se * gc));
              }
            } else {
              const GlobalPtr<T> gp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + (int64_t(n) * C_total + c0) * plane) + (int64_t(iy_nw) * w + ix_nw);
#pragma unroll
              for (int c = 0; c < CN; ++c) {
                const T gc = go[c] * alpha;
                if (gc == T(0)) continue;
                atomic_add_g1(gp + c * plane, q_nw * gc);
                atomic_add_g1(gp + c * plane + 1, q_ne * gc);
                atomic_add_g1(gp + c * plane + w, q_sw * gc);
                atomic_add_g1(gp + c * plane + w + 1, q_se * gc);
              }
            }
            continue;
          }
        }
        const Quad<T> q = bilinear_quad<T>(x, y, h, w, padding, align_corners); // (a deferred tap on the border of its level: rare)
        const int cell = slot_cell(l, q.ix_nw, q.iy_nw);
        const int stride = slot_stride(l), chan = slot_chan(l);
        if (cell < 0 && !last) continue; // stays pending: the next round's windows
        pending &= ~(1u << bit);
        if (cell < 0) DRTK_MIP_STAT(10, 1);
        if (cell < 0) DRTK_MIP_STAT(12 + min(max(d - ref, 0), 3), 1);
        if (cell < 0) DRTK_MIP_DUMP(static_cast<unsigned>(n) << 24 | static_cast<unsigned>(tile), static_cast<unsigned>(tid) << 16 | static_cast<unsigned>(d), static_cast<unsigned>(q.ix_nw), static_cast<unsigned>(q.iy_nw));
        if (cell < 0 && DRTK_DBG(dbg, 256)) continue; // (ablation: what the left-over global atomics cost)
        const GlobalPtr<T> ginp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + (int64_t(n) * C_total + c0) * plane);
#pragma unroll
        for (int c = 0; c < CN; ++c) {
          const T gc = go[c] * alpha;
          if (gc == T(0)) continue;
          if (cell >= 0) {
            double* wp = s_win + slot_base(l) + c * chan + cell;
            if (q.o_nw >= 0) lds_add(wp, static_cast<double>(q.nw * gc));
            if (q.o_ne >= 0) lds_add(wp + 1, static_cast<double>(q.ne * gc));
            if (q.o_sw >= 0) lds_add(wp + stride, static_cast<double>(q.sw * gc));
            if (q.o_se >= 0) lds_add(wp + stride + 1, static_cast<double>(q.se * gc));
          } else {
            const GlobalPtr<T> gp = ginp + c * plane;
            if (q.o_nw >= 0) atomic_add_g1(gp + q.o_nw, q.nw * gc);
            if (q.o_ne >= 0) atomic_add_g1(gp + q.o_ne, q.ne * gc);
            if (q.o_sw >= 0) atomic_add_g1(gp + q.o_sw, q.sw * gc);
            if (q.o_se >= 0) atomic_add_g1(gp + q.o_se, q.se * gc);
          }
        }
      }
    }
  }
}

// ---- Backward, bilinear, f32, any C: WAVE-PRIVATE windows (round 4) -------------------------------------------------
// Round 4's tile kernel was bound by its chain of dependent round trips at 3 waves per SIMD: 48 KB of
// workgroup-shared windows and 157 VGPRs hold the occupancy there, and every phase of a tile (window placement, taps,
// flush, each further round) is fenced by workgroup barriers -- 0.85 of its 1.95 ms remain with no accumulation at all
// (profiles/r03/mipmap_pmc_sq.txt).  Here a WAVE owns its 16 x 4 pixels and a window of its own:
//   * no workgroup barrier after the level table is staged: placement, taps, flush and the further rounds of a wave are
//     ordered by the wave's own LDS counter, so 20+ independent waves per CU overlap each other's round trips;
//   * kWaveCells cells per channel (double accumulators, ds_add_f64) shaped like the bounding box of the wave's taps --
//     2 KB per channel and wave instead of 16 KB per channel and workgroup; channels go in blocks of four, so any C runs
//     here (the level pair, tap geometry and grid gradient of a pixel are per block; the reference's sums regrouped);
//   * per-level sizes and pointers are re-read from the LDS table inside the tap loop instead of being carried in
//     registers; taps that touch the border of a level (not all four corners inside) go straight to global memory.
// Taps that find no cell stay pending and get the window in a further round, exactly as in the tiled kernel.
#ifndef DRTK_MIP_WAVE_CELLS
#define DRTK_MIP_WAVE_CELLS 256
#endif
constexpr int kWaveCells = DRTK_MIP_WAVE_CELLS; // window cells per channel and wave, both levels together
constexpr int kWaveWinW = 32;                   // widest window row (the flush walks two rows of <= 32 cells per step)
#ifndef DRTK_MIP_WAVE_OCC
#define DRTK_MIP_WAVE_OCC 3
#endif

__device__ __forceinline__ void wave_minmax2(int x, int y, bool on, int& x0, int& y0, int& x1, int& y1) {
  x0 = wave_min_i32(on ? x : INT32_MAX), y0 = wave_min_i32(on ? y : INT32_MAX);
  x1 = wave_max_i32(on ? x : INT32_MIN), y1 = wave_max_i32(on ? y : INT32_MIN);
}

#ifndef DRTK_BICUBIC_COMPACT_SLOW
#define DRTK_BICUBIC_COMPACT_SLOW 1
#endif
#ifndef DRTK_BICUBIC_ROWS
#define DRTK_BICUBIC_ROWS 1
#endif
#ifndef DRTK_MIP_BICUBIC_OCC
#define DRTK_MIP_BICUBIC_OCC 3 // (round 6: the compact border path fits 168 registers; 2 until then)
#endif
// MODE 0: bilinear (2 x 2 texels per tap); MODE 2: bicubic (4 x 4: mipmap_grid_sampler_kernel.cu:806-861) -- the same windows with
// a span of four cells, interior taps (sixteen consecutive texels inside the level) windowed, the others corner by corner.
template <typename T, int PAD, bool ALIGN, int MODE = 0>
__global__ __launch_bounds__(kMipBlock, sizeof(T) == 8 ? 1 : (MODE == 2 ? DRTK_MIP_BICUBIC_OCC : DRTK_MIP_WAVE_OCC)) void mipmap_backward_wave_kernel(
    LevelTable lv, int mipmaps, const T* __restrict__ grad_out, const T* __restrict__ grid, GridLayout gl,
    const T* __restrict__ vt, int H, int W, int C, int tiles_x, int max_aniso,
    bool force_max_aniso, bool clip_grad, T* __restrict__ grad_grid, GridLayout ggl, int strip, int dbg) {
  constexpr int padding = PAD;
  constexpr bool align_corners = ALIGN;
  constexpr int kSpan = MODE == 2 ? 4 : 2; // cells per tap and axis
  static_assert(kTileW == 16 && kWave == 64, "a wave covers 16 x 4 pixels");
  __shared__ double s_f[kTapTab * kTapTab];
  __shared__ const void* s_ptr[kMaxLevels];
  __shared__ void* s_grad[kMaxLevels];
  __shared__ int s_h[kMaxLevels], s_w[kMaxLevels];
  __shared__ long long s_sn[kMaxLevels];
  extern __shared__ __attribute__((aligned(16))) unsigned char s_win_raw[]; // [wave][channel of the block][kWaveCells] doubles
  const int tid = threadIdx.x;
  const int wave = tid / kWave, lane = tid & (kWave - 1);
  const int CB = C < kChBlock ? C : kChBlock; // channels per block
  double* const win = reinterpret_cast<double*>(s_win_raw) + wave * CB * kWaveCells;
  stage_tap_table(s_f);
  stage_levels(lv, mipmaps, s_ptr, s_grad, s_h, s_w, s_sn); // the kernel's only workgroup barrier
  for (int i = lane; i < CB * kWaveCells; i += kWave) win[i] = 0.0; // every flush leaves the window zeroed again

  const int n = blockIdx.y;
  const int tile = tile_index(strip);
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int px = tx * kTileW + (lane & (kTileW - 1)), py = ty * kTileH + wave * (kWave / kTileW) + lane / kTileW;
  const bool valid = px < W && py < H;
  const int64_t HW = int64_t(H) * W;
  const int64_t pix = int64_t(py) * W + px;
  const int64_t index = int64_t(n) * HW + pix;
  PixelUV<T> uv = {};
  if (valid) uv = load_pixel_uv<T>(grid, gl, vt, n, pix, index);
  const int n_lv = mipmaps > 1 ? 2 : 1;
  Taps<T> t = {};
  bool have_taps = false;
  T acc_x = T(0), acc_y = T(0);
  auto tap_xy = [&](int i, T& x, T& y) {
    const double f = tap_f(s_f, i, t.n);
    x = t.u + static_cast<T>(t.du * f);
    y = t.v + static_cast<T>(t.dv * f);
  };
  auto texel_floor = [&](T coord, int size) -> int { // (as in the tiled kernel: any origin is correct)
    T unused;
    const T c = unnormalize(coord, size, align_corners, &unused);
    const T lo = padding == 0 ? T(-1) : T(0);
    return static_cast<int>(floor(fmin(fmax(c, lo), static_cast<T>(size - 1)))) - (MODE == 2 ? 1 : 0);
  };

  for (int c0 = 0; c0 < C; c0 += kChBlock) {
    const int cc = min(kChBlock, C - c0);
    T go[kChBlock] = {T(0), T(0), T(0), T(0)};
    if (valid) {
      const T* gout_px = grad_out + (int64_t(n) * C + c0) * HW + pix;
#pragma unroll
      for (int c = 0; c < kChBlock; ++c) {
        if (c < cc) go[c] = gout_px[int64_t(c) * HW];
      }
    }
    const bool has_go = go[0] != T(0) || go[1] != T(0) || go[2] != T(0) || go[3] != T(0);
    if (__ballot(has_go) == 0) continue; // masked background: nothing to add anywhere, grid gradient +0
    if (has_go && !have_taps) {
      t = setup_taps<T>(uv, s_h[0], s_w[0], mipmaps, max_aniso, force_max_aniso, clip_grad);
      have_taps = true;
    }
    const T alpha_1 = has_go ? t.a / t.n : T(0);
    const T alpha_2 = has_go ? static_cast<T>((1.0 - t.a) / t.n) : T(0);
    bool live[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const T alpha = s == 0 ? alpha_2 : alpha_1;
      live[s] = false;
#pragma unroll
      for (int c = 0; c < kChBlock; ++c) live[s] = live[s] || (go[c] * alpha != T(0));
      live[s] = live[s] && s < n_lv;
    }
    // ---- window placement: the finest live level of the wave and the next one, each window the bounding box of the
    // north-west texels of the wave's first and last taps (the taps of a pixel are collinear), clipped to its share of
    // the cells.  Wave-uniform values, no LDS.
    int ref, ox[2], oy[2], ww[2], wh[2], base[2];
    auto place = [&](const bool (&on)[2], const int (&lo_x)[2], const int (&lo_y)[2], const int (&hi_x)[2], const int (&hi_y)[2]) {
      // lane values: per level s of the pixel, the extreme north-west texels it needs (on[s]); d = t.d1 + s
      ref = wave_min_i32(on[0] ? t.d1 : on[1] ? t.d1 + 1 : kMaxLevels);
      int x0[2], y0[2], x1[2], y1[2];
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const bool m0 = on[0] && t.d1 - ref == l, m1 = on[1] && t.d1 + 1 - ref == l;
        const int lx = min(m0 ? lo_x[0] : INT32_MAX, m1 ? lo_x[1] : INT32_MAX), ly = min(m0 ? lo_y[0] : INT32_MAX, m1 ? lo_y[1] : INT32_MAX);
        const int hx = max(m0 ? hi_x[0] : INT32_MIN, m1 ? hi_x[1] : INT32_MIN), hy = max(m0 ? hi_y[0] : INT32_MIN, m1 ? hi_y[1] : INT32_MIN);
        x0[l] = wave_min_i32(lx), y0[l] = wave_min_i32(ly), x1[l] = wave_max_i32(hx), y1[l] = wave_max_i32(hy);
      }
      const bool both = x0[1] != INT32_MAX;
      const int budget[2] = {both ? kWaveCells - kWaveCells / 4 : kWaveCells, kWaveCells / 4};
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        ox[l] = x0[l], oy[l] = y0[l];
        base[l] = l == 0 ? 0 : budget[0];
        if (x0[l] == INT32_MAX) {
          ww[l] = wh[l] = 0;
          continue;
        }
        // + kSpan: the east / south cells of the extreme taps; at most a flush row
        const long long need_w = static_cast<long long>(x1[l]) - x0[l] + kSpan, need_h = static_cast<long long>(y1[l]) - y0[l] + kSpan;
        ww[l] = static_cast<int>(need_w < kWaveWinW ? need_w : kWaveWinW);
        const int rows = budget[l] / ww[l];
        wh[l] = static_cast<int>(need_h < rows ? need_h : rows);
      }
    };
    {
      bool on[2] = {live[0], live[1]};
      int lo_x[2] = {INT32_MAX, INT32_MAX}, lo_y[2] = {INT32_MAX, INT32_MAX}, hi_x[2] = {INT32_MIN, INT32_MIN}, hi_y[2] = {INT32_MIN, INT32_MIN};
      if (live[0] || live[1]) {
        for (int e = 0; e < 2; ++e) {
          T x, y;
          tap_xy(e == 0 ? 0 : t.n - 1, x, y);
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (!live[s]) continue;
            const int fx = texel_floor(x, s_w[t.d1 + s]), fy = texel_floor(y, s_h[t.d1 + s]);
            lo_x[s] = min(lo_x[s], fx), lo_y[s] = min(lo_y[s], fy), hi_x[s] = max(hi_x[s], fx), hi_y[s] = max(hi_y[s], fy);
          }
        }
      }
      place(on, lo_x, lo_y, hi_x, hi_y);
    }
    // window cell of a tap's north-west corner on level d (the other three are +1 in x / y), or -1
    auto cell_of = [&](int d, int ix_nw, int iy_nw, int& stride) -> int {
      const int l = d - ref;
      stride = 0;
      if (l < 0 || l > 1) return -1;
      const int wx = ix_nw - ox[l], wy = iy_nw - oy[l];
      stride = ww[l];
      return (wx >= 0 && wx < ww[l] - (kSpan - 1) && wy >= 0 && wy < wh[l] - (kSpan - 1)) ? base[l] + wy * ww[l] + wx : -1;
    };
    // the wave's windows to global memory; every cell is left zero.  Two rows of <= 32 cells per step: lanes of a half
    // wave = consecutive texels of a row, so the atomics of a row form one or two requests.
    auto flush = [&]() {
      wave_lds_sync();
      if (DRTK_DBG(dbg, 4)) return;
#pragma unroll 1
      for (int l = 0; l < 2; ++l) {
        if (ww[l] == 0) continue;
        const int d = ref + l;
        const int w_tex = s_w[d];
        const int64_t plane = int64_t(s_h[d]) * w_tex;
        const GlobalPtr<T> ginp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + (int64_t(n) * C + c0) * plane);
        const int col = lane & (kWaveWinW - 1), half = lane / kWaveWinW;
        for (int r = half; r < wh[l]; r += kWave / kWaveWinW) {
          if (col < ww[l]) {
            double* wp = win + base[l] + r * ww[l] + col;
            const int64_t o = int64_t(oy[l] + r) * w_tex + ox[l] + col;
            for (int c = 0; c < cc; ++c) {
              const double q = wp[c * kWaveCells];
              if (q != 0.0) {
                wp[c * kWaveCells] = 0.0;
                atomic_add_g1(ginp + c * plane + o, static_cast<T>(q));
              }
            }
          }
        }
      }
      wave_lds_sync();
    };

    bool miss[2] = {false, false};
    int miss_x0[2] = {INT32_MAX, INT32_MAX}, miss_y0[2] = {INT32_MAX, INT32_MAX}, miss_x1[2] = {INT32_MIN, INT32_MIN}, miss_y1[2] = {INT32_MIN, INT32_MIN};
    uint32_t pending = 0; // bit 2 i + s: tap i on the pixel's level s found no window cell yet (taps >= 16 are never deferred)
    auto note_miss = [&](int s, int ix_nw, int iy_nw) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (k == s) {
          miss[k] = true;
          miss_x0[k] = min(miss_x0[k], ix_nw), miss_y0[k] = min(miss_y0[k], iy_nw);
          miss_x1[k] = max(miss_x1[k], ix_nw), miss_y1[k] = max(miss_y1[k], iy_nw);
        }
      }
    };
    wave_lds_sync(); // (the zero-fill before the first block)
    if (live[0] || live[1]) {
      for (int i = 0; i < t.n; ++i) {
        T x, y;
        tap_xy(i, x, y);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (!live[s]) continue;
          const int d = t.d1 + s;
          const int h = s_h[d], w = s_w[d];
          const int plane = h * w; // < 2^31, checked by fill_table()
          const GlobalPtr<const T> inp = (GlobalPtr<const T>)(static_cast<const T*>(s_ptr[d]) + int64_t(n) * s_sn[d]) + int64_t(c0) * plane;
          const T alpha = s == 0 ? alpha_2 : alpha_1;
          if constexpr (MODE == 2) {
            const Cubic<T> cb = bicubic_footprint<T>(x, y, h, w, padding, align_corners);
            T xc[4], yc[4], xg[4], yg[4];
            cubic_coeffs(xc, cb.tx);
            cubic_coeffs(yc, cb.ty);
            cubic_coeffs_grad(xg, cb.tx);
            cubic_coeffs_grad(yg, cb.ty);
            const GlobalPtr<T> gp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + (int64_t(n) * C + c0) * plane);
            T gix = T(0), giy = T(0);
            const int bx = cb.xi[0], by = cb.yi[0];
            const bool interior = bx >= 0 && by >= 0 && cb.xi[1] == bx + 1 && cb.xi[2] == bx + 2 && cb.xi[3] == bx + 3 &&
                cb.yi[1] == by + 1 && cb.yi[2] == by + 2 && cb.yi[3] == by + 3;
            int stride = 0;
            const int cell = interior ? cell_of(d, bx, by, stride) : -1;
            const bool defer = interior && cell < 0 && i < 16 && !DRTK_DBG(dbg, 1);
            if (defer) {
              pending |= 1u << (2 * i + s);
              note_miss(s, bx, by);
            }
#pragma unroll 1
            for (int c = 0; c < cc; ++c) {
              const T gOut = go[c] * alpha;
              if (gOut == T(0)) continue; // every term below would be +-0 * finite
              const GlobalPtr<const T> pc = inp + c * plane;
              if (DRTK_BICUBIC_ROWS && sizeof(T) == 4 && interior && !DRTK_DBG(dbg, 2)) { // (double: 12 registers spilled with the rows in flight)
                // Round 6: the sixteen texels of an interior footprint as FOUR 16-byte row loads (element-aligned), like the
                // forward -- they were sixteen predicated 4-byte loads, each in its own exec-mask region with its own wait
                // (6.87 ms for RGB on the textured benchmark's inputs against 1.29 bilinear).  Same products, same order.
                typedef T Quad4 __attribute__((ext_vector_type(4), aligned(sizeof(T))));
                Quad4 row[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) row[r] = *(GlobalPtr<const Quad4>)(pc + (by + r) * w + bx);
                // ... and ROW BY ROW: the footprint's weights are an outer product xc (x) yc, so a row's four cells share
                // gOut yc[j] and the grid gradient is  -gOut sum_j yc[j] (row_j . xg)  /  -gOut sum_j yg[j] (row_j . xc): 8
                // multiply-adds per row instead of six operations per cell with their 48 hoisted coefficient products
                // (the registers that held the kernel at two waves per SIMD)
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                  const T gy = gOut * yc[j2];
                  if (DRTK_DBG(dbg, 1)) {
                  } else if (cell >= 0) {
                    double* wp = win + cell + c * kWaveCells + j2 * stride;
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) lds_add(wp + i2, static_cast<double>(gy * xc[i2]));
                  } else if (!defer) {
                    const GlobalPtr<T> gq = gp + c * plane + ((by + j2) * w + bx);
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) atomic_add_g1(gq + i2, gy * xc[i2]);
                  }
                  const T sx = row[j2].x * xg[0] + row[j2].y * xg[1] + row[j2].z * xg[2] + row[j2].w * xg[3];
                  const T sc = row[j2].x * xc[0] + row[j2].y * xc[1] + row[j2].z * xc[2] + row[j2].w * xc[3];
                  gix -= gy * sx;
                  giy -= (gOut * yg[j2]) * sc;
                }
                continue;
              }
              if constexpr (DRTK_BICUBIC_COMPACT_SLOW && sizeof(T) == 4) {
                // a footprint that touches the border of its level (rare): the sixteen cells in a loop that is NOT unrolled,
                // coefficients and indices picked by selects -- the unrolled form below (sixteen predicated loads and their
                // products in flight) is what kept this kernel at 225-243 registers = two waves per SIMD
                auto sel4 = [](const auto (&a)[4], int k) { return k == 0 ? a[0] : k == 1 ? a[1] : k == 2 ? a[2] : a[3]; };
#pragma unroll 1
                for (int e = 0; e < 16; ++e) {
                  const int i2 = e >> 2, j2 = e & 3;
                  const int xi = sel4(cb.xi, i2), yi = sel4(cb.yi, j2);
                  const bool ok = xi >= 0 && yi >= 0;
                  const int o = ok ? yi * w + xi : 0;
                  const T cxw = sel4(xc, i2), cyw = sel4(yc, j2);
                  const T wgt = gOut * cxw * cyw;
                  if (DRTK_DBG(dbg, 1)) {
                  } else if (cell >= 0) { // (an interior footprint whose rows were not taken above: the ablation build only)
                    lds_add(win + cell + c * kWaveCells + j2 * stride + i2, static_cast<double>(wgt));
                  } else if (ok && !defer) {
                    atomic_add_g1(gp + c * plane + o, wgt);
                  }
                  const T val = (ok && !DRTK_DBG(dbg, 2)) ? pc[o] : T(0);
                  gix -= gOut * val * (sel4(xg, i2) * cyw);
                  giy -= gOut * val * (sel4(yg, j2) * cxw);
                }
                continue;
              }
#pragma unroll
              for (int i2 = 0; i2 < 4; ++i2) {
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                  const bool ok = cb.xi[i2] >= 0 && cb.yi[j2] >= 0;
                  const int o = ok ? cb.yi[j2] * w + cb.xi[i2] : 0;
                  const T wgt = gOut * xc[i2] * yc[j2];
                  if (DRTK_DBG(dbg, 1)) {
                  } else if (cell >= 0) {
                    lds_add(win + cell + c * kWaveCells + j2 * stride + i2, static_cast<double>(wgt));
                  } else if (ok && !defer) {
                    atomic_add_g1(gp + c * plane + o, wgt);
                  }
                  const T val = (ok && !DRTK_DBG(dbg, 2)) ? pc[o] : T(0);
                  gix -= gOut * val * (xg[i2] * yc[j2]);
                  giy -= gOut * val * (yg[j2] * xc[i2]);
                }
              }
            }
            acc_x += cb.mx * gix;
            acc_y += cb.my * giy;
            continue;
          }
          const Quad<T> q = bilinear_quad<T>(x, y, h, w, padding, align_corners);
          const int ix_se = q.ix_nw + 1, iy_se = q.iy_nw + 1;
          T g[kChBlock];
#pragma unroll
          for (int c = 0; c < kChBlock; ++c) g[c] = c < cc ? go[c] * alpha : T(0);
          T gix = T(0), giy = T(0);
          if ((q.o_nw | q.o_ne | q.o_sw | q.o_se) >= 0) {
            // interior tap (nearly all): 2 cc 8-byte texel loads, 4 cc window adds, the grid-gradient products
            Pair<T> top[kChBlock], bot[kChBlock];
#pragma unroll
            for (int c = 0; c < kChBlock; ++c) {
              top[c] = bot[c] = Pair<T>{T(0), T(0)};
              if (c < cc && !DRTK_DBG(dbg, 2)) {
                top[c] = *(GlobalPtr<const Pair<T>>)(inp + c * plane + q.o_nw);
                bot[c] = *(GlobalPtr<const Pair<T>>)(inp + c * plane + q.o_sw);
              }
            }
            int stride;
            const int cell = cell_of(d, q.ix_nw, q.iy_nw, stride);
            if (DRTK_DBG(dbg, 1)) {
            } else if (cell >= 0) {
              double* wp = win + cell;
#pragma unroll
              for (int c = 0; c < kChBlock; ++c) {
                if (c >= cc) break;
                lds_add(wp + c * kWaveCells, static_cast<double>(q.nw * g[c]));
                lds_add(wp + c * kWaveCells + 1, static_cast<double>(q.ne * g[c]));
                lds_add(wp + c * kWaveCells + stride, static_cast<double>(q.sw * g[c]));
                lds_add(wp + c * kWaveCells + stride + 1, static_cast<double>(q.se * g[c]));
              }
            } else if (i < 16) {
              pending |= 1u << (2 * i + s);
              note_miss(s, q.ix_nw, q.iy_nw);
            } else {
              const GlobalPtr<T> gp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + (int64_t(n) * C + c0) * plane);
#pragma unroll
              for (int c = 0; c < kChBlock; ++c) {
                if (c >= cc) break;
                atomic_add_g1(gp + c * plane + q.o_nw, q.nw * g[c]);
                atomic_add_g1(gp + c * plane + q.o_ne, q.ne * g[c]);
                atomic_add_g1(gp + c * plane + q.o_sw, q.sw * g[c]);
                atomic_add_g1(gp + c * plane + q.o_se, q.se * g[c]);
              }
            }
            const T fy1 = iy_se - q.iy, fy0 = q.iy - q.iy_nw, fx1 = ix_se - q.ix, fx0 = q.ix - q.ix_nw;
#pragma unroll
            for (int c = 0; c < kChBlock; ++c) {
              if (c >= cc) break;
              const T gOut = g[c];
              // with a zero upstream gradient every term is +-0 * finite: the texels count as 0
              const T v_nw = gOut != T(0) ? top[c].x : T(0), v_ne = gOut != T(0) ? top[c].y : T(0);
              const T v_sw = gOut != T(0) ? bot[c].x : T(0), v_se = gOut != T(0) ? bot[c].y : T(0);
              gix -= v_nw * fy1 * gOut;
              giy -= v_nw * fx1 * gOut;
              gix += v_ne * fy1 * gOut;
              giy -= v_ne * fx0 * gOut;
              gix -= v_sw * fy0 * gOut;
              giy += v_sw * fx1 * gOut;
              gix += v_se * fy0 * gOut;
              giy += v_se * fx0 * gOut;
            }
          } else {
            // a tap on the border of its level: corner by corner, straight to global memory (the reference's form)
            const GlobalPtr<T> gp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + (int64_t(n) * C + c0) * plane);
#pragma unroll 1
            for (int c = 0; c < cc; ++c) {
              const T gOut = g[c];
              if (gOut == T(0)) continue; // every term below would be +-0 * finite
              const GlobalPtr<const T> p = inp + c * plane;
              T v_nw = T(0), v_ne = T(0), v_sw = T(0), v_se = T(0);
              if (!DRTK_DBG(dbg, 2)) {
                if (q.o_nw >= 0) v_nw = p[q.o_nw];
                if (q.o_ne >= 0) v_ne = p[q.o_ne];
                if (q.o_sw >= 0) v_sw = p[q.o_sw];
                if (q.o_se >= 0) v_se = p[q.o_se];
              }
              if (!DRTK_DBG(dbg, 1)) {
                if (q.o_nw >= 0) atomic_add_g1(gp + c * plane + q.o_nw, q.nw * gOut);
                if (q.o_ne >= 0) atomic_add_g1(gp + c * plane + q.o_ne, q.ne * gOut);
                if (q.o_sw >= 0) atomic_add_g1(gp + c * plane + q.o_sw, q.sw * gOut);
                if (q.o_se >= 0) atomic_add_g1(gp + c * plane + q.o_se, q.se * gOut);
              }
              if (q.o_nw >= 0) {
                gix -= v_nw * (iy_se - q.iy) * gOut;
                giy -= v_nw * (ix_se - q.ix) * gOut;
              }
              if (q.o_ne >= 0) {
                gix += v_ne * (iy_se - q.iy) * gOut;
                giy -= v_ne * (q.ix - q.ix_nw) * gOut;
              }
              if (q.o_sw >= 0) {
                gix -= v_sw * (q.iy - q.iy_nw) * gOut;
                giy += v_sw * (ix_se - q.ix) * gOut;
              }
              if (q.o_se >= 0) {
                gix += v_se * (q.iy - q.iy_nw) * gOut;
                giy += v_se * (q.ix - q.ix_nw) * gOut;
              }
            }
          }
          acc_x += q.mx * gix;
          acc_y += q.my * giy;
        }
      }
    }
    flush();
    // ---- further rounds: the window moves onto the taps that are still pending (texture gradient only: the grid
    // gradient of the block is complete); what is pending after the last round goes to global memory
#ifndef DRTK_MIP_WAVE_ROUNDS
#define DRTK_MIP_WAVE_ROUNDS 6
#endif
    for (int round = 1; __ballot(pending != 0) != 0 && !DRTK_DBG(dbg, 8); ++round) {
      const bool last = round >= DRTK_MIP_WAVE_ROUNDS - 1;
      place(miss, miss_x0, miss_y0, miss_x1, miss_y1);
      miss[0] = miss[1] = false;
#pragma unroll
      for (int k = 0; k < 2; ++k) miss_x0[k] = miss_y0[k] = INT32_MAX, miss_x1[k] = miss_y1[k] = INT32_MIN;
      uint32_t todo = pending;
      while (todo) {
        const int bit = __builtin_ctz(todo);
        todo &= todo - 1;
        const int i = bit >> 1, s2 = bit & 1;
        T x, y;
        tap_xy(i, x, y);
        const int d = t.d1 + s2;
        const int h = s_h[d], w = s_w[d];
        const int plane = h * w;
        const T alpha = s2 == 0 ? alpha_2 : alpha_1;
        if constexpr (MODE == 2) {
          const Cubic<T> cb = bicubic_footprint<T>(x, y, h, w, padding, align_corners); // (interior: only those are deferred)
          T xc[4], yc[4];
          cubic_coeffs(xc, cb.tx);
          cubic_coeffs(yc, cb.ty);
          const int bx = cb.xi[0], by = cb.yi[0];
          int stride;
          const int cell = cell_of(d, bx, by, stride);
          if (cell < 0 && !last) {
            note_miss(s2, bx, by);
            continue;
          }
          pending &= ~(1u << bit);
          const GlobalPtr<T> gp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + (int64_t(n) * C + c0) * plane);
#pragma unroll 1
          for (int c = 0; c < cc; ++c) {
            const T gOut = go[c] * alpha;
            if (gOut == T(0)) continue;
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
#pragma unroll
              for (int j2 = 0; j2 < 4; ++j2) {
                const T wgt = gOut * xc[i2] * yc[j2];
                if (cell >= 0) {
                  lds_add(win + cell + c * kWaveCells + j2 * stride + i2, static_cast<double>(wgt));
                } else {
                  atomic_add_g1(gp + c * plane + (by + j2) * w + bx + i2, wgt);
                }
              }
            }
          }
          continue;
        }
        const Quad<T> q = bilinear_quad<T>(x, y, h, w, padding, align_corners); // (interior: only those are deferred)
        int stride;
        const int cell = cell_of(d, q.ix_nw, q.iy_nw, stride);
        if (cell < 0 && !last) { // stays pending: the next round's window
          note_miss(s2, q.ix_nw, q.iy_nw);
          continue;
        }
        pending &= ~(1u << bit);
        const GlobalPtr<T> gp = (GlobalPtr<T>)(static_cast<T*>(s_grad[d]) + (int64_t(n) * C + c0) * plane);
#pragma unroll
        for (int c = 0; c < kChBlock; ++c) {
          if (c >= cc) break;
          const T gc = go[c] * alpha;
          if (cell >= 0) {
            double* wp = win + cell + c * kWaveCells;
            lds_add(wp, static_cast<double>(q.nw * gc));
            lds_add(wp + 1, static_cast<double>(q.ne * gc));
            lds_add(wp + stride, static_cast<double>(q.sw * gc));
            lds_add(wp + stride + 1, static_cast<double>(q.se * gc));
          } else {
            atomic_add_g1(gp + c * plane + q.o_nw, q.nw * gc);
            atomic_add_g1(gp + c * plane + q.o_ne, q.ne * gc);
            atomic_add_g1(gp + c * plane + q.o_sw, q.sw * gc);
            atomic_add_g1(gp + c * plane + q.o_se, q.se * gc);
          }
        }
      }
      flush();
    }
  }
  if (valid) store_grid_grad<T>(grad_grid, ggl, n, pix, acc_x, acc_y);
}

int fill_table(
    LevelTable& lv, const void* const* levels, void* const* grad_levels, const int64_t* level_h,
    const int64_t* level_w, const int64_t* level_sN, int mipmaps, int64_t N, int64_t C) {
  if (mipmaps < 1 || mipmaps > kMaxLevels || !levels || !level_h || !level_w) return DRTK_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < kMaxLevels; ++i) {
    const int j = i < mipmaps ? i : 0;
    if (level_h[j] <= 0 || level_w[j] <= 0 || level_h[j] * level_w[j] >= (int64_t(1) << 31)) return DRTK_ERR_INVALID_ARGUMENT;
    if (N * C > 0 && !levels[j]) return DRTK_ERR_INVALID_ARGUMENT;
    lv.ptr[i] = levels[j];
    lv.grad[i] = grad_levels ? grad_levels[j] : nullptr;
    lv.h[i] = static_cast<int>(level_h[j]);
    lv.w[i] = static_cast<int>(level_w[j]);
    lv.sn[i] = level_sN ? level_sN[j] : C * level_h[j] * level_w[j];
    if (lv.sn[i] != 0 && lv.sn[i] < C * level_h[j] * level_w[j]) return DRTK_ERR_INVALID_ARGUMENT; // overlapping views
  }
  return DRTK_OK;
}

// The lean kernels address a texel as `channel_in_block * plane + offset` in 32 bits (at most three planes + one): levels
// beyond this size (a 23 170^2 level) take the general kernels, whose plane offsets are 64-bit.
constexpr int64_t kLeanMaxPlane = ((int64_t(1) << 31) - 1) / 4;
bool lean_planes_ok(const LevelTable& lv, int mipmaps) {
  for (int i = 0; i < mipmaps; ++i) {
    if (int64_t(lv.h[i]) * lv.w[i] > kLeanMaxPlane) return false;
  }
  return true;
}

// grid_layout = {sN, sP, sC} in elements (NULL: contiguous [N,H,W,2]); the pair access needs sC = 1, even strides and a
// base aligned to two elements
int make_grid_layout(GridLayout& gl, const int64_t* layout, const void* base, int64_t H, int64_t W, size_t elem) {
  gl.sN = layout ? layout[0] : 2 * H * W;
  gl.sP = layout ? layout[1] : 2;
  gl.sC = layout ? layout[2] : 1;
  if (gl.sN < 0 || gl.sP <= 0 || gl.sC <= 0) return DRTK_ERR_INVALID_ARGUMENT;
  gl.pair = gl.sC == 1 && gl.sP % 2 == 0 && gl.sN % 2 == 0 && reinterpret_cast<uintptr_t>(base) % (2 * elem) == 0;
  return DRTK_OK;
}

} // namespace
} // namespace drtk_amd

using namespace drtk_amd;

extern "C" int drtk_amd_mipmap_grid_sampler_2d(
    drtk_dtype_t dtype, const void* const* levels, const int64_t* level_h, const int64_t* level_w, const int64_t* level_sN, int mipmaps,
    const void* grid, const int64_t* grid_layout, const void* vt_dxdy_img, int64_t N, int64_t C, int64_t H, int64_t W, int max_aniso,
    int padding_mode, int interpolation_mode, int align_corners, int force_max_aniso, int clip_grad, void* out,
    drtk_stream_t stream) {
  (void)align_corners; // ignored by the reference's forward kernel (:423)
  if (N < 0 || C < 0 || H < 0 || W < 0 || C >= (1 << 20) || max_aniso < 1 || padding_mode < 0 || padding_mode > 2 ||
      (interpolation_mode != 0 && interpolation_mode != 2) || (dtype != DRTK_F32 && dtype != DRTK_F64))
    return DRTK_ERR_INVALID_ARGUMENT;
  LevelTable lv;
  const int st = fill_table(lv, levels, nullptr, level_h, level_w, level_sN, mipmaps, N, C);
  if (st != DRTK_OK) return st;
  const int64_t count = N * H * W;
  if (count == 0 || C == 0) return DRTK_OK;
  if (!grid || !vt_dxdy_img || !out) return DRTK_ERR_INVALID_ARGUMENT;
  GridLayout gl;
  if (make_grid_layout(gl, grid_layout, grid, H, W, dtype == DRTK_F32 ? 4 : 8) != DRTK_OK) return DRTK_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid_dim(static_cast<unsigned>(ceil_div(count, kBlock)));
#define LAUNCH_P(T, MODE, PAD)                                                                                 \
  DRTK_LAUNCH(                                                                                                 \
      (mipmap_forward_kernel<T, MODE, PAD>), grid_dim, dim3(kBlock), 0, s, lv, mipmaps, static_cast<const T*>(grid), gl, \
      static_cast<const T*>(vt_dxdy_img), count, (int)C, H * W, max_aniso, force_max_aniso != 0, clip_grad != 0, \
      static_cast<T*>(out), xcd_strip(ceil_div(16 * W, kBlock)))
#define LAUNCH(T, MODE)                                                                    \
  if (padding_mode == 0) LAUNCH_P(T, MODE, 0); else if (padding_mode == 1) LAUNCH_P(T, MODE, 1); else LAUNCH_P(T, MODE, 2)
#ifndef DRTK_MIP_FWD_LEAN
#define DRTK_MIP_FWD_LEAN 1
#endif
#ifndef DRTK_MIP_FWD_LEAN_F64
#define DRTK_MIP_FWD_LEAN_F64 1 // double takes the lean forward too (round 6)
#endif
#define LAUNCH_L(T, PAD, CB)                                                                                      \
  DRTK_LAUNCH(                                                                                                 \
      (mipmap_forward_lean_kernel<T, PAD, CB>), dim3(static_cast<unsigned>(ceil_div(H * W, kBlock)), static_cast<unsigned>(N)), \
      dim3(kBlock), 0, s, lv, mipmaps, static_cast<const T*>(grid), gl, static_cast<const T*>(vt_dxdy_img), (int)C, \
      H * W, max_aniso, force_max_aniso != 0, clip_grad != 0, static_cast<T*>(out), xcd_strip(ceil_div(16 * W, kBlock)))
#define LAUNCH_LC(T, PAD)                                                                  \
  if (C % 4 == 0) LAUNCH_L(T, PAD, 4); else if (C % 3 == 0) LAUNCH_L(T, PAD, 3); else if (C % 2 == 0) LAUNCH_L(T, PAD, 2); else LAUNCH_L(T, PAD, 1)
  if (dtype == DRTK_F32 && interpolation_mode == 0 && N <= kMaxViewsPerLaunch && DRTK_MIP_FWD_LEAN && lean_planes_ok(lv, mipmaps)) {
    if (padding_mode == 0) { LAUNCH_LC(float, 0); } else if (padding_mode == 1) { LAUNCH_LC(float, 1); } else { LAUNCH_LC(float, 2); }
  } else if (dtype == DRTK_F64 && interpolation_mode == 0 && N <= kMaxViewsPerLaunch && DRTK_MIP_FWD_LEAN && DRTK_MIP_FWD_LEAN_F64 && lean_planes_ok(lv, mipmaps)) {
    if (padding_mode == 0) { LAUNCH_LC(double, 0); } else if (padding_mode == 1) { LAUNCH_LC(double, 1); } else { LAUNCH_LC(double, 2); }
  } else if (dtype == DRTK_F32) {
    if (interpolation_mode == 0) { LAUNCH(float, 0); } else { LAUNCH(float, 2); }
  } else {
    if (interpolation_mode == 0) { LAUNCH(double, 0); } else { LAUNCH(double, 2); }
  }
#undef LAUNCH_LC
#undef LAUNCH_L
#undef LAUNCH
#undef LAUNCH_P
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

extern "C" int drtk_amd_mipmap_grid_sampler_2d_backward(
    drtk_dtype_t dtype, const void* grad_out, const void* const* levels, const int64_t* level_h,
    const int64_t* level_w, const int64_t* level_sN, int mipmaps, const void* grid, const int64_t* grid_layout, const void* vt_dxdy_img,
    int64_t N, int64_t C, int64_t H, int64_t W, int max_aniso, int padding_mode, int interpolation_mode, int align_corners,
    int force_max_aniso, int clip_grad, void* const* grad_levels, void* grad_grid, const int64_t* grad_grid_layout,
    drtk_stream_t stream) {
  if (N < 0 || C < 0 || H < 0 || W < 0 || C >= (1 << 20) || max_aniso < 1 || padding_mode < 0 || padding_mode > 2 ||
      (interpolation_mode != 0 && interpolation_mode != 2) || (dtype != DRTK_F32 && dtype != DRTK_F64) || !grad_levels)
    return DRTK_ERR_INVALID_ARGUMENT;
  LevelTable lv;
  const int st = fill_table(lv, levels, grad_levels, level_h, level_w, level_sN, mipmaps, N, C);
  if (st != DRTK_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t es = dtype == DRTK_F32 ? 4 : 8;
  // :1120-1123 zeros_like.  Levels that lie back to back in memory (a caller that carves the gradient pyramid out of one
  // buffer, like the torch shim) are zeroed by one launch: the coarse levels are a few KB each and a launch costs more
  // than their fill.
  for (int i = 0; i < mipmaps;) {
    size_t bytes = es * N * C * level_h[i] * level_w[i];
    if (bytes > 0 && !grad_levels[i]) return DRTK_ERR_INVALID_ARGUMENT;
    int j = i + 1;
    while (j < mipmaps && bytes > 0 && grad_levels[j] == static_cast<unsigned char*>(grad_levels[i]) + bytes &&
           es * N * C * level_h[j] * level_w[j] > 0) {
      bytes += es * N * C * level_h[j] * level_w[j];
      ++j;
    }
    if (bytes > 0 && fill_bytes_async(grad_levels[i], 0, bytes, s) != DRTK_OK) return DRTK_ERR_LAUNCH;
    i = j;
  }
  const int64_t count = N * H * W;
  if (count == 0) return DRTK_OK;
  if (!grid || !vt_dxdy_img || !grad_grid || (C > 0 && !grad_out)) return DRTK_ERR_INVALID_ARGUMENT;
  GridLayout gl, ggl;
  if (make_grid_layout(gl, grid_layout, grid, H, W, es) != DRTK_OK || make_grid_layout(ggl, grad_grid_layout, grad_grid, H, W, es) != DRTK_OK)
    return DRTK_ERR_INVALID_ARGUMENT;
  // The windowed kernels (LDS accumulators in double whatever the element type) serve float and double alike: the
  // reference dispatches both (kernel_utils.h:35-57).  Returns 1 when the shape is not theirs.
  auto windowed = [&](auto tag) -> int {
    using T = decltype(tag);
#ifndef DRTK_MIP_BWD_LEAN
#define DRTK_MIP_BWD_LEAN 1
#endif
#ifndef DRTK_MIP_BWD_LEAN_F64
#define DRTK_MIP_BWD_LEAN_F64 1 // double takes the lean tile kernel too (round 6; before: tiled2 at 240-256 registers)
#endif
#ifndef DRTK_MIP_BWD_LEAN_WIDE
#define DRTK_MIP_BWD_LEAN_WIDE 1 // C > 4: the lean tile kernel once per block of four channels instead of the wave-private kernel
#endif
    if constexpr ((sizeof(T) == 4 || DRTK_MIP_BWD_LEAN_F64) && DRTK_MIP_BWD_LEAN) {
      if (interpolation_mode == 0 && C >= 1 && N <= 65535 && !DRTK_DBG(debug_flags(), 512) &&
          (C <= 4 || DRTK_MIP_BWD_LEAN_WIDE) && lean_planes_ok(lv, mipmaps)) { // bilinear, any padding, float and double: the lean tap loop, four channels a launch
        const int tiles_x = static_cast<int>(ceil_div(W, kTileW)), tiles_y = static_cast<int>(ceil_div(H, kTileH));
#define LEANK(PAD, ALIGN, CN)                                                                                           \
  DRTK_LAUNCH(                                                                                                          \
      (mipmap_backward_lean_kernel<T, PAD, ALIGN, CN>), dim3(static_cast<unsigned>(tiles_x * tiles_y), static_cast<unsigned>(N)), \
      dim3(kMipBlock), sizeof(double) * CN * kWinLevels * lean_slot_cells<CN>(), s, lv, mipmaps, static_cast<const T*>(grad_out), \
      static_cast<const T*>(grid), gl, static_cast<const T*>(vt_dxdy_img), (int)H, (int)W, tiles_x, max_aniso, \
      force_max_aniso != 0, clip_grad != 0, static_cast<T*>(grad_grid), ggl, xcd_strip(tiles_x), debug_flags(), (int)C, c0)
#define LEANC(PAD, ALIGN)                                                                       \
  switch (cn) {                                                                                 \
    case 1: LEANK(PAD, ALIGN, 1); break;                                                        \
    case 2: LEANK(PAD, ALIGN, 2); break;                                                        \
    case 3: LEANK(PAD, ALIGN, 3); break;                                                        \
    default: LEANK(PAD, ALIGN, 4); break;                                                       \
  }
        for (int c0 = 0; c0 < C; c0 += 4) {
          const int cn = static_cast<int>(C - c0 < 4 ? C - c0 : 4);
          if (align_corners) {
            if (padding_mode == 0) { LEANC(0, true) } else if (padding_mode == 1) { LEANC(1, true) } else { LEANC(2, true) }
          } else {
            if (padding_mode == 0) { LEANC(0, false) } else if (padding_mode == 1) { LEANC(1, false) } else { LEANC(2, false) }
          }
          DRTK_RETURN_IF_LAUNCH_FAILED();
        }
#undef LEANC
#undef LEANK
        return DRTK_OK;
      }
    }
  // bilinear: C <= 4 takes the workgroup-tiled kernel, wider textures (neural textures, 8-16 channels) the
  // wave-private one in blocks of four channels -- before round 4 they fell to the direct kernel, bound by the float-atomic
  // request rate.  (-DDRTK_MIP_BACKWARD_WAVE_ALL: the wave kernel for every C, for A/B; at C = 3 it is slower than the
  // tiled one -- 2.52 against 1.95 ms on the textured benchmark at the same 3 waves per SIMD.)
#ifdef DRTK_MIP_BACKWARD_WAVE_ALL
  constexpr int64_t kWaveFromC = 1;
#else
  constexpr int64_t kWaveFromC = 5;
#endif
  // bicubic (any C >= 1): the same wave-private windows with 4 x 4 cells per tap (before round 4: the direct kernel,
  // sixteen global float atomics per tap, level and channel -- 89 ms against 1.96 ms bilinear on the textured benchmark)
  if (((interpolation_mode == 0 && C >= kWaveFromC) || (interpolation_mode == 2 && C >= 1)) && N <= 65535 &&
      !DRTK_DBG(debug_flags(), 512)) {
    const int tiles_x = static_cast<int>(ceil_div(W, kTileW)), tiles_y = static_cast<int>(ceil_div(H, kTileH));
    const size_t lds = sizeof(double) * (C < kChBlock ? C : kChBlock) * kWaveCells * (kMipBlock / kWave);
#define WAVEK(PAD, ALIGN)                                                                                               \
  if (interpolation_mode == 2)                                                                                          \
    DRTK_LAUNCH(                                                                                                        \
        (mipmap_backward_wave_kernel<T, PAD, ALIGN, 2>), dim3(static_cast<unsigned>(tiles_x * tiles_y), static_cast<unsigned>(N)), \
        dim3(kMipBlock), lds, s, lv, mipmaps, static_cast<const T*>(grad_out), static_cast<const T*>(grid), gl, \
        static_cast<const T*>(vt_dxdy_img), (int)H, (int)W, (int)C, tiles_x, max_aniso, force_max_aniso != 0,     \
        clip_grad != 0, static_cast<T*>(grad_grid), ggl, xcd_strip(tiles_x), debug_flags());                        \
  else                                                                                                                  \
  DRTK_LAUNCH(                                                                                                          \
      (mipmap_backward_wave_kernel<T, PAD, ALIGN>), dim3(static_cast<unsigned>(tiles_x * tiles_y), static_cast<unsigned>(N)), \
      dim3(kMipBlock), lds, s, lv, mipmaps, static_cast<const T*>(grad_out), static_cast<const T*>(grid), gl,  \
      static_cast<const T*>(vt_dxdy_img), (int)H, (int)W, (int)C, tiles_x, max_aniso, force_max_aniso != 0,        \
      clip_grad != 0, static_cast<T*>(grad_grid), ggl, xcd_strip(tiles_x), debug_flags())
    if (align_corners) {
      if (padding_mode == 0) WAVEK(0, true); else if (padding_mode == 1) WAVEK(1, true); else WAVEK(2, true);
    } else {
      if (padding_mode == 0) WAVEK(0, false); else if (padding_mode == 1) WAVEK(1, false); else WAVEK(2, false);
    }
#undef WAVEK
    DRTK_RETURN_IF_LAUNCH_FAILED();
    return DRTK_OK;
  }
  if (interpolation_mode == 0 && C <= 4 && N <= 65535 && !DRTK_DBG(debug_flags(), 512)) {
    const int tiles_x = static_cast<int>(ceil_div(W, kTileW)), tiles_y = static_cast<int>(ceil_div(H, kTileH));
#define TILED(PAD, ALIGN)                                                                                                \
  DRTK_LAUNCH(                                                                                                           \
      (mipmap_backward_tiled2_kernel<T, PAD, ALIGN>), dim3(static_cast<unsigned>(tiles_x * tiles_y), static_cast<unsigned>(N)), \
      dim3(kMipBlock), sizeof(double) * C * kWinCells, s, lv, mipmaps, static_cast<const T*>(grad_out),              \
      static_cast<const T*>(grid), gl, static_cast<const T*>(vt_dxdy_img), (int)H, (int)W, (int)C, tiles_x, max_aniso, \
      force_max_aniso != 0, clip_grad != 0, static_cast<T*>(grad_grid), ggl, xcd_strip(tiles_x), debug_flags())
    if (align_corners) {
      if (padding_mode == 0) TILED(0, true); else if (padding_mode == 1) TILED(1, true); else TILED(2, true);
    } else {
      if (padding_mode == 0) TILED(0, false); else if (padding_mode == 1) TILED(1, false); else TILED(2, false);
    }
#undef TILED
    DRTK_RETURN_IF_LAUNCH_FAILED();
    return DRTK_OK;
  }
    return 1;
  };
  {
    const int w = dtype == DRTK_F32 ? windowed(float{}) : windowed(double{});
    if (w != 1) return w;
  }
  const dim3 grid_dim(static_cast<unsigned>(ceil_div(count, kBlock)));
#define LAUNCH(T, MODE)                                                                                       \
  DRTK_LAUNCH(                                                                                         \
      (mipmap_backward_kernel<T, MODE>), grid_dim, dim3(kBlock), 0, s, lv, mipmaps, static_cast<const T*>(grad_out), \
      static_cast<const T*>(grid), gl, static_cast<const T*>(vt_dxdy_img), count, (int)C, H * W, max_aniso, padding_mode, \
      align_corners != 0, force_max_aniso != 0, clip_grad != 0, static_cast<T*>(grad_grid), ggl,                  \
      xcd_strip(ceil_div(16 * W, kBlock)))
  if (dtype == DRTK_F32) {
    if (interpolation_mode == 0) LAUNCH(float, 0); else LAUNCH(float, 2);
  } else {
    if (interpolation_mode == 0) LAUNCH(double, 0); else LAUNCH(double, 2);
  }
#undef LAUNCH
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}
