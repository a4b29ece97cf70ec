// Wireframe rasterization -- `wireframe=True` of rasterize_ext::rasterize.
//
// Reference: src/rasterize/rasterize_kernel.cu:170-400 (rasterize_lines_kernel; CUDA-only -- the CPU
// twin refuses it, rasterize_kernel_cpu.cpp:257; no reference test): every triangle walks its bbox
// padded by 2 pixels; a pixel is hit by an edge when the edge crosses the pixel's diamond
// (|x-px| + |y-py| = 0.5) and that edge's bit in the top nibble of vi[...,0] is set; hit pixels write
// (depth, id), merely covered pixels write (depth, 0xFFFFFFFF) -- they occlude but show index -1 --
// through the same packed 64-bit atomicMin as the triangle mode.  PARITY UNPINNED: the oracle is a
// restatement of the device code; arithmetic here is strict IEEE in source order so that kernel and
// restatement agree bit for bit.
//
// This mode is a debugging / visualisation aid, not part of the benchmarked path, so the structure is
// kept simple: one wave per triangle, lanes stride over the padded bbox, global packed buffer
// (8 B/px workspace), unpack pass.
// Round 6 measured what bounds it (8 x 100k triangles at 2048^2, every edge on: 4.5 ms against 0.33 ms for triangle mode):
// the per-fragment ARITHMETIC -- three edges x four diamond sides x two IEEE divisions, ~1000 instructions per pixel of the
// padded bounding box -- not the memory side: skipping the global atomicMin wherever a plain read already proves it a no-op
// changed nothing (4.50 vs 4.50 ms), and the triangle mode's recipe (tile bins + a packed z-buffer in LDS, no global buffer,
// no unpack pass; profiles/r06/rasterize_lines_binned.hip.txt) was SLOWER: 11.4 ms with 64 x 64 tiles and 4 waves, 6.9 ms
// with 32 x 32 tiles and 8 waves -- a limb tile's ~1000 list entries are serial work for the few waves of its workgroup,
// where one wave per triangle balances perfectly.  A conservative pre-test ("this pixel is too far from edge k to cross it")
// would have to allow for the float error of the Cramer intersection itself, ~0.25 px at 2048^2 and growing with the square
// of the coordinates: a band of +-1.5 to 2 px around each edge keeps most pixels of a 7-pixel triangle's padded box.
#include "common.hpp"

namespace drtk_amd {
namespace {

template <typename T>
struct MaxOf;
template <>
struct MaxOf<float> {
  static __device__ constexpr float value() { return 3.402823466e+38f; }
};
template <>
struct MaxOf<double> {
  static __device__ constexpr double value() { return 1.7976931348623157e+308; }
};

template <typename T>
__device__ __forceinline__ bool in_segment(T p1x, T p1y, T p2x, T p2y, T cx, T cy) {
  return (((p2x >= cx) && (cx >= p1x)) || ((p2x <= cx) && (cx <= p1x))) &&
      (((p2y >= cy) && (cy >= p1y)) || ((p2y <= cy) && (cy <= p1y)));
}

// one diamond side (s0,s1) against the line a0 x + b0 y + c0 = 0 through the edge (p1,p2): :193-215
template <typename T>
__device__ __forceinline__ bool diamond_side(
    T a0, T b0, T c0, T p1x, T p1y, T p2x, T p2y, T s0x, T s0y, T s1x, T s1y) {
  const T a2 = s0y - s1y, b2 = s1x - s0x, c2 = s0x * s1y - s1x * s0y;
  const T d = a0 * b2 - a2 * b0;
  T cx, cy;
  if (d == T(0)) {
    cx = MaxOf<T>::value();
    cy = T(0);
  } else {
    cx = (b0 * c2 - b2 * c0) / d;
    cy = (a2 * c0 - a0 * c2) / d;
  }
  return in_segment(s0x, s0y, s1x, s1y, cx, cy) && in_segment(p1x, p1y, p2x, p2y, cx, cy);
}

template <typename T>
__device__ __forceinline__ bool crossing_diamond(T p1x, T p1y, T p2x, T p2y, T px, T py) { // :217-260
  const T a0 = p1y - p2y, b0 = p2x - p1x, c0 = p1x * p2y - p2x * p1y;
  const T h = T(0.5);
  bool hit = diamond_side(a0, b0, c0, p1x, p1y, p2x, p2y, px, py - h, px + h, py);
  hit |= diamond_side(a0, b0, c0, p1x, p1y, p2x, p2y, px + h, py, px, py + h);
  hit |= diamond_side(a0, b0, c0, p1x, p1y, p2x, p2y, px, py + h, px - h, py);
  hit |= diamond_side(a0, b0, c0, p1x, p1y, p2x, p2y, px - h, py, px, py - h);
  return hit;
}

template <typename T>
__device__ __forceinline__ T edge_fn(T ax, T ay, T bx, T by, T px, T py) { // :19-27
  return (py - ay) * (bx - ax) - (px - ax) * (by - ay);
}
template <typename T>
__device__ __forceinline__ T canon_edge(int32_t ia, int32_t ib, T ax, T ay, T bx, T by, T px, T py) { // :29-40
  return ia <= ib ? edge_fn(ax, ay, bx, by, px, py) : -edge_fn(bx, by, ax, ay, px, py);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void rasterize_lines_kernel(
    const T* __restrict__ v, const int32_t* __restrict__ vi, int F, int64_t V, int64_t vi_sN, int H, int W,
    unsigned long long* __restrict__ packed) {
  const int n = blockIdx.y;
  const int id = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave; // one wave per triangle
  if (id >= F) return;
  const int lane = threadIdx.x & (kWave - 1);
  const T* v_n = v + int64_t(n) * V * 3;
  const int32_t* face = vi + int64_t(n) * vi_sN + int64_t(id) * 3;
  const uint32_t raw0 = static_cast<uint32_t>(face[0]);
  const int32_t flag = static_cast<int32_t>((raw0 & 0xF0000000u) >> 28);
  const int32_t vi_0 = static_cast<int32_t>(raw0 & 0x0FFFFFFFu);
  const int32_t vi_1 = face[1], vi_2 = face[2];
  if (vi_0 == vi_1 && vi_1 == vi_2) return;
  const bool e0_vis = (flag & 1) != 0, e1_vis = (flag & 2) != 0, e2_vis = (flag & 4) != 0;
  const T p0x = v_n[3 * (int64_t)vi_0 + 0], p0y = v_n[3 * (int64_t)vi_0 + 1], p0z = v_n[3 * (int64_t)vi_0 + 2];
  const T p1x = v_n[3 * (int64_t)vi_1 + 0], p1y = v_n[3 * (int64_t)vi_1 + 1], p1z = v_n[3 * (int64_t)vi_1 + 2];
  const T p2x = v_n[3 * (int64_t)vi_2 + 0], p2y = v_n[3 * (int64_t)vi_2 + 1], p2z = v_n[3 * (int64_t)vi_2 + 2];
  if (!(p0z > T(1e-8f) && p1z > T(1e-8f) && p2z > T(1e-8f))) return;
  const T min_x = min3(p0x, p1x, p2x), min_y = min3(p0y, p1y, p2y);
  const T max_x = max3(p0x, p1x, p2x), max_y = max3(p0y, p1y, p2y);
  if (!(min_x <= T(W - 1) && min_y <= T(H - 1) && max_x > T(0) && max_y > T(0))) return;
  const T v01x = p1x - p0x, v01y = p1y - p0y, v02x = p2x - p0x, v02y = p2y - p0y, v12x = p2x - p1x, v12y = p2y - p1y;
  const T den = v01x * v02y - v01y * v02x;
  if (den == T(0)) return;
  const T sgn = den > T(0) ? T(1) : T(-1), abs_den = den > T(0) ? den : -den;
  // padded bbox (:321-325), in 64-bit from the saturating conversion (the source's 32-bit `int(x) - 2`
  // is undefined beyond +-2^31)
  long long bx0 = static_cast<long long>(trunc_i32(min_x)) - 2, by0 = static_cast<long long>(trunc_i32(min_y)) - 2;
  long long bx1 = static_cast<long long>(trunc_i32(max_x)) + 2, by1 = static_cast<long long>(trunc_i32(max_y)) + 2;
  bx0 = bx0 < 1 ? 1 : bx0;
  by0 = by0 < 1 ? 1 : by0;
  bx1 = bx1 > W - 2 ? W - 2 : bx1;
  by1 = by1 > H - 2 ? H - 2 : by1;
  if (bx0 > bx1 || by0 > by1) return;
  const bool pos = den > T(0);
  const bool tl0 = pos ? (v12y < T(0) || (v12y == T(0) && v12x > T(0))) : (v12y > T(0) || (v12y == T(0) && v12x < T(0)));
  const bool tl1 = pos ? (v02y > T(0) || (v02y == T(0) && v02x < T(0))) : (v02y < T(0) || (v02y == T(0) && v02x > T(0)));
  const bool tl2 = pos ? (v01y < T(0) || (v01y == T(0) && v01x > T(0))) : (v01y > T(0) || (v01y == T(0) && v01x < T(0)));
  const T dinv0 = T(1) / epsclamp(p0z), dinv1 = T(1) / epsclamp(p1z), dinv2 = T(1) / epsclamp(p2z);
  const long long bw = bx1 - bx0 + 1, total = bw * (by1 - by0 + 1);
  unsigned long long* packed_n = packed + int64_t(n) * H * W;
  for (long long i = lane; i < total; i += kWave) {
    const int y = static_cast<int>(by0 + i / bw), x = static_cast<int>(bx0 + i % bw);
    const T px = static_cast<T>(x), py = static_cast<T>(y);
    bool intersecting = crossing_diamond(p0x, p0y, p1x, p1y, px, py) && e0_vis;
    intersecting |= crossing_diamond(p1x, p1y, p2x, p2y, px, py) && e1_vis;
    intersecting |= crossing_diamond(p0x, p0y, p2x, p2y, px, py) && e2_vis;
    T b0 = canon_edge(vi_1, vi_2, p1x, p1y, p2x, p2y, px, py) * sgn;
    T b1 = canon_edge(vi_2, vi_0, p2x, p2y, p0x, p0y, px, py) * sgn;
    T b2 = canon_edge(vi_0, vi_1, p0x, p0y, p1x, p1y, px, py) * sgn;
    const bool inside = (b0 >= T(0)) && (b1 >= T(0)) && (b2 >= T(0));
    const bool covered = inside && !(((b0 == T(0)) && !tl0) || ((b1 == T(0)) && !tl1) || ((b2 == T(0)) && !tl2));
    if (!(covered || intersecting)) continue;
    b0 /= abs_den, b1 /= abs_den, b2 /= abs_den; // :376-379
    b0 = b0 > T(0) ? b0 : T(0), b1 = b1 > T(0) ? b1 : T(0), b2 = b2 > T(0) ? b2 : T(0);
    b0 = b0 < T(1) ? b0 : T(1), b1 = b1 < T(1) ? b1 : T(1), b2 = b2 < T(1) ? b2 : T(1);
    const T sum = b0 + b1 + b2;
    b0 = b0 / sum, b1 = b1 / sum, b2 = b2 / sum;
    const T depth_inverse = dinv0 * b0 + dinv1 * b1 + dinv2 * b2;
    const float depth = static_cast<float>(T(1) / epsclamp(depth_inverse));
    const unsigned long long packed_val = (static_cast<unsigned long long>(__float_as_uint(depth)) << 32) |
        (intersecting ? static_cast<unsigned long long>(static_cast<uint32_t>(id)) : 0xFFFFFFFFull);
    atomicMin(packed_n + int64_t(y) * W + x, packed_val);
  }
}

__global__ __launch_bounds__(kBlock) void unpack_lines_kernel( // rasterize_kernel.cu:402-415
    const unsigned long long* __restrict__ packed, int64_t count, float* __restrict__ depth_img,
    int32_t* __restrict__ index_img) {
  const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= count) return;
  const unsigned long long pv = packed[i];
  const uint32_t hi = static_cast<uint32_t>(pv >> 32);
  depth_img[i] = (hi == 0xFFFFFFFFu) ? 0.0f : __uint_as_float(hi);
  index_img[i] = static_cast<int32_t>(static_cast<uint32_t>(pv & 0xFFFFFFFFull));
}

} // namespace

// called by drtk_amd_rasterize (rasterize.hip) when wireframe != 0
int rasterize_lines_dispatch(
    drtk_dtype_t dtype, const void* v, const int32_t* vi, int64_t N, int64_t V, int64_t F, int64_t vi_sN, int64_t H,
    int64_t W, float* depth_img, int32_t* index_img, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const int64_t count = N * H * W;
  if (count == 0) return DRTK_OK;
  if (workspace_bytes < sizeof(unsigned long long) * static_cast<size_t>(count) || !workspace) return DRTK_ERR_WORKSPACE_TOO_SMALL;
  if (F >= (int64_t(1) << 31)) return DRTK_ERR_INVALID_ARGUMENT;
  auto* packed = static_cast<unsigned long long*>(workspace);
  if (fill_bytes_async(packed, 0xFF, sizeof(unsigned long long) * count, stream) != DRTK_OK) return DRTK_ERR_LAUNCH; // :484-488
  if (N * F > 0) {
    const dim3 grid(static_cast<unsigned>(ceil_div(F, kBlock / kWave)), static_cast<unsigned>(N));
    if (dtype == DRTK_F32) {
      DRTK_LAUNCH((rasterize_lines_kernel<float>), grid, dim3(kBlock), 0, stream, static_cast<const float*>(v), vi, (int)F, V, vi_sN, (int)H, (int)W, packed);
    } else {
      DRTK_LAUNCH((rasterize_lines_kernel<double>), grid, dim3(kBlock), 0, stream, static_cast<const double*>(v), vi, (int)F, V, vi_sN, (int)H, (int)W, packed);
    }
    DRTK_RETURN_IF_LAUNCH_FAILED();
  }
  DRTK_LAUNCH(unpack_lines_kernel, dim3(static_cast<unsigned>(ceil_div(count, kBlock))), dim3(kBlock), 0, stream, packed, count, depth_img, index_img);
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

} // namespace drtk_amd

extern "C" int drtk_amd_rasterize_lines_workspace_bytes(int64_t N, int64_t H, int64_t W, size_t* bytes) {
  if (!bytes || N < 0 || H <= 0 || W <= 0) return DRTK_ERR_INVALID_ARGUMENT;
  const size_t b = sizeof(unsigned long long) * static_cast<size_t>(N) * H * W;
  *bytes = b > 0 ? b : 16;
  return DRTK_OK;
}
