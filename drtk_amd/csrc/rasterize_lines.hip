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
// Structure (round 6; rounds 2-5: one wave per triangle, a global packed [N,H,W] 64-bit buffer filled with 0xFF, a global
// atomicMin per fragment and an unpack pass -- the reference's shape, 28 B/px of passes around ~50 M global 64-bit atomics
// at the bench shape): the triangle mode's recipe with a binner of its own --
//   lines_bin_kernel<COUNT>   cull (:285-316) + the PADDED, clamped bbox (:321-325) -> tile range; a triangle on <= 4 tiles
//                             bumps their counters (wave-aggregated), one on more goes to its view's big list
//   lines_scan_kernel         exclusive scan of the tile counters (one workgroup)
//   lines_bin_kernel<FILL>    tile lists
//   lines_tile_kernel         one workgroup per 64 x 64 (32 x 32 for small jobs) tile: packed (depth, id) z-buffer in LDS
//                             (ds_min_u64), a wave per list entry, lanes over the entry's bbox clipped to the tile, then
//                             the tile is unpacked (:402-415) and stored -- no fill pass, no unpack pass, no global atomic.
// Fragments and their packed minimum are the reference's whatever the order: images are what they were.
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

// cull + padded bbox of one triangle (:285-325).  false: the reference drops it.
template <typename T>
struct LineTri {
  int32_t vi_0, vi_1, vi_2, flag;
  T p0x, p0y, p0z, p1x, p1y, p1z, p2x, p2y, p2z;
  T den;
  int bx0, by0, bx1, by1; // padded by 2, clamped to [1, W-2] x [1, H-2]
};
template <typename T>
__device__ __forceinline__ bool line_tri(const T* __restrict__ v_n, const int32_t* __restrict__ face, int H, int W, LineTri<T>& t) {
  const uint32_t raw0 = static_cast<uint32_t>(face[0]);
  t.flag = static_cast<int32_t>((raw0 & 0xF0000000u) >> 28);
  t.vi_0 = static_cast<int32_t>(raw0 & 0x0FFFFFFFu);
  t.vi_1 = face[1], t.vi_2 = face[2];
  if (t.vi_0 == t.vi_1 && t.vi_1 == t.vi_2) return false;
  t.p0x = v_n[3 * (int64_t)t.vi_0 + 0], t.p0y = v_n[3 * (int64_t)t.vi_0 + 1], t.p0z = v_n[3 * (int64_t)t.vi_0 + 2];
  t.p1x = v_n[3 * (int64_t)t.vi_1 + 0], t.p1y = v_n[3 * (int64_t)t.vi_1 + 1], t.p1z = v_n[3 * (int64_t)t.vi_1 + 2];
  t.p2x = v_n[3 * (int64_t)t.vi_2 + 0], t.p2y = v_n[3 * (int64_t)t.vi_2 + 1], t.p2z = v_n[3 * (int64_t)t.vi_2 + 2];
  if (!(t.p0z > T(1e-8f) && t.p1z > T(1e-8f) && t.p2z > T(1e-8f))) return false;
  const T min_x = min3(t.p0x, t.p1x, t.p2x), min_y = min3(t.p0y, t.p1y, t.p2y);
  const T max_x = max3(t.p0x, t.p1x, t.p2x), max_y = max3(t.p0y, t.p1y, t.p2y);
  if (!(min_x <= T(W - 1) && min_y <= T(H - 1) && max_x > T(0) && max_y > T(0))) return false;
  const T v01x = t.p1x - t.p0x, v01y = t.p1y - t.p0y, v02x = t.p2x - t.p0x, v02y = t.p2y - t.p0y;
  t.den = v01x * v02y - v01y * v02x;
  if (t.den == T(0)) return false;
  // padded bbox (:321-325), in 64-bit from the saturating conversion (the source's 32-bit `int(x) - 2`
  // is undefined beyond +-2^31)
  long long bx0 = static_cast<long long>(trunc_i32(min_x)) - 2, by0 = static_cast<long long>(trunc_i32(min_y)) - 2;
  long long bx1 = static_cast<long long>(trunc_i32(max_x)) + 2, by1 = static_cast<long long>(trunc_i32(max_y)) + 2;
  bx0 = bx0 < 1 ? 1 : bx0;
  by0 = by0 < 1 ? 1 : by0;
  bx1 = bx1 > W - 2 ? W - 2 : bx1;
  by1 = by1 > H - 2 ? H - 2 : by1;
  if (bx0 > bx1 || by0 > by1) return false;
  t.bx0 = static_cast<int>(bx0), t.by0 = static_cast<int>(by0), t.bx1 = static_cast<int>(bx1), t.by1 = static_cast<int>(by1);
  return true;
}

constexpr int kLinesMaxSmallTiles = 4; // a triangle on more tiles goes to its view's big list

// COUNT: cull, bbox -> record + tile counters / big list.  !COUNT (fill): the tile lists, from the records.
template <typename T, bool COUNT>
__global__ __launch_bounds__(kBlock) void lines_bin_kernel(
    const T* __restrict__ v, const int32_t* __restrict__ vi, int F, int64_t V, int64_t vi_sN, int H, int W, int tile_shift,
    int tiles_x, int tiles_per_view, int4* __restrict__ bbox, int32_t* __restrict__ tile_count,
    const int32_t* __restrict__ tile_offset, int32_t* __restrict__ tile_cursor, int32_t* __restrict__ big_count,
    int32_t* __restrict__ big_list, int32_t* __restrict__ pairs) {
  const int n = blockIdx.y;
  const int f = blockIdx.x * kBlock + threadIdx.x;
  int4 bb = make_int4(1, 1, 0, 0); // empty
  if (f < F) {
    if constexpr (COUNT) {
      LineTri<T> t;
      if (line_tri<T>(v + int64_t(n) * V * 3, vi + int64_t(n) * vi_sN + int64_t(f) * 3, H, W, t)) bb = make_int4(t.bx0, t.by0, t.bx1, t.by1);
      bbox[int64_t(n) * F + f] = bb;
    } else {
      bb = bbox[int64_t(n) * F + f];
    }
  }
  const bool valid = bb.x <= bb.z;
  const int tx0 = bb.x >> tile_shift, ty0 = bb.y >> tile_shift;
  const int ntx = valid ? (bb.z >> tile_shift) - tx0 + 1 : 0, nty = valid ? (bb.w >> tile_shift) - ty0 + 1 : 0;
  const long long ntiles = static_cast<long long>(ntx) * nty;
  const bool small = valid && ntiles <= kLinesMaxSmallTiles;
  int32_t* count_n = tile_count + int64_t(n) * tiles_per_view;
  int32_t* cursor_n = tile_cursor + int64_t(n) * tiles_per_view;
#pragma unroll
  for (int k = 0; k < kLinesMaxSmallTiles; ++k) { // (every lane of the wave takes part in the aggregation)
    const bool on = small && k < ntiles;
    const int ntx1 = ntx > 0 ? ntx : 1;
    const int tile = on ? (ty0 + k / ntx1) * tiles_x + tx0 + k % ntx1 : 0;
    if constexpr (COUNT) {
      wave_agg_inc<false>(count_n, tile, on);
    } else {
      const int pos = wave_agg_inc<true>(cursor_n, tile, on);
      if (on) pairs[tile_offset[int64_t(n) * tiles_per_view + tile] + pos] = f;
    }
  }
  if constexpr (COUNT) {
    const bool big = valid && !small;
    const int pos = wave_agg_inc<true>(big_count, n, big);
    if (big) big_list[int64_t(n) * F + pos] = f;
  }
}

// tile_offset[i] = sum of tile_count[0 .. i-1], tile_offset[num_tiles] = the total: one workgroup, 1024 tiles a round
__global__ __launch_bounds__(1024) void lines_scan_kernel(const int32_t* __restrict__ tile_count, int32_t* __restrict__ tile_offset, int64_t num_tiles) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < num_tiles; base += 1024) {
    const int64_t i = base + tid;
    const int c = i < num_tiles ? tile_count[i] : 0;
    int x = c;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == kWave - 1) s_wave[wave] = x;
    __syncthreads();
    if (wave == 0) {
      const int w = lane < 16 ? s_wave[lane] : 0;
      int xs = w;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const int y = __shfl_up(xs, d);
        if (lane >= d) xs += y;
      }
      if (lane < 16) s_wave[lane] = xs - w; // exclusive
    }
    __syncthreads();
    const int excl = x - c + s_wave[wave] + s_carry;
    if (i < num_tiles) tile_offset[i] = excl;
    __syncthreads();
    if (tid == 1023) s_carry = excl + c;
    __syncthreads();
  }
  if (tid == 0) tile_offset[num_tiles] = s_carry;
}

// One workgroup = one tile.  Per-fragment arithmetic: rasterize_kernel.cu:327-397, strict IEEE in source order (the oracle's).
template <typename T, int TILE_SHIFT>
__global__ __launch_bounds__(kBlock) void lines_tile_kernel(
    const T* __restrict__ v, const int32_t* __restrict__ vi, int F, int64_t V, int64_t vi_sN, int H, int W, int tiles_x,
    int tiles_per_view, const int32_t* __restrict__ tile_offset, const int32_t* __restrict__ tile_count,
    const int32_t* __restrict__ pairs, const int32_t* __restrict__ big_count, const int32_t* __restrict__ big_list,
    const int4* __restrict__ bbox, float* __restrict__ depth_img, int32_t* __restrict__ index_img) {
  constexpr int TILE = 1 << TILE_SHIFT;
  constexpr int NPIX = TILE * TILE;
  constexpr int kWaves = kBlock / kWave;
  __shared__ unsigned long long zbuf[NPIX];
  const int n = blockIdx.y, tile = blockIdx.x;
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int x0 = txi << TILE_SHIFT, y0 = tyi << TILE_SHIFT;
  const int x1 = min(x0 + TILE - 1, W - 1), y1 = min(y0 + TILE - 1, H - 1);
  for (int i = threadIdx.x; i < NPIX; i += kBlock) zbuf[i] = ~0ull; // :484-488
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int lane = threadIdx.x & (kWave - 1);
  const T* v_n = v + int64_t(n) * V * 3;
  const int32_t* vi_n = vi + int64_t(n) * vi_sN;

  auto shade = [&](int id) { // wave-uniform triangle: every lane sets it up (scalar loads), lanes stride over its pixels
    LineTri<T> t;
    if (!line_tri<T>(v_n, vi_n + int64_t(id) * 3, H, W, t)) return; // (cannot happen for a binned triangle)
    const int cx0 = max(t.bx0, x0), cy0 = max(t.by0, y0), cx1 = min(t.bx1, x1), cy1 = min(t.by1, y1);
    if (cx0 > cx1 || cy0 > cy1) return;
    const bool e0_vis = (t.flag & 1) != 0, e1_vis = (t.flag & 2) != 0, e2_vis = (t.flag & 4) != 0;
    const T p0x = t.p0x, p0y = t.p0y, p1x = t.p1x, p1y = t.p1y, p2x = t.p2x, p2y = t.p2y;
    const T v01x = p1x - p0x, v01y = p1y - p0y, v02x = p2x - p0x, v02y = p2y - p0y, v12x = p2x - p1x, v12y = p2y - p1y;
    const T den = t.den;
    const T sgn = den > T(0) ? T(1) : T(-1), abs_den = den > T(0) ? den : -den;
    const bool pos = den > T(0);
    const bool tl0 = pos ? (v12y < T(0) || (v12y == T(0) && v12x > T(0))) : (v12y > T(0) || (v12y == T(0) && v12x < T(0)));
    const bool tl1 = pos ? (v02y > T(0) || (v02y == T(0) && v02x < T(0))) : (v02y < T(0) || (v02y == T(0) && v02x > T(0)));
    const bool tl2 = pos ? (v01y < T(0) || (v01y == T(0) && v01x > T(0))) : (v01y > T(0) || (v01y == T(0) && v01x < T(0)));
    const T dinv0 = T(1) / epsclamp(t.p0z), dinv1 = T(1) / epsclamp(t.p1z), dinv2 = T(1) / epsclamp(t.p2z);
    const int cw = cx1 - cx0 + 1, total = cw * (cy1 - cy0 + 1); // <= 4096
    // pixel i of the clipped bbox -> (i % cw, i / cw) through ONE IEEE float division: (i + 1/2) / cw is never within
    // 1 / (2 cw) >= 1/128 of an integer and the quotient is correctly rounded (i < 4096): the truncation is exact
    const float cwf = static_cast<float>(cw);
    for (int i = lane; i < total; i += kWave) {
      const int ly = static_cast<int>((static_cast<float>(i) + 0.5f) / cwf);
      const int lx = i - ly * cw;
      const int x = cx0 + lx, y = cy0 + ly;
      const T px = static_cast<T>(x), py = static_cast<T>(y);
      bool intersecting = crossing_diamond(p0x, p0y, p1x, p1y, px, py) && e0_vis;
      intersecting |= crossing_diamond(p1x, p1y, p2x, p2y, px, py) && e1_vis;
      intersecting |= crossing_diamond(p0x, p0y, p2x, p2y, px, py) && e2_vis;
      T b0 = canon_edge(t.vi_1, t.vi_2, p1x, p1y, p2x, p2y, px, py) * sgn;
      T b1 = canon_edge(t.vi_2, t.vi_0, p2x, p2y, p0x, p0y, px, py) * sgn;
      T b2 = canon_edge(t.vi_0, t.vi_1, p0x, p0y, p1x, p1y, px, py) * sgn;
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
      atomicMin(&zbuf[((y - y0) << TILE_SHIFT) + (x - x0)], packed_val);
    }
  };

  {
    const int64_t t_glob = int64_t(n) * tiles_per_view + tile;
    const int off = tile_offset[t_glob], cnt = tile_count[t_glob];
    for (int e = wave; e < cnt; e += kWaves) shade(__builtin_amdgcn_readfirstlane(pairs[off + e]));
    const int nb = big_count[n];
    for (int e = wave; e < nb; e += kWaves) {
      const int id = __builtin_amdgcn_readfirstlane(big_list[int64_t(n) * F + e]);
      const int4 bb = bbox[int64_t(n) * F + id];
      if (bb.x <= x1 && bb.z >= x0 && bb.y <= y1 && bb.w >= y0) shade(id);
    }
  }
  __syncthreads();
  // unpack (:402-415) and store: consecutive lanes = consecutive pixels of a tile row
  for (int i = threadIdx.x; i < NPIX; i += kBlock) {
    const int x = x0 + (i & (TILE - 1)), y = y0 + (i >> TILE_SHIFT);
    if (x < W && y < H) {
      const unsigned long long pv = zbuf[i];
      const uint32_t hi = static_cast<uint32_t>(pv >> 32);
      const int64_t o = (int64_t(n) * H + y) * W + x;
      depth_img[o] = (hi == 0xFFFFFFFFu) ? 0.0f : __uint_as_float(hi);
      index_img[o] = static_cast<int32_t>(static_cast<uint32_t>(pv & 0xFFFFFFFFull));
    }
  }
}

struct LinesLayout {
  int tile_shift, tiles_x, tiles_y;
  int64_t tiles_per_view, num_tiles;
  size_t off_count, off_cursor, off_big_count, zero_bytes, off_offset, off_bbox, off_big_list, off_pairs, total_bytes;
};
inline size_t lines_align(size_t x) { return (x + 255) / 256 * 256; }
inline LinesLayout make_lines_layout(int64_t N, int64_t F, int64_t H, int64_t W) {
  LinesLayout L;
  const int64_t t64 = N * ceil_div(W, 64) * ceil_div(H, 64);
  L.tile_shift = (t64 >= 2048) ? 6 : 5; // (as the triangle mode: fewer than ~8 workgroups per CU -> 32 x 32 tiles)
  const int64_t ts = int64_t(1) << L.tile_shift;
  L.tiles_x = static_cast<int>(ceil_div(W, ts));
  L.tiles_y = static_cast<int>(ceil_div(H, ts));
  L.tiles_per_view = int64_t(L.tiles_x) * L.tiles_y;
  L.num_tiles = N * L.tiles_per_view;
  size_t o = 0;
  L.off_count = o, o += lines_align(sizeof(int32_t) * L.num_tiles);
  L.off_cursor = o, o += lines_align(sizeof(int32_t) * L.num_tiles);
  L.off_big_count = o, o += lines_align(sizeof(int32_t) * (N > 0 ? N : 1));
  L.zero_bytes = o;
  L.off_offset = o, o += lines_align(sizeof(int32_t) * (L.num_tiles + 1));
  L.off_bbox = o, o += lines_align(sizeof(int4) * N * F);
  L.off_big_list = o, o += lines_align(sizeof(int32_t) * N * F);
  L.off_pairs = o, o += lines_align(sizeof(int32_t) * kLinesMaxSmallTiles * N * F);
  L.total_bytes = o > 0 ? o : 256;
  return L;
}

} // namespace

// called by drtk_amd_rasterize (rasterize.hip) when wireframe != 0
int rasterize_lines_dispatch(
    drtk_dtype_t dtype, const void* v, const int32_t* vi, int64_t N, int64_t V, int64_t F, int64_t vi_sN, int64_t H,
    int64_t W, float* depth_img, int32_t* index_img, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const int64_t count = N * H * W;
  if (count == 0) return DRTK_OK;
  // list positions and tile counters are 32-bit
  if (F >= (int64_t(1) << 31) || kLinesMaxSmallTiles * N * F >= (int64_t(1) << 31)) return DRTK_ERR_INVALID_ARGUMENT;
  const LinesLayout L = make_lines_layout(N, F, H, W);
  if (workspace_bytes < L.total_bytes || !workspace) return DRTK_ERR_WORKSPACE_TOO_SMALL;
  char* ws = static_cast<char*>(workspace);
  auto* tile_count = reinterpret_cast<int32_t*>(ws + L.off_count);
  auto* tile_cursor = reinterpret_cast<int32_t*>(ws + L.off_cursor);
  auto* big_count = reinterpret_cast<int32_t*>(ws + L.off_big_count);
  auto* tile_offset = reinterpret_cast<int32_t*>(ws + L.off_offset);
  auto* bbox = reinterpret_cast<int4*>(ws + L.off_bbox);
  auto* big_list = reinterpret_cast<int32_t*>(ws + L.off_big_list);
  auto* pairs = reinterpret_cast<int32_t*>(ws + L.off_pairs);
  if (fill_bytes_async(ws, 0, L.zero_bytes, stream) != DRTK_OK) return DRTK_ERR_LAUNCH;
  const dim3 tri_grid(static_cast<unsigned>(ceil_div(F > 0 ? F : 1, kBlock)), static_cast<unsigned>(N));
  const dim3 tile_grid(static_cast<unsigned>(L.tiles_per_view), static_cast<unsigned>(N));
#define DRTK_LINES(T)                                                                                                          \
  do {                                                                                                                         \
    const T* vt = static_cast<const T*>(v);                                                                                    \
    if (N * F > 0) {                                                                                                           \
      DRTK_LAUNCH((lines_bin_kernel<T, true>), tri_grid, dim3(kBlock), 0, stream, vt, vi, (int)F, V, vi_sN, (int)H, (int)W, L.tile_shift, \
                  L.tiles_x, (int)L.tiles_per_view, bbox, tile_count, tile_offset, tile_cursor, big_count, big_list, pairs);  \
      DRTK_RETURN_IF_LAUNCH_FAILED();                                                                                          \
    }                                                                                                                          \
    DRTK_LAUNCH(lines_scan_kernel, dim3(1), dim3(1024), 0, stream, tile_count, tile_offset, L.num_tiles);                      \
    DRTK_RETURN_IF_LAUNCH_FAILED();                                                                                            \
    if (N * F > 0) {                                                                                                           \
      DRTK_LAUNCH((lines_bin_kernel<T, false>), tri_grid, dim3(kBlock), 0, stream, vt, vi, (int)F, V, vi_sN, (int)H, (int)W, L.tile_shift, \
                  L.tiles_x, (int)L.tiles_per_view, bbox, tile_count, tile_offset, tile_cursor, big_count, big_list, pairs);  \
      DRTK_RETURN_IF_LAUNCH_FAILED();                                                                                          \
    }                                                                                                                          \
    if (L.tile_shift == 6) {                                                                                                   \
      DRTK_LAUNCH((lines_tile_kernel<T, 6>), tile_grid, dim3(kBlock), 0, stream, vt, vi, (int)F, V, vi_sN, (int)H, (int)W, L.tiles_x, \
                  (int)L.tiles_per_view, tile_offset, tile_count, pairs, big_count, big_list, bbox, depth_img, index_img);   \
    } else {                                                                                                                   \
      DRTK_LAUNCH((lines_tile_kernel<T, 5>), tile_grid, dim3(kBlock), 0, stream, vt, vi, (int)F, V, vi_sN, (int)H, (int)W, L.tiles_x, \
                  (int)L.tiles_per_view, tile_offset, tile_count, pairs, big_count, big_list, bbox, depth_img, index_img);   \
    }                                                                                                                          \
    DRTK_RETURN_IF_LAUNCH_FAILED();                                                                                            \
  } while (0)
  if (dtype == DRTK_F32) {
    DRTK_LINES(float);
  } else {
    DRTK_LINES(double);
  }
#undef DRTK_LINES
  return DRTK_OK;
}

} // namespace drtk_amd

extern "C" int drtk_amd_rasterize_lines_workspace_bytes(int64_t N, int64_t F, int64_t H, int64_t W, size_t* bytes) {
  if (!bytes || N < 0 || F < 0 || H <= 0 || W <= 0) return DRTK_ERR_INVALID_ARGUMENT;
  *bytes = drtk_amd::make_lines_layout(N, F, H, W).total_bytes;
  return DRTK_OK;
}
