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
//   4. tile_raster one workgroup per (view, tile): the tile's packed (depth_bits<<32 | id) z-buffer
//                  lives in LDS (64x64x8 B = 32 KiB), small triangles are rasterized one per lane,
//                  big triangles pixel-parallel across the whole workgroup, both with LDS 64-bit
//                  atomicMin (ds_min_u64); the finished tile is unpacked and stored once.
//
// HBM traffic is therefore the 8 B/px of real output plus the bins (~28 B per triangle): no
// memset, no global atomics, no unpack pass (the reference moves >= 24 B/px).
// The min-reduction over (depth bits, id) is order independent, so the result is deterministic
// and bit-identical to the reference regardless of bin order.
#include "common.hpp"

namespace drtk_amd {
namespace {

constexpr uint32_t kCulled = 0xFFFFFFFFu;
constexpr int kMaxSmallTiles = 4; // bins hold at most this many entries per triangle

// Exact per-triangle setup (rasterize_kernel.cu:73-113,133-141).
template <typename T>
struct TriSetup {
  T p0x, p0y, p1x, p1y, p2x, p2y;
  T dinv0, dinv1, dinv2;
  T sign_denom, abs_denom;
  int bb_min_x, bb_min_y, bb_max_x, bb_max_y;
  bool c0, c1, c2;    // canonical edge orientation flags (vi_a <= vi_b)
  bool tl0, tl1, tl2; // top-left classification
};

// Culling + bounding box only (used by the binning passes).  Returns false if the triangle is
// dropped by the reference (:81 degenerate indices, :96 near plane, :97-98 off canvas, :107 zero
// area).  On success the pixel bbox (already clamped to the canvas) is returned.
template <typename T>
__device__ __forceinline__ bool tri_bbox(
    const T* __restrict__ v_n, const int32_t* __restrict__ vi_face, int H, int W, int& bx0,
    int& by0, int& bx1, int& by1) {
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

struct BinLayout {
  int tile_shift; // log2(tile size in pixels)
  int tiles_x, tiles_y;
  int64_t tiles_per_view, num_tiles; // per view / total
  size_t off_count, off_cursor, off_big_count, zero_bytes; // zero-filled prefix
  size_t off_offset, off_range, off_big_list, off_pairs, total_bytes;
};

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
  L.off_count = o;
  o += align_up(sizeof(int32_t) * L.num_tiles, 256);
  L.off_cursor = o;
  o += align_up(sizeof(int32_t) * L.num_tiles, 256);
  L.off_big_count = o;
  o += align_up(sizeof(int32_t) * (N > 0 ? N : 1), 256);
  L.zero_bytes = o;
  L.off_offset = o;
  o += align_up(sizeof(int32_t) * (L.num_tiles + 1), 256);
  L.off_range = o;
  o += align_up(sizeof(uint2) * N * F, 256);
  L.off_big_list = o;
  o += align_up(sizeof(int32_t) * N * F, 256);
  L.off_pairs = o;
  o += align_up(sizeof(int32_t) * kMaxSmallTiles * N * F, 256);
  L.total_bytes = o > 0 ? o : 256;
  return L;
}

// ---- pass 1: cull, bbox -> tile range, count -------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void bin_count_kernel(
    const T* __restrict__ v, const int32_t* __restrict__ vi, int64_t total, int F, int64_t V,
    int64_t vi_sN, int H, int W, int tile_shift, int tiles_x, int tiles_per_view,
    int32_t* __restrict__ tile_count, int32_t* __restrict__ big_count,
    int32_t* __restrict__ big_list, uint2* __restrict__ tri_range) {
  const int64_t idx = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (idx >= total) return;
  const int n = static_cast<int>(idx / F);
  const int f = static_cast<int>(idx - int64_t(n) * F);
  int bx0, by0, bx1, by1;
  uint2 r = make_uint2(kCulled, kCulled);
  if (tri_bbox<T>(v + int64_t(n) * V * 3, vi + int64_t(n) * vi_sN + int64_t(f) * 3, H, W, bx0, by0, bx1, by1)) {
    const int tx0 = bx0 >> tile_shift, tx1 = bx1 >> tile_shift;
    const int ty0 = by0 >> tile_shift, ty1 = by1 >> tile_shift;
    r.x = static_cast<uint32_t>(tx0) | (static_cast<uint32_t>(tx1) << 16);
    r.y = static_cast<uint32_t>(ty0) | (static_cast<uint32_t>(ty1) << 16);
    const int ntiles = (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
    if (ntiles <= kMaxSmallTiles) {
      int32_t* cnt = tile_count + int64_t(n) * tiles_per_view;
      for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(cnt + ty * tiles_x + tx, 1);
    } else {
      const int pos = atomicAdd(big_count + n, 1);
      big_list[int64_t(n) * F + pos] = f;
    }
  }
  tri_range[idx] = r;
}

// ---- pass 2: exclusive scan over all tile counters (single workgroup) -------------------------
__global__ __launch_bounds__(1024) void bin_scan_kernel(
    const int32_t* __restrict__ tile_count, int32_t* __restrict__ tile_offset, int num_tiles) {
  __shared__ int32_t part[1024];
  const int tid = threadIdx.x;
  const int chunk = (num_tiles + 1023) / 1024;
  const int begin = tid * chunk;
  const int end = min(begin + chunk, num_tiles);
  int32_t sum = 0;
  for (int i = begin; i < end; ++i) sum += tile_count[i];
  part[tid] = sum;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 1024 partial sums
  for (int off = 1; off < 1024; off <<= 1) {
    int32_t add = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  int32_t run = part[tid] - sum; // exclusive prefix of this thread's chunk
  for (int i = begin; i < end; ++i) {
    tile_offset[i] = run;
    run += tile_count[i];
  }
  if (tid == 1023) tile_offset[num_tiles] = part[1023];
}

// ---- pass 3: write triangle ids into the tile lists -----------------------------------------
__global__ __launch_bounds__(kBlock) void bin_fill_kernel(
    const uint2* __restrict__ tri_range, int64_t total, int F, int tiles_x, int tiles_per_view,
    const int32_t* __restrict__ tile_offset, int32_t* __restrict__ tile_cursor,
    int32_t* __restrict__ pairs) {
  const int64_t idx = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (idx >= total) return;
  const uint2 r = tri_range[idx];
  if (r.x == kCulled) return;
  const int tx0 = r.x & 0xFFFF, tx1 = r.x >> 16, ty0 = r.y & 0xFFFF, ty1 = r.y >> 16;
  if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > kMaxSmallTiles) return;
  const int n = static_cast<int>(idx / F);
  const int f = static_cast<int>(idx - int64_t(n) * F);
  const int64_t base = int64_t(n) * tiles_per_view;
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) {
      const int64_t t = base + ty * tiles_x + tx;
      const int pos = atomicAdd(tile_cursor + t, 1);
      pairs[tile_offset[t] + pos] = f;
    }
}

// ---- pass 4: per-tile rasterization with the z-buffer in LDS -----------------------------------
template <typename T, int TILE_SHIFT>
__global__ __launch_bounds__(kBlock) void tile_raster_kernel(
    const T* __restrict__ v, const int32_t* __restrict__ vi, int F, int64_t V, int64_t vi_sN,
    int H, int W, int tiles_x, int tiles_per_view, const int32_t* __restrict__ tile_offset,
    const int32_t* __restrict__ pairs, const int32_t* __restrict__ big_count,
    const int32_t* __restrict__ big_list, const uint2* __restrict__ tri_range,
    float* __restrict__ depth_img, int32_t* __restrict__ index_img) {
  constexpr int TILE = 1 << TILE_SHIFT;
  constexpr int NPIX = TILE * TILE;
  __shared__ unsigned long long zbuf[NPIX];

  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int n = tile / tiles_per_view;
  const int t_in_view = tile - n * tiles_per_view;
  const int ty = t_in_view / tiles_x, tx = t_in_view - ty * tiles_x;
  const int x0 = tx << TILE_SHIFT, y0 = ty << TILE_SHIFT;
  const int x1 = min(x0 + TILE - 1, W - 1), y1 = min(y0 + TILE - 1, H - 1);

  for (int i = tid; i < NPIX; i += kBlock) zbuf[i] = ~0ull; // rasterize_kernel.cu:484-488
  __syncthreads();

  const T* v_n = v + int64_t(n) * V * 3;
  const int32_t* vi_n = vi + int64_t(n) * vi_sN;

  // small triangles: one per lane
  const int begin = tile_offset[tile], end = tile_offset[tile + 1];
  for (int i = begin + tid; i < end; i += kBlock) {
    const int f = pairs[i];
    TriSetup<T> s;
    if (!tri_setup<T>(v_n, vi_n + int64_t(f) * 3, H, W, s)) continue;
    const int bx0 = max(s.bb_min_x, x0), bx1 = min(s.bb_max_x, x1);
    const int by0 = max(s.bb_min_y, y0), by1 = min(s.bb_max_y, y1);
    for (int y = by0; y <= by1; ++y) {
      for (int x = bx0; x <= bx1; ++x) {
        uint32_t dbits;
        if (fragment(s, x, y, dbits)) {
          const unsigned long long packed =
              (static_cast<unsigned long long>(dbits) << 32) | static_cast<uint32_t>(f);
          atomicMin(&zbuf[((y - y0) << TILE_SHIFT) + (x - x0)], packed);
        }
      }
    }
  }

  // big triangles: the whole workgroup sweeps bbox ∩ tile pixel-parallel
  const int nbig = big_count[n];
  const int32_t* big_n = big_list + int64_t(n) * F;
  const uint2* range_n = tri_range + int64_t(n) * F;
  for (int b = 0; b < nbig; ++b) {
    const int f = __builtin_amdgcn_readfirstlane(big_n[b]);
    const uint2 r = range_n[f];
    const int rtx0 = r.x & 0xFFFF, rtx1 = r.x >> 16, rty0 = r.y & 0xFFFF, rty1 = r.y >> 16;
    if (tx < rtx0 || tx > rtx1 || ty < rty0 || ty > rty1) continue;
    TriSetup<T> s;
    if (!tri_setup<T>(v_n, vi_n + int64_t(f) * 3, H, W, s)) continue;
    const int bx0 = max(s.bb_min_x, x0), bx1 = min(s.bb_max_x, x1);
    const int by0 = max(s.bb_min_y, y0), by1 = min(s.bb_max_y, y1);
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    if (bw <= 0 || bh <= 0) continue;
    const int npx = bw * bh;
    for (int k = tid; k < npx; k += kBlock) {
      const int yy = k / bw;
      const int x = bx0 + (k - yy * bw), y = by0 + yy;
      uint32_t dbits;
      if (fragment(s, x, y, dbits)) {
        const unsigned long long packed =
            (static_cast<unsigned long long>(dbits) << 32) | static_cast<uint32_t>(f);
        atomicMin(&zbuf[((y - y0) << TILE_SHIFT) + (x - x0)], packed);
      }
    }
  }
  __syncthreads();

  // unpack + store (rasterize_kernel.cu:402-415)
  const int64_t img_base = int64_t(n) * H * W;
  const bool vec_ok = (W & 3) == 0;
  for (int q = tid; q < NPIX / 4; q += kBlock) {
    const int row = q >> (TILE_SHIFT - 2);
    const int col = (q & ((TILE >> 2) - 1)) << 2;
    const int y = y0 + row, x = x0 + col;
    if (y > y1 || x > x1) continue;
    int32_t idx4[4];
    float dep4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned long long pv = zbuf[(row << TILE_SHIFT) + col + j];
      const uint32_t hi = static_cast<uint32_t>(pv >> 32);
      dep4[j] = (hi == 0xFFFFFFFFu) ? 0.0f : __uint_as_float(hi);
      idx4[j] = static_cast<int32_t>(static_cast<uint32_t>(pv & 0xFFFFFFFFu));
    }
    const int64_t o = img_base + int64_t(y) * W + x;
    if (vec_ok) { // x % 4 == 0 and W % 4 == 0 -> x+3 < W and 16-byte aligned
      *reinterpret_cast<int4*>(index_img + o) = make_int4(idx4[0], idx4[1], idx4[2], idx4[3]);
      *reinterpret_cast<float4*>(depth_img + o) = make_float4(dep4[0], dep4[1], dep4[2], dep4[3]);
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

template <typename T>
int rasterize_impl(
    const T* v, const int32_t* vi, int64_t N, int64_t V, int64_t F, int64_t vi_sN, int64_t H,
    int64_t W, float* depth_img, int32_t* index_img, void* workspace, size_t workspace_bytes,
    hipStream_t stream) {
  const BinLayout L = make_layout(N, F, H, W);
  if (workspace_bytes < L.total_bytes) return DRTK_ERR_WORKSPACE_TOO_SMALL;
  if (N * H * W == 0) return DRTK_OK;
  char* ws = static_cast<char*>(workspace);
  auto* tile_count = reinterpret_cast<int32_t*>(ws + L.off_count);
  auto* tile_cursor = reinterpret_cast<int32_t*>(ws + L.off_cursor);
  auto* big_count = reinterpret_cast<int32_t*>(ws + L.off_big_count);
  auto* tile_offset = reinterpret_cast<int32_t*>(ws + L.off_offset);
  auto* tri_range = reinterpret_cast<uint2*>(ws + L.off_range);
  auto* big_list = reinterpret_cast<int32_t*>(ws + L.off_big_list);
  auto* pairs = reinterpret_cast<int32_t*>(ws + L.off_pairs);

  if (hipMemsetAsync(ws, 0, L.zero_bytes, stream) != hipSuccess) return DRTK_ERR_LAUNCH;
  const int64_t total = N * F;
  if (total > 0) {
    const unsigned blocks = static_cast<unsigned>(ceil_div(total, kBlock));
    hipLaunchKernelGGL(
        bin_count_kernel<T>, dim3(blocks), dim3(kBlock), 0, stream, v, vi, total, (int)F, V, vi_sN,
        (int)H, (int)W, L.tile_shift, L.tiles_x, (int)L.tiles_per_view, tile_count, big_count,
        big_list, tri_range);
    DRTK_RETURN_IF_LAUNCH_FAILED();
  }
  hipLaunchKernelGGL(
      bin_scan_kernel, dim3(1), dim3(1024), 0, stream, tile_count, tile_offset, (int)L.num_tiles);
  DRTK_RETURN_IF_LAUNCH_FAILED();
  if (total > 0) {
    const unsigned blocks = static_cast<unsigned>(ceil_div(total, kBlock));
    hipLaunchKernelGGL(
        bin_fill_kernel, dim3(blocks), dim3(kBlock), 0, stream, tri_range, total, (int)F,
        L.tiles_x, (int)L.tiles_per_view, tile_offset, tile_cursor, pairs);
    DRTK_RETURN_IF_LAUNCH_FAILED();
  }
  const unsigned tiles = static_cast<unsigned>(L.num_tiles);
  if (L.tile_shift == 6) {
    hipLaunchKernelGGL(
        (tile_raster_kernel<T, 6>), dim3(tiles), dim3(kBlock), 0, stream, v, vi, (int)F, V, vi_sN,
        (int)H, (int)W, L.tiles_x, (int)L.tiles_per_view, tile_offset, pairs, big_count, big_list,
        tri_range, depth_img, index_img);
  } else {
    hipLaunchKernelGGL(
        (tile_raster_kernel<T, 5>), dim3(tiles), dim3(kBlock), 0, stream, v, vi, (int)F, V, vi_sN,
        (int)H, (int)W, L.tiles_x, (int)L.tiles_per_view, tile_offset, pairs, big_count, big_list,
        tri_range, depth_img, index_img);
  }
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

} // namespace
} // namespace drtk_amd

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
  if (V >= 0x10000000LL) return DRTK_ERR_TOO_MANY_VERTICES;                           // :459-462
  if (wireframe) return DRTK_ERR_UNSUPPORTED;
  if (H > 65535LL * 32 || W > 65535LL * 32 || N * F >= (int64_t(1) << 31) / kMaxSmallTiles ||
      N * H * W >= (int64_t(1) << 40))
    return DRTK_ERR_INVALID_ARGUMENT;
  if (N * H * W > 0 && (!depth_img || !index_img || !workspace)) return DRTK_ERR_INVALID_ARGUMENT;
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
