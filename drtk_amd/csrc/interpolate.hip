// interpolate -- barycentric interpolation of per-vertex attributes, forward and backward.
//
// Reference: src/interpolate/interpolate_kernel.cu:38-111 (forward), :113-299 (backward).
// Forward: one lane = 4 adjacent pixels, so index/bary loads and every channel-plane store are
// 16-byte vectors; attribute rows are gathered through L1/L2 as float4 when C % 4 == 0.
// Backward: bary_grad is a per-pixel dot product; the vertex-attribute gradient goes through the
// wave-level run reduction of segscatter.hpp in chunks of 16 channels (3*16 = 48 of the wave's 64
// lanes own one (corner, channel) pair each), replacing the reference's per-channel
// WarpReduce + __syncthreads + atomics loop.
#include "common.hpp"
#include "segscatter.hpp"

namespace drtk_amd {
namespace {


template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
  using type = float4;
};
template <>
struct Vec4<double> {
  using type = double4;
};

// (Two pixels per lane instead of four -- half the registers, twice the waves in flight for a kernel whose waves spend
// four gather round trips each -- was measured slower, 0.652 vs 0.543-0.577 ms: the 8-byte stores cost more than the
// occupancy buys.)
// PREFETCH (VEC = CV = 4, more than one channel chunk): the attribute rows of the NEXT chunk are requested before this
// chunk's products and stores.  A wave's life was four dependent gather round trips (SQ counters: 39 k cycles alive for
// 1.7 k cycles of vector issue); with the next chunk in flight the kernel takes 158 instead of 88 VGPRs -- 3 waves per
// SIMD instead of 5 -- and is still faster: 0.573 -> 0.548-0.560 ms on the bench shape, 2.46 -> 2.18 at 8 x 4096^2,
// 1.46 -> 1.30 at C = 32.  With a single chunk (C = 4) there is nothing to prefetch and only the registers are paid
// (0.216 -> 0.232 ms): that case keeps the plain loop.
// ANYW (VEC = 4): any W >= 4 and any element-aligned placement of index / bary / out -- the lane's four consecutive
// pixels may run over the end of a row (each gets its own coordinates for the background sweep), the accesses need the
// element's alignment only, and the last lane of a view whose pixel count is not a multiple of four goes pixel by pixel
// (round 4; before, such images took one pixel per lane: 1.2x the time at 2048 x 2046).
// (round 6: bounded to 4 waves per SIMD -- 128 registers -- the prefetching instantiation spills 30-38 of its 158)
#ifndef DRTK_INTERP_FWD_DIAG
#define DRTK_INTERP_FWD_DIAG 0 // 1 / 2: diagnostic builds of the prefetching forward kernel (profiles/NOTES.md R6.12); never the product
#endif
#ifndef DRTK_INTERP_FWD_NT
#define DRTK_INTERP_FWD_NT 1 // index / barycentric quads as non-temporal loads: read once (0.524 -> 0.508 ms, four interleaved rounds on one box, round 6)
#endif
template <typename T, int VEC, int CV, bool PREFETCH = false, bool ANYW = false>
__global__ __launch_bounds__(kBlock) void interpolate_kernel(
    const T* __restrict__ attrs, const int32_t* __restrict__ vi,
    const int32_t* __restrict__ index_img, const T* __restrict__ bary_img, int64_t V, int C,
    int64_t vi_sN, int H, int W, T* __restrict__ out, int zero_background, int strip) {
  using V4 = typename Vec4<T>::type;
  const int64_t HW = int64_t(H) * W;
  const int n = blockIdx.y;
  const int64_t pix0 = (int64_t(tile_index(strip)) * kBlock + threadIdx.x) * VEC;
  if (pix0 >= HW) return;
  const T* attrs_n = attrs + int64_t(n) * V * C;
  const int32_t* vi_n = vi + int64_t(n) * vi_sN;
  const int32_t* idx_p = index_img + int64_t(n) * HW + pix0;
  const T* bary_p = bary_img + int64_t(n) * 3 * HW + pix0;
  T* out_p = out + int64_t(n) * C * HW + pix0;

  typedef int32_t IQuad __attribute__((ext_vector_type(4), aligned(4)));
  typedef T TQuad __attribute__((ext_vector_type(4), aligned(sizeof(T))));
  static_assert(!ANYW || VEC == 4, "ANYW is a variant of the four-pixel lane");
  const bool whole = !ANYW || pix0 + 4 <= HW; // all four pixels belong to this view
  int32_t tr[VEC];
  T B0[VEC], B1[VEC], B2[VEC];
  if constexpr (ANYW) {
    if (whole) {
      const IQuad t4 = *reinterpret_cast<const IQuad*>(idx_p);
      tr[0] = t4.x, tr[1] = t4.y, tr[2] = t4.z, tr[3] = t4.w;
      const TQuad a = *reinterpret_cast<const TQuad*>(bary_p);
      const TQuad b = *reinterpret_cast<const TQuad*>(bary_p + HW);
      const TQuad c = *reinterpret_cast<const TQuad*>(bary_p + 2 * HW);
      B0[0] = a.x, B0[1] = a.y, B0[2] = a.z, B0[3] = a.w;
      B1[0] = b.x, B1[1] = b.y, B1[2] = b.z, B1[3] = b.w;
      B2[0] = c.x, B2[1] = c.y, B2[2] = c.z, B2[3] = c.w;
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const bool in = pix0 + j < HW;
        tr[j] = in ? idx_p[j] : -1;
        B0[j] = in ? bary_p[j] : T(0), B1[j] = in ? bary_p[HW + j] : T(0), B2[j] = in ? bary_p[2 * HW + j] : T(0);
      }
    }
  } else if constexpr (VEC == 4) {
#if DRTK_INTERP_FWD_NT
    const NtQuad<int32_t> t4 = nt_load4(idx_p);
    tr[0] = t4.x, tr[1] = t4.y, tr[2] = t4.z, tr[3] = t4.w;
    const NtQuad<T> a = nt_load4(bary_p), b = nt_load4(bary_p + HW), c = nt_load4(bary_p + 2 * HW);
#else
    const int4 t4 = *reinterpret_cast<const int4*>(idx_p);
    tr[0] = t4.x, tr[1] = t4.y, tr[2] = t4.z, tr[3] = t4.w;
    const V4 a = *reinterpret_cast<const V4*>(bary_p);
    const V4 b = *reinterpret_cast<const V4*>(bary_p + HW);
    const V4 c = *reinterpret_cast<const V4*>(bary_p + 2 * HW);
#endif
    B0[0] = a.x, B0[1] = a.y, B0[2] = a.z, B0[3] = a.w;
    B1[0] = b.x, B1[1] = b.y, B1[2] = b.z, B1[3] = b.w;
    B2[0] = c.x, B2[1] = c.y, B2[2] = c.z, B2[3] = c.w;
  } else {
    tr[0] = idx_p[0];
    B0[0] = bary_p[0], B1[0] = bary_p[HW], B2[0] = bary_p[2 * HW];
  }
  const int y = static_cast<int>(pix0 / W);
  const int x0 = static_cast<int>(pix0 - int64_t(y) * W);

  const T* a0[VEC];
  const T* a1[VEC];
  const T* a2[VEC];
  T bgx[VEC], bgy[VEC];
  // corner ids of the lane's pixels: UNCONDITIONAL loads (a background pixel reads face 0 and discards it) so that the
  // VEC x 3 of them form one batch -- under `if (covered)` every pixel's three loads sat in their own exec-masked region
  // with an s_waitcnt behind them: four dependent round trips per lane before the first attribute row was requested
  // (a lane with no covered pixel loads nothing -- with F = 0 there is no face to read --; in a lane that has one, its
  // background pixels read that pixel's triangle)
  int32_t t_any = tr[0];
#pragma unroll
  for (int j = 1; j < VEC; ++j) t_any = max(t_any, tr[j]);
  const bool any_cov = t_any != -1;
  int32_t fid[VEC][3];
#pragma unroll
  for (int j = 0; j < VEC; ++j) fid[j][0] = fid[j][1] = fid[j][2] = 0;
  if (any_cov) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int32_t* face = vi_n + int64_t(tr[j] != -1 ? tr[j] : t_any) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) fid[j][k] = face[k];
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    a0[j] = attrs_n + int64_t(fid[j][0]) * C;
    a1[j] = attrs_n + int64_t(fid[j][1]) * C;
    a2[j] = attrs_n + int64_t(fid[j][2]) * C;
    // "undefined region" sweep (interpolate_kernel.cu:104-108, CPU twin interpolate_kernel_cpu.cpp:99-104)
    int xj = x0 + j, yj = y;
    if constexpr (ANYW) {
      if (xj >= W) xj -= W, yj += 1; // W >= 4: at most one row further
    }
    bgx[j] = (static_cast<T>(xj) * T(2.0) + T(1.0)) / static_cast<T>(W) - T(1.0);
    bgy[j] = (static_cast<T>(yj) * T(2.0) + T(1.0)) / static_cast<T>(H) - T(1.0);
  }
  // one row of four pixels' values to a channel plane
  auto store4 = [&](T* o, T r0, T r1, T r2, T r3) {
    if constexpr (ANYW) {
      if (whole) {
        *reinterpret_cast<TQuad*>(o) = TQuad{r0, r1, r2, r3};
      } else {
        const T r[4] = {r0, r1, r2, r3};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (pix0 + j < HW) o[j] = r[j];
      }
    } else {
      *reinterpret_cast<V4*>(o) = V4{r0, r1, r2, r3}; // (non-temporal stores: 0.534 vs 0.538 ms, round 6 -- nothing)
    }
  };

  if constexpr (PREFETCH) {
    static_assert(CV == 4 && VEC == 4, "the prefetching loop is written for 4 pixels x 4 channels");
    V4 Q[VEC][3], Qn[VEC][3];
    // rows as 32-bit element offsets from attrs_n (V * C < 2^31 on this path, checked by the launcher): 12 registers
    // instead of 24 for the addresses
    uint32_t off[VEC][3];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      off[j][0] = static_cast<uint32_t>(a0[j] - attrs_n);
      off[j][1] = static_cast<uint32_t>(a1[j] - attrs_n);
      off[j][2] = static_cast<uint32_t>(a2[j] - attrs_n);
    }
    auto fetch = [&](int c0, V4 (&q)[VEC][3]) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (tr[j] != -1) {
#if DRTK_INTERP_FWD_DIAG == 2 // (diagnostic build: no attribute gathers -- the stream alone)
#pragma unroll
          for (int k = 0; k < 3; ++k) q[j][k] = V4{B0[j], B1[j], B2[j], T(c0 + k)};
#else
#pragma unroll
          for (int k = 0; k < 3; ++k) q[j][k] = *reinterpret_cast<const V4*>(attrs_n + off[j][k] + c0);
#endif
        }
      }
    };
    fetch(0, Q);
    for (int c0 = 0; c0 < C; c0 += CV) {
      if (c0 + CV < C) fetch(c0 + CV, Qn);
      T r[CV][VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (tr[j] != -1) {
          const T u0[4] = {Q[j][0].x, Q[j][0].y, Q[j][0].z, Q[j][0].w};
          const T u1[4] = {Q[j][1].x, Q[j][1].y, Q[j][1].z, Q[j][1].w};
          const T u2[4] = {Q[j][2].x, Q[j][2].y, Q[j][2].z, Q[j][2].w};
#pragma unroll
          for (int cc = 0; cc < CV; ++cc) r[cc][j] = u0[cc] * B0[j] + u1[cc] * B1[j] + u2[cc] * B2[j];
        } else {
#pragma unroll
          for (int cc = 0; cc < CV; ++cc) r[cc][j] = zero_background ? T(0) : (((c0 + cc) & 1) ? bgy[j] : bgx[j]);
        }
      }
#pragma unroll
      for (int cc = 0; cc < CV; ++cc) {
#if DRTK_INTERP_FWD_DIAG == 1 // (diagnostic build: the planes are computed and not stored -- reads and gathers alone)
        if (r[cc][0] == T(123456.75) && r[cc][3] == T(-98765.25)) store4(out_p + int64_t(c0 + cc) * HW, r[cc][0], r[cc][1], r[cc][2], r[cc][3]);
#else
        store4(out_p + int64_t(c0 + cc) * HW, r[cc][0], r[cc][1], r[cc][2], r[cc][3]);
#endif
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
#pragma unroll
        for (int k = 0; k < 3; ++k) Q[j][k] = Qn[j][k];
      }
    }
    return;
  }
  for (int c0 = 0; c0 < C; c0 += CV) {
    T r[CV][VEC];
    // the block's attribute rows of all the lane's pixels first (unconditional: a background pixel reads vertex 0), the
    // products after: one round trip per channel block instead of one per pixel
    T u0[VEC][CV], u1[VEC][CV], u2[VEC][CV];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (!any_cov) break;
      if constexpr (CV == 4) {
        const V4 q0 = *reinterpret_cast<const V4*>(a0[j] + c0);
        const V4 q1 = *reinterpret_cast<const V4*>(a1[j] + c0);
        const V4 q2 = *reinterpret_cast<const V4*>(a2[j] + c0);
        u0[j][0] = q0.x, u0[j][1] = q0.y, u0[j][2] = q0.z, u0[j][3] = q0.w;
        u1[j][0] = q1.x, u1[j][1] = q1.y, u1[j][2] = q1.z, u1[j][3] = q1.w;
        u2[j][0] = q2.x, u2[j][1] = q2.y, u2[j][2] = q2.z, u2[j][3] = q2.w;
      } else {
        u0[j][0] = a0[j][c0], u1[j][0] = a1[j][c0], u2[j][0] = a2[j][c0];
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (tr[j] != -1) {
#pragma unroll
        for (int cc = 0; cc < CV; ++cc) r[cc][j] = u0[j][cc] * B0[j] + u1[j][cc] * B1[j] + u2[j][cc] * B2[j];
      } else {
#pragma unroll
        for (int cc = 0; cc < CV; ++cc) r[cc][j] = zero_background ? T(0) : (((c0 + cc) & 1) ? bgy[j] : bgx[j]);
      }
    }
#pragma unroll
    for (int cc = 0; cc < CV; ++cc) {
      T* o = out_p + int64_t(c0 + cc) * HW;
      if constexpr (VEC == 4) {
        store4(o, r[cc][0], r[cc][1], r[cc][2], r[cc][3]);
      } else {
        o[0] = r[cc][0];
      }
    }
  }
}

// Backward.  A workgroup owns a 64 x 16 pixel tile; each of its 4 waves walks 4 adjacent rows of 64
// pixels (lane = pixel in phase 1) with a wave-private vertex table.  Channels are processed in
// chunks of 16 (outer loop).  No workgroup barrier is needed anywhere.
// CHUNK = channels staged per round.  CHUNK = 4 (C <= 4, e.g. the v_pix route of edge_grad) keeps
// a wave-private vertex table; CHUNK = 16 sends each run's sums straight to global memory -- its 16
// lanes per corner form one 64-byte request, and the smaller LDS footprint buys the occupancy
// that hides phase 1's gather latency (measured faster than the table at C = 16).
template <typename T, bool HAS_VERT, bool HAS_BARY, int CV, int CHUNK>
__global__ __launch_bounds__(kBlock) void interpolate_backward_kernel(
    const T* __restrict__ grad_out, const T* __restrict__ attrs, const int32_t* __restrict__ vi,
    const int32_t* __restrict__ index_img, const T* __restrict__ bary_img, int64_t V, int C,
    int64_t vi_sN, int H, int W, int tiles_x, T* __restrict__ attr_grad, T* __restrict__ bary_grad,
    int dbg, int strip) {
  using V4 = typename Vec4<T>::type;
  constexpr int kWaves = kBlock / kWave;
  constexpr int kPasses = kTileRows / kWaves;
  constexpr bool TABLE = HAS_VERT && CHUNK <= 4;
  __shared__ __attribute__((aligned(16))) T s_g[HAS_VERT ? kWaves : 1][HAS_VERT ? CHUNK * kRunPad : 4];
  __shared__ __attribute__((aligned(16))) T s_b[HAS_VERT ? kWaves : 1][HAS_VERT ? 3 * kRunPad : 4];
  __shared__ int32_t s_vid[HAS_VERT ? kWaves : 1][HAS_VERT ? 3 * kRunPad : 1];
  __shared__ int32_t s_slot[HAS_VERT ? kWaves : 1][HAS_VERT ? 3 * kRunPad : 1];
  __shared__ int32_t t_keys[TABLE ? kWaves : 1][TABLE ? kTableSlots : 1];
  __shared__ TableAcc t_vals[TABLE ? kWaves : 1][TABLE ? kTableSlots * CHUNK : 1];

  const int64_t HW = int64_t(H) * W;
  const int n = blockIdx.y;
  const int tile = tile_index(strip);
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int wave = threadIdx.x / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int x = txi * kWave + lane;
  const T* attrs_n = attrs + int64_t(n) * V * C;
  const int32_t* vi_n = vi + int64_t(n) * vi_sN;
  T* attr_grad_n = HAS_VERT ? attr_grad + int64_t(n) * V * C : nullptr;

  if constexpr (TABLE) table_init(t_keys[wave]);

  for (int c0 = 0; c0 < C; c0 += CHUNK) {
    const int CC = min(CHUNK, C - c0);
    if constexpr (TABLE) {
      wave_lds_sync(); // previous chunk flushed, keys initialised
      for (int i = lane; i < kTableSlots * CHUNK; i += kWave) t_vals[wave][i] = 0;
      wave_lds_sync();
    }
    // software pipeline over the 4 row passes: the index of pass p+1 is requested at the top of
    // pass p and its triangle's vertex ids right before phase 2, so both dependent gathers fly
    // under the current pass instead of in front of the next one.
    auto load_tr = [&](int ps) -> int32_t {
      const int yy = tyi * kTileRows + wave * kPasses + ps;
      return (x < W && yy < H) ? index_img[int64_t(n) * HW + int64_t(yy) * W + x] : -1;
    };
    int32_t tr_next = load_tr(0);
    int32_t vn0 = 0, vn1 = 0, vn2 = 0;
    if (tr_next != -1) {
      const int32_t* face = vi_n + int64_t(tr_next) * 3;
      vn0 = face[0], vn1 = face[1], vn2 = face[2];
    }
#pragma unroll 1
    for (int ps = 0; ps < kPasses; ++ps) {
      const int y = tyi * kTileRows + wave * kPasses + ps;
      const bool in_range = x < W && y < H;
      const int64_t pix = int64_t(y) * W + x;
      const int32_t tr = tr_next;
      const bool covered = tr != -1;
      const int32_t vid0 = vn0, vid1 = vn1, vid2 = vn2;
      if (ps + 1 < kPasses) tr_next = load_tr(ps + 1);
      unsigned long long heads = 0, cov = 0;
      if constexpr (HAS_VERT) {
        T B0 = T(0), B1 = T(0), B2 = T(0);
        if (covered) {
          const T* bp = bary_img + int64_t(n) * 3 * HW + pix;
          B0 = bp[0], B1 = bp[HW], B2 = bp[2 * HW];
        }
        s_b[wave][0 * kRunPad + lane] = B0;
        s_b[wave][1 * kRunPad + lane] = B1;
        s_b[wave][2 * kRunPad + lane] = B2;
        s_vid[wave][0 * kRunPad + lane] = vid0;
        s_vid[wave][1 * kRunPad + lane] = vid1;
        s_vid[wave][2 * kRunPad + lane] = vid2;
        const bool use_table = TABLE && covered && !DRTK_DBG(dbg, 2) && vid0 != vid1 && vid0 != vid2 && vid1 != vid2;
        s_slot[wave][0 * kRunPad + lane] = use_table ? table_slot(t_keys[wave], vid0) : -1;
        s_slot[wave][1 * kRunPad + lane] = use_table ? table_slot(t_keys[wave], vid1) : -1;
        s_slot[wave][2 * kRunPad + lane] = use_table ? table_slot(t_keys[wave], vid2) : -1;
        run_masks(tr, heads, cov);
      }
      T bg0 = T(0), bg1 = T(0), bg2 = T(0);
      if constexpr (HAS_VERT) {
        if (!covered) { // phase 2 sums whole runs of the staging rows: uncovered pixels must read as 0
          for (int cb = 0; cb < CC; ++cb) s_g[wave][cb * kRunPad + lane] = T(0);
        }
      }
      if (covered) {
        const T* go_p = grad_out + int64_t(n) * C * HW + pix;
        const T* a0 = attrs_n + int64_t(vid0) * C;
        const T* a1 = attrs_n + int64_t(vid1) * C;
        const T* a2 = attrs_n + int64_t(vid2) * C;
        for (int cb = 0; cb < CC; cb += CV) {
          T g[CV];
#pragma unroll
          for (int cc = 0; cc < CV; ++cc) g[cc] = DRTK_DBG(dbg, 4) ? T(1) : go_p[int64_t(c0 + cb + cc) * HW];
          if constexpr (HAS_VERT) {
#pragma unroll
            for (int cc = 0; cc < CV; ++cc) s_g[wave][(cb + cc) * kRunPad + lane] = g[cc];
          }
          if (HAS_BARY && !DRTK_DBG(dbg, 8)) {
            T u0[CV], u1[CV], u2[CV];
            if constexpr (CV == 4) {
              const V4 q0 = *reinterpret_cast<const V4*>(a0 + c0 + cb);
              const V4 q1 = *reinterpret_cast<const V4*>(a1 + c0 + cb);
              const V4 q2 = *reinterpret_cast<const V4*>(a2 + c0 + cb);
              u0[0] = q0.x, u0[1] = q0.y, u0[2] = q0.z, u0[3] = q0.w;
              u1[0] = q1.x, u1[1] = q1.y, u1[2] = q1.z, u1[3] = q1.w;
              u2[0] = q2.x, u2[1] = q2.y, u2[2] = q2.z, u2[3] = q2.w;
            } else {
              u0[0] = a0[c0 + cb], u1[0] = a1[c0 + cb], u2[0] = a2[c0 + cb];
            }
#pragma unroll
            for (int cc = 0; cc < CV; ++cc) { // interpolate_kernel.cu:238-246 accumulation order
              bg0 += g[cc] * u0[cc];
              bg1 += g[cc] * u1[cc];
              bg2 += g[cc] * u2[cc];
            }
          }
        }
      }
      if constexpr (HAS_BARY) {
        if (in_range) { // channel chunks accumulate in ascending order, like the reference's loop
          T* bgp = bary_grad + int64_t(n) * 3 * HW + pix;
          if (c0 == 0) {
            bgp[0] = bg0, bgp[HW] = bg1, bgp[2 * HW] = bg2;
          } else if (covered) {
            bgp[0] += bg0, bgp[HW] += bg1, bgp[2 * HW] += bg2;
          }
        }
      }
      if (ps + 1 < kPasses) {
        vn0 = vn1 = vn2 = 0;
        if (tr_next != -1) {
          const int32_t* face = vi_n + int64_t(tr_next) * 3;
          vn0 = face[0], vn1 = face[1], vn2 = face[2];
        }
      }
      if constexpr (HAS_VERT) {
        wave_lds_sync();
        if (cov != 0 && !DRTK_DBG(dbg, 1)) {
          const T* sg = s_g[wave];
          const T* sb = s_b[wave];
          scatter_runs<T>(
              heads, cov, s_slot[wave], s_vid[wave], 3 * CC, CC, t_vals[wave], CHUNK, attr_grad_n, C, c0,
              [=](int k, int c, int g4, T* x) {
                if (DRTK_DBG(dbg, 64)) {
                  x[0] = x[1] = x[2] = x[3] = T(1);
                  return;
                }
                const V4 a = *reinterpret_cast<const V4*>(sg + c * kRunPad + 4 * g4);
                const V4 b = *reinterpret_cast<const V4*>(sb + k * kRunPad + 4 * g4);
                x[0] = a.x * b.x, x[1] = a.y * b.y, x[2] = a.z * b.z, x[3] = a.w * b.w;
              }, dbg);
        }
        wave_lds_sync();
      }
    }
    if constexpr (TABLE) {
      if (!DRTK_DBG(dbg, 16)) table_flush<T>(t_keys[wave], t_vals[wave], CHUNK, CC, attr_grad_n, C, c0);
    }
  }
}

// Backward for C <= 4 (the v_pix route of edge_grad_estimator with a hook, uv attributes, RGB attributes): lane = pixel
// like render backward; the 3 * CN products grad_out[c] * bary[k] stay in registers, a segmented scan over the 16-lane
// rows leaves each run's sums in its last lane (segscatter.hpp: run_sums_rows16), which alone updates the wave's vertex
// table (double accumulators) -- no LDS staging, no lane flip, no barrier in the loop.  The bary gradient is the
// reference's per-pixel dot products (interpolate_kernel.cu:238-246, channels ascending).
template <typename T, bool HAS_VERT, bool HAS_BARY, int CN>
__global__ __launch_bounds__(kBlock) void interpolate_backward_small_kernel(
    const T* __restrict__ grad_out, const T* __restrict__ attrs, const int32_t* __restrict__ vi,
    const int32_t* __restrict__ index_img, const T* __restrict__ bary_img, int64_t V, int64_t vi_sN, int H, int W,
    int tiles_x, T* __restrict__ attr_grad, T* __restrict__ bary_grad, int strip) {
  constexpr int kWaves = kBlock / kWave;
  constexpr int kPasses = kTileRows / kWaves;
  constexpr int kSlots = kTableSlots;
  constexpr int kStride = CN <= 4 ? 4 : CN <= 8 ? 8 : 12; // table entries per vertex
  static_assert(CN >= 1 && CN <= 12, "channels in registers");
  __shared__ int32_t t_keys[HAS_VERT ? kWaves : 1][HAS_VERT ? kSlots : 1];
  __shared__ TableAcc t_vals[HAS_VERT ? kWaves : 1][HAS_VERT ? kSlots * kStride : 1];

  const int64_t HW = int64_t(H) * W;
  const int n = blockIdx.y;
  const int tile = tile_index(strip);
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int wave = threadIdx.x / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int x = txi * kWave + lane;
  const T* attrs_n = attrs + int64_t(n) * V * CN;
  const int32_t* vi_n = vi + int64_t(n) * vi_sN;
  T* attr_grad_n = HAS_VERT ? attr_grad + int64_t(n) * V * CN : nullptr;
  const T* go_n = grad_out + int64_t(n) * CN * HW;
  const T* bary_n = bary_img + int64_t(n) * 3 * HW;

  if constexpr (HAS_VERT) {
    table_init<kSlots>(t_keys[wave]);
    for (int i = lane; i < kSlots * kStride; i += kWave) t_vals[wave][i] = 0;
    wave_lds_sync();
  }
  auto load_tr = [&](int ps) -> int32_t {
    const int yy = tyi * kTileRows + wave * kPasses + ps;
    return (x < W && yy < H) ? index_img[int64_t(n) * HW + int64_t(yy) * W + x] : -1;
  };
  auto load_face = [&](int32_t t, int32_t (&f)[3]) {
    f[0] = f[1] = f[2] = 0;
    if (t != -1) {
      const int32_t* face = vi_n + int64_t(t) * 3;
      f[0] = face[0], f[1] = face[1], f[2] = face[2];
    }
  };
  int32_t tr_next = load_tr(0);
  int32_t vn[3];
  load_face(tr_next, vn);
#pragma unroll 1
  for (int ps = 0; ps < kPasses; ++ps) {
    const int y = tyi * kTileRows + wave * kPasses + ps;
    const bool in_range = x < W && y < H;
    const int64_t pix = int64_t(y) * W + x;
    const int32_t tr = tr_next;
    const bool covered = tr != -1;
    const int32_t cur[3] = {vn[0], vn[1], vn[2]};
    if (ps + 1 < kPasses) tr_next = load_tr(ps + 1);
    T g[CN], B[3] = {T(0), T(0), T(0)};
#pragma unroll
    for (int c = 0; c < CN; ++c) g[c] = T(0);
    if (covered) {
#pragma unroll
      for (int c = 0; c < CN; ++c) g[c] = go_n[int64_t(c) * HW + pix];
      if constexpr (HAS_VERT) {
#pragma unroll
        for (int k = 0; k < 3; ++k) B[k] = bary_n[int64_t(k) * HW + pix];
      }
    }
    if constexpr (HAS_BARY) {
      T bg[3] = {T(0), T(0), T(0)};
      if (covered) {
#pragma unroll
        for (int c = 0; c < CN; ++c) { // interpolate_kernel.cu:238-246 accumulation order
#pragma unroll
          for (int k = 0; k < 3; ++k) bg[k] += g[c] * attrs_n[int64_t(cur[k]) * CN + c];
        }
      }
      if (in_range) {
        T* bgp = bary_grad + int64_t(n) * 3 * HW + pix;
        bgp[0] = bg[0], bgp[HW] = bg[1], bgp[2 * HW] = bg[2];
      }
    }
    if (ps + 1 < kPasses) load_face(tr_next, vn);
    if constexpr (HAS_VERT) {
      int dist;
      bool tail;
      run_rows16(tr, dist, tail);
      if (__ballot(covered) != 0) {
        T p[3 * CN];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
          for (int c = 0; c < CN; ++c) p[k * CN + c] = g[c] * B[k];
        }
        run_sums_rows16<T, 3 * CN>(p, dist);
        if (tail && covered) table_add<T, 3, CN, kSlots>(t_keys[wave], t_vals[wave], kStride, cur, p, attr_grad_n, CN);
      }
    }
  }
  if constexpr (HAS_VERT) {
    wave_lds_sync();
    table_flush<T, TableAcc, kSlots>(t_keys[wave], t_vals[wave], kStride, CN, attr_grad_n, CN, 0);
  }
}

// Backward, wide-channel path: the vertex gradient (with or without the bary gradient) for any C % 4 == 0, C >= 8,
// f32, 16-byte aligned attribute rows.  (Round 4: rounds 1-3 had this pipeline for C % 16 == 0 with both gradients only;
// every other shape fell back to the generic kernel above.)
//
// Same tiling as interpolate_backward_kernel -- a workgroup owns 64 x 16 pixels, each wave walks 4 adjacent rows of 64
// pixels, lane = pixel in phase 1, lane = (corner, channel) in phase 2 -- but written so that no memory round trip is
// ever waited for in isolation (the generic kernel's phase 1 is a chain of up to nine dependent global waits per row --
// index, corners, bary, then go/attribute loads per 4-channel block -- and PMC shows its waves parked on s_waitcnt
// 72 % of their life):
//   * ROW-outer, channel-chunk-inner: index, corner ids and barycentrics of a row are fetched and staged ONCE, whatever
//     C is, and the bary gradient of a pixel accumulates in registers over all chunks in the reference's channel order
//     (interpolate_kernel.cu:238-246) and is stored once -- a chunk-outer loop re-reads 16 B/px and read-modify-writes
//     24 B/px of bary_grad per extra chunk, i.e. costs more per channel at C = 64 than at C = 16;
//   * channels go in chunks of 16 with a last chunk of 4, 8 or 12 (scatter_runs folds 12 / 24 (corner, channel) pairs
//     into 4 / 2 pixel slices of the wave, so the lanes stay busy);
//   * every global load of a step (<= 16 grad_out planes, first half of the attribute rows) is issued in one batch, and
//     the NEXT step's grad_out planes (next chunk of this row, or first chunk + barycentrics of the next row) are
//     requested before the current step's phase 2, so they land while the wave is busy in LDS;
//   * the per-pixel bary-gradient dot products run after phase 2 and read grad_out back from the LDS staging rows, which
//     frees the registers for the prefetch.
//   * TABLE (round 4): the run sums do not go to global memory run by run but into a table of the WORKGROUP in LDS --
//     2^log2_slots vertices x C double accumulators, filled by all four waves with ds_add_f64, slots looked up once per
//     row by the lanes that start a run -- and the tile's 64 x 16 pixels cost one global atomic per vertex and channel at
//     the end.  Per-run atomics are what the counters show as write traffic beyond bary_grad (864 MB against 403 MB at
//     the bench shape), and on rows that are not a multiple of 64 bytes (C = 12, 24) they are several times slower still.
//     A vertex that finds no slot keeps the direct atomics.
#ifndef DRTK_INTERP_WIDE_ONLY
#define DRTK_INTERP_WIDE_ONLY false
#endif
#ifndef DRTK_INTERP_ANYC_QA
#define DRTK_INTERP_ANYC_QA 1
#endif
#ifndef DRTK_INTERP_ANYC_SPLIT
#define DRTK_INTERP_ANYC_SPLIT 1
#endif
#ifndef DRTK_INTERP_ANYC_WAVES
#define DRTK_INTERP_ANYC_WAVES 4
#endif
#ifndef DRTK_INTERP_CH8
#define DRTK_INTERP_CH8 1
#endif
#ifndef DRTK_INTERP_CH12
#define DRTK_INTERP_CH12 1
#endif
#ifndef DRTK_INTERP_CH12_MAXC
#define DRTK_INTERP_CH12_MAXC 12
#endif
#ifndef DRTK_INTERP_CH8_ANYC
#define DRTK_INTERP_CH8_ANYC 0 // 9 <= C <= 15, C % 4 != 0: chunks of 8 + a tail (4 waves per SIMD) instead of one chunk (3)
#endif
#ifndef DRTK_INTERP_CH8_WAVES
#define DRTK_INTERP_CH8_WAVES 4 // (with the bary gradient 104 VGPRs; 4-12 spilled at 5)
#endif
#ifndef DRTK_INTERP_F64_WAVES
#define DRTK_INTERP_F64_WAVES 3 // waves per SIMD the double instantiations are compiled for (144-152 VGPRs; 32-50 spilled at 4)
#endif
//   * ANYC (round 5): channel counts that are not a multiple of four (C >= 5: position + normal = 6, RGB + k features) and
//     attribute tensors that are only element-aligned.  The last chunk is then 1 ... 15 channels: its grad_out planes are
//     still fetched in groups of four, the planes beyond C through a descriptor of ZERO records (they read 0.0 without
//     touching memory); the attribute rows' last, partial group of four is read as the four channels that END the row
//     (window shifted back by 4 - r: never a byte beyond the row, element alignment only) and its dot products skip the
//     channels the previous group already took; phase 2 is channel-count agnostic (J = 3 cc lanes).  Rows of such a C
//     are never whole 64-byte segments, so the run sums go through the workgroup's vertex table.
//   * double (round 5): the same pipeline in chunks of CH = 8 channels (the registers of 16 floats), 8-byte buffer loads.
#ifndef DRTK_INTERP_BWD_AUX
#define DRTK_INTERP_BWD_AUX 2 // cache policy of the grad_out / bary plane loads: 2 = nt (each plane is read once per launch: 0.600 -> 0.581 ms
                               // at C = 16, same box, round 6; 0 = default policy)
#endif
template <typename T, bool HAS_BARY, bool TABLE, int CH, bool ANYC>
__global__ __launch_bounds__(kBlock, (sizeof(T) == 8 ? DRTK_INTERP_F64_WAVES : CH == 8 || CH == 12 ? DRTK_INTERP_CH8_WAVES : ANYC && HAS_BARY ? DRTK_INTERP_ANYC_WAVES : 4)) void interpolate_backward_wide_kernel(
    const T* __restrict__ grad_out, const T* __restrict__ attrs, const int32_t* __restrict__ vi,
    const int32_t* __restrict__ index_img, const T* __restrict__ bary_img, int64_t V, int C,
    int64_t vi_sN, int H, int W, int tiles_x, T* __restrict__ attr_grad, T* __restrict__ bary_grad,
    int dbg, int strip, int log2_slots, int CG) { // CG: elements between the rows of attr_grad (C, or the padded pitch of the workspace)
  using V4 = typename Vec4<T>::type;
  static_assert(CH % 4 == 0 && CH >= 8 && CH <= 16, "chunks of 8 or 16 channels");
  constexpr int kWaves = kBlock / kWave;
  constexpr int kPasses = kTileRows / kWaves;
  static_assert(kPasses == 4, "row pipeline below is written for 4 rows per wave");
  __shared__ __attribute__((aligned(16))) T s_g[kWaves][CH * kRunPad];
  __shared__ __attribute__((aligned(16))) T s_b[kWaves][3 * kRunPad];
  __shared__ int32_t s_vid[kWaves][3 * kRunPad];
  __shared__ int32_t s_slot[TABLE ? kWaves : 1][TABLE ? 3 * kRunPad : 1];
  extern __shared__ __attribute__((aligned(16))) unsigned char s_table_raw[]; // TABLE: [slots][C] doubles, then [slots] keys
  TableAcc* const t_vals = reinterpret_cast<TableAcc*>(s_table_raw);
  int32_t* const t_keys = reinterpret_cast<int32_t*>(t_vals + (TABLE ? (size_t(C) << log2_slots) : 0));

  const int64_t HW = int64_t(H) * W;
  const int n = blockIdx.y;
  const int tile = tile_index(strip);
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int wave = threadIdx.x / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int x = txi * kWave + lane;
  const int y0 = tyi * kTileRows + wave * kPasses;
  const T* attrs_n = attrs + int64_t(n) * V * C;
  const int32_t* vi_n = vi + int64_t(n) * vi_sN;
  T* attr_grad_n = attr_grad + int64_t(n) * V * CG;
  const T* go_n = grad_out + int64_t(n) * C * HW;
  const T* bary_n = bary_img + int64_t(n) * 3 * HW;
  T* bgrad_n = HAS_BARY ? bary_grad + int64_t(n) * 3 * HW : nullptr;
  if constexpr (TABLE) {
    for (int i = threadIdx.x; i < (C << log2_slots); i += kBlock) t_vals[i] = 0;
    for (int i = threadIdx.x; i < (1 << log2_slots); i += kBlock) t_keys[i] = -1;
  }

  auto load_tr = [&](int ps) -> int32_t {
    const int y = y0 + ps;
    return (ps < kPasses && x < W && y < H) ? index_img[int64_t(n) * HW + int64_t(y) * W + x] : -1;
  };
  auto load_face = [&](int32_t t, int32_t& a0, int32_t& a1, int32_t& a2) {
    a0 = a1 = a2 = 0;
    if (t != -1) {
      const int32_t* face = vi_n + int64_t(t) * 3;
      a0 = face[0], a1 = face[1], a2 = face[2];
    }
  };

  // A wave whose 64 x 4 pixels are all background (43 % of them at the benchmark's coverage) has nothing to
  // scatter: it zeroes its part of bary_grad and leaves (TABLE: stays for the barriers of a tile that has foreground).
  bool wave_fg;
  {
    int32_t t4[kPasses];
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) t4[ps] = load_tr(ps);
    const bool any_fg = (t4[0] & t4[1] & t4[2] & t4[3]) != -1;
    wave_fg = __ballot(any_fg) != 0;
    bool tile_fg = wave_fg;
    if constexpr (TABLE) tile_fg = __syncthreads_or(wave_fg) != 0; // (also orders the table's zero-fill before its first use)
    if (!wave_fg) {
      if constexpr (HAS_BARY) {
        int yz = y0; // (recomputed from a laundered copy: kept from the index loads above, the row predicates were spilled)
        asm volatile("" : "+v"(yz));
#pragma unroll
        for (int ps = 0; ps < kPasses; ++ps) {
          const int y = yz + ps;
          if (x < W && y < H) {
            T* bgp = bgrad_n + int64_t(y) * W + x;
            bgp[0] = T(0), bgp[HW] = T(0), bgp[2 * HW] = T(0);
          }
        }
      }
      if (!TABLE || !tile_fg) return;
    }
  }

  if (wave_fg) {
  T G[CH], B[3];
  // grad_out planes c0 .. c0 + cc - 1 of row ps (cc = 4, 8, 12 or 16: wave-uniform); zeros where uncovered
  // Loads of a row's operands: UNCONDITIONAL per lane (the offset of a lane off the canvas is clamped; what an uncovered
  // pixel reads is never used: phase 2 flushes covered runs only and restarts its sums at every run head, the bary
  // gradient is computed for covered pixels only), skipped for a row with no covered pixel at all (wave-uniform).  A
  // per-lane `covered ? load : 0` puts every group of loads into its own exec-masked region with its own s_waitcnt --
  // the prefetch became four dependent round trips (0.65 -> 0.72 ms at C = 16 when the tail chunks were introduced).
  // Plane bases are wave-uniform and the in-plane offset is a 32-bit lane value (H W < 2^31): SGPR base + VGPR offset.
  // The 16 + 3 plane loads of a row go through BUFFER loads: one descriptor per plane (wave-uniform: SGPRs) and ONE
  // 32-bit byte offset register for all of them (H W < 2^30 on this path, checked by the launcher; a lane off the canvas
  // is out of the plane's range and reads 0).  As global loads the compiler built a 64-bit address per load and, at the
  // register bound, recycled registers that were still load destinations as address temporaries -- an s_waitcnt
  // vmcnt(0) in the middle of the batch.
  const uint32_t plane_bytes = static_cast<uint32_t>(HW * int64_t(sizeof(T)));
  auto plane_load = [&](const T* plane, uint32_t byte_offset, bool exists) -> T { // `exists`: wave-uniform (ANYC: a plane < C)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(plane), 0, exists ? static_cast<int>(plane_bytes) : 0, 0x00020000);
    if constexpr (sizeof(T) == 4) {
      return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, byte_offset, 0, DRTK_INTERP_BWD_AUX));
    } else {
      return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(r, byte_offset, 0, DRTK_INTERP_BWD_AUX));
    }
  };
  auto row_offset = [&](int ps) -> uint32_t { // beyond the plane for a lane off the canvas
    return (x < W && y0 + ps < H) ? static_cast<uint32_t>((y0 + ps) * W + x) * static_cast<uint32_t>(sizeof(T)) : plane_bytes;
  };
  // one call site, one offset register for grad_out and barycentrics (with an offset of its own the bary loads got a
  // freshly written register, and the compiler waited for the grad_out loads before writing it)
  // (plane bases by repeated addition of H W: written as go_n + (c0 + c) H W every plane cost a 64-bit scalar multiply,
  // eight scalar instructions -- the kernel issued more scalar than vector instructions, 152 M against 126 M per launch)
  auto load_row = [&](int ps, int c0, int cc, bool row_any, bool with_bary) {
    if (!row_any) return;
    const uint32_t bo = row_offset(ps);
    const T* plane = go_n + int64_t(c0) * HW;
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) {
      if (4 * q < cc) {
#pragma unroll
        for (int c = 4 * q; c < 4 * q + 4; ++c) {
          G[c] = plane_load(plane, bo, !ANYC || c < cc);
          plane += HW;
        }
      }
    }
    if (with_bary) {
      const T* bplane = bary_n;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        B[k] = plane_load(bplane, bo, true);
        bplane += HW;
      }
    }
  };
  // rotating state: row ps (tr, v*), row ps+1 (tr_n, vn*), row ps+2 (tr_nn)
  int32_t tr = load_tr(0), tr_n = load_tr(1), tr_nn = -1;
  int32_t v0, v1, v2, vn0 = 0, vn1 = 0, vn2 = 0;
  load_face(tr, v0, v1, v2);
#pragma unroll
  for (int c = 0; c < CH; ++c) G[c] = T(0);
  B[0] = B[1] = B[2] = T(0);
  load_row(0, 0, min(CH, C), __ballot(tr != -1) != 0, true);
  // ONE loop over (row, chunk) steps, rows outer: as two nested loops the compiler kept ~45 more values live across the
  // inner one (173 instead of 128 VGPRs: spills at 4 waves per SIMD)
  unsigned long long heads = 0, cov = 0;
  T bg0 = T(0), bg1 = T(0), bg2 = T(0);
  int ps = 0, c0 = 0;
#pragma unroll 1
  for (;;) {
    const bool covered = tr != -1;
    if (c0 == 0) {
      // 0. the two CONDITIONAL fetches of the rows ahead come first: `cond ? load : default` compiles to a branch around
      //    the load with an s_waitcnt vmcnt(0) inside it; issued after the next row's 19 loads (as they used to be) that
      //    wait held the wave until all of them had landed, before the phase 2 they were meant to fly under (0.659 ->
      //    0.650 ms; making the two fetches unconditional instead cost registers and was slower, 0.675)
      if (ps + 1 < kPasses) {
        load_face(tr_n, vn0, vn1, vn2);
        tr_nn = load_tr(ps + 2);
      }
      // 1. what the row's chunks share: barycentrics, corner ids, run structure
#pragma unroll
      for (int k = 0; k < 3; ++k) s_b[wave][k * kRunPad + lane] = B[k];
      s_vid[wave][0 * kRunPad + lane] = v0;
      s_vid[wave][1 * kRunPad + lane] = v1;
      s_vid[wave][2 * kRunPad + lane] = v2;
      run_masks(tr, heads, cov);
      bg0 = bg1 = bg2 = T(0);
      if constexpr (TABLE) {
        // table slots of the row's runs, looked up by the lanes that START a covered run (phase 2 reads them there)
        int s0 = -1, s1 = -1, s2 = -1;
        if (covered && ((heads >> lane) & 1ull) && !DRTK_DBG(dbg, 2)) {
          s0 = table_slot_rt(t_keys, v0, log2_slots);
          s1 = table_slot_rt(t_keys, v1, log2_slots);
          s2 = table_slot_rt(t_keys, v2, log2_slots);
        }
        s_slot[wave][0 * kRunPad + lane] = s0;
        s_slot[wave][1 * kRunPad + lane] = s1;
        s_slot[wave][2 * kRunPad + lane] = s2;
      }
    }
    const int cc = min(CH, C - c0); // 4, 8, 12 or 16 (ANYC: anything from 1)
    const bool last_chunk = c0 + CH >= C;
    // 2. stage this chunk's grad_out
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) {
      if (4 * q < cc) {
#pragma unroll
        for (int c = 4 * q; c < 4 * q + 4; ++c) s_g[wave][c * kRunPad + lane] = G[c];
      }
    }
    // 3. first half of the attribute rows of this row's triangles (consumed after phase 2)
    const T* a0 = attrs_n + int64_t(v0) * C + c0;
    const T* a1 = attrs_n + int64_t(v1) * C + c0;
    const T* a2 = attrs_n + int64_t(v2) * C + c0;
#ifndef DRTK_INTERP_QA
#define DRTK_INTERP_QA 1
#endif
    // float4 per corner requested before / after phase 2 (ANYC + bary gradient: the window arithmetic of the partial group spills
    // 4-10 registers at 4 waves per SIMD, and 2-4 with every request moved behind phase 2; rounds 5a-b ran it at 3 waves, 138
    // registers.  With the LAST group of a 16-channel chunk requested after the products of the others -- DRTK_INTERP_ANYC_SPLIT,
    // below -- it is 124 registers and runs at 4: C = 13 ... 15 0.855 -> 0.745 ms, 17 1.19 -> 1.05, 21 1.27 -> 1.12, 37 1.90 -> 1.65)
    constexpr int QA = ANYC ? DRTK_INTERP_ANYC_QA : DRTK_INTERP_QA, QD = CH / 4 - QA;
    // group q of a row's chunk: channels 4q .. 4q+3, or (ANYC, the chunk's partial last group) the four that end the row
    typedef T TQuadU __attribute__((ext_vector_type(4), aligned(sizeof(T))));
    auto attr4 = [&](const T* row, int q) -> V4 {
      if constexpr (ANYC) {
        const int ws = 4 * q + 4 <= cc ? 4 * q : cc - 4; // wave-uniform; c0 + cc - 4 >= 1 (C >= 5)
        const TQuadU u = *reinterpret_cast<const TQuadU*>(row + ws);
        return V4{u.x, u.y, u.z, u.w};
      } else {
        return *reinterpret_cast<const V4*>(row + 4 * q);
      }
    };
    V4 A0[QA > 0 ? QA : 1], A1[QA > 0 ? QA : 1], A2[QA > 0 ? QA : 1];
    if constexpr (HAS_BARY) {
      if (covered) {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
          if (4 * q < cc) {
            A0[q] = attr4(a0, q);
            A1[q] = attr4(a1, q);
            A2[q] = attr4(a2, q);
          }
        }
      }
    }
    // 4. the next step's operands fly under phase 2: the next chunk of this row, or the first chunk (and the
    //    barycentrics) of the next row
    //    (ONE load site for both cases: with two, the compiler loaded into temporaries and copied them into G behind an
    //    s_waitcnt vmcnt(0) straight after the loads -- the prefetch waited for itself)
    {
      const int n_ps = last_chunk ? ps + 1 : ps, n_c0 = last_chunk ? 0 : c0 + CH;
      const bool n_any = last_chunk ? (ps + 1 < kPasses && __ballot(tr_n != -1) != 0) : cov != 0;
      load_row(n_ps, n_c0, min(CH, C - n_c0), n_any, last_chunk);
    }
    wave_lds_sync();
    // 5. phase 2: run sums per (corner, channel) lane -> one 64-byte atomic request per corner and run
    //    (a per-vertex LDS table in front of the atomics was measured slower here, twice: 1.01 vs 0.83 ms in round 1;
    //    1.07 vs 0.65 ms in round 2 with a float table and slots looked up at run heads only -- the flush it saves is
    //    0.11 ms, but the table's bookkeeping takes the kernel from 126 to 171 VGPRs, i.e. from 4 to 2 waves per SIMD.
    //    A per-wave cache keyed by TRIANGLE instead -- the run loop is scalar, so the lookup is one v_cmp + s_ff1 on a
    //    VGPR of keys and costs no registers (128 VGPRs, 4 waves kept) -- merges the rows of the wave's 64 x 4 pixels
    //    and was also slower: 0.666 vs 0.654 ms with 16 float entries updated by read-add-write (the LDS round trip
    //    sits on every run's critical path), 0.692 with 8, 0.74 with 16 double entries and fire-and-forget ds_add_f64
    //    (51 KB of LDS per workgroup: 3 waves per SIMD).  Round 3: CARRYING the sums of the vertices two consecutive
    //    runs share (neighbouring triangles share an edge) from run to run instead of flushing them -- which corner
    //    continues as which decided by the scalar unit, the sums moved between the corner rows by one ds_bpermute, a
    //    third of the atomic requests -- is correct and much slower: 0.901 vs 0.648 ms (1.096 vs 0.731 at 250k
    //    triangles); the per-run scalar chain (vertex ids to SGPRs, nine compares, the permute's round trip) sits on
    //    the one wave's critical path, the atomics it saves were fire-and-forget)
    if (cov != 0 && !DRTK_DBG(dbg, 1)) {
      const T* sg = s_g[wave];
      const T* sb = s_b[wave];
      // (the table + bary variant sits at its register bound: it keeps the unpipelined reads)
      scatter_runs_rows<T, TableAcc, DRTK_INTERP_WIDE_ONLY, TABLE, 16, !(TABLE && HAS_BARY)>(
          heads, cov, TABLE ? s_slot[wave] : nullptr, s_vid[wave], 3 * cc, cc, t_vals, C, attr_grad_n, CG, c0, sg, sb, dbg,
          TABLE ? c0 : 0);
    }
    // 6. bary gradient: interpolate_kernel.cu:238-246 accumulation order (channels ascending, over all chunks)
    if constexpr (HAS_BARY) {
      if (covered && !DRTK_DBG(dbg, 8)) {
        V4 D0[QD], D1[QD], D2[QD]; // the rest: requested now, used after the first part's products
        // (row addresses rebuilt from laundered ids: carried over phase 2 as three 64-bit pointers they were spilled)
        int32_t w0 = v0, w1 = v1, w2 = v2;
        asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));
        const T* d0 = attrs_n + int64_t(w0) * C + c0;
        const T* d1 = attrs_n + int64_t(w1) * C + c0;
        const T* d2 = attrs_n + int64_t(w2) * C + c0;
        // (ANYC at four waves per SIMD: the last group is requested only after the products of the others -- twelve registers
        // less in flight, one more round trip per chunk)
        constexpr int QD1 = (ANYC && CH == 16 && DRTK_INTERP_ANYC_SPLIT && QD > 1) ? QD - 1 : QD; // (12-channel chunks fit without: 0.69 vs 0.71 ms at C = 11)
#pragma unroll
        for (int q = 0; q < QD1; ++q) {
          if (4 * (QA + q) < cc) {
            D0[q] = attr4(d0, QA + q);
            D1[q] = attr4(d1, QA + q);
            D2[q] = attr4(d2, QA + q);
          }
        }
        auto dot4 = [&](int q4, const V4& u0, const V4& u1, const V4& u2) {
          if constexpr (ANYC) {
            if (4 * q4 + 4 > cc) { // the partial group: the window holds channels cc-4 .. cc-1, the first `skip` are done
              const int base = cc - 4, skip = 4 * q4 - base;
              const T* sgp = &s_g[wave][lane] + base * kRunPad;
              if (skip <= 1) {
                const T g1 = sgp[1 * kRunPad];
                bg0 += g1 * u0.y, bg1 += g1 * u1.y, bg2 += g1 * u2.y;
              }
              if (skip <= 2) {
                const T g2 = sgp[2 * kRunPad];
                bg0 += g2 * u0.z, bg1 += g2 * u1.z, bg2 += g2 * u2.z;
              }
              const T g3 = sgp[3 * kRunPad];
              bg0 += g3 * u0.w, bg1 += g3 * u1.w, bg2 += g3 * u2.w;
              return;
            }
          }
          const T g0 = s_g[wave][(4 * q4 + 0) * kRunPad + lane], g1 = s_g[wave][(4 * q4 + 1) * kRunPad + lane];
          const T g2 = s_g[wave][(4 * q4 + 2) * kRunPad + lane], g3 = s_g[wave][(4 * q4 + 3) * kRunPad + lane];
          bg0 += g0 * u0.x, bg1 += g0 * u1.x, bg2 += g0 * u2.x;
          bg0 += g1 * u0.y, bg1 += g1 * u1.y, bg2 += g1 * u2.y;
          bg0 += g2 * u0.z, bg1 += g2 * u1.z, bg2 += g2 * u2.z;
          bg0 += g3 * u0.w, bg1 += g3 * u1.w, bg2 += g3 * u2.w;
        };
#pragma unroll
        for (int q = 0; q < QA; ++q) {
          if (4 * q < cc) dot4(q, A0[q], A1[q], A2[q]);
        }
#pragma unroll
        for (int q = 0; q < QD1; ++q) {
          if (4 * (QA + q) < cc) dot4(QA + q, D0[q], D1[q], D2[q]);
        }
        if constexpr (QD1 < QD) {
          if (4 * (QA + QD1) < cc) {
            asm volatile("" ::: "memory"); // (keeps the request behind the products above)
            const V4 e0 = attr4(d0, QA + QD1), e1 = attr4(d1, QA + QD1), e2 = attr4(d2, QA + QD1);
            dot4(QA + QD1, e0, e1, e2);
          }
        }
      }
    }
    wave_lds_sync(); // the next step overwrites the staging rows
    c0 += CH;
    if (last_chunk) {
      if constexpr (HAS_BARY) {
        const int y = y0 + ps;
        if (x < W && y < H) {
          T* bgp = bgrad_n + int64_t(y) * W + x;
          bgp[0] = bg0, bgp[HW] = bg1, bgp[2 * HW] = bg2;
        }
      }
      tr = tr_n, v0 = vn0, v1 = vn1, v2 = vn2;
      tr_n = tr_nn;
      c0 = 0;
      if (++ps == kPasses) break;
    }
  }
  } // wave_fg
  if constexpr (TABLE) {
    // the tile's vertices to global memory: consecutive lanes = consecutive channels of a vertex (one request per 16)
    __syncthreads();
    if (!DRTK_DBG(dbg, 16)) {
      for (int e = threadIdx.x; e < (C << log2_slots); e += kBlock) {
        const int sl = e / C, c = e - sl * C;
        const int32_t key = t_keys[sl];
        if (key >= 0) {
          const T xv = static_cast<T>(t_vals[e]);
          if (xv != T(0)) atomic_add_global(attr_grad_n + int64_t(key) * CG + c, xv);
        }
      }
    }
  }
}

// rows of CG elements -> rows of C (the padded gradient workspace of the wide pipeline back into attr_grad)
template <typename T>
__global__ __launch_bounds__(kBlock) void compact_rows_kernel(const T* __restrict__ ws, T* __restrict__ out, int64_t count, int C, int CG) {
  const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= count) return;
  const int64_t r = i / C;
  out[i] = ws[r * CG + (i - r * C)];
}

// elements between the rows of the padded gradient workspace: whole 64-byte segments
template <typename T>
inline int64_t padded_row(int64_t C) {
  const int64_t seg = 64 / int64_t(sizeof(T));
  return (C + seg - 1) / seg * seg;
}
template <typename T>
inline bool rows_are_segments(int64_t C) {
  const size_t row_bytes = sizeof(T) * C;
  return row_bytes % 64 == 0 || 64 % row_bytes == 0;
}

template <typename T>
int interpolate_impl(
    const T* attrs, const int32_t* vi, const int32_t* index_img, const T* bary_img, int64_t N,
    int64_t V, int64_t C, int64_t vi_sN, int64_t H, int64_t W, T* out, int zero_background, hipStream_t stream) {
  const int64_t HW = H * W;
  if (N * HW * C == 0) return DRTK_OK;
  const bool pvec = (W % 4 == 0) && (reinterpret_cast<uintptr_t>(index_img) % 16 == 0) &&
      (reinterpret_cast<uintptr_t>(bary_img) % (4 * sizeof(T)) == 0) &&
      (reinterpret_cast<uintptr_t>(out) % (4 * sizeof(T)) == 0);
  const bool cvec = (C % 4 == 0) && (reinterpret_cast<uintptr_t>(attrs) % (4 * sizeof(T)) == 0);
  const dim3 block(kBlock);
#define LAUNCH(VEC, CV, ...)                                                                    \
  DRTK_LAUNCH(                                                                           \
      (interpolate_kernel<T, VEC, CV, ##__VA_ARGS__>),                                          \
      dim3(static_cast<unsigned>(ceil_div(ceil_div(HW, VEC), kBlock)), static_cast<unsigned>(N)), block, \
      0, stream, attrs, vi, index_img, bary_img, V, (int)C, vi_sN, (int)H, (int)W, out, zero_background, \
      xcd_strip(ceil_div(16 * W, int64_t(kBlock) * VEC)))
  // (double: the prefetching loop needs 254 VGPRs -- one wave per SIMD -- and loses to the plain one at two: 1.26 vs
  // 1.15 ms at 8 x 2048^2, C = 16, same box)
#ifndef DRTK_INTERP_F64_PREFETCH
#define DRTK_INTERP_F64_PREFETCH 0
#endif
  const bool prefetch = sizeof(T) == 4 || DRTK_INTERP_F64_PREFETCH;
  if (pvec && cvec && C > 4 && V * C < (int64_t(1) << 31) && prefetch) // (32-bit row offsets in the prefetching loop)
    LAUNCH(4, 4, true);
  else if (pvec && cvec)
    LAUNCH(4, 4);
  else if (pvec)
    LAUNCH(4, 1);
  else if (W >= 4 && cvec && C > 4 && V * C < (int64_t(1) << 31) && prefetch) // widths that are not a multiple of four, views into flat buffers
    LAUNCH(4, 4, true, true);
  else if (W >= 4 && cvec)
    LAUNCH(4, 4, false, true);
  else if (W >= 4)
    LAUNCH(4, 1, false, true);
  else if (cvec)
    LAUNCH(1, 4);
  else
    LAUNCH(1, 1);
#undef LAUNCH
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

template <typename T>
int interpolate_backward_impl(
    const T* grad_out, const T* attrs, const int32_t* vi, const int32_t* index_img,
    const T* bary_img, int64_t N, int64_t V, int64_t C, int64_t vi_sN, int64_t H, int64_t W,
    T* attr_grad, T* bary_grad, T* ws, size_t ws_bytes, hipStream_t stream) {
  const int64_t HW = H * W;
  // Round 6: where the rows of attr_grad are not whole 64-byte segments (C = 12, 20, 24 ...; every C that is not a multiple of
  // four) the run sums of the wide pipeline went through the workgroup's vertex table, which costs about a quarter more than
  // the per-run atomics aligned rows take (C = 12 0.61 ms against 0.48 for its bytes at C = 16's rate; C = 13 ... 15 0.745).
  // With a workspace the gradient is accumulated in rows PADDED to whole segments -- every such C takes the aligned rows'
  // path -- and compacted into attr_grad afterwards (N V C elements: 21 MB at C = 13 on the bench mesh).
  const bool pad_possible = attr_grad && ws && !rows_are_segments<T>(C) && N * V * C > 0 &&
      ws_bytes >= sizeof(T) * size_t(N * V * padded_row<T>(C)) && padded_row<T>(C) < (int64_t(1) << 30);
  bool padded = false; // decided below, with the pipeline; the zero-fill goes where the atomics will
  auto zero_grad = [&](bool to_ws) -> int {
    if (!attr_grad || N * V * C == 0) return DRTK_OK;
    return to_ws ? fill_bytes_async(ws, 0, sizeof(T) * N * V * padded_row<T>(C), stream) : fill_bytes_async(attr_grad, 0, sizeof(T) * N * V * C, stream);
  };
  if (N * HW == 0 || C == 0) {
    if (zero_grad(false) != DRTK_OK) return DRTK_ERR_LAUNCH;
  }
  if (N * HW == 0) return DRTK_OK;
  if (C == 0) {
    if (bary_grad && fill_bytes_async(bary_grad, 0, sizeof(T) * N * 3 * HW, stream) != DRTK_OK) return DRTK_ERR_LAUNCH;
    return DRTK_OK;
  }
  const bool cvec = (C % 4 == 0) && (reinterpret_cast<uintptr_t>(attrs) % (4 * sizeof(T)) == 0);
  const int tiles_x = static_cast<int>(ceil_div(W, kWave)), tiles_y = static_cast<int>(ceil_div(H, kTileRows));
  const dim3 grid(static_cast<unsigned>(int64_t(tiles_x) * tiles_y), static_cast<unsigned>(N));
  const dim3 block(kBlock);
  const int strip = xcd_strip(int64_t(tiles_x) * (16 / kTileRows));
#define LAUNCH(HV, HB, CV, CH)                                                                  \
  DRTK_LAUNCH(                                                                           \
      (interpolate_backward_kernel<T, HV, HB, CV, CH>), grid, block, 0, stream, grad_out, attrs, \
      vi, index_img, bary_img, V, (int)C, vi_sN, (int)H, (int)W, tiles_x, attr_grad, bary_grad,  \
      debug_flags(), strip)
  // The register-scan kernel (interpolate_backward_small_kernel: lane = pixel, the 3 C products of a pixel in registers, a
  // segmented DPP scan, a wave-private vertex table) beats the wide pipeline while 3 C values per lane are few: the wide
  // pipeline's phase 2 runs one (corner, channel) pair per lane, so a row of 8 channels keeps 24 of 64 lanes busy and costs
  // what a row of 16 does.  Round 5, 8 x 2048^2, both gradients, ms (register scan / wide): C = 5 0.33 / 0.54, 6 0.38 / 0.59,
  // 7 0.44 / 0.60, 8 0.50 / 0.55, 9 0.55 / 0.67, 10 0.58 / 0.67, 11 0.67 / 0.68, 12 0.69 / 0.64; double: 6 0.61 / 0.82,
  // 7 0.65 / 0.82, 8 0.80 / 0.80 (attributes only 0.60 / 0.46), 9 0.88 / 1.11.  The bary gradient alone (no scatter): register
  // scan up to 12 (C = 11: 0.32 against the generic kernel's 0.46).
#ifndef DRTK_INTERP_SMALL_MAXC_F32
#define DRTK_INTERP_SMALL_MAXC_F32 10
#endif
#ifndef DRTK_INTERP_SMALL_MAXC_F64
#define DRTK_INTERP_SMALL_MAXC_F64 7
#endif
#ifndef DRTK_INTERP_SMALL_INST
#define DRTK_INTERP_SMALL_INST 12
#endif
  constexpr int kSmallMaxC = sizeof(T) == 4 ? DRTK_INTERP_SMALL_MAXC_F32 : DRTK_INTERP_SMALL_MAXC_F64;
  static_assert(kSmallMaxC <= DRTK_INTERP_SMALL_INST && DRTK_INTERP_SMALL_INST <= 12, "instantiated channel counts");
  const bool small_c = C <= kSmallMaxC || (!attr_grad && C <= DRTK_INTERP_SMALL_INST) ||
      (sizeof(T) == 8 && C % 4 != 0 && C <= 11 && C <= DRTK_INTERP_SMALL_INST); // (double, rows not 32-byte aligned: 9 measured, 10 / 11 by the trend)
  // wide path: the vertex gradient (+ bary gradient) for any C >= 5, float and double (round 5; round 4: C % 4 == 0,
  // C >= 8, float only -- every other shape took the generic kernel, a chain of dependent waits per row)
  const bool wide = attr_grad && !small_c && HW * int64_t(sizeof(T)) < (int64_t(1) << 32) && !DRTK_DBG(debug_flags(), 128);
#ifndef DRTK_INTERP_PAD
#define DRTK_INTERP_PAD 1
#endif
  // ... measured (profiles/r06/interp_bwd_by_C*.json, same box, padded / table route, ms): with both gradients C = 11 0.616 / 0.665,
  // 12 0.554 / 0.617, 13 0.693 / 0.726, 15 0.702 / 0.735 -- but attributes only 0.497 / 0.468 at C = 11 (the lighter kernel hides
  // the table better than the atomics), and with more than one chunk the tail chunk's few channels are a request of their own
  // per run and corner where the table merges a vertex's chunks: C = 17 1.093 / 1.023, 20 1.071 / 0.860 (attributes only 1.039 /
  // 0.668).  So: one chunk, both gradients, float.
  padded = wide && pad_possible && DRTK_INTERP_PAD && bary_grad && sizeof(T) == 4 && C <= 16;
  if (zero_grad(padded) != DRTK_OK) return DRTK_ERR_LAUNCH;
  T* const grad_dst = padded ? ws : attr_grad;
  const int CG = static_cast<int>(padded ? padded_row<T>(C) : C);
  // (the bary gradient alone stays with the generic kernel: no scatter, covered pixels only -- 0.95 of the HBM peak on
  // SURVEY 8d's bytes at the bench coverage; a forward-shaped streaming kernel with four pixels per lane, which cannot
  // skip the background of a partly covered quad, was measured slower: 0.55 vs 0.40 ms at C = 16)
  if (wide) {
    // the workgroup's vertex table: as many slots as fit DRTK_INTERP_TABLE_BYTES of LDS (the staging rows take 27 KB; 4
    // workgroups per CU leave 40 KB each), at most 128, at least 16 -- else per-run atomics as before
#ifndef DRTK_INTERP_TABLE_BYTES
#define DRTK_INTERP_TABLE_BYTES 12800
#endif
    // -- and only where the rows of attr_grad are not made of whole 64-byte segments (C = 12, 20, 24, 28, ..., every C that
    // is not a multiple of four): there the per-run atomics are several times slower (8 x 2048^2, both gradients: C = 12
    // 0.92 -> 0.65 ms, C = 24 1.30 -> 1.04 with the table), while on aligned rows (C = 8, 16, 32, 64) the table's lookups
    // and barriers cost more than the fire-and-forget atomics they replace (C = 16: 0.63 -> 0.70 ms)
    int log2_slots = 7;
    while (log2_slots > 0 && (sizeof(TableAcc) * C + 4) * (size_t(1) << log2_slots) > DRTK_INTERP_TABLE_BYTES) --log2_slots;
    const bool rows_aligned = padded || rows_are_segments<T>(C);
#ifdef DRTK_INTERP_NO_TABLE
    const bool table = false;
#else
    const bool table = log2_slots >= 4 && !rows_aligned && !DRTK_DBG(debug_flags(), 4096);
#endif
    const size_t lds = table ? (sizeof(TableAcc) * C + 4) * (size_t(1) << log2_slots) : 0;
    // chunks of 16 channels (float; 12 for the element-aligned 11 ... 12, below), 8 for double.  Float chunks of 8 are what
    // C <= 8 would take -- but the register-scan kernel serves every float C <= DRTK_INTERP_SMALL_MAXC_F32 = 10 since round 5,
    // so they are instantiated only where a build makes them reachable (a lower DRTK_INTERP_SMALL_MAXC_F32, or
    // DRTK_INTERP_CH8_ANYC for the unaligned 9 ... 15): the default library carries none of them.
    constexpr int CH = sizeof(T) == 4 ? 16 : 8;
    constexpr bool kCh8Reachable = sizeof(T) == 4 && DRTK_INTERP_CH8 && (DRTK_INTERP_SMALL_MAXC_F32 < 8 || DRTK_INTERP_CH8_ANYC);
    const bool ch8 = kCh8Reachable && (C <= 8 || (DRTK_INTERP_CH8_ANYC && !cvec && C < 16));
    // 9 <= C <= 12 on element-aligned rows: ONE chunk of 12 (the 16-channel chunk's registers put the kernel at 3 waves per
    // SIMD there: 0.81 against 0.63 ms at C = 12, aligned vs not, same loads)
    const bool ch12 = sizeof(T) == 4 && DRTK_INTERP_CH12 && !cvec && C > 8 && C <= DRTK_INTERP_CH12_MAXC && bary_grad;
#define WIDE(HB, TB, AC)                                                                                                   \
  do {                                                                                                                     \
    bool launched8_ = false;                                                                                               \
    if constexpr (kCh8Reachable) {                                                                                         \
      if (ch8) {                                                                                                           \
        DRTK_LAUNCH(                                                                                                       \
            (interpolate_backward_wide_kernel<T, HB, TB, 8, AC>), grid, block, lds, stream, grad_out, attrs, vi, index_img, \
            bary_img, V, (int)C, vi_sN, (int)H, (int)W, tiles_x, grad_dst, bary_grad, debug_flags(), strip, log2_slots, CG); \
        launched8_ = true;                                                                                                 \
      }                                                                                                                    \
    }                                                                                                                      \
    if (launched8_) break;                                                                                                 \
    if (ch12 && AC && HB)                                                                                                  \
      DRTK_LAUNCH(                                                                                                         \
          (interpolate_backward_wide_kernel<T, HB, TB, (sizeof(T) == 4 && AC && HB ? 12 : CH), AC>), grid, block, lds, stream, grad_out, attrs, vi, index_img, \
          bary_img, V, (int)C, vi_sN, (int)H, (int)W, tiles_x, grad_dst, bary_grad, debug_flags(), strip, log2_slots, CG);   \
    else                                                                                                                   \
      DRTK_LAUNCH(                                                                                                         \
          (interpolate_backward_wide_kernel<T, HB, TB, CH, AC>), grid, block, lds, stream, grad_out, attrs, vi, index_img, \
          bary_img, V, (int)C, vi_sN, (int)H, (int)W, tiles_x, grad_dst, bary_grad, debug_flags(), strip, log2_slots, CG);   \
  } while (0)
#define WIDE_T(HB, AC) \
  if (table) WIDE(HB, true, AC); else WIDE(HB, false, AC)
    if (bary_grad) {
      if (cvec) { WIDE_T(true, false); } else { WIDE_T(true, true); }
    } else {
      if (cvec) { WIDE_T(false, false); } else { WIDE_T(false, true); }
    }
#undef WIDE_T
#undef WIDE
  } else if (small_c && !DRTK_DBG(debug_flags(), 128)) {
#define SMALL(HV, HB, CN)                                                                                                   \
  DRTK_LAUNCH(                                                                                                              \
      (interpolate_backward_small_kernel<T, HV, HB, CN>), grid, block, 0, stream, grad_out, attrs, vi, index_img, bary_img, \
      V, vi_sN, (int)H, (int)W, tiles_x, attr_grad, bary_grad, strip)
#define SMALL_CASE(HV, HB, K) \
  case K:                     \
    if constexpr (K <= DRTK_INTERP_SMALL_INST) SMALL(HV, HB, (K <= DRTK_INTERP_SMALL_INST ? K : 1)); \
    break;
#define SMALL_C(HV, HB)          \
  switch (C) {                   \
    SMALL_CASE(HV, HB, 1) SMALL_CASE(HV, HB, 2) SMALL_CASE(HV, HB, 3) SMALL_CASE(HV, HB, 4) SMALL_CASE(HV, HB, 5) SMALL_CASE(HV, HB, 6) \
    SMALL_CASE(HV, HB, 7) SMALL_CASE(HV, HB, 8) SMALL_CASE(HV, HB, 9) SMALL_CASE(HV, HB, 10) SMALL_CASE(HV, HB, 11) SMALL_CASE(HV, HB, 12) \
    default: break;              \
  }
    if (attr_grad && bary_grad) {
      SMALL_C(true, true)
    } else if (attr_grad) {
      SMALL_C(true, false)
    } else {
      SMALL_C(false, true)
    }
#undef SMALL_C
#undef SMALL_CASE
#undef SMALL
  } else if (attr_grad && bary_grad) { // (the generic kernel: the profiling build's flag 128, shapes beyond the pipelines)
    if (C <= 4) {
      if (cvec) LAUNCH(true, true, 4, 4); else LAUNCH(true, true, 1, 4);
    } else {
      if (cvec) LAUNCH(true, true, 4, 16); else LAUNCH(true, true, 1, 16);
    }
  } else if (attr_grad) {
    if (C <= 4) LAUNCH(true, false, 1, 4); else LAUNCH(true, false, 1, 16);
  } else {
    if (cvec) LAUNCH(false, true, 4, 16); else LAUNCH(false, true, 1, 16);
  }
#undef LAUNCH
  DRTK_RETURN_IF_LAUNCH_FAILED();
  if (padded) {
    const int64_t count = N * V * C;
    DRTK_LAUNCH((compact_rows_kernel<T>), dim3(static_cast<unsigned>(ceil_div(count, kBlock))), dim3(kBlock), 0, stream, ws, attr_grad, count, (int)C, CG);
    DRTK_RETURN_IF_LAUNCH_FAILED();
  }
  return DRTK_OK;
}

bool bad_common(int64_t N, int64_t V, int64_t C, int64_t F, int64_t vi_sN, int64_t H, int64_t W) {
  return N < 0 || V < 0 || C < 0 || F < 0 || H < 0 || W < 0 || C >= (1 << 20) ||
      (vi_sN != 0 && vi_sN != F * 3) || H * W >= (int64_t(1) << 31);
}

} // namespace
} // namespace drtk_amd

using namespace drtk_amd;

static int interpolate_entry(
    drtk_dtype_t dtype, const void* attrs, const int32_t* vi, const int32_t* index_img,
    const void* bary_img, int64_t N, int64_t V, int64_t C, int64_t F, int64_t vi_sN, int64_t H,
    int64_t W, void* out, int zero_background, drtk_stream_t stream) {
  if (bad_common(N, V, C, F, vi_sN, H, W)) return DRTK_ERR_INVALID_ARGUMENT;
  if (N * H * W * C > 0 && (!index_img || !bary_img || !out)) return DRTK_ERR_INVALID_ARGUMENT;
  if ((N * V * C > 0 && !attrs) || (F > 0 && !vi)) return DRTK_ERR_INVALID_ARGUMENT;
  if (dtype != DRTK_F32 && dtype != DRTK_F64) return DRTK_ERR_INVALID_ARGUMENT;
  const size_t es = dtype_size(dtype);
  DRTK_FOR_VIEW_SLICES(N, n0, n, interpolate_entry(
      dtype, advance(attrs, n0 * V * C, es), advance_typed(vi, n0 * vi_sN), advance_typed(index_img, n0 * H * W),
      advance(bary_img, n0 * 3 * H * W, es), n, V, C, F, vi_sN, H, W, advance(out, n0 * C * H * W, es), zero_background, stream))
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case DRTK_F32:
      return interpolate_impl<float>(static_cast<const float*>(attrs), vi, index_img, static_cast<const float*>(bary_img), N, V, C, vi_sN, H, W, static_cast<float*>(out), zero_background, s);
    case DRTK_F64:
      return interpolate_impl<double>(static_cast<const double*>(attrs), vi, index_img, static_cast<const double*>(bary_img), N, V, C, vi_sN, H, W, static_cast<double*>(out), zero_background, s);
    default:
      return DRTK_ERR_INVALID_ARGUMENT;
  }
}

extern "C" int drtk_amd_interpolate(
    drtk_dtype_t dtype, const void* attrs, const int32_t* vi, const int32_t* index_img,
    const void* bary_img, int64_t N, int64_t V, int64_t C, int64_t F, int64_t vi_sN, int64_t H,
    int64_t W, void* out, drtk_stream_t stream) {
  return interpolate_entry(dtype, attrs, vi, index_img, bary_img, N, V, C, F, vi_sN, H, W, out, 0, stream);
}

extern "C" int drtk_amd_interpolate_masked(
    drtk_dtype_t dtype, const void* attrs, const int32_t* vi, const int32_t* index_img,
    const void* bary_img, int64_t N, int64_t V, int64_t C, int64_t F, int64_t vi_sN, int64_t H,
    int64_t W, void* out, drtk_stream_t stream) {
  return interpolate_entry(dtype, attrs, vi, index_img, bary_img, N, V, C, F, vi_sN, H, W, out, 1, stream);
}

extern "C" int drtk_amd_interpolate_backward_workspace_bytes(drtk_dtype_t dtype, int64_t N, int64_t V, int64_t C, size_t* bytes) {
  if (!bytes || N < 0 || V < 0 || C < 0 || (dtype != DRTK_F32 && dtype != DRTK_F64)) return DRTK_ERR_INVALID_ARGUMENT;
  const bool f32 = dtype == DRTK_F32;
  const bool segs = f32 ? rows_are_segments<float>(C) : rows_are_segments<double>(C);
  const int64_t row = f32 ? padded_row<float>(C) : padded_row<double>(C);
  // only the wide pipeline's one-chunk float shapes pad (interpolate_backward_impl: 11 <= C <= 15); anything else needs no workspace
  const bool pads = f32 && C > DRTK_INTERP_SMALL_MAXC_F32 && C <= 16;
  *bytes = (!segs && pads && N * V > 0) ? dtype_size(dtype) * size_t(N * V * row) : 0;
  return DRTK_OK;
}

extern "C" int drtk_amd_interpolate_backward_ws(
    drtk_dtype_t dtype, const void* grad_out, const void* attrs, const int32_t* vi,
    const int32_t* index_img, const void* bary_img, int64_t N, int64_t V, int64_t C, int64_t F,
    int64_t vi_sN, int64_t H, int64_t W, void* attr_grad, void* bary_grad, void* workspace, size_t workspace_bytes,
    drtk_stream_t stream) {
  if (bad_common(N, V, C, F, vi_sN, H, W)) return DRTK_ERR_INVALID_ARGUMENT;
  if (!attr_grad && !bary_grad) {
    // A null output is one that was not requested OR is empty (N*V*C == 0 / N*H*W == 0; an empty tensor has no
    // storage): either way there is nothing to write.  Requesting nothing is an error only when both gradients
    // would have had elements.
    return (N * V * C > 0 && N * H * W > 0) ? DRTK_ERR_INVALID_ARGUMENT : DRTK_OK;
  }
  if (N * H * W * C > 0 && (!grad_out || !index_img || !bary_img)) return DRTK_ERR_INVALID_ARGUMENT;
  if ((N * V * C > 0 && !attrs) || (F > 0 && !vi)) return DRTK_ERR_INVALID_ARGUMENT;
  if (dtype != DRTK_F32 && dtype != DRTK_F64) return DRTK_ERR_INVALID_ARGUMENT;
  if (workspace && reinterpret_cast<uintptr_t>(workspace) % 64 != 0) return DRTK_ERR_INVALID_ARGUMENT; // rows of whole 64-byte segments
  const size_t es = dtype_size(dtype);
  // (slices of views reuse the workspace: each call leaves nothing in it)
  DRTK_FOR_VIEW_SLICES(N, n0, n, drtk_amd_interpolate_backward_ws(
      dtype, advance(grad_out, n0 * C * H * W, es), advance(attrs, n0 * V * C, es), advance_typed(vi, n0 * vi_sN),
      advance_typed(index_img, n0 * H * W), advance(bary_img, n0 * 3 * H * W, es), n, V, C, F, vi_sN, H, W,
      advance(attr_grad, n0 * V * C, es), advance(bary_grad, n0 * 3 * H * W, es), workspace, workspace_bytes, stream))
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case DRTK_F32:
      return interpolate_backward_impl<float>(static_cast<const float*>(grad_out), static_cast<const float*>(attrs), vi, index_img, static_cast<const float*>(bary_img), N, V, C, vi_sN, H, W, static_cast<float*>(attr_grad), static_cast<float*>(bary_grad), static_cast<float*>(workspace), workspace_bytes, s);
    case DRTK_F64:
      return interpolate_backward_impl<double>(static_cast<const double*>(grad_out), static_cast<const double*>(attrs), vi, index_img, static_cast<const double*>(bary_img), N, V, C, vi_sN, H, W, static_cast<double*>(attr_grad), static_cast<double*>(bary_grad), static_cast<double*>(workspace), workspace_bytes, s);
    default:
      return DRTK_ERR_INVALID_ARGUMENT;
  }
}

extern "C" int drtk_amd_interpolate_backward(
    drtk_dtype_t dtype, const void* grad_out, const void* attrs, const int32_t* vi,
    const int32_t* index_img, const void* bary_img, int64_t N, int64_t V, int64_t C, int64_t F,
    int64_t vi_sN, int64_t H, int64_t W, void* attr_grad, void* bary_grad, drtk_stream_t stream) {
  return drtk_amd_interpolate_backward_ws(dtype, grad_out, attrs, vi, index_img, bary_img, N, V, C, F, vi_sN, H, W, attr_grad, bary_grad, nullptr, 0, stream);
}
