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
template <typename T, int VEC, int CV, bool PREFETCH = false>
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

  int32_t tr[VEC];
  T B0[VEC], B1[VEC], B2[VEC];
  if constexpr (VEC == 4) {
    const int4 t4 = *reinterpret_cast<const int4*>(idx_p);
    tr[0] = t4.x, tr[1] = t4.y, tr[2] = t4.z, tr[3] = t4.w;
    const V4 a = *reinterpret_cast<const V4*>(bary_p);
    const V4 b = *reinterpret_cast<const V4*>(bary_p + HW);
    const V4 c = *reinterpret_cast<const V4*>(bary_p + 2 * HW);
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
  T bgx[VEC];
  // "undefined region" sweep (interpolate_kernel.cu:104-108, CPU twin interpolate_kernel_cpu.cpp:99-104)
  const T bgy = (static_cast<T>(y) * T(2.0) + T(1.0)) / static_cast<T>(H) - T(1.0);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    if (tr[j] != -1) {
      const int32_t* face = vi_n + int64_t(tr[j]) * 3;
      a0[j] = attrs_n + int64_t(face[0]) * C;
      a1[j] = attrs_n + int64_t(face[1]) * C;
      a2[j] = attrs_n + int64_t(face[2]) * C;
    } else {
      a0[j] = a1[j] = a2[j] = attrs_n;
    }
    bgx[j] = (static_cast<T>(x0 + j) * T(2.0) + T(1.0)) / static_cast<T>(W) - T(1.0);
  }

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
#pragma unroll
          for (int k = 0; k < 3; ++k) q[j][k] = *reinterpret_cast<const V4*>(attrs_n + off[j][k] + c0);
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
          for (int cc = 0; cc < CV; ++cc) r[cc][j] = zero_background ? T(0) : (((c0 + cc) & 1) ? bgy : bgx[j]);
        }
      }
#pragma unroll
      for (int cc = 0; cc < CV; ++cc) {
        T* o = out_p + int64_t(c0 + cc) * HW;
        *reinterpret_cast<V4*>(o) = V4{r[cc][0], r[cc][1], r[cc][2], r[cc][3]};
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
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (tr[j] != -1) {
        T u0[CV], u1[CV], u2[CV];
        if constexpr (CV == 4) {
          const V4 q0 = *reinterpret_cast<const V4*>(a0[j] + c0);
          const V4 q1 = *reinterpret_cast<const V4*>(a1[j] + c0);
          const V4 q2 = *reinterpret_cast<const V4*>(a2[j] + c0);
          u0[0] = q0.x, u0[1] = q0.y, u0[2] = q0.z, u0[3] = q0.w;
          u1[0] = q1.x, u1[1] = q1.y, u1[2] = q1.z, u1[3] = q1.w;
          u2[0] = q2.x, u2[1] = q2.y, u2[2] = q2.z, u2[3] = q2.w;
        } else {
          u0[0] = a0[j][c0], u1[0] = a1[j][c0], u2[0] = a2[j][c0];
        }
#pragma unroll
        for (int cc = 0; cc < CV; ++cc) r[cc][j] = u0[cc] * B0[j] + u1[cc] * B1[j] + u2[cc] * B2[j];
      } else {
#pragma unroll
        for (int cc = 0; cc < CV; ++cc) r[cc][j] = zero_background ? T(0) : (((c0 + cc) & 1) ? bgy : bgx[j]);
      }
    }
#pragma unroll
    for (int cc = 0; cc < CV; ++cc) {
      T* o = out_p + int64_t(c0 + cc) * HW;
      if constexpr (VEC == 4) {
        *reinterpret_cast<V4*>(o) = V4{r[cc][0], r[cc][1], r[cc][2], r[cc][3]};
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
  __shared__ int32_t t_keys[HAS_VERT ? kWaves : 1][HAS_VERT ? kSlots : 1];
  __shared__ TableAcc t_vals[HAS_VERT ? kWaves : 1][HAS_VERT ? kSlots * 4 : 1];

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
    for (int i = lane; i < kSlots * 4; i += kWave) t_vals[wave][i] = 0;
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
        if (tail && covered) table_add<T, 3, CN, kSlots>(t_keys[wave], t_vals[wave], 4, cur, p, attr_grad_n, CN);
      }
    }
  }
  if constexpr (HAS_VERT) {
    wave_lds_sync();
    table_flush<T, TableAcc, kSlots>(t_keys[wave], t_vals[wave], 4, CN, attr_grad_n, CN, 0);
  }
}

// Backward, wide-channel fast path (vertex + bary gradients, C % 16 == 0, 16-byte aligned rows).
//
// Same tiling and phase structure as interpolate_backward_kernel<.., 16>, but written so that no
// memory round trip is ever waited for in isolation (the generic kernel's phase 1 is a chain of up
// to nine dependent global waits per row -- index, corners, bary, then go/attribute loads per
// 4-channel block -- and PMC shows its waves parked on s_waitcnt 72 % of their life):
//   * the triangle and corner ids of all four rows are fetched once, up front;
//   * every global load of a row (16 grad_out planes, 3 bary planes, 12 attribute float4) is issued
//     in one batch, and the NEXT row's grad_out/bary batch is issued before the current row's
//     phase 2, so it lands while the wave is busy in LDS;
//   * the per-pixel bary-gradient dot products run after phase 2 and read grad_out back from the
//     LDS staging rows, which frees the registers for the prefetch.
template <typename T>
__global__ __launch_bounds__(kBlock, 4) void interpolate_backward_wide_kernel(
    const T* __restrict__ grad_out, const T* __restrict__ attrs, const int32_t* __restrict__ vi,
    const int32_t* __restrict__ index_img, const T* __restrict__ bary_img, int64_t V, int C,
    int64_t vi_sN, int H, int W, int tiles_x, T* __restrict__ attr_grad, T* __restrict__ bary_grad,
    int dbg, int strip) {
  using V4 = typename Vec4<T>::type;
  constexpr int CH = 16;
  constexpr int kWaves = kBlock / kWave;
  constexpr int kPasses = kTileRows / kWaves;
  static_assert(kPasses == 4, "row pipeline below is written for 4 rows per wave");
  __shared__ __attribute__((aligned(16))) T s_g[kWaves][CH * kRunPad];
  __shared__ __attribute__((aligned(16))) T s_b[kWaves][3 * kRunPad];
  __shared__ int32_t s_vid[kWaves][3 * kRunPad];

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
  T* attr_grad_n = attr_grad + int64_t(n) * V * C;
  const T* go_n = grad_out + int64_t(n) * C * HW;
  const T* bary_n = bary_img + int64_t(n) * 3 * HW;
  T* bgrad_n = bary_grad + int64_t(n) * 3 * HW;

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
  // scatter: it zeroes its part of bary_grad and leaves.
  {
    int32_t t4[kPasses];
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) t4[ps] = load_tr(ps);
    const bool any_fg = (t4[0] & t4[1] & t4[2] & t4[3]) != -1;
    if (__ballot(any_fg) == 0) {
#pragma unroll
      for (int ps = 0; ps < kPasses; ++ps) {
        const int y = y0 + ps;
        if (x < W && y < H) {
          T* bgp = bgrad_n + int64_t(y) * W + x;
          bgp[0] = T(0), bgp[HW] = T(0), bgp[2 * HW] = T(0);
        }
      }
      return;
    }
  }

  for (int c0 = 0; c0 < C; c0 += CH) {
    T G[CH], B[3];
    auto load_row = [&](int ps, bool cov) { // grad_out + bary of row ps; zeros where uncovered
      const int64_t pix = int64_t(y0 + ps) * W + x;
#pragma unroll
      for (int c = 0; c < CH; ++c) G[c] = cov ? go_n[int64_t(c0 + c) * HW + pix] : T(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) B[k] = cov ? bary_n[int64_t(k) * HW + pix] : T(0);
    };
    // rotating state: row ps (tr, v*), row ps+1 (tr_n, vn*), row ps+2 (tr_nn)
    int32_t tr = load_tr(0), tr_n = load_tr(1), tr_nn = -1;
    int32_t v0, v1, v2, vn0 = 0, vn1 = 0, vn2 = 0;
    load_face(tr, v0, v1, v2);
    load_row(0, tr != -1);
#pragma unroll 1
    for (int ps = 0; ps < kPasses; ++ps) {
      const int y = y0 + ps;
      const int64_t pix = int64_t(y) * W + x;
      const bool covered = tr != -1;
      // 0. the two CONDITIONAL fetches of the rows ahead come first: `cond ? load : default` compiles to a branch around the
      //    load with an s_waitcnt vmcnt(0) inside it; issued after the next row's 19 loads (as they used to be) that wait
      //    held the wave until all of them had landed, before the phase 2 they were meant to fly under (0.659 -> 0.650 ms;
      //    making the two fetches unconditional instead cost registers and was slower, 0.675)
      if (ps + 1 < kPasses) {
        load_face(tr_n, vn0, vn1, vn2);
        tr_nn = load_tr(ps + 2);
      }
      // 1. stage this row
#pragma unroll
      for (int c = 0; c < CH; ++c) s_g[wave][c * kRunPad + lane] = G[c];
#pragma unroll
      for (int k = 0; k < 3; ++k) s_b[wave][k * kRunPad + lane] = B[k];
      s_vid[wave][0 * kRunPad + lane] = v0;
      s_vid[wave][1 * kRunPad + lane] = v1;
      s_vid[wave][2 * kRunPad + lane] = v2;
      unsigned long long heads, cov;
      run_masks(tr, heads, cov);
      // 2. first half of the attribute rows of this row's triangles (consumed after phase 2)
      constexpr int QH = CH / 8; // float4 per corner and half
      const T* a0 = attrs_n + int64_t(v0) * C + c0;
      const T* a1 = attrs_n + int64_t(v1) * C + c0;
      const T* a2 = attrs_n + int64_t(v2) * C + c0;
      V4 A0[QH], A1[QH], A2[QH];
      if (covered) {
#pragma unroll
        for (int q = 0; q < QH; ++q) {
          A0[q] = *reinterpret_cast<const V4*>(a0 + 4 * q);
          A1[q] = *reinterpret_cast<const V4*>(a1 + 4 * q);
          A2[q] = *reinterpret_cast<const V4*>(a2 + 4 * q);
        }
      }
      // 3. the next row's operands (and the ids of the row after it) fly under phase 2
      if (ps + 1 < kPasses) load_row(ps + 1, tr_n != -1);
      wave_lds_sync();
      // 4. phase 2: run sums per (corner, channel) lane -> one 64-byte atomic request per corner and run
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
        scatter_runs<T>(
            heads, cov, nullptr, s_vid[wave], 3 * CH, CH, static_cast<TableAcc*>(nullptr), 0, attr_grad_n, C, c0,
            [sg, sb](int k, int c, int g4, T* xv) {
              const V4 a = *reinterpret_cast<const V4*>(sg + c * kRunPad + 4 * g4);
              const V4 b = *reinterpret_cast<const V4*>(sb + k * kRunPad + 4 * g4);
              xv[0] = a.x * b.x, xv[1] = a.y * b.y, xv[2] = a.z * b.z, xv[3] = a.w * b.w;
            }, dbg);
      }
      // 5. bary gradient: interpolate_kernel.cu:238-246 accumulation order (channels ascending)
      T bg0 = T(0), bg1 = T(0), bg2 = T(0);
      if (covered && !DRTK_DBG(dbg, 8)) {
        V4 D0[QH], D1[QH], D2[QH]; // second half: requested now, used after the first half's products
#pragma unroll
        for (int q = 0; q < QH; ++q) {
          D0[q] = *reinterpret_cast<const V4*>(a0 + 4 * (QH + q));
          D1[q] = *reinterpret_cast<const V4*>(a1 + 4 * (QH + q));
          D2[q] = *reinterpret_cast<const V4*>(a2 + 4 * (QH + q));
        }
        auto dot4 = [&](int q4, const V4& u0, const V4& u1, const V4& u2) {
          const T g0 = s_g[wave][(4 * q4 + 0) * kRunPad + lane], g1 = s_g[wave][(4 * q4 + 1) * kRunPad + lane];
          const T g2 = s_g[wave][(4 * q4 + 2) * kRunPad + lane], g3 = s_g[wave][(4 * q4 + 3) * kRunPad + lane];
          bg0 += g0 * u0.x, bg1 += g0 * u1.x, bg2 += g0 * u2.x;
          bg0 += g1 * u0.y, bg1 += g1 * u1.y, bg2 += g1 * u2.y;
          bg0 += g2 * u0.z, bg1 += g2 * u1.z, bg2 += g2 * u2.z;
          bg0 += g3 * u0.w, bg1 += g3 * u1.w, bg2 += g3 * u2.w;
        };
#pragma unroll
        for (int q = 0; q < QH; ++q) dot4(q, A0[q], A1[q], A2[q]);
#pragma unroll
        for (int q = 0; q < QH; ++q) dot4(QH + q, D0[q], D1[q], D2[q]);
      }
      if (x < W && y < H) { // channel chunks accumulate in ascending order, like the reference's loop
        T* bgp = bgrad_n + pix;
        if (c0 == 0) {
          bgp[0] = bg0, bgp[HW] = bg1, bgp[2 * HW] = bg2;
        } else if (covered) {
          bgp[0] += bg0, bgp[HW] += bg1, bgp[2 * HW] += bg2;
        }
      }
      wave_lds_sync();
      tr = tr_n, v0 = vn0, v1 = vn1, v2 = vn2;
      tr_n = tr_nn;
    }
  }
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
      dim3(static_cast<unsigned>(ceil_div(HW / VEC, kBlock)), static_cast<unsigned>(N)), block, \
      0, stream, attrs, vi, index_img, bary_img, V, (int)C, vi_sN, (int)H, (int)W, out, zero_background, \
      xcd_strip(ceil_div(16 * W, int64_t(kBlock) * VEC)))
  if (pvec && cvec && C > 4 && V * C < (int64_t(1) << 31)) // (32-bit row offsets in the prefetching loop)
    LAUNCH(4, 4, true);
  else if (pvec && cvec)
    LAUNCH(4, 4);
  else if (pvec)
    LAUNCH(4, 1);
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
    T* attr_grad, T* bary_grad, hipStream_t stream) {
  if (attr_grad && N * V * C > 0) {
    if (fill_bytes_async(attr_grad, 0, sizeof(T) * N * V * C, stream) != DRTK_OK) return DRTK_ERR_LAUNCH;
  }
  const int64_t HW = H * W;
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
  const bool small_c = C <= 4;
  // float only: the double instantiation would need > 256 VGPRs for the same pipeline
  const bool wide = sizeof(T) == 4 && attr_grad && bary_grad && cvec && (C % 16 == 0) && !DRTK_DBG(debug_flags(), 128);
  if (wide) {
    if constexpr (sizeof(T) == 4) {
      DRTK_LAUNCH(
          (interpolate_backward_wide_kernel<T>), grid, block, 0, stream, grad_out, attrs, vi, index_img, bary_img,
          V, (int)C, vi_sN, (int)H, (int)W, tiles_x, attr_grad, bary_grad, debug_flags(), strip);
    }
  } else if (small_c && !DRTK_DBG(debug_flags(), 128)) {
#define SMALL(HV, HB, CN)                                                                                                   \
  DRTK_LAUNCH(                                                                                                              \
      (interpolate_backward_small_kernel<T, HV, HB, CN>), grid, block, 0, stream, grad_out, attrs, vi, index_img, bary_img, \
      V, vi_sN, (int)H, (int)W, tiles_x, attr_grad, bary_grad, strip)
#define SMALL_C(HV, HB)          \
  switch (C) {                   \
    case 1: SMALL(HV, HB, 1); break; \
    case 2: SMALL(HV, HB, 2); break; \
    case 3: SMALL(HV, HB, 3); break; \
    default: SMALL(HV, HB, 4); break; \
  }
    if (attr_grad && bary_grad) {
      SMALL_C(true, true)
    } else if (attr_grad) {
      SMALL_C(true, false)
    } else {
      SMALL_C(false, true)
    }
#undef SMALL_C
#undef SMALL
  } else if (attr_grad && bary_grad) {
    if (small_c) {
      if (cvec) LAUNCH(true, true, 4, 4); else LAUNCH(true, true, 1, 4);
    } else {
      if (cvec) LAUNCH(true, true, 4, 16); else LAUNCH(true, true, 1, 16);
    }
  } else if (attr_grad) {
    if (small_c) LAUNCH(true, false, 1, 4); else LAUNCH(true, false, 1, 16);
  } else {
    if (cvec) LAUNCH(false, true, 4, 16); else LAUNCH(false, true, 1, 16);
  }
#undef LAUNCH
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

bool bad_common(int64_t N, int64_t V, int64_t C, int64_t F, int64_t vi_sN, int64_t H, int64_t W) {
  return N < 0 || V < 0 || C < 0 || F < 0 || H < 0 || W < 0 || N > 65535 || C >= (1 << 20) ||
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

extern "C" int drtk_amd_interpolate_backward(
    drtk_dtype_t dtype, const void* grad_out, const void* attrs, const int32_t* vi,
    const int32_t* index_img, const void* bary_img, int64_t N, int64_t V, int64_t C, int64_t F,
    int64_t vi_sN, int64_t H, int64_t W, void* attr_grad, void* bary_grad, drtk_stream_t stream) {
  if (bad_common(N, V, C, F, vi_sN, H, W)) return DRTK_ERR_INVALID_ARGUMENT;
  if (!attr_grad && !bary_grad) {
    // A null output is one that was not requested OR is empty (N*V*C == 0 / N*H*W == 0; an empty tensor has no
    // storage): either way there is nothing to write.  Requesting nothing is an error only when both gradients
    // would have had elements.
    return (N * V * C > 0 && N * H * W > 0) ? DRTK_ERR_INVALID_ARGUMENT : DRTK_OK;
  }
  if (N * H * W * C > 0 && (!grad_out || !index_img || !bary_img)) return DRTK_ERR_INVALID_ARGUMENT;
  if ((N * V * C > 0 && !attrs) || (F > 0 && !vi)) return DRTK_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case DRTK_F32:
      return interpolate_backward_impl<float>(static_cast<const float*>(grad_out), static_cast<const float*>(attrs), vi, index_img, static_cast<const float*>(bary_img), N, V, C, vi_sN, H, W, static_cast<float*>(attr_grad), static_cast<float*>(bary_grad), s);
    case DRTK_F64:
      return interpolate_backward_impl<double>(static_cast<const double*>(grad_out), static_cast<const double*>(attrs), vi, index_img, static_cast<const double*>(bary_img), N, V, C, vi_sN, H, W, static_cast<double*>(attr_grad), static_cast<double*>(bary_grad), s);
    default:
      return DRTK_ERR_INVALID_ARGUMENT;
  }
}
