// render -- per-pixel perspective-correct barycentrics + depth, and its analytic backward.
//
// Reference: src/render/render_kernel.cu:19-117 (forward), :119-281 (backward).  Forward is a
// pure stream: 4 B/px of index in, 16 B/px out, triangle data gathered through L2.  Each lane
// handles 4 horizontally adjacent pixels so that every global access is a 16-byte vector
// (index int4 in; depth + three bary planes float4 out).
// Backward produces 9 values per covered pixel that the reference scatters with 9 atomics per
// pixel; here they go through the wave-level run reduction of segscatter.hpp.
#include "common.hpp"
#include "segscatter.hpp"

namespace drtk_amd {
namespace {

// Everything the forward and the backward share (render_kernel.cu:68-101 == :174-218).
template <typename T>
struct RenderPix {
  int32_t vi0, vi1, vi2;
  T v01x, v01y, v02x, v02y;
  T den_raw, den;
  T vp0x, vp0y;
  T b0, b1, b2;
  T z0e, z1e, z2e;
  bool z0c, z1c, z2c;
  T dinv0, dinv1, dinv2;
  T depth_inverse, depth_inverse_eps, depth;
};

// EXACT = true: IEEE divisions (the forward's outputs are compared bit for bit).  EXACT = false (backward,
// float): a * rcp(b), 1-2 ulp -- the gradients are sums of thousands of such terms accumulated in
// arbitrary order and are compared at 1e-5, and the backward kernel is bound by instruction issue
// (an IEEE float division is a 10-instruction sequence, 12 of them per pixel).
template <bool EXACT, typename T>
__device__ __forceinline__ T quotient(T a, T b) {
  if constexpr (!EXACT && sizeof(T) == 4) {
    return a * __builtin_amdgcn_rcpf(b);
  } else {
    return a / b;
  }
}

template <typename T>
struct TriVerts { // the three pixel-space vertices of one triangle
  T p0x, p0y, p0z, p1x, p1y, p1z, p2x, p2y, p2z;
};

template <typename T>
__device__ __forceinline__ TriVerts<T> load_tri_verts(const T* __restrict__ v_n, int32_t vi0, int32_t vi1, int32_t vi2) {
  const T* q0 = v_n + 3 * (int64_t)vi0;
  const T* q1 = v_n + 3 * (int64_t)vi1;
  const T* q2 = v_n + 3 * (int64_t)vi2;
  return TriVerts<T>{q0[0], q0[1], q0[2], q1[0], q1[1], q1[2], q2[0], q2[1], q2[2]};
}

template <typename T, bool EXACT = true>
__device__ __forceinline__ void render_math(
    const TriVerts<T>& q, int32_t vi0, int32_t vi1, int32_t vi2, int x, int y, RenderPix<T>& r) {
  r.vi0 = vi0;
  r.vi1 = vi1;
  r.vi2 = vi2;
  const T p0x = q.p0x, p0y = q.p0y, p0z = q.p0z;
  const T p1x = q.p1x, p1y = q.p1y, p1z = q.p1z;
  const T p2x = q.p2x, p2y = q.p2y, p2z = q.p2z;
  r.v01x = p1x - p0x;
  r.v01y = p1y - p0y;
  r.v02x = p2x - p0x;
  r.v02y = p2y - p0y;
  r.den_raw = r.v01x * r.v02y - r.v01y * r.v02x;
  r.den = epsclamp(r.den_raw);
  r.vp0x = static_cast<T>(x) - p0x;
  r.vp0y = static_cast<T>(y) - p0y;
  const T b1_pre = r.vp0x * r.v02y - r.vp0y * r.v02x;
  const T b2_pre = r.vp0y * r.v01x - r.vp0x * r.v01y;
  r.b1 = quotient<EXACT>(b1_pre, r.den);
  r.b2 = quotient<EXACT>(b2_pre, r.den);
  r.b0 = T(1.0) - r.b1 - r.b2;
  r.z0e = epsclamp(p0z);
  r.z1e = epsclamp(p1z);
  r.z2e = epsclamp(p2z);
  r.z0c = r.z0e != p0z;
  r.z1c = r.z1e != p1z;
  r.z2c = r.z2e != p2z;
  r.dinv0 = quotient<EXACT>(T(1.0), r.z0e);
  r.dinv1 = quotient<EXACT>(T(1.0), r.z1e);
  r.dinv2 = quotient<EXACT>(T(1.0), r.z2e);
  r.depth_inverse = r.dinv0 * r.b0 + r.dinv1 * r.b1 + r.dinv2 * r.b2;
  r.depth_inverse_eps = epsclamp(r.depth_inverse);
  r.depth = quotient<EXACT>(T(1.0), r.depth_inverse_eps);
}

template <typename T, bool EXACT = true>
__device__ __forceinline__ void render_pix(
    const T* __restrict__ v_n, int32_t vi0, int32_t vi1, int32_t vi2, int x, int y, RenderPix<T>& r) {
  render_math<T, EXACT>(load_tri_verts<T>(v_n, vi0, vi1, vi2), vi0, vi1, vi2, x, y, r);
}

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

// VEC = 4: one lane = 4 consecutive pixels of the view.  ANYW = false: W % 4 == 0 and 16-byte aligned images, the four lie
// in one row and move as aligned vectors.  ANYW = true: any W >= 4 and any element-aligned placement -- the four may run
// over the end of a row (each pixel gets its own coordinates), the accesses need the element's alignment only, and the
// last lane of a view whose pixel count is not a multiple of four goes pixel by pixel.  VEC = 1: W < 4.
template <typename T, int VEC, bool ANYW = false>
__global__ __launch_bounds__(kBlock) void render_kernel(
    const T* __restrict__ v, const int32_t* __restrict__ vi, const int32_t* __restrict__ index_img,
    int64_t V, int64_t vi_sN, int H, int W, T* __restrict__ depth_img, T* __restrict__ bary_img, int strip) {
  const int64_t HW = int64_t(H) * W;
  const int n = blockIdx.y;
  const int64_t pix0 = (int64_t(tile_index(strip)) * kBlock + threadIdx.x) * VEC; // pixel within the view
  if (pix0 >= HW) return;
  const T* v_n = v + int64_t(n) * V * 3;
  const int32_t* vi_n = vi + int64_t(n) * vi_sN;
  const int32_t* idx_p = index_img + int64_t(n) * HW + pix0;
  T* depth_p = depth_img + int64_t(n) * HW + pix0;
  T* bary_p = bary_img + int64_t(n) * 3 * HW + pix0;

  typedef int32_t IQuad __attribute__((ext_vector_type(4), aligned(4)));
  typedef T TQuad __attribute__((ext_vector_type(4), aligned(sizeof(T))));
  const bool whole = !ANYW || pix0 + 4 <= HW; // all four pixels belong to this view
  int32_t tr[VEC];
  if constexpr (VEC == 4 && ANYW) {
    if (whole) {
      const IQuad t4 = *reinterpret_cast<const IQuad*>(idx_p);
      tr[0] = t4.x, tr[1] = t4.y, tr[2] = t4.z, tr[3] = t4.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) tr[j] = pix0 + j < HW ? idx_p[j] : -1;
    }
  } else if constexpr (VEC == 4) {
    const int4 t4 = *reinterpret_cast<const int4*>(idx_p); // (non-temporal index / upstream-gradient loads in this file: within the noise, round 6)
    tr[0] = t4.x;
    tr[1] = t4.y;
    tr[2] = t4.z;
    tr[3] = t4.w;
  } else {
    tr[0] = idx_p[0];
  }
  const int y = static_cast<int>(pix0 / W);
  const int x0 = static_cast<int>(pix0 - int64_t(y) * W);

  // Every memory round trip of the lane's VEC pixels is issued as ONE batch -- the faces of all of them, then their nine
  // vertices -- instead of one `if (covered) { face -> vertices -> arithmetic }` block per pixel: a conditional load
  // compiles to a branch with its own s_waitcnt, i.e. 2 VEC dependent round trips per lane after the index load
  // (0.166 -> 0.157 ms at the benchmark shape, same box).  To make the loads unconditional a background pixel reads the
  // triangle of a covered pixel of its lane, or of its wave (its results are discarded); a wave with no covered pixel
  // stores zeros.  Measured on top of this and NOT kept: the six IEEE divisions per pixel (11 instructions each)
  // replaced by the rasterizer's correctly rounded fast forms (hardware reciprocal + Markstein corrections, one
  // wave-uniform IEEE fallback) -- bit-identical, a third fewer instructions, and no faster: 0.166 ms, the registers of
  // the second code path take the kernel from 8 to 6 waves per SIMD; and fetching a triangle only where it differs from
  // the previous pixel's (a third of the gathers, but one conditional block per pixel again): 0.160 ms = the old kernel.
  // 0.150-0.155 ms is 20 B/px at 4.4-4.5 TB/s: the kernel sits at what a mixed read/write stream reaches here.
  bool fg[VEC];
  int32_t sub = -1;
#pragma unroll
  for (int j = VEC - 1; j >= 0; --j) {
    fg[j] = tr[j] != -1;
    sub = fg[j] ? tr[j] : sub;
  }
  const unsigned long long any = __ballot(sub != -1);
  T d[VEC], b0[VEC], b1[VEC], b2[VEC];
  if (any != 0) {
    const int32_t wave_sub = __builtin_amdgcn_readlane(sub, __builtin_amdgcn_readfirstlane(__builtin_ctzll(any)));
    if (sub == -1) sub = wave_sub;
    int32_t f0[VEC], f1[VEC], f2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int32_t* face = vi_n + int64_t(fg[j] ? tr[j] : sub) * 3;
      f0[j] = face[0], f1[j] = face[1], f2[j] = face[2];
    }
    TriVerts<T> q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) q[j] = load_tri_verts<T>(v_n, f0[j], f1[j], f2[j]);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      RenderPix<T> r;
      int xj = x0 + j, yj = y;
      if constexpr (ANYW) {
        if (xj >= W) xj -= W, yj += 1; // W >= 4: at most one row further
      }
      render_math<T>(q[j], f0[j], f1[j], f2[j], xj, yj, r);
      b0[j] = fg[j] ? r.dinv0 * r.b0 * r.depth : T(0);
      b1[j] = fg[j] ? r.dinv1 * r.b1 * r.depth : T(0);
      b2[j] = fg[j] ? r.dinv2 * r.b2 * r.depth : T(0);
      d[j] = fg[j] ? r.depth : T(0);
    }
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) b0[j] = b1[j] = b2[j] = d[j] = T(0);
  }
  if constexpr (VEC == 4 && ANYW) {
    if (whole) {
      *reinterpret_cast<TQuad*>(depth_p) = TQuad{d[0], d[1], d[2], d[3]};
      *reinterpret_cast<TQuad*>(bary_p) = TQuad{b0[0], b0[1], b0[2], b0[3]};
      *reinterpret_cast<TQuad*>(bary_p + HW) = TQuad{b1[0], b1[1], b1[2], b1[3]};
      *reinterpret_cast<TQuad*>(bary_p + 2 * HW) = TQuad{b2[0], b2[1], b2[2], b2[3]};
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (pix0 + j < HW) depth_p[j] = d[j], bary_p[j] = b0[j], bary_p[HW + j] = b1[j], bary_p[2 * HW + j] = b2[j];
      }
    }
  } else if constexpr (VEC == 4) {
    using V4 = typename Vec4<T>::type;
    *reinterpret_cast<V4*>(depth_p) = V4{d[0], d[1], d[2], d[3]};
    *reinterpret_cast<V4*>(bary_p) = V4{b0[0], b0[1], b0[2], b0[3]};
    *reinterpret_cast<V4*>(bary_p + HW) = V4{b1[0], b1[1], b1[2], b1[3]};
    *reinterpret_cast<V4*>(bary_p + 2 * HW) = V4{b2[0], b2[1], b2[2], b2[3]};
  } else {
    depth_p[0] = d[0];
    bary_p[0] = b0[0];
    bary_p[HW] = b1[0];
    bary_p[2 * HW] = b2[0];
  }
}

// The nine vertex-position gradients of one covered pixel (render_kernel.cu:220-262) from the shared forward quantities
// `r` and the upstream gradients of the three barycentrics and the depth: g = {v0.x, v0.y, v0.z, v1.x, ..., v2.z}.
template <typename T>
__device__ __forceinline__ void render_backward_pixel(const RenderPix<T>& r, T dL_B0, T dL_B1, T dL_B2, T gD, T (&g)[9]) {
  const bool den_clamped = r.den != r.den_raw;
  const bool dinv_clamped = r.depth_inverse_eps != r.depth_inverse;

  // render_kernel.cu:225 : *grad_depth + dot(dL_bary_3D * d_inv, bary)
  const T dL_depth = gD + dL_B0 * r.dinv0 * r.b0 +
      dL_B1 * r.dinv1 * r.b1 + dL_B2 * r.dinv2 * r.b2;
  // (1 / s^2 as depth^2 and 1 / z_k^2 as dinv_k^2 -- four quarter-rate reciprocals less per pixel -- measured nothing:
  // 0.2035 vs 0.2040 ms, round 6)
  const T dL_dinv_s =
      dinv_clamped ? T(0) : quotient<false>(-dL_depth, r.depth_inverse * r.depth_inverse);

  const T dL_dinv0 = dL_B0 * r.b0 * r.depth + dL_dinv_s * r.b0;
  const T dL_dinv1 = dL_B1 * r.b1 * r.depth + dL_dinv_s * r.b1;
  const T dL_dinv2 = dL_B2 * r.b2 * r.depth + dL_dinv_s * r.b2;
  g[2] = r.z0c ? T(0) : quotient<false>(-dL_dinv0, r.z0e * r.z0e);
  g[5] = r.z1c ? T(0) : quotient<false>(-dL_dinv1, r.z1e * r.z1e);
  g[8] = r.z2c ? T(0) : quotient<false>(-dL_dinv2, r.z2e * r.z2e);

  const T dL_b0 = dL_B0 * r.dinv0 * r.depth + dL_dinv_s * r.dinv0;
  const T dL_b1 = dL_B1 * r.dinv1 * r.depth + dL_dinv_s * r.dinv1;
  const T dL_b2 = dL_B2 * r.dinv2 * r.depth + dL_dinv_s * r.dinv2;
  const T dL_b12x = -dL_b0 + dL_b1;
  const T dL_b12y = -dL_b0 + dL_b2;
  const T prex = quotient<false>(dL_b12x, r.den);
  const T prey = quotient<false>(dL_b12y, r.den);
  const T dL_den = den_clamped ? T(0) : -(prex * r.b1 + prey * r.b2);

  const T dL_vp0x = prex * r.v02y - prey * r.v01y;
  const T dL_vp0y = -prex * r.v02x + prey * r.v01x;
  const T dL_v02x = -prex * r.vp0y - dL_den * r.v01y;
  const T dL_v02y = prex * r.vp0x + dL_den * r.v01x;
  const T dL_v01x = prey * r.vp0y + dL_den * r.v02y;
  const T dL_v01y = -prey * r.vp0x - dL_den * r.v02x;

  g[0] = -dL_v02x - dL_v01x - dL_vp0x;
  g[1] = -dL_v02y - dL_v01y - dL_vp0y;
  g[3] = dL_v01x;
  g[4] = dL_v01y;
  g[6] = dL_v02x;
  g[7] = dL_v02y;
}

// Backward: a workgroup owns a 64 x 16 pixel tile; each of its 4 waves walks 4 adjacent rows of 64 pixels.  Lane = pixel:
// the nine per-pixel terms stay in registers, a segmented scan over the 16-lane rows leaves each run's sums in its
// last lane (segscatter.hpp: run_sums_rows16), and only those lanes update the wave's vertex table -- so every vertex
// costs one global atomic per component per 64 x 4 pixel tile.  No LDS staging, no barrier inside the loop.
// (Round 6: the same tile with FOUR pixels per lane -- five 16-byte loads per lane, one scan per 256 pixels, corner ids and
// vertices kept across a lane's pixels, ~25 % fewer instructions -- was built and is parity-green, and runs 0.226-0.237 ms
// against this kernel's 0.223-0.232 on the benchmark shape: its three dependent round trips per tile at five waves per
// SIMD cost what the instructions save.  Kernel text, ablation and the K-tiles-per-wave variants:
// profiles/r06/render_backward_quad.hip.txt.)
template <typename T>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? 8 : 4) void render_backward_kernel(
    const T* __restrict__ v, const int32_t* __restrict__ vi, const int32_t* __restrict__ index_img,
    const T* __restrict__ grad_depth_img, const T* __restrict__ grad_bary_img, int64_t V,
    int64_t vi_sN, int H, int W, int tiles_x, T* __restrict__ grad_v, int strip, int dbg) {
  constexpr int kWaves = kBlock / kWave;
  __shared__ int32_t t_keys[kWaves][kTableSlots];
  __shared__ TableAcc t_vals[kWaves][kTableSlots * 4];

  const int64_t HW = int64_t(H) * W;
  const int n = blockIdx.y;
  const int tile = tile_index(strip);
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  // wave-uniform by construction: saying so (readfirstlane) lets the row base pointers live in SGPRs and the loads take
  // a 32-bit lane offset instead of a 64-bit address computed per lane
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int lane = threadIdx.x & (kWave - 1);
  const int x = txi * kWave + lane;
  const T* v_n = v + int64_t(n) * V * 3;
  const int32_t* vi_n = vi + int64_t(n) * vi_sN;
  T* grad_v_n = grad_v + int64_t(n) * V * 3;

  table_init(t_keys[wave]);
  for (int i = lane; i < kTableSlots * 4; i += kWave) t_vals[wave][i] = 0;
  wave_lds_sync();

  // software pipeline over the 4 rows: index of row p+1 is requested at the top of row p, its
  // triangle's vertex ids before the reduction -- both dependent gathers fly under the current row.
  // (All four rows' indices and ids requested up front, the loop unrolled -- 62 registers, still eight waves: 0.210-0.218
  // against 0.215-0.234 ms on one box, within the noise of the comparison; not kept, round 6.)
  auto load_tr = [&](int pass) -> int32_t {
    const int yy = tyi * kTileRows + wave * (kTileRows / kWaves) + pass;
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
  for (int pass = 0; pass < kTileRows / kWaves; ++pass) {
    const int y = tyi * kTileRows + wave * (kTileRows / kWaves) + pass;
    const int64_t pix = int64_t(y) * W + x;
    const int32_t tr = tr_next;
    const int32_t cur[3] = {vn[0], vn[1], vn[2]};
    if (pass + 1 < kTileRows / kWaves) tr_next = load_tr(pass + 1);
    if (__ballot(tr != -1) == 0) { // whole row segment is background: nothing to do but keep the pipeline fed
      if (pass + 1 < kTileRows / kWaves) load_face(tr_next, vn);
      continue;
    }
    T g[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) g[j] = T(0);

    if (tr != -1 && DRTK_DBG(dbg, 4)) {
#pragma unroll
      for (int j = 0; j < 9; ++j) g[j] = T(1);
    } else if (tr != -1) {
      RenderPix<T> r;
      render_pix<T, false>(v_n, cur[0], cur[1], cur[2], x, y, r);
      const T* gb = grad_bary_img + int64_t(n) * 3 * HW + pix;
      render_backward_pixel<T>(r, gb[0], gb[HW], gb[2 * HW], grad_depth_img[int64_t(n) * HW + pix], g);
    }
    if (pass + 1 < kTileRows / kWaves) load_face(tr_next, vn);

    int dist;
    bool tail;
    run_rows16(tr, dist, tail);
    if (!DRTK_DBG(dbg, 1)) {
      run_sums_rows16<T, 9>(g, dist);
      if (tail && tr != -1 && !DRTK_DBG(dbg, 32)) {
        table_add<T, 3, 3>(t_keys[wave], t_vals[wave], 4, cur, g, grad_v_n, 3);
      }
    }
  }
  wave_lds_sync();
  table_flush<T>(t_keys[wave], t_vals[wave], 4, 3, grad_v_n, 3, 0);
}

template <typename T>
int render_impl(
    const T* v, const int32_t* vi, const int32_t* index_img, int64_t N, int64_t V, int64_t vi_sN,
    int64_t H, int64_t W, T* depth_img, T* bary_img, hipStream_t stream) {
  const int64_t HW = H * W;
  if (N * HW == 0) return DRTK_OK;
  const bool vec = (W % 4 == 0) && (reinterpret_cast<uintptr_t>(index_img) % 16 == 0) &&
      (reinterpret_cast<uintptr_t>(depth_img) % (4 * sizeof(T)) == 0) &&
      (reinterpret_cast<uintptr_t>(bary_img) % (4 * sizeof(T)) == 0);
  if (vec) {
    dim3 grid(static_cast<unsigned>(ceil_div(HW / 4, kBlock)), static_cast<unsigned>(N));
    DRTK_LAUNCH((render_kernel<T, 4>), grid, dim3(kBlock), 0, stream, v, vi, index_img, V, vi_sN, (int)H, (int)W, depth_img, bary_img, xcd_strip(ceil_div(16 * W, kBlock * 4)));
  } else if (W >= 4) {
    // widths that are not a multiple of four, views into flat buffers (round 4; before: one pixel per lane, 1.4x the time)
    dim3 grid(static_cast<unsigned>(ceil_div(ceil_div(HW, 4), kBlock)), static_cast<unsigned>(N));
    DRTK_LAUNCH((render_kernel<T, 4, true>), grid, dim3(kBlock), 0, stream, v, vi, index_img, V, vi_sN, (int)H, (int)W, depth_img, bary_img, xcd_strip(ceil_div(16 * W, kBlock * 4)));
  } else {
    dim3 grid(static_cast<unsigned>(ceil_div(HW, kBlock)), static_cast<unsigned>(N));
    DRTK_LAUNCH((render_kernel<T, 1>), grid, dim3(kBlock), 0, stream, v, vi, index_img, V, vi_sN, (int)H, (int)W, depth_img, bary_img, xcd_strip(ceil_div(16 * W, kBlock)));
  }
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

template <typename T>
int render_backward_impl(
    const T* v, const int32_t* vi, const int32_t* index_img, const T* grad_depth_img,
    const T* grad_bary_img, int64_t N, int64_t V, int64_t vi_sN, int64_t H, int64_t W, T* grad_v,
    hipStream_t stream) {
  if (N * V > 0) {
    if (fill_bytes_async(grad_v, 0, sizeof(T) * N * V * 3, stream) != DRTK_OK) return DRTK_ERR_LAUNCH;
  }
  const int64_t HW = H * W;
  if (N * HW == 0) return DRTK_OK;
  const int tiles_x = static_cast<int>(ceil_div(W, kWave)), tiles_y = static_cast<int>(ceil_div(H, kTileRows));
  dim3 grid(static_cast<unsigned>(int64_t(tiles_x) * tiles_y), static_cast<unsigned>(N));
  DRTK_LAUNCH((render_backward_kernel<T>), grid, dim3(kBlock), 0, stream, v, vi, index_img, grad_depth_img, grad_bary_img, V, vi_sN, (int)H, (int)W, tiles_x, grad_v, xcd_strip(int64_t(tiles_x) * (16 / kTileRows)), debug_flags());
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

bool bad_common(int64_t N, int64_t V, int64_t F, int64_t vi_sN, int64_t H, int64_t W) {
  return N < 0 || V < 0 || F < 0 || H < 0 || W < 0 || (vi_sN != 0 && vi_sN != F * 3) ||
      H * W >= (int64_t(1) << 31);
}

} // namespace
} // namespace drtk_amd

using namespace drtk_amd;

extern "C" int drtk_amd_render(
    drtk_dtype_t dtype, const void* v, const int32_t* vi, const int32_t* index_img, int64_t N,
    int64_t V, int64_t F, int64_t vi_sN, int64_t H, int64_t W, void* depth_img, void* bary_img,
    drtk_stream_t stream) {
  if (bad_common(N, V, F, vi_sN, H, W)) return DRTK_ERR_INVALID_ARGUMENT;
  if (N * H * W > 0 && (!index_img || !depth_img || !bary_img)) return DRTK_ERR_INVALID_ARGUMENT;
  if ((N * V > 0 && !v) || (F > 0 && !vi)) return DRTK_ERR_INVALID_ARGUMENT;
  if (dtype != DRTK_F32 && dtype != DRTK_F64) return DRTK_ERR_INVALID_ARGUMENT;
  const size_t es = dtype_size(dtype);
  DRTK_FOR_VIEW_SLICES(N, n0, n, drtk_amd_render(
      dtype, advance(v, n0 * V * 3, es), advance_typed(vi, n0 * vi_sN), advance_typed(index_img, n0 * H * W), n, V, F, vi_sN, H, W,
      advance(depth_img, n0 * H * W, es), advance(bary_img, n0 * 3 * H * W, es), stream))
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case DRTK_F32:
      return render_impl<float>(static_cast<const float*>(v), vi, index_img, N, V, vi_sN, H, W, static_cast<float*>(depth_img), static_cast<float*>(bary_img), s);
    case DRTK_F64:
      return render_impl<double>(static_cast<const double*>(v), vi, index_img, N, V, vi_sN, H, W, static_cast<double*>(depth_img), static_cast<double*>(bary_img), s);
    default:
      return DRTK_ERR_INVALID_ARGUMENT;
  }
}

extern "C" int drtk_amd_render_backward(
    drtk_dtype_t dtype, const void* v, const int32_t* vi, const int32_t* index_img,
    const void* grad_depth_img, const void* grad_bary_img, int64_t N, int64_t V, int64_t F,
    int64_t vi_sN, int64_t H, int64_t W, void* grad_v, drtk_stream_t stream) {
  if (bad_common(N, V, F, vi_sN, H, W)) return DRTK_ERR_INVALID_ARGUMENT;
  if (N * V > 0 && !grad_v) return DRTK_ERR_INVALID_ARGUMENT;
  if (N * H * W > 0 && (!index_img || !grad_depth_img || !grad_bary_img)) return DRTK_ERR_INVALID_ARGUMENT;
  if ((N * V > 0 && !v) || (F > 0 && !vi)) return DRTK_ERR_INVALID_ARGUMENT;
  if (dtype != DRTK_F32 && dtype != DRTK_F64) return DRTK_ERR_INVALID_ARGUMENT;
  const size_t es = dtype_size(dtype);
  DRTK_FOR_VIEW_SLICES(N, n0, n, drtk_amd_render_backward(
      dtype, advance(v, n0 * V * 3, es), advance_typed(vi, n0 * vi_sN), advance_typed(index_img, n0 * H * W),
      advance(grad_depth_img, n0 * H * W, es), advance(grad_bary_img, n0 * 3 * H * W, es), n, V, F, vi_sN, H, W,
      advance(grad_v, n0 * V * 3, es), stream))
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case DRTK_F32:
      return render_backward_impl<float>(static_cast<const float*>(v), vi, index_img, static_cast<const float*>(grad_depth_img), static_cast<const float*>(grad_bary_img), N, V, vi_sN, H, W, static_cast<float*>(grad_v), s);
    case DRTK_F64:
      return render_backward_impl<double>(static_cast<const double*>(v), vi, index_img, static_cast<const double*>(grad_depth_img), static_cast<const double*>(grad_bary_img), N, V, vi_sN, H, W, static_cast<double*>(grad_v), s);
    default:
      return DRTK_ERR_INVALID_ARGUMENT;
  }
}
