// edge_grad backward -- gradients of the rasterized image at visibility discontinuities
// (Pidhorskyi et al., arXiv 2405.02508), written per pixel into grad_v_pix_img [N,3,H,W].
//
// Reference: src/edge_grad/edge_grad_kernel.cu:217-449.  There every interior pixel acts as the
// centre of a (centre, right, down) stencil, reads img / grad_output for both pixels of a pair
// only where the triangle index changes (sparse 4-byte loads), and scatters 9 atomicAdds per
// pixel (zeros included) into a zero-filled output.
//
// CDNA4 mapping, two passes, no atomics, no zero-fill:
//   A. edge_dots_kernel  -- dense stream over img and grad_output with 16-byte loads: for every
//      pixel the two pair terms  gdx = sum_c (img_R - img_C) * 0.5 (g_R + g_C)  and the same
//      downwards (gdy).  A wave walks a 252-pixel-wide strip down R rows keeping the previous row in
//      registers, so each row is read (R+1)/R times instead of twice.  8 B/px of scratch out.
//   B. edge_gather_kernel -- every pixel OWNS its output: it classifies the four pairs it takes
//      part in (as centre of its own stencil, as the right pixel of its left neighbour's, as the
//      down pixel of its upper neighbour's) exactly like the reference and sums the at most four
//      contributions in the single-threaded reference order.
// Pairs are classified twice (once by each member), which costs ALU only.
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

// Four adjacent pixels of an image row as ONE 16 / 32-byte access that needs the alignment of its element only: rows
// of an image whose width is not a multiple of four start at any element, and so do contiguous views into a flat buffer.
// The hardware serves dword-aligned wide accesses (a wave's 64 such loads touch one cache line more than aligned ones).
template <typename T>
using Quad = T __attribute__((ext_vector_type(4), aligned(sizeof(T))));

// ---------------------------------------------------------------------------------------------
// Pass A
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
struct Row {
  T v[VEC];
  T next; // pixel x + VEC (first pixel of the next lane)
};

template <typename T, int VEC>
__device__ __forceinline__ Row<T, VEC> load_row(
    const T* __restrict__ plane, int64_t row_off, int x, int W, bool x_ok, int lane) {
  Row<T, VEC> r;
  if (x_ok) {
    if constexpr (VEC == 4) {
      if (x + 4 <= W) {
        // (as non-temporal loads -- both images are read once per launch -- the fused route ran 0.90 instead of 0.74 ms,
        // round 6: the row below a wave's strip is its sibling wave's first row and comes out of the cache the hint gives up)
        const Quad<T> q = *reinterpret_cast<const Quad<T>*>(plane + row_off + x);
        r.v[0] = q.x, r.v[1] = q.y, r.v[2] = q.z, r.v[3] = q.w;
      } else { // the lane that holds the end of a row whose width is not a multiple of four
#pragma unroll
        for (int j = 0; j < 4; ++j) r.v[j] = x + j < W ? plane[row_off + x + j] : T(0);
      }
    } else {
      r.v[0] = plane[row_off + x];
    }
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) r.v[j] = T(0);
  }
  r.next = __shfl_down(r.v[0], 1); // lane 63's is never used: it is the halo lane (see edge_dots_kernel)
  return r;
}

// One wave = strip of 63*VEC pixels (+ VEC halo pixels in lane 63) x R rows; the WAVES waves of a workgroup are stacked vertically
// on the same pixel columns (WAVES*R rows per workgroup), so the extra "row below" every wave needs
// is the first row of its sibling wave and is served by this CU's L1 -- HBM sees each row
// (WAVES*R+1)/(WAVES*R) times.
// gdx[y][x] pairs (x,y)-(x+1,y), gdy[y][x] pairs (x,y)-(x,y+1); accumulation over channels in
// ascending order (edge_grad_kernel.cu:353-380).
template <typename T, int VEC, int R, int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void edge_dots_kernel(
    const T* __restrict__ img, const T* __restrict__ grad_output, const int32_t* __restrict__ index_img, int C,
    int H, int W, int strips_x, T* __restrict__ gdx, T* __restrict__ gdy, int strip, T* __restrict__ zero_out,
    int64_t zero_count) {
  // the accumulator of the pass that follows this one (grad_v_pix of the fused route) is cleared here, spread over the
  // whole grid: one launch less per backward pass, which is what a small scene's step is made of
  if (zero_out) {
    const int64_t step = int64_t(gridDim.x) * gridDim.y * blockDim.x;
    for (int64_t i = (int64_t(blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < zero_count; i += step) zero_out[i] = T(0);
  }
  const int64_t HW = int64_t(H) * W;
  const int n = blockIdx.y;
  const int tile = tile_index(strip);
  const int by = tile / strips_x, sx = tile - by * strips_x;
  const int lane = threadIdx.x & (kWave - 1);
  const int y0 = (by * WAVES + threadIdx.x / kWave) * R;
  if (y0 >= H) return;
  // strips are 63 lanes wide: lane 63 holds the first pixels of the next strip and only feeds lane 62's pair term.  (A
  // one-lane load of that pixel per row, plane and channel was half of this kernel's memory instructions: 0.783-0.801 ->
  // 0.772-0.773 ms for the fused route on the bench shape.)
  const int x = (sx * (kWave - 1) + lane) * VEC;
  const T* img_n = img + int64_t(n) * C * HW;
  const T* go_n = grad_output + int64_t(n) * C * HW;

  // Pair terms are only ever read where the triangle index changes, and a pair of two background
  // pixels never does: a lane whose pixels, their left / right neighbours and the halo row are all
  // background neither loads nor stores anything (its part of the workspace is never read), so the
  // background of the image costs 4 B/px of index instead of 8C B/px of img + grad_output.
  bool any_fg = false;
  {
    const int32_t* idx_n = index_img + int64_t(n) * HW;
#pragma unroll
    for (int r = 0; r <= R; ++r) {
      const int y = y0 + r;
      if (y < H) { // wave-uniform
        bool fg = false;
        int32_t first = -1, last = -1;
        if (x < W) {
          if constexpr (VEC == 4) {
            Quad<int32_t> q;
            if (x + 4 <= W) {
              q = *reinterpret_cast<const Quad<int32_t>*>(idx_n + int64_t(y) * W + x);
            } else {
              const int32_t* rp = idx_n + int64_t(y) * W + x;
              q.x = rp[0], q.y = x + 1 < W ? rp[1] : -1, q.z = x + 2 < W ? rp[2] : -1, q.w = -1;
            }
            fg = (q.x & q.y & q.z & q.w) != -1;
            first = q.x, last = q.w;
          } else {
            first = last = idx_n[int64_t(y) * W + x];
            fg = first != -1;
          }
        }
        int32_t left = __shfl_up(last, 1), right = __shfl_down(first, 1);
        if (lane == 0) left = (x >= 1 && x < W) ? idx_n[int64_t(y) * W + x - 1] : -1;
        if (lane == kWave - 1) right = (x + VEC < W) ? idx_n[int64_t(y) * W + x + VEC] : -1;
        any_fg = any_fg || fg || left != -1 || right != -1;
      }
    }
  }
  if (__ballot(any_fg) == 0) return;
  const bool x_ok = x < W && any_fg;

  T ax[R][VEC], ay[R][VEC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < VEC; ++j) ax[r][j] = ay[r][j] = T(0);

  for (int c = 0; c < C; ++c) {
    const T* ip = img_n + int64_t(c) * HW;
    const T* gp = go_n + int64_t(c) * HW;
    Row<T, VEC> ri[R + 1], rg[R + 1];
#pragma unroll
    for (int r = 0; r <= R; ++r) {
      const bool y_ok = (y0 + r) < H; // wave-uniform
      // (round 6: rows 1 .. R-1, read by this wave alone, as non-temporal loads: 0.763 vs 0.750 ms for the fused route -- no)
      ri[r] = load_row<T, VEC>(ip, int64_t(y0 + r) * W, x, W, x_ok && y_ok, lane);
      rg[r] = load_row<T, VEC>(gp, int64_t(y0 + r) * W, x, W, x_ok && y_ok, lane);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const T inext = (j + 1 < VEC) ? ri[r].v[(j + 1) % VEC] : ri[r].next;
        const T gnext = (j + 1 < VEC) ? rg[r].v[(j + 1) % VEC] : rg[r].next;
        ax[r][j] += (inext - ri[r].v[j]) * (T(0.5) * (gnext + rg[r].v[j]));
        ay[r][j] += (ri[r + 1].v[j] - ri[r].v[j]) * (T(0.5) * (rg[r + 1].v[j] + rg[r].v[j]));
      }
    }
  }
  if (!x_ok || lane == kWave - 1) return;
  T* ox = gdx + int64_t(n) * HW;
  T* oy = gdy + int64_t(n) * HW;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int y = y0 + r;
    if (y >= H) break;
    const int64_t o = int64_t(y) * W + x;
    if constexpr (VEC == 4) {
      if (x + 4 <= W) {
        *reinterpret_cast<Quad<T>*>(ox + o) = Quad<T>{ax[r][0], ax[r][1], ax[r][2], ax[r][3]};
        *reinterpret_cast<Quad<T>*>(oy + o) = Quad<T>{ay[r][0], ay[r][1], ay[r][2], ay[r][3]};
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x + j < W) ox[o + j] = ax[r][j], oy[o + j] = ay[r][j];
      }
    } else {
      ox[o] = ax[r][0];
      oy[o] = ay[r][0];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Pass B
// ---------------------------------------------------------------------------------------------
// edge_grad_kernel.cu:18-28,72-87
template <typename T>
struct TriInfo {
  T p0x, p0y, p1x, p1y;
  T v01x, v01y, v02x, v02y, v12x, v12y;
  T den;
  T nx, ny, nz; // face normal, filled lazily
  int32_t i0, i1, i2;
};

template <typename T>
__device__ __forceinline__ void load_tri(
    const T* __restrict__ v_n, const int32_t* __restrict__ vi_n, int32_t idx, TriInfo<T>& t) {
  // invalid triangles use vertex (0,0,0) like the reference (edge_grad_kernel.cu:296-301)
  t.i0 = t.i1 = t.i2 = 0;
  if (idx >= 0) {
    const int32_t* f = vi_n + int64_t(idx) * 3;
    t.i0 = f[0], t.i1 = f[1], t.i2 = f[2];
  }
  t.p0x = v_n[3 * (int64_t)t.i0 + 0];
  t.p0y = v_n[3 * (int64_t)t.i0 + 1];
  t.p1x = v_n[3 * (int64_t)t.i1 + 0];
  t.p1y = v_n[3 * (int64_t)t.i1 + 1];
  const T p2x = v_n[3 * (int64_t)t.i2 + 0], p2y = v_n[3 * (int64_t)t.i2 + 1];
  t.v01x = t.p1x - t.p0x;
  t.v01y = t.p1y - t.p0y;
  t.v02x = p2x - t.p0x;
  t.v02y = p2y - t.p0y;
  t.v12x = p2x - t.p1x;
  t.v12y = p2y - t.p1y;
  t.den = t.v01x * t.v02y - t.v01y * t.v02x;
}

// edge_grad_kernel.cu:30-70 : rasterizer's top-left rule on NON-canonical edge functions
template <typename T>
__device__ __forceinline__ bool pix_in_tri(const TriInfo<T>& t, int x, int y) {
  if (t.den != T(0)) {
    const T px = static_cast<T>(x), py = static_cast<T>(y);
    const T vp0x = px - t.p0x, vp0y = py - t.p0y;
    const T vp1x = px - t.p1x, vp1y = py - t.p1y;
    T b0 = vp1y * t.v12x - vp1x * t.v12y;
    T b1 = vp0x * t.v02y - vp0y * t.v02x;
    T b2 = vp0y * t.v01x - vp0x * t.v01y;
    const T s = t.den > T(0) ? T(1) : (t.den < T(0) ? T(-1) : T(0));
    b0 *= s;
    b1 *= s;
    b2 *= s;
    const bool inside = (b0 >= T(0)) && (b1 >= T(0)) && (b2 >= T(0));
    const bool on0 = b0 == T(0), on1 = b1 == T(0), on2 = b2 == T(0);
    const bool pos = t.den > T(0);
    const bool tl0 = pos ? (t.v12y < T(0) || (t.v12y == T(0) && t.v12x > T(0)))
                         : (t.v12y > T(0) || (t.v12y == T(0) && t.v12x < T(0)));
    const bool tl1 = pos ? (t.v02y > T(0) || (t.v02y == T(0) && t.v02x < T(0)))
                         : (t.v02y < T(0) || (t.v02y == T(0) && t.v02x > T(0)));
    const bool tl2 = pos ? (t.v01y < T(0) || (t.v01y == T(0) && t.v01x > T(0)))
                         : (t.v01y > T(0) || (t.v01y == T(0) && t.v01x < T(0)));
    return inside && !((on0 && !tl0) || (on1 && !tl1) || (on2 && !tl2));
  }
  return false;
}

// Correctly rounded square roots, as the reference's host path gets from libm.  NOT `__fsqrt_rn`: without
// OCML_BASIC_ROUNDED_OPERATIONS the toolchain defines it as __ocml_native_sqrt_f32 (the bare 1-ulp v_sqrt_f32,
// __clang_hip_math.h), and one ulp in a normal's length is enough to flip the sign that get_dp_dr takes from a
// near-zero `d` -- a difference of 2 * max_dp_dr in the output of that pixel.  The builtin is lowered with the
// compiler's correction steps (HIP's default -fhip-fp32-correctly-rounded-divide-sqrt).
__device__ __forceinline__ float sqrt_t(float x) {
  return __builtin_sqrtf(x);
}
__device__ __forceinline__ double sqrt_t(double x) {
  return __builtin_sqrt(x);
}

// edge_grad_kernel.cu:89-100 ; normalize = v * (1 / sqrt(dot)) as on the reference's host path
template <typename T>
__device__ __forceinline__ void tri_normal(const T* __restrict__ v_n, TriInfo<T>& t) {
  const T* p0 = v_n + 3 * (int64_t)t.i0;
  const T* p1 = v_n + 3 * (int64_t)t.i1;
  const T* p2 = v_n + 3 * (int64_t)t.i2;
  const T ax = p0[0] - p2[0], ay = p0[1] - p2[1], az = p0[2] - p2[2];
  const T bx = p1[0] - p0[0], by = p1[1] - p0[1], bz = p1[2] - p0[2];
  const T cx = ay * bz - az * by;
  const T cy = az * bx - ax * bz;
  const T cz = ax * by - ay * bx;
  const T inv = T(1.0) / sqrt_t(cx * cx + cy * cy + cz * cz);
  t.nx = cx * inv;
  t.ny = cy * inv;
  t.nz = cz * inv;
}

// edge_grad_kernel.cu:102-203
template <typename T>
__device__ __forceinline__ void get_dp_dr(T nvx, T nvy, T nfx, T nfy, T M, T& ox, T& oy) {
  const T inv_v = T(1.0) / sqrt_t(nvx * nvx + nvy * nvy);
  const T nv_x = nvx * inv_v, nv_y = nvy * inv_v;
  const T inv_f = T(1.0) / sqrt_t(nfx * nfx + nfy * nfy);
  const T nf_x = nfx * inv_f, nf_y = nfy * inv_f;
  const T bx = -nf_y, by = nf_x;
  const T d = bx * nv_x + by * nv_y;
  T q;
  if (M > T(0)) {
    const T abs_d = d < T(0) ? -d : d;
    const T abs_bx = bx < T(0) ? -bx : bx;
    const T lim = abs_bx / M;
    const T m = abs_d < lim ? lim : abs_d;
    const T safe_d = (d >= T(0) ? T(1) : T(-1)) * epsclamp(m);
    q = bx / safe_d;
  } else {
    q = bx / epsclamp(d);
  }
  ox = q * nv_x;
  oy = q * nv_y;
}

// Contributions of one pixel pair A (centre role) - B (right/down role).
//   AXIS 0: B = A + (1,0), normals use (x,z);  AXIS 1: B = A + (0,1), normals use (y,z).
// gA/gB: in-plane component (x or y), zA/zB: z component.  edge_grad_kernel.cu:303-425.
template <typename T, int AXIS>
__device__ __forceinline__ void eval_pair(
    const T* __restrict__ v_n, TriInfo<T>& tA, TriInfo<T>& tB, int32_t idxA, int32_t idxB, int xa,
    int ya, T grad_dot, T M, T& gA, T& zA, T& gB, T& zB) {
  gA = zA = gB = zB = T(0);
  const bool a_valid = idxA >= 0, b_valid = idxB >= 0;
  const bool both = a_valid && b_valid;
  const int xb = xa + (AXIS == 0 ? 1 : 0), yb = ya + (AXIS == 1 ? 1 : 0);
  const bool a_in_b = both && pix_in_tri(tB, xa, ya);
  const bool b_in_a = both && pix_in_tri(tA, xb, yb);
  const bool a_over_b = a_in_b && !b_in_a;
  const bool b_over_a = b_in_a && !a_in_b;
  const bool inter = a_in_b && b_in_a;
  const bool adjacent = both && !a_in_b && !b_in_a;
  if (!inter) {
    gA = (!a_valid || b_over_a || adjacent) ? T(0) : grad_dot;
    gB = (!b_valid || a_over_b || adjacent) ? T(0) : grad_dot;
  } else {
    tri_normal(v_n, tA);
    tri_normal(v_n, tB);
    const T a_in = AXIS == 0 ? tA.nx : tA.ny;
    const T b_in = AXIS == 0 ? tB.nx : tB.ny;
    T dx, dz;
    get_dp_dr<T>(a_in, tA.nz, b_in, tB.nz, M, dx, dz);
    gA = grad_dot * dx;
    zA = grad_dot * dz;
    get_dp_dr<T>(b_in, tB.nz, a_in, tA.nz, M, dx, dz);
    gB = grad_dot * dx;
    zB = grad_dot * dz;
  }
}

// Output of one pixel that takes part in at least one pair with differing indices ("edge pixel").
// ic/ir/id/il/iu: indices of the pixel and its right/down/left/up neighbours, already forced equal
// to ic where the pair is outside the reference's stencil domain.
template <typename T>
__device__ __forceinline__ void edge_pixel(
    const T* __restrict__ v_n, const int32_t* __restrict__ vi_n, const T* __restrict__ gdx_n,
    const T* __restrict__ gdy_n, int64_t pix, int x, int y, int W, int32_t ic, int32_t ir, int32_t id,
    int32_t il, int32_t iu, T M, T& ox, T& oy, T& oz, int32_t* centre_vids = nullptr) {
  T cx = T(0), cy = T(0), cz = T(0); // own centre contributions (gc)
  T rx = T(0), rz = T(0);            // as right pixel of the left neighbour's stencil (gr)
  T dy = T(0), dz = T(0);            // as down pixel of the upper neighbour's stencil (gd)
  TriInfo<T> tc;
  load_tri<T>(v_n, vi_n, ic, tc);
  if (centre_vids) {
    centre_vids[0] = tc.i0, centre_vids[1] = tc.i1, centre_vids[2] = tc.i2;
  }
  T gA, zA, gB, zB;
  if (ic != ir) {
    TriInfo<T> tn;
    load_tri<T>(v_n, vi_n, ir, tn);
    eval_pair<T, 0>(v_n, tc, tn, ic, ir, x, y, gdx_n[pix], M, gA, zA, gB, zB);
    cx += gA;
    cz += zA;
  }
  if (ic != id) {
    TriInfo<T> tn;
    load_tri<T>(v_n, vi_n, id, tn);
    eval_pair<T, 1>(v_n, tc, tn, ic, id, x, y, gdy_n[pix], M, gA, zA, gB, zB);
    cy += gA;
    cz += zA;
  }
  if (il != ic) {
    TriInfo<T> tn;
    load_tri<T>(v_n, vi_n, il, tn);
    eval_pair<T, 0>(v_n, tn, tc, il, ic, x - 1, y, gdx_n[pix - 1], M, gA, zA, gB, zB);
    rx += gB;
    rz += zB;
  }
  if (iu != ic) {
    TriInfo<T> tn;
    load_tri<T>(v_n, vi_n, iu, tn);
    eval_pair<T, 1>(v_n, tn, tc, iu, ic, x, y - 1, gdy_n[pix - W], M, gA, zA, gB, zB);
    dy += gB;
    dz += zB;
  }
  // edge_grad_kernel.cu:427-445 negates and accumulates; order = the single-threaded reference
  // order (upper neighbour's stencil, left neighbour's stencil, own stencil).
  ox = (T(0) + (-rx)) + (-cx);
  oy = (T(0) + (-dy)) + (-cy);
  oz = ((T(0) + (-dz)) + (-rz)) + (-cz);
}

// Generic fallback (any W): one pixel per lane.
template <typename T>
__global__ __launch_bounds__(kBlock) void edge_gather_kernel(
    const T* __restrict__ v_pix, const int32_t* __restrict__ vi,
    const int32_t* __restrict__ index_img, const T* __restrict__ gdx, const T* __restrict__ gdy,
    int64_t V, int64_t vi_sN, int H, int W, T M, T* __restrict__ out) {
  const int64_t HW = int64_t(H) * W;
  const int n = blockIdx.y;
  const int64_t pix = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (pix >= HW) return;
  const int y = static_cast<int>(pix / W);
  const int x = static_cast<int>(pix - int64_t(y) * W);
  const int32_t* idx_n = index_img + int64_t(n) * HW;

  // stencil domain of the reference: centres with x < W-1 && y < H-1 (edge_grad_kernel.cu:270)
  const bool own = (x < W - 1) && (y < H - 1);
  const bool left = (x >= 1) && (y < H - 1); // pair (x-1,y)-(x,y), centre (x-1,y)
  const bool up = (y >= 1) && (x < W - 1);   // pair (x,y-1)-(x,y), centre (x,y-1)
  const int32_t ic = idx_n[pix];
  const int32_t ir = own ? idx_n[pix + 1] : ic;
  const int32_t id = own ? idx_n[pix + W] : ic;
  const int32_t il = left ? idx_n[pix - 1] : ic;
  const int32_t iu = up ? idx_n[pix - W] : ic;
  T ox = T(0), oy = T(0), oz = T(0);
  if (ic != ir || ic != id || ic != il || ic != iu) {
    edge_pixel<T>(
        v_pix + int64_t(n) * V * 3, vi + int64_t(n) * vi_sN, gdx + int64_t(n) * HW, gdy + int64_t(n) * HW, pix,
        x, y, W, ic, ir, id, il, iu, M, ox, oy, oz);
  }
  T* o = out + int64_t(n) * 3 * HW + pix;
  o[0] = ox;
  o[HW] = oy;
  o[2 * HW] = oz;
}

// W % 4 == 0: a wave owns 256 consecutive pixels (4 per lane, 16-byte index loads and output
// stores).  Only ~1 pixel in 5 is an edge pixel, scattered, so running the heavy classification
// per lane-pixel would execute it for every wave at ~20 % lane utilisation; instead the edge
// pixels of the 256 are COMPACTED (ballot + popcount ranks -> LDS list) and classified 64 at a
// time at near-full utilisation; results go through an LDS tile and leave as dense float4 stores.
template <typename T>
__global__ __launch_bounds__(kBlock) void edge_gather4_kernel(
    const T* __restrict__ v_pix, const int32_t* __restrict__ vi,
    const int32_t* __restrict__ index_img, const T* __restrict__ gdx, const T* __restrict__ gdy,
    int64_t V, int64_t vi_sN, int H, int W, T M, T* __restrict__ out, int strip) {
  using V4 = typename Vec4<T>::type;
  constexpr int kWaves = kBlock / kWave;
  constexpr int kPix = kWave * 4;
  __shared__ __attribute__((aligned(16))) T s_out[kWaves][3][kPix];
  __shared__ uint16_t s_list[kWaves][kPix];
  __shared__ int32_t s_nb[kWaves][4][kPix]; // right/down/left/up neighbour index per pixel

  const int64_t HW = int64_t(H) * W;
  const int n = blockIdx.y;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  const int64_t wave_pix0 = (int64_t(tile_index(strip)) * kWaves + wave) * kPix;
  if (wave_pix0 >= HW) return;
  const int64_t pix0 = wave_pix0 + lane * 4;
  const bool in_range = pix0 < HW;
  const int32_t* idx_n = index_img + int64_t(n) * HW;
  const int y = in_range ? static_cast<int>(pix0 / W) : 0;
  const int x0 = in_range ? static_cast<int>(pix0 - int64_t(y) * W) : 0;

  int32_t c[4] = {-1, -1, -1, -1}, u[4], d[4];
  if (in_range) {
    const int4 q = *reinterpret_cast<const int4*>(idx_n + pix0);
    c[0] = q.x, c[1] = q.y, c[2] = q.z, c[3] = q.w;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) u[j] = d[j] = c[j];
  if (in_range && y >= 1) {
    const int4 q = *reinterpret_cast<const int4*>(idx_n + pix0 - W);
    u[0] = q.x, u[1] = q.y, u[2] = q.z, u[3] = q.w;
  }
  if (in_range && y < H - 1) {
    const int4 q = *reinterpret_cast<const int4*>(idx_n + pix0 + W);
    d[0] = q.x, d[1] = q.y, d[2] = q.z, d[3] = q.w;
  }
  // horizontal neighbours across lanes: previous lane's last / next lane's first pixel
  int32_t lprev = __shfl_up(c[3], 1), rnext = __shfl_down(c[0], 1);
  if (in_range && lane == 0 && x0 >= 1) lprev = idx_n[pix0 - 1];
  if (in_range && lane == kWave - 1 && x0 + 4 < W) rnext = idx_n[pix0 + 4];

  bool e[4];
  int32_t nr[4], nd[4], nl[4], nu[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int x = x0 + j;
    // stencil domain of the reference: centres with x < W-1 && y < H-1 (edge_grad_kernel.cu:270)
    const bool own = in_range && (x < W - 1) && (y < H - 1);
    const bool left = in_range && (x >= 1) && (y < H - 1);
    const bool up = in_range && (y >= 1) && (x < W - 1);
    nr[j] = own ? (j < 3 ? c[(j + 1) & 3] : rnext) : c[j];
    nd[j] = own ? d[j] : c[j];
    nl[j] = left ? (j > 0 ? c[(j + 3) & 3] : lprev) : c[j];
    nu[j] = up ? u[j] : c[j];
    e[j] = c[j] != nr[j] || c[j] != nd[j] || c[j] != nl[j] || c[j] != nu[j];
  }

  // zero the output tile, publish neighbour indices, compact the edge pixels
  const V4 zero4 = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
  for (int k = 0; k < 3; ++k) *reinterpret_cast<V4*>(&s_out[wave][k][lane * 4]) = zero4;
  *reinterpret_cast<int4*>(&s_nb[wave][0][lane * 4]) = make_int4(nr[0], nr[1], nr[2], nr[3]);
  *reinterpret_cast<int4*>(&s_nb[wave][1][lane * 4]) = make_int4(nd[0], nd[1], nd[2], nd[3]);
  *reinterpret_cast<int4*>(&s_nb[wave][2][lane * 4]) = make_int4(nl[0], nl[1], nl[2], nl[3]);
  *reinterpret_cast<int4*>(&s_nb[wave][3][lane * 4]) = make_int4(nu[0], nu[1], nu[2], nu[3]);
  const unsigned long long lt = (1ull << lane) - 1ull;
  int total = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned long long m = __ballot(e[j]);
    if (e[j]) s_list[wave][total + __popcll(m & lt)] = static_cast<uint16_t>(lane * 4 + j);
    total += __popcll(m);
  }
  wave_lds_sync();

  const T* v_n = v_pix + int64_t(n) * V * 3;
  const int32_t* vi_n = vi + int64_t(n) * vi_sN;
  const T* gdx_n = gdx + int64_t(n) * HW;
  const T* gdy_n = gdy + int64_t(n) * HW;
  for (int t0 = 0; t0 < total; t0 += kWave) {
    const int t = t0 + lane;
    if (t < total) {
      const int local = s_list[wave][t];
      const int64_t pix = wave_pix0 + local;
      const int py = static_cast<int>(pix / W);
      const int px = static_cast<int>(pix - int64_t(py) * W);
      const int32_t ic = idx_n[pix];
      T ox, oy, oz;
      edge_pixel<T>(
          v_n, vi_n, gdx_n, gdy_n, pix, px, py, W, ic, s_nb[wave][0][local], s_nb[wave][1][local],
          s_nb[wave][2][local], s_nb[wave][3][local], M, ox, oy, oz);
      s_out[wave][0][local] = ox;
      s_out[wave][1][local] = oy;
      s_out[wave][2][local] = oz;
    }
  }
  wave_lds_sync();
  if (in_range) {
    T* o = out + int64_t(n) * 3 * HW + pix0;
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<V4*>(o + int64_t(k) * HW) = *reinterpret_cast<const V4*>(&s_out[wave][k][lane * 4]);
  }
}

// Fused pass B for the default route of drtk.edge_grad_estimator (no v_pix_img hook): instead of
// materialising grad_v_pix_img [N,3,H,W] (80 % exact zeros) for a separate C=3 interpolate backward to
// scatter, the contributions are multiplied by the pixel's barycentrics and scattered to the triangle's
// vertices right away (edge_grad_estimator.py:168-176 + interpolate_kernel.cu:271-279 fused).
// A per-pixel formulation (as in edge_gather4_kernel, which must own its output pixel) evaluates every
// differing pixel pair twice -- once from each end -- inside four divergent branches; that arithmetic was
// 60 % of such a kernel, which is instruction-issue bound (1.27 ms fused; this kernel: 0.96 ms).
// Here the unit of work is the PAIR: a wave compacts the horizontal pairs (x,y)-(x+1,y) and the vertical
// pairs (x,y)-(x,y+1) of its 256 x kPairRows tile whose indices differ (same stencil domain, edge_grad_kernel.cu:270),
// one pair per lane, evaluates it ONCE with the axis as a compile-time constant, and scatters both ends'
// contributions (pixel A: -gA on the axis, -zA on z; pixel B: -gB, -zB; each times its own pixel's
// barycentrics) through the run reduction of segscatter.hpp with six corner slots (A0..A2, B0..B2) and
// two components each.
// Any width, any element-aligned pointers: the index rows are the only wide global access of this kernel (Quad, above),
// everything else is per pair.  (Rounds 2-3 took this route for W % 4 == 0 only and sent other widths through the
// unfused one: 2.2x the time at 2048 x 2046.)
#ifndef DRTK_PAIR_ROWS
#define DRTK_PAIR_ROWS 2
#endif
// Image rows per wave of edge_scatter_pairs_kernel (its tile: 256 x kPairRows pixels).  The kernel is bound by the
// dependent loads of each 64-pair round (list -> index -> corners -> vertices) at six waves per SIMD (80 registers), and its waves'
// work varies with the number of pairs in their tile: more, shorter waves keep more rounds in flight and even out the
// tail.  Fused route on one box (kernel_bench): 4 rows 0.799-0.804 ms, 2 rows 0.759-0.762, 1 row 0.800 (the halo row
// doubles the index traffic and the lists get short), 8 rows 0.830; 1M triangles at 2 x 4096^2: 0.468 -> 0.409 ms.
// Narrower tiles instead (2 pixels per lane, 128 x 2 or 128 x 4): 0.80-0.84 ms against 0.72-0.76 -- the 8-byte index
// loads and half-empty pair rounds cost more than the extra waves bring.
// (Round 6: the tile's index rows staged in LDS so that a round's two triangle ids are LDS reads -- one of the three dependent
// round trips less -- for 12.5 KB more LDS and four more registers: five waves per SIMD instead of six, 0.771-0.780 against
// 0.750-0.759 ms, 250k triangles 0.819 against 0.791.  Slower; not kept.)
constexpr int kPairRows = DRTK_PAIR_ROWS;
template <typename T>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? 4 : 2) void edge_scatter_pairs_kernel(
    const T* __restrict__ v_pix, const int32_t* __restrict__ vi, const int32_t* __restrict__ index_img,
    const T* __restrict__ bary_img, const T* __restrict__ gdx, const T* __restrict__ gdy, int64_t V, int64_t vi_sN,
    int H, int W, int strips_x, T M, T* __restrict__ grad_v_pix, int strip) {
  constexpr int kWaves = kBlock / kWave;
  constexpr int kRows = kPairRows;
  constexpr int kCap = kWave * 4 * 2 * 2; // pairs of two rows, both axes: the most one list build can hold
  __shared__ uint16_t s_list[kWaves][kCap];
#ifndef DRTK_EDGE_SLOTS
#define DRTK_EDGE_SLOTS 64 // A/B on one box (build.py --variant): 128 slots 0.810-0.815 ms, 64 slots 0.812-0.814 -- no difference
#endif
  constexpr int kSlots = DRTK_EDGE_SLOTS;
  __shared__ int32_t t_keys[kWaves][kSlots];
  __shared__ TableAcc t_vals[kWaves][kSlots * 4];

  const int64_t HW = int64_t(H) * W;
  const int n = blockIdx.y;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  const int wg = tile_index(strip) * kWaves + wave;
  const int ry = wg / strips_x, sx = wg - ry * strips_x;
  const int y_base = ry * kRows;
  if (y_base >= H) return;
  const int x0 = (sx * kWave + lane) * 4;
  const bool in_x = x0 < W;
  const int32_t* idx_n = index_img + int64_t(n) * HW;
  const T* v_n = v_pix + int64_t(n) * V * 3;
  const int32_t* vi_n = vi + int64_t(n) * vi_sN;
  const T* gdx_n = gdx + int64_t(n) * HW;
  const T* gdy_n = gdy + int64_t(n) * HW;
  const T* bary_n = bary_img + int64_t(n) * 3 * HW;
  T* grad_n = grad_v_pix + int64_t(n) * V * 3;

  table_init<kSlots>(t_keys[wave]);
  for (int i = lane; i < kSlots * 4; i += kWave) t_vals[wave][i] = 0;

  // index rows y_base .. y_base+kRows of this lane's 4 pixels, one batch
  int32_t row[kRows + 1][4];
#pragma unroll
  for (int r = 0; r <= kRows; ++r) {
    const int y = y_base + r;
    if (in_x && y < H) {
      const int32_t* rp = idx_n + int64_t(y) * W + x0;
      if (x0 + 4 <= W) {
        const Quad<int32_t> q = *reinterpret_cast<const Quad<int32_t>*>(rp);
        row[r][0] = q.x, row[r][1] = q.y, row[r][2] = q.z, row[r][3] = q.w;
      } else { // the end of a row whose width is not a multiple of four
#pragma unroll
        for (int j = 0; j < 4; ++j) row[r][j] = x0 + j < W ? rp[j] : -1;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) row[r][j] = -1;
    }
  }
  // pair flags of the lane's pixels: bit j of hf[r] / vf[r] = horizontal / vertical pair starting at pixel j of row r
  uint32_t hf[kRows], vf[kRows];
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const int y = y_base + r;
    int32_t rnext = __shfl_down(row[r][0], 1);
    if (in_x && y < H && lane == kWave - 1 && x0 + 4 < W) rnext = idx_n[int64_t(y) * W + x0 + 4];
    hf[r] = vf[r] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool own = in_x && (x0 + j < W - 1) && (y < H - 1); // stencil domain of the reference
      const int32_t nr = j < 3 ? row[r][(j + 1) & 3] : rnext;
      if (own && row[r][j] != nr) hf[r] |= 1u << j;
      if (own && row[r][j] != row[r + 1][j]) vf[r] |= 1u << j;
    }
  }
  const unsigned long long lt = (1ull << lane) - 1ull;
  // exclusive prefix over the lanes of a 0..4 count, and its wave total, from three ballots
  auto prefix4 = [&](int cnt, int& total) -> int {
    const unsigned long long b0 = __ballot(cnt & 1), b1 = __ballot(cnt & 2), b2 = __ballot(cnt & 4);
    total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
    return __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt);
  };

  // rows are processed in groups whose pair lists fit the LDS list: all four rows normally, two and two
  // when the tile is so dense that they would not
  int grand = 0;
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    int t;
    prefix4(__popc(hf[r]), t);
    grand += t;
    prefix4(__popc(vf[r]), t);
    grand += t;
  }
  const int rows_per_group = grand <= kCap ? kRows : (kRows < 2 ? kRows : 2);
  for (int r0 = 0; r0 < kRows; r0 += rows_per_group) {
    // list layout: [horizontal pairs of the group, row-major][vertical pairs, row-major]
    int n_h = 0, n_all = 0;
    wave_lds_sync(); // the previous group's list and staging are no longer read
#pragma unroll
    for (int axis = 0; axis < 2; ++axis) {
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        if (r < r0 || r >= r0 + rows_per_group) continue;
        const uint32_t f = axis == 0 ? hf[r] : vf[r];
        int t;
        int pos = n_all + prefix4(__popc(f), t);
        n_all += t;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (f & (1u << j)) s_list[wave][pos++] = static_cast<uint16_t>((r << 8) | (lane * 4 + j));
        }
      }
      if (axis == 0) n_h = n_all;
    }
    wave_lds_sync();

    // one pair per lane; a chunk of 64 lanes never mixes the two axes
    for (int axis = 0; axis < 2; ++axis) {
      const int l_begin = axis == 0 ? 0 : n_h, l_end = axis == 0 ? n_h : n_all;
      for (int t0 = l_begin; t0 < l_end; t0 += kWave) {
        const int t = t0 + lane;
        const bool act = t < l_end;
        int32_t ia = -1, ib = -1;
        T ga = T(0), za = T(0), gb = T(0), zb = T(0); // -(in-plane), -(z) contributions to pixel A / pixel B
        T Ba[3] = {T(0), T(0), T(0)}, Bb[3] = {T(0), T(0), T(0)};
        int32_t va[3] = {0, 0, 0}, vb[3] = {0, 0, 0};
        if (act) {
          const int entry = s_list[wave][t];
          const int y = y_base + (entry >> 8);
          const int px = sx * (kWave * 4) + (entry & 255);
          const int64_t pa = int64_t(y) * W + px;
          const int64_t pb = axis == 0 ? pa + 1 : pa + W;
          ia = idx_n[pa];
          ib = idx_n[pb];
          TriInfo<T> ta, tb;
          load_tri<T>(v_n, vi_n, ia, ta);
          load_tri<T>(v_n, vi_n, ib, tb);
          T gA, zA, gB, zB;
          if (axis == 0) {
            eval_pair<T, 0>(v_n, ta, tb, ia, ib, px, y, gdx_n[pa], M, gA, zA, gB, zB);
          } else {
            eval_pair<T, 1>(v_n, ta, tb, ia, ib, px, y, gdy_n[pa], M, gA, zA, gB, zB);
          }
          ga = -gA, za = -zA, gb = -gB, zb = -zB; // edge_grad_kernel.cu:427-445 negates
          va[0] = ta.i0, va[1] = ta.i1, va[2] = ta.i2;
          vb[0] = tb.i0, vb[1] = tb.i1, vb[2] = tb.i2;
          if (ia >= 0) Ba[0] = bary_n[pa], Ba[1] = bary_n[HW + pa], Ba[2] = bary_n[2 * HW + pa];
          if (ib >= 0) Bb[0] = bary_n[pb], Bb[1] = bary_n[HW + pb], Bb[2] = bary_n[2 * HW + pb];
        }
        // a side takes part if its pixel is foreground and it received something
        const bool a_on = ia >= 0 && (ga != T(0) || za != T(0));
        const bool b_on = ib >= 0 && (gb != T(0) || zb != T(0));
        // twelve terms per pair: corner slots A0..A2, B0..B2, two components each -- the pair's axis (x or y) and z
        T g[12];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          g[2 * k + 0] = a_on ? ga * Ba[k] : T(0);
          g[2 * k + 1] = a_on ? za * Ba[k] : T(0);
          g[6 + 2 * k + 0] = b_on ? gb * Bb[k] : T(0);
          g[6 + 2 * k + 1] = b_on ? zb * Bb[k] : T(0);
        }
        // a run = consecutive lanes with the same two triangles and the same participation; its sums end up in its
        // last lane (segscatter.hpp: run_sums_rows16), which alone touches the vertex table
        const int32_t key_a = a_on ? ia : -1, key_b = b_on ? ib : -1;
        int dist;
        bool tail;
        const int32_t left_a = __shfl_up(key_a, 1), left_b = __shfl_up(key_b, 1); // both BEFORE the ||: a shuffle under a
        run_rows16_heads(key_a != left_a || key_b != left_b, dist, tail);           // short-circuit runs with lanes switched off
        if (__ballot(a_on || b_on) != 0) {
          run_sums_rows16<T, 12>(g, dist);
          // components {axis, z}: x,z = 0 + c*2 ; y,z = 1 + c*1
          const int c_off = axis == 0 ? 0 : 1, c_step = axis == 0 ? 2 : 1;
          if (tail && a_on) table_add<T, 3, 2, kSlots>(t_keys[wave], t_vals[wave], 4, va, g, grad_n, 3, c_off, c_step);
          if (tail && b_on) table_add<T, 3, 2, kSlots>(t_keys[wave], t_vals[wave], 4, vb, g + 6, grad_n, 3, c_off, c_step);
        }
      }
    }
  }
  wave_lds_sync();
  table_flush<T, TableAcc, kSlots>(t_keys[wave], t_vals[wave], 4, 3, grad_n, 3, 0);
}

#ifndef DRTK_DOTS_ROWS
#define DRTK_DOTS_ROWS 2
#endif
#ifndef DRTK_DOTS_WAVES
#define DRTK_DOTS_WAVES 4
#endif
// rows per wave x vertically stacked waves per workgroup.  Fused route on one box, round 2 (build.py --variant):
// 2 x 4 0.711-0.717 ms, 2 x 8 0.769, 1 x 4 0.778, 1 x 8 0.777, 3 x 4 0.838, 4 x 2 0.910, 4 x 4 0.941.
// Requesting the NEXT channel's rows before this channel's products (what pays in interpolate forward): 98 instead of 76
// VGPRs, 4 instead of 6 waves per SIMD, 0.748-0.749 vs 0.736-0.754 ms -- nothing.  Nor does it matter that every load of
// load_row sits in a divergent branch with an s_waitcnt vmcnt(0) behind it (six round trips per channel, one after the
// other): with unconditional loads from clamped addresses the compiler batches them four and two, and the kernel takes
// the same 0.701 ms -- at 21 waves per CU the memory system is busy either way; this is what a 32-plane stream reaches.
constexpr int kStripRows = DRTK_DOTS_ROWS;
constexpr int kDotsWaves = DRTK_DOTS_WAVES;

template <typename T>
int edge_grad_backward_impl(
    const T* v_pix, const T* img, const int32_t* index_img, const int32_t* vi, const T* grad_output,
    int64_t N, int64_t V, int64_t C, int64_t vi_sN, int64_t H, int64_t W, double max_dp_dr, T* out,
    void* workspace, hipStream_t stream) {
  const int64_t HW = H * W;
  if (N * HW == 0) return DRTK_OK;
  T* gdx = static_cast<T*>(workspace);
  T* gdy = gdx + N * HW;
  const bool vec = (W % 4 == 0) && (reinterpret_cast<uintptr_t>(img) % (4 * sizeof(T)) == 0) &&
      (reinterpret_cast<uintptr_t>(grad_output) % (4 * sizeof(T)) == 0) &&
      (reinterpret_cast<uintptr_t>(workspace) % (4 * sizeof(T)) == 0) &&
      (reinterpret_cast<uintptr_t>(index_img) % 16 == 0);
  // pass A takes any width and any element-aligned placement (Quad); `vec` only picks pass B's kernel
  const int px_per_wave = (kWave - 1) * 4; // lane 63 = halo
  const int strips_x = static_cast<int>(ceil_div(W, px_per_wave));
  const int bands_y = static_cast<int>(ceil_div(H, kStripRows * kDotsWaves));
  const dim3 gridA(static_cast<unsigned>(int64_t(strips_x) * bands_y), static_cast<unsigned>(N));
  DRTK_LAUNCH((edge_dots_kernel<T, 4, kStripRows, kDotsWaves>), gridA, dim3(kDotsWaves * kWave), 0, stream, img, grad_output, index_img, (int)C, (int)H, (int)W, strips_x, gdx, gdy, xcd_strip(int64_t(strips_x) * (16 / (kStripRows * kDotsWaves))), static_cast<T*>(nullptr), int64_t(0));
  DRTK_RETURN_IF_LAUNCH_FAILED();
  const bool vec_out = vec && (reinterpret_cast<uintptr_t>(index_img) % 16 == 0) &&
      (reinterpret_cast<uintptr_t>(out) % (4 * sizeof(T)) == 0);
  if (vec_out) {
    const dim3 gridB(static_cast<unsigned>(ceil_div(HW, kBlock * 4)), static_cast<unsigned>(N));
    DRTK_LAUNCH((edge_gather4_kernel<T>), gridB, dim3(kBlock), 0, stream, v_pix, vi, index_img, gdx, gdy, V, vi_sN, (int)H, (int)W, static_cast<T>(max_dp_dr), out, xcd_strip(ceil_div(16 * W, kBlock * 4)));
  } else {
    const dim3 gridB(static_cast<unsigned>(ceil_div(HW, kBlock)), static_cast<unsigned>(N));
    DRTK_LAUNCH((edge_gather_kernel<T>), gridB, dim3(kBlock), 0, stream, v_pix, vi, index_img, gdx, gdy, V, vi_sN, (int)H, (int)W, static_cast<T>(max_dp_dr), out);
  }
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

// pass A shared by both routes
template <typename T>
int launch_edge_dots(const T* img, const T* grad_output, const int32_t* index_img, int64_t N, int64_t C, int64_t H, int64_t W, T* gdx, T* gdy, hipStream_t stream, T* zero_out, int64_t zero_count) {
  const int px_per_wave = (kWave - 1) * 4; // lane 63 = halo
  const int strips_x = static_cast<int>(ceil_div(W, px_per_wave));
  const int bands_y = static_cast<int>(ceil_div(H, kStripRows * kDotsWaves));
  const dim3 gridA(static_cast<unsigned>(int64_t(strips_x) * bands_y), static_cast<unsigned>(N));
  DRTK_LAUNCH((edge_dots_kernel<T, 4, kStripRows, kDotsWaves>), gridA, dim3(kDotsWaves * kWave), 0, stream, img, grad_output, index_img, (int)C, (int)H, (int)W, strips_x, gdx, gdy, xcd_strip(int64_t(strips_x) * (16 / (kStripRows * kDotsWaves))), zero_out, zero_count);
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

inline bool aligned_to(const void* p, size_t a) {
  return reinterpret_cast<uintptr_t>(p) % a == 0;
}

template <typename T>
int edge_grad_backward_fused_impl(
    drtk_dtype_t dtype, const T* v_pix, const T* img, const int32_t* index_img, const int32_t* vi,
    const T* bary_img, const T* grad_output, int64_t N, int64_t V, int64_t C, int64_t F, int64_t vi_sN,
    int64_t H, int64_t W, double max_dp_dr, T* grad_v_pix, void* workspace, size_t workspace_bytes,
    hipStream_t stream) {
  const int64_t HW = H * W;
  if (N * HW == 0) { // (otherwise grad_v_pix is cleared inside the first kernel)
    if (N * V > 0 && fill_bytes_async(grad_v_pix, 0, sizeof(T) * N * V * 3, stream) != DRTK_OK) return DRTK_ERR_LAUNCH;
    return DRTK_OK;
  }
  // One route for every shape and every element-aligned placement of the tensors (two planes of workspace): both kernels
  // fetch four adjacent pixels per access with the alignment of the element and handle the end of a row whose width is
  // not a multiple of four lane by lane.
  T* gdx = static_cast<T*>(workspace);
  T* gdy = gdx + N * HW;
  const int st = launch_edge_dots<T>(img, grad_output, index_img, N, C, H, W, gdx, gdy, stream, grad_v_pix, N * V * 3);
  if (st != DRTK_OK) return st;
  const int strips_x = static_cast<int>(ceil_div(W, kWave * 4));
  const int64_t waves = int64_t(strips_x) * ceil_div(H, kPairRows);
  const dim3 grid(static_cast<unsigned>(ceil_div(waves, kBlock / kWave)), static_cast<unsigned>(N));
  const int strip = xcd_strip(ceil_div(int64_t(strips_x) * (16 / kPairRows), kBlock / kWave));
  DRTK_LAUNCH((edge_scatter_pairs_kernel<T>), grid, dim3(kBlock), 0, stream, v_pix, vi, index_img, bary_img, gdx, gdy, V, vi_sN, (int)H, (int)W, strips_x, static_cast<T>(max_dp_dr), grad_v_pix, strip);
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

} // namespace
} // namespace drtk_amd

using namespace drtk_amd;

extern "C" int drtk_amd_edge_grad_backward_workspace_bytes(
    drtk_dtype_t dtype, int64_t N, int64_t H, int64_t W, size_t* bytes) {
  if (!bytes || N < 0 || H < 0 || W < 0 || (dtype != DRTK_F32 && dtype != DRTK_F64)) return DRTK_ERR_INVALID_ARGUMENT;
  const size_t es = dtype == DRTK_F32 ? 4 : 8;
  const size_t b = size_t(2) * N * H * W * es;
  *bytes = b > 0 ? b : 16;
  return DRTK_OK;
}

extern "C" int drtk_amd_edge_grad_backward(
    drtk_dtype_t dtype, const void* v_pix, const void* img, const int32_t* index_img,
    const int32_t* vi, const void* grad_output, int64_t N, int64_t V, int64_t C, int64_t F,
    int64_t vi_sN, int64_t H, int64_t W, double max_dp_dr, void* grad_v_pix_img, void* workspace,
    size_t workspace_bytes, drtk_stream_t stream) {
  if (N < 0 || V < 0 || C < 0 || F < 0 || H < 0 || W < 0 ||
      (vi_sN != 0 && vi_sN != F * 3) || H * W >= (int64_t(1) << 31) || (dtype != DRTK_F32 && dtype != DRTK_F64))
    return DRTK_ERR_INVALID_ARGUMENT;
  if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return DRTK_ERR_INVALID_ARGUMENT; // include/drtk_amd.h, alignment
  size_t need = 0;
  if (drtk_amd_edge_grad_backward_workspace_bytes(dtype, N, H, W, &need) != DRTK_OK) return DRTK_ERR_INVALID_ARGUMENT;
  if (N * H * W > 0) {
    if (!index_img || !grad_v_pix_img || !workspace) return DRTK_ERR_INVALID_ARGUMENT;
    if ((N * V > 0 && !v_pix) || (F > 0 && !vi)) return DRTK_ERR_INVALID_ARGUMENT;
    if (C > 0 && (!img || !grad_output)) return DRTK_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < need) return DRTK_ERR_WORKSPACE_TOO_SMALL;
  }
  {
    const size_t es = dtype_size(dtype); // (every slice reuses the front of the workspace: stream order)
    DRTK_FOR_VIEW_SLICES(N, n0, n, drtk_amd_edge_grad_backward(
        dtype, advance(v_pix, n0 * V * 3, es), advance(img, n0 * C * H * W, es), advance_typed(index_img, n0 * H * W),
        advance_typed(vi, n0 * vi_sN), advance(grad_output, n0 * C * H * W, es), n, V, C, F, vi_sN, H, W, max_dp_dr,
        advance(grad_v_pix_img, n0 * 3 * H * W, es), workspace, workspace_bytes, stream))
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case DRTK_F32:
      return edge_grad_backward_impl<float>(static_cast<const float*>(v_pix), static_cast<const float*>(img), index_img, vi, static_cast<const float*>(grad_output), N, V, C, vi_sN, H, W, max_dp_dr, static_cast<float*>(grad_v_pix_img), workspace, s);
    case DRTK_F64:
      return edge_grad_backward_impl<double>(static_cast<const double*>(v_pix), static_cast<const double*>(img), index_img, vi, static_cast<const double*>(grad_output), N, V, C, vi_sN, H, W, max_dp_dr, static_cast<double*>(grad_v_pix_img), workspace, s);
    default:
      return DRTK_ERR_INVALID_ARGUMENT;
  }
}

extern "C" int drtk_amd_edge_grad_backward_fused_workspace_bytes(
    drtk_dtype_t dtype, int64_t N, int64_t H, int64_t W, size_t* bytes) {
  if (!bytes || N < 0 || H < 0 || W < 0 || (dtype != DRTK_F32 && dtype != DRTK_F64)) return DRTK_ERR_INVALID_ARGUMENT;
  const size_t es = dtype == DRTK_F32 ? 4 : 8;
  const size_t b = size_t(2) * N * H * W * es;
  *bytes = b > 0 ? b : 16;
  return DRTK_OK;
}

extern "C" int drtk_amd_edge_grad_backward_fused(
    drtk_dtype_t dtype, const void* v_pix, const void* img, const int32_t* index_img,
    const int32_t* vi, const void* bary_img, const void* grad_output, int64_t N, int64_t V, int64_t C,
    int64_t F, int64_t vi_sN, int64_t H, int64_t W, double max_dp_dr, void* grad_v_pix,
    void* workspace, size_t workspace_bytes, drtk_stream_t stream) {
  if (N < 0 || V < 0 || C < 0 || F < 0 || H < 0 || W < 0 ||
      (vi_sN != 0 && vi_sN != F * 3) || H * W >= (int64_t(1) << 31) || (dtype != DRTK_F32 && dtype != DRTK_F64))
    return DRTK_ERR_INVALID_ARGUMENT;
  if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return DRTK_ERR_INVALID_ARGUMENT; // include/drtk_amd.h, alignment
  size_t need = 0;
  if (drtk_amd_edge_grad_backward_fused_workspace_bytes(dtype, N, H, W, &need) != DRTK_OK) return DRTK_ERR_INVALID_ARGUMENT;
  if (N * V > 0 && !grad_v_pix) return DRTK_ERR_INVALID_ARGUMENT;
  if (N * H * W > 0) {
    if (!index_img || !bary_img || !workspace) return DRTK_ERR_INVALID_ARGUMENT;
    if ((N * V > 0 && !v_pix) || (F > 0 && !vi)) return DRTK_ERR_INVALID_ARGUMENT;
    if (C > 0 && (!img || !grad_output)) return DRTK_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < need) return DRTK_ERR_WORKSPACE_TOO_SMALL;
  }
  {
    const size_t es = dtype_size(dtype);
    DRTK_FOR_VIEW_SLICES(N, n0, n, drtk_amd_edge_grad_backward_fused(
        dtype, advance(v_pix, n0 * V * 3, es), advance(img, n0 * C * H * W, es), advance_typed(index_img, n0 * H * W),
        advance_typed(vi, n0 * vi_sN), advance(bary_img, n0 * 3 * H * W, es), advance(grad_output, n0 * C * H * W, es), n, V, C, F,
        vi_sN, H, W, max_dp_dr, advance(grad_v_pix, n0 * V * 3, es), workspace, workspace_bytes, stream))
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case DRTK_F32:
      return edge_grad_backward_fused_impl<float>(dtype, static_cast<const float*>(v_pix), static_cast<const float*>(img), index_img, vi, static_cast<const float*>(bary_img), static_cast<const float*>(grad_output), N, V, C, F, vi_sN, H, W, max_dp_dr, static_cast<float*>(grad_v_pix), workspace, workspace_bytes, s);
    case DRTK_F64:
      return edge_grad_backward_fused_impl<double>(dtype, static_cast<const double*>(v_pix), static_cast<const double*>(img), index_img, vi, static_cast<const double*>(bary_img), static_cast<const double*>(grad_output), N, V, C, F, vi_sN, H, W, max_dp_dr, static_cast<double*>(grad_v_pix), workspace, workspace_bytes, s);
    default:
      return DRTK_ERR_INVALID_ARGUMENT;
  }
}
