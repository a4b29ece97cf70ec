// Two-level segmented scatter-add used by the vertex-gradient backward kernels
// (render_backward: 9 values per pixel, interpolate_backward: 3*C values per pixel).
//
// Reference behaviour: one fastAtomicAdd per (pixel, corner, channel) (render_kernel.cu:233-278)
// or, for interpolate, a per-channel cub::WarpReduce::TailSegmentedSum over runs of equal vertex
// id in adjacent lanes followed by head-lane atomics, with a __syncthreads per channel
// (interpolate_kernel.cu:249-281).  Measured on MI355X the global float atomics are what bounds
// such a kernel (~10 G atomic requests/s, device scope), so the number of REQUESTS is what is
// minimised here:
//
//   level 1 (wave):  a wave owns 64 consecutive pixels of one image row.  Phase 1 (pixel-major,
//     done by the caller) leaves the per-pixel operands in LDS; the run structure -- where the
//     triangle id changes along the 64 pixels -- is a 64-bit ballot, i.e. wave-uniform SGPR state.
//     Phase 2 flips the wave to (corner, channel)-major: a lane owns one (k, c) pair (and, for
//     small channel counts, one slice of the pixels), and sums each run in a register.
//   level 2 (wave tile): run sums go into a small wave-private LDS hash table keyed by VERTEX id
//     that lives for the wave's whole 64 x 4 pixel tile, so every vertex touched by the tile costs
//     one global atomic per channel at the end -- one contiguous segment per vertex -- instead of
//     one per run.  The table is private to the wave, so the wide (C = 16) path updates it with
//     plain read-modify-write (measured: LDS float atomics retire ~1 lane per clock and were the
//     bottleneck of a workgroup-shared table).  Vertices that do not fit (table full) fall back to
//     a direct global atomic, so any input is handled.
#pragma once

#include <type_traits>

#include "common.hpp"

namespace drtk_amd {

constexpr int kRunPad = kWave + 4; // LDS row stride: rows stay 16-byte aligned (ds_read_b128), +4 staggers banks
constexpr int kTableSlots = 64;    // vertices per wave-tile table (power of two).  The table functions take the size as a
                                   // template argument; 128 slots were A/B-tested on ONE box (profiles/kernel_bench.py
                                   // --lib, build.py --variant) for edge_scatter_pairs: 0.810-0.815 vs 0.812-0.814 ms --
                                   // no difference (a first comparison across two gpurun boxes had shown 4 %: boxes
                                   // differ by that much, so kernel variants are only ever compared within one call)
constexpr int log2_of(int n) { return n <= 1 ? 0 : 1 + log2_of(n / 2); }
constexpr int kTileRows = 16;      // a workgroup (4 waves) covers 64 x 16 pixels: 4 adjacent rows per wave
constexpr int kTableProbes = 6;

// Orders this wave's LDS writes before its later LDS reads by other lanes (wave-private staging
// areas need no workgroup barrier: a wave's DS operations execute in order).  The hardware side is the
// lgkmcnt(0) wait; the two wavefront-scope fences are for the COMPILER -- the barrier and waitcnt builtins carry no
// memory semantics of their own, so without a release before and an acquire after nothing would forbid moving a
// lane's plain LDS store below, or another lane's load above, the barrier (fence / wave_barrier / fence, the idiom
// of HIP's own __syncwarp-style helpers).  Wavefront scope emits no instruction: disassembly unchanged.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0) only: do NOT drain outstanding global stores
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int SLOTS = kTableSlots>
__device__ __forceinline__ void table_init(int32_t* keys) { // wave-private table
  for (int i = lane_id(); i < SLOTS; i += kWave) keys[i] = -1;
}

// Slot of vertex `vid` in the tile table (inserting it if needed); -1 if the table is full along
// its probe sequence.  Safe under concurrent calls from the lanes of the owning wave.
template <int SLOTS = kTableSlots>
__device__ __forceinline__ int table_slot(int32_t* keys, int32_t vid) {
  uint32_t h = (static_cast<uint32_t>(vid) * 2654435761u) >> (32 - log2_of(SLOTS));
  for (int probe = 0; probe < kTableProbes; ++probe) {
    const int32_t cur = keys[h];
    if (cur == vid) return static_cast<int>(h);
    if (cur == -1) {
      const int32_t old = atomicCAS(&keys[h], -1, vid);
      if (old == -1 || old == vid) return static_cast<int>(h);
    }
    h = (h + 1) & (SLOTS - 1);
  }
  return -1;
}

// The same with a table size chosen at run time (2^log2_slots entries): interpolate backward's workgroup-shared table,
// whose rows hold all C components of a vertex and whose size follows from the LDS it may take.
__device__ __forceinline__ int table_slot_rt(int32_t* keys, int32_t vid, int log2_slots) {
  const uint32_t mask = (1u << log2_slots) - 1u;
  uint32_t h = (static_cast<uint32_t>(vid) * 2654435761u) >> (32 - log2_slots);
#pragma unroll 1
  for (int probe = 0; probe < kTableProbes; ++probe) {
    const int32_t cur = keys[h];
    if (cur == vid) return static_cast<int>(h);
    if (cur == -1) {
      const int32_t old = atomicCAS(&keys[h], -1, vid);
      if (old == -1 || old == vid) return static_cast<int>(h);
    }
    h = (h + 1) & mask;
  }
  return -1;
}

// heads  bit p set  <=>  pixel p starts a new run (p == 0 or triangle differs from pixel p-1)
// cov    bit p set  <=>  pixel p is covered (index != -1); constant within a run
// slot   LDS [3][kRunPad] table slot of every pixel's triangle corners (-1: use `vid` + global atomic)
// vid    LDS [3][kRunPad] vertex ids of every pixel's triangle corners
// val4(k, c, g, out[4]) yields the contributions of pixels 4g..4g+3 to pair (k, c); it must be 0
// for uncovered pixels (phase 1 stores zeros there).
// pair j = k * CC + c  ->  table vals[slot * stride + c]   (or dst_n[vid * C_total + c_base + c])
// LDS float/double atomic add on an explicitly LDS-typed pointer: guarantees ds_add_f32/f64 (a
// generic pointer that may alias global memory is lowered to a much slower flat_atomic).
template <typename T>
__device__ __forceinline__ void lds_atomic_add(T* p, T v) {
  using LdsPtr = __attribute__((address_space(3))) T*;
  __hip_atomic_fetch_add((LdsPtr)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The vertex tables accumulate in DOUBLE whatever T is.  Measured on MI355X (profiles/micro/lds_atomics.hip, every CU
// busy): ds_add_f32 retires 0.33 lanes per clock per CU -- three clocks per LANE, with or without return, distinct
// addresses or not -- while ds_add_f64 retires 6.9 (ds_add_u32: 11; a plain read-add-write: 6.9).  The f32 atomic was
// the floor of every C <= 4 scatter (render backward: 0.08 of 0.26 ms; edge_scatter_pairs: most of its 0.29 ms).  A
// double accumulator costs 8 instead of 4 bytes of LDS per entry and rounds once, at the flush, instead of at every add.
using TableAcc = double;

// A = type of the table entries: TableAcc where entries are updated with LDS atomics (J <= 32), or T itself where the
// J > 32 path's plain read-modify-write applies (see `exclusive` below).
// WIDE_ONLY: always take the one-slice form (lanes j >= J idle), whatever J is -- for callers that cannot afford the
// registers of both forms in one kernel.  SLICE_MAX_J: the sliced form for J <= SLICE_MAX_J only (interpolate backward's
// wide kernel: 16 -- its 8-channel chunks, J = 24, measured faster on the one-slice form than in two slices of 32 pixels:
// 0.53 vs 0.56 ms at C = 8; its 4-channel tails, J = 12, faster in four slices: 0.95 vs 1.00 ms at C = 20).
// SHARED_TABLE: the table is shared by the waves of the workgroup -- its entries are updated with LDS atomics even on the
// one-slice form (whose plain read-modify-write is only safe for a wave-private table); `tab_off` is added to the
// component index of a table entry (a table whose rows hold all C_total components while the call scatters the chunk
// that starts at c_base: tab_off = c_base).
// A val4 callable for scatter_runs: plain lambdas (k, c, g, out[4]) are wrapped in Val4Plain; Val4Rows is the pipelined
// form for "pixel values = elementwise products of two staged LDS rows" (rows of kRunPad elements: a[c], b[k]).
template <typename F>
struct Val4Plain {
  static constexpr bool pipelined = false;
  struct Raw {};
  F f;
  template <typename T>
  __device__ __forceinline__ void operator()(int k, int c, int g, T* x) const { f(k, c, g, x); }
  template <typename R>
  __device__ __forceinline__ void load(int, int, int, R&) const {}
  template <typename R, typename T>
  __device__ __forceinline__ void finish(const R&, T*) const {}
};
template <typename T, bool PIPELINED = true>
struct Val4Rows {
  static constexpr bool pipelined = PIPELINED;
  using V4 = typename std::conditional<sizeof(T) == 4, float4, double4>::type;
  struct Raw {
    V4 a, b;
  };
  const T* rows_a; // [c][kRunPad]
  const T* rows_b; // [k][kRunPad]
  __device__ __forceinline__ void load(int k, int c, int g, Raw& r) const {
    r.a = *reinterpret_cast<const V4*>(rows_a + c * kRunPad + 4 * g);
    r.b = *reinterpret_cast<const V4*>(rows_b + k * kRunPad + 4 * g);
  }
  __device__ __forceinline__ void finish(const Raw& r, T* x) const {
    x[0] = r.a.x * r.b.x, x[1] = r.a.y * r.b.y, x[2] = r.a.z * r.b.z, x[3] = r.a.w * r.b.w;
  }
  __device__ __forceinline__ void operator()(int k, int c, int g, T* x) const { // (the sliced form)
    Raw r;
    load(k, c, g, r);
    finish(r, x);
  }
};

template <typename T, typename A, bool WIDE_ONLY, bool SHARED_TABLE, int SLICE_MAX_J, typename Val4Fn>
__device__ __forceinline__ void scatter_runs_impl(
    unsigned long long heads, unsigned long long cov, const int32_t* slot, const int32_t* vid, int J,
    int CC, A* vals, int stride, T* __restrict__ dst_n, int C_total, int c_base, Val4Fn val4, int dbg = 0,
    int c_off = 0, int c_step = 1, int tab_off = 0) {
  const int lane = lane_id();
  using LdsPtr = __attribute__((address_space(3))) A*;
  // `exclusive`: no two lanes of one call target the same table entry (one run at a time, and the
  // three corners of a triangle are distinct vertices -- phase 1 sends triangles with repeated
  // vertex ids to the global fallback), so a plain read-modify-write is race free.
  // lane channel c addresses component c_off + c * c_step of the table entry / destination row
  // (default: c itself), so callers can scatter a strided subset of the components
  auto flush = [&](int k, int c_lane, int start, T acc, bool exclusive) {
    if (DRTK_DBG(dbg, 32)) return;
    const int c = c_off + c_lane * c_step;
    const int s = slot ? slot[k * kRunPad + start] : -1; // slot == nullptr: no table, always direct
    if (s < -1) return;                                   // -2: this corner receives nothing in this run
    if (s >= 0) {
      if (exclusive && !SHARED_TABLE) {
        LdsPtr q = (LdsPtr)(vals + s * stride + tab_off + c);
        *q = *q + static_cast<A>(acc);
      } else {
        lds_atomic_add(vals + s * stride + tab_off + c, static_cast<A>(acc));
      }
    }
    if (s < 0) { // table full for this vertex: direct global atomic (rare)
      // (rows of C_total elements that are not a multiple of 64 bytes -- C = 12, 24 -- make these atomics several times
      // slower: interpolate backward's flush 0.51 ms at C = 12 against 0.125 at C = 16, 8 x 2048^2.  Sending the lanes of
      // even and odd 64-byte segments in two instructions did not help (measured, round 4): it is not the straddling)
      atomic_add_global(dst_n + int64_t(vid[k * kRunPad + start]) * C_total + c_base + c, acc);
    }
  };
  if (WIDE_ONLY || J > SLICE_MAX_J) {
    // one slice of 64 pixels: run boundaries are wave-uniform -> scalar tests, no divergence
    for (int j0 = 0; j0 < J; j0 += kWave) {
      const int j = j0 + lane;
      const bool active = j < J;
      const int k = active ? j / CC : 0;
      const int c = active ? j - k * CC : 0;
      T acc = T(0);
      int run_start = 0; // wave-uniform
      // PIPELINED operands (val4 objects that offer load() / finish(): interpolate backward's products of two staged
      // rows): group g + 1's two 16-byte LDS reads are issued before group g's run logic, whose branches otherwise keep
      // every read next to its s_waitcnt lgkmcnt(0) -- sixteen exposed LDS round trips per row and chunk.
      constexpr bool kPipelined = Val4Fn::pipelined;
      typename Val4Fn::Raw raw;
      if constexpr (kPipelined) val4.load(k, c, 0, raw);
#pragma unroll 4
      for (int g = 0; g < kWave / 4; ++g) {
        T x[4];
        if constexpr (kPipelined) {
          val4.finish(raw, x);
          if (g + 1 < kWave / 4) val4.load(k, c, g + 1, raw);
        } else {
          val4(k, c, g, x);
        }
        // run heads among pixels 4g..4g+3 (pixel 0 never flushes)
        const uint32_t hb = __builtin_amdgcn_readfirstlane(
            static_cast<uint32_t>(heads >> (4 * g)) & (g == 0 ? 0xEu : 0xFu));
        if (hb == 0) {
          acc += x[0];
          acc += x[1];
          acc += x[2];
          acc += x[3];
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (hb & (1u << q)) {
              if (active && ((cov >> run_start) & 1ull)) flush(k, c, run_start, acc, true);
              acc = T(0);
              run_start = __builtin_amdgcn_readfirstlane(4 * g + q);
            }
            acc += x[q];
          }
        }
      }
      if (active && ((cov >> run_start) & 1ull)) flush(k, c, run_start, acc, true);
    }
    return;
  }
  const int jp = J <= 16 ? 16 : 32;   // lanes per slice
  const int span = J <= 16 ? 16 : 32; // pixels per slice (64 / number of slices)
  const int p0 = (lane / jp) * span;
  const int j = lane & (jp - 1);
  const bool active = j < J;
  const int k = active ? j / CC : 0;
  const int c = active ? j - k * CC : 0;
  T acc = T(0);
  int run_start = p0;
  for (int g = 0; g < span / 4; ++g) { // uniform trip count; pixels differ per slice
    T x[4];
    val4(k, c, p0 / 4 + g, x);
    const uint32_t hb = static_cast<uint32_t>(heads >> (p0 + 4 * g)) & (g == 0 ? 0xEu : 0xFu);
    if (__ballot(hb != 0) == 0) { // no slice has a run head in this group
      acc += x[0];
      acc += x[1];
      acc += x[2];
      acc += x[3];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (hb & (1u << q)) {
          if (active && ((cov >> run_start) & 1ull)) flush(k, c, run_start, acc, false);
          acc = T(0);
          run_start = p0 + 4 * g + q;
        }
        acc += x[q];
      }
    }
  }
  if (active && ((cov >> run_start) & 1ull)) flush(k, c, run_start, acc, false);
}

template <typename T, typename A = TableAcc, bool WIDE_ONLY = false, bool SHARED_TABLE = false, int SLICE_MAX_J = 32, typename Val4Fn>
__device__ __forceinline__ void scatter_runs(
    unsigned long long heads, unsigned long long cov, const int32_t* slot, const int32_t* vid, int J,
    int CC, A* vals, int stride, T* __restrict__ dst_n, int C_total, int c_base, Val4Fn val4, int dbg = 0,
    int c_off = 0, int c_step = 1, int tab_off = 0) {
  scatter_runs_impl<T, A, WIDE_ONLY, SHARED_TABLE, SLICE_MAX_J>(
      heads, cov, slot, vid, J, CC, vals, stride, dst_n, C_total, c_base, Val4Plain<Val4Fn>{val4}, dbg, c_off, c_step, tab_off);
}
// ... with the pipelined operand form
template <typename T, typename A = TableAcc, bool WIDE_ONLY = false, bool SHARED_TABLE = false, int SLICE_MAX_J = 32, bool PIPELINED = true>
__device__ __forceinline__ void scatter_runs_rows(
    unsigned long long heads, unsigned long long cov, const int32_t* slot, const int32_t* vid, int J,
    int CC, A* vals, int stride, T* __restrict__ dst_n, int C_total, int c_base, const T* rows_a, const T* rows_b,
    int dbg = 0, int tab_off = 0) {
  scatter_runs_impl<T, A, WIDE_ONLY, SHARED_TABLE, SLICE_MAX_J>(
      heads, cov, slot, vid, J, CC, vals, stride, dst_n, C_total, c_base, Val4Rows<T, PIPELINED>{rows_a, rows_b}, dbg, 0, 1, tab_off);
}

// Wave-wide: add every occupied entry of the wave's table to dst_n[key * C_total + c_base + c], c < CC.
template <typename T, typename A = TableAcc, int SLOTS = kTableSlots>
__device__ __forceinline__ void table_flush(
    const int32_t* keys, const A* vals, int stride, int CC, T* __restrict__ dst_n, int C_total,
    int c_base) {
  for (int e = lane_id(); e < SLOTS * CC; e += kWave) {
    const int s = e / CC, c = e - s * CC;
    const int32_t key = keys[s];
    if (key >= 0) {
      const T x = static_cast<T>(vals[s * stride + c]);
      if (x != T(0)) atomic_add_global(dst_n + int64_t(key) * C_total + c_base + c, x);
    }
  }
}

// Run-head / coverage ballots of a wave whose lane l holds triangle id `tr` (-1 = background or
// out of range) of pixel l.
__device__ __forceinline__ void run_masks(int32_t tr, unsigned long long& heads, unsigned long long& cov) {
  const int lane = lane_id();
  const int32_t prev = __shfl_up(tr, 1);
  heads = __ballot(lane == 0 || tr != prev);
  cov = __ballot(tr != -1);
}

// ---- run reduction in registers (C <= 4 scatters: render backward, the v_pix routes) ---------------------------------
// A wave holds one pixel per lane and J values per pixel; pixels of one triangle that are adjacent in the row form a
// run, and what the vertex table needs is the SUM of each value over each run.  Instead of staging the values in LDS
// and flipping the wave to (corner, channel) lanes (scatter_runs above), the sums are formed where the values are: a
// segmented Hillis-Steele scan over the 16-lane DPP rows -- four steps of `x += row_shr:d(x)` gated by "lane - d is
// still in my run" -- leaves every run's total in its last lane, and only those tail lanes touch the table.  Runs are
// cut at the DPP row boundaries (lanes 0, 16, 32, 48 always start a run): a row-crossing run is flushed in pieces,
// which the table merges again.  Per 64 pixels: 3 J VALU per step and J table updates, no LDS staging, no barrier.
template <int D>
__device__ __forceinline__ float dpp_row_shr(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 | D, 0xF, 0xF, true));
}
template <int D>
__device__ __forceinline__ double dpp_row_shr(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(u & 0xFFFFFFFFull), 0x110 | D, 0xF, 0xF, true);
  const unsigned hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(u >> 32), 0x110 | D, 0xF, 0xF, true);
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}

// Run structure of a wave whose lane l holds triangle id `tr` of pixel l (-1 = background / out of range), with runs cut
// at the 16-lane row boundaries: `dist` = this lane's distance from the first lane of its run (0..15), `tail` = this
// lane is the last of its run.
__device__ __forceinline__ void run_rows16_heads(bool differs_from_left, int& dist, bool& tail) {
  const int lane = lane_id();
  const unsigned long long heads = __ballot(differs_from_left) | 0x0001000100010001ull;
  const unsigned long long upto = heads & (~0ull >> (63 - lane)); // heads at or below this lane (bit of the row start is set)
  dist = lane - (63 - __builtin_clzll(upto));
  tail = lane == kWave - 1 || ((heads >> (lane + 1)) & 1ull);
}
__device__ __forceinline__ void run_rows16(int32_t tr, int& dist, bool& tail) {
  run_rows16_heads(tr != __shfl_up(tr, 1), dist, tail);
}

template <int D>
__device__ __forceinline__ int dpp_row_shl_i32(int x) { // lane l <- lane l + D of its 16-lane row, 0 beyond the row
  return __builtin_amdgcn_update_dpp(0, x, 0x100 | D, 0xF, 0xF, true);
}
__device__ __forceinline__ float masked(float x, int m) {
  return __builtin_bit_cast(float, __builtin_bit_cast(int, x) & m);
}
__device__ __forceinline__ double masked(double x, int m) {
  const unsigned long long mm = (static_cast<unsigned long long>(static_cast<unsigned>(m)) << 32) | static_cast<unsigned>(m);
  return __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, x) & mm);
}

// Step D of the scan adds lane l - D's value to lane l iff l - D is still in l's run (dist[l] >= D).  The gate is
// applied on the SOURCE side -- lane s is zeroed unless dist[s + D] >= D, a mask that is itself one row_shl of the
// predicate and is shared by all J values -- so that the add can take its operand straight through DPP
// (v_and + v_add_f32_dpp: 2 instructions per value and step instead of mov_dpp + cndmask + add).
template <typename T, int J>
__device__ __forceinline__ void run_sums_rows16(T (&g)[J], int dist) {
#define DRTK_SCAN_STEP(D)                                                      \
  {                                                                            \
    const int m = dpp_row_shl_i32<D>(dist >= D ? -1 : 0);                      \
    _Pragma("unroll") for (int j = 0; j < J; ++j) g[j] += dpp_row_shr<D>(masked(g[j], m)); \
  }
  DRTK_SCAN_STEP(1)
  DRTK_SCAN_STEP(2)
  DRTK_SCAN_STEP(4)
  DRTK_SCAN_STEP(8)
#undef DRTK_SCAN_STEP
}

// First-probe lookup of K vertices at once: the K hash slots are read in one batch (one LDS round trip instead of K
// dependent ones); a vertex found there -- the common case after the first pixels of a tile -- is done, the others go
// through table_slot's probe-and-insert loop.
template <int K, int SLOTS = kTableSlots>
__device__ __forceinline__ void table_slots(int32_t* keys, const int32_t (&vid)[K], int (&slot)[K]) {
  uint32_t h[K];
  int32_t cur[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    h[k] = (static_cast<uint32_t>(vid[k]) * 2654435761u) >> (32 - log2_of(SLOTS));
    cur[k] = keys[h[k]];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) slot[k] = cur[k] == vid[k] ? static_cast<int>(h[k]) : table_slot<SLOTS>(keys, vid[k]);
}

// Tail lanes only: add `CC` components of each of the K corners `vid[k]` (values x[k * CC + c]) into the wave's vertex
// table (LDS atomics: neighbouring triangles share vertices, so two tail lanes of one instruction may hit the same
// entry), or straight into dst_n[vid * C_total + c] when the table has no room for the vertex.
template <typename T, int K, int CC, int SLOTS = kTableSlots>
__device__ __forceinline__ void table_add(
    int32_t* keys, TableAcc* vals, int stride, const int32_t (&vid)[K], const T* x, T* __restrict__ dst_n, int C_total,
    int c_off = 0, int c_step = 1) { // value c of a corner goes to component c_off + c * c_step of the vertex
  int slot[K];
  table_slots<K, SLOTS>(keys, vid, slot);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (slot[k] >= 0) {
#pragma unroll
      for (int c = 0; c < CC; ++c) lds_atomic_add(vals + slot[k] * stride + c_off + c * c_step, static_cast<TableAcc>(x[k * CC + c]));
    } else {
#pragma unroll
      for (int c = 0; c < CC; ++c) atomic_add_global(dst_n + int64_t(vid[k]) * C_total + c_off + c * c_step, x[k * CC + c]);
    }
  }
}

} // namespace drtk_amd
