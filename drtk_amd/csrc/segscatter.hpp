// Wave-level segmented scatter-add used by the two vertex-gradient backward kernels
// (render_backward: 9 values per pixel, interpolate_backward: 3*C values per pixel).
//
// Reference behaviour: one fastAtomicAdd per (pixel, corner, channel) (render_kernel.cu:233-278)
// or, for interpolate, a per-channel cub::WarpReduce::TailSegmentedSum over runs of equal vertex
// id in adjacent lanes followed by head-lane atomics, with a __syncthreads per channel
// (interpolate_kernel.cu:249-281).
//
// CDNA4 mapping: a wave owns 64 consecutive pixels of one view.  Phase 1 (pixel-major, done by
// the caller) leaves the per-pixel operands in LDS; the run structure -- where the triangle id
// changes along the 64 pixels -- is a 64-bit ballot, i.e. wave-uniform SGPR state.  Phase 2
// (this file) flips the wave to (corner, channel)-major: a lane owns one (k, c) pair and one
// contiguous slice of the 64 pixels, sums each run of equal triangle id inside its slice in a
// register and issues ONE atomic per run -- the CC lanes of a corner hit CC consecutive floats of
// dst[n, vi_k, c_base:c_base+CC], i.e. one contiguous segment per corner per run.
//   J = 3*CC pairs.  J <= 16: 4 pixel slices of 16 (lanes = 4 x 16);  J <= 32: 2 slices of 32;
//   otherwise 1 slice of 64 pixels and ceil(J/64) rounds.
#pragma once

#include "common.hpp"

namespace drtk_amd {

constexpr int kRunPad = kWave + 1; // LDS row stride: +1 breaks the 32-bank alignment of rows

// heads  bit p set  <=>  pixel p starts a new run (p == 0 or triangle differs from pixel p-1)
// cov    bit p set  <=>  pixel p is covered (index != -1); constant within a run
// vidx   LDS [3][kRunPad] vertex ids of every pixel's triangle corners
// pair j = k * CC + c  ->  dst_n[vidx[k] * C_total + c_base + c]
template <typename T, typename ValFn>
__device__ __forceinline__ void scatter_runs(
    unsigned long long heads, unsigned long long cov, const int32_t* vidx, int J, int CC,
    T* __restrict__ dst_n, int C_total, int c_base, ValFn val) {
  const int lane = lane_id();
  if (J > 32) {
    // one slice: run bounds are wave-uniform -> scalar loops, no exec-mask divergence
    for (int j0 = 0; j0 < J; j0 += kWave) {
      const int j = j0 + lane;
      const bool active = j < J;
      const int k = active ? j / CC : 0;
      const int c = active ? j - k * CC : 0;
      unsigned long long h = heads;
      while (h) {
        const int start = __builtin_ctzll(h);
        h &= h - 1;
        const int end = h ? __builtin_ctzll(h) : kWave;
        if (!((cov >> start) & 1ull)) continue; // background run
        if (active) {
          T acc = T(0);
          for (int p = start; p < end; ++p) acc += val(k, c, p);
          atomic_add_global(dst_n + int64_t(vidx[k * kRunPad + start]) * C_total + c_base + c, acc);
        }
      }
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
  T* const out = dst_n + c_base + c;
  T acc = T(0);
  int run_start = p0;
  for (int i = 0; i < span; ++i) { // uniform trip count; p differs per slice
    const int p = p0 + i;
    if (i > 0 && ((heads >> p) & 1ull)) {
      if (active && ((cov >> run_start) & 1ull))
        atomic_add_global(out + int64_t(vidx[k * kRunPad + run_start]) * C_total, acc);
      acc = T(0);
      run_start = p;
    }
    if (active && ((cov >> p) & 1ull)) acc += val(k, c, p);
  }
  if (active && ((cov >> run_start) & 1ull))
    atomic_add_global(out + int64_t(vidx[k * kRunPad + run_start]) * C_total, acc);
}

// Run-head / coverage ballots of a wave whose lane l holds triangle id `tr` (-1 = background or
// out of range) of pixel l.
__device__ __forceinline__ void run_masks(int32_t tr, unsigned long long& heads, unsigned long long& cov) {
  const int lane = lane_id();
  const int32_t prev = __shfl_up(tr, 1);
  heads = __ballot(lane == 0 || tr != prev);
  cov = __ballot(tr != -1);
}

} // namespace drtk_amd
