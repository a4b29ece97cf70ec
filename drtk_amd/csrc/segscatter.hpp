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
// (this file) flips the wave to (corner, channel)-major: lane j owns one (k, c) pair, walks each
// run with scalar loop bounds (no exec-mask divergence), sums that run's contributions in a
// register and issues ONE atomic per run -- the 16 lanes of a corner hit 16 consecutive floats of
// attr_grad[n, vi_k, :], i.e. one 64-byte segment per corner per run.
#pragma once

#include "common.hpp"

namespace drtk_amd {

constexpr int kRunPad = kWave + 1; // LDS row stride: +1 breaks the 32-bank alignment of rows

// heads  bit p set  <=>  pixel p starts a new run (p == 0 or triangle differs from pixel p-1)
// cov    bit p set  <=>  pixel p is covered (index != -1)
// vidx   LDS [3][kRunPad] vertex ids of every pixel's triangle corners
// J = 3 * CC pairs, pair j = k * CC + c  ->  dst_n[vidx[k] * C_total + c_base + c]
template <typename T, typename ValFn>
__device__ __forceinline__ void scatter_runs(
    unsigned long long heads, unsigned long long cov, const int32_t* vidx, int J, int CC,
    T* __restrict__ dst_n, int C_total, int c_base, ValFn val) {
  const int lane = lane_id();
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
        const int32_t vid = vidx[k * kRunPad + start];
        atomic_add_global(dst_n + int64_t(vid) * C_total + c_base + c, acc);
      }
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

} // namespace drtk_amd
