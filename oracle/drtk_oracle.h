/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, strict IEEE, source op order) of the reference's hot path
 *   rasterize -> render -> interpolate -> edge_grad        (forward and backward)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path (drtk_amd/) never does.
 *
 * Parity pinning: every function here is checked bit-for-bit (index_img, depth, all forward floats)
 * or to <=1e-6 relative (atomically-accumulated gradients) against the reference's own CPU kernels
 * built by oracle/ref_build.py ("strict" variant), and against the committed fixtures in
 * tests/golden/ that oracle/gen_golden.py produced from that build.
 *
 * Conventions (all tensors contiguous, row-major):
 *   v / v_pix        [N,V,3]      vi        [N or 1,F,3] int32 with batch stride vi_sN (0 = shared)
 *   index_img        [N,H,W] i32  depth_img [N,H,W]      bary_img [N,3,H,W]
 *   attrs            [N,V,C]      img/out   [N,C,H,W]
 * nthreads <= 1 runs single-threaded (deterministic accumulation order = pixel order, as the
 * reference's at::parallel_for does with one thread); nthreads > 1 uses OpenMP with atomics
 * (used only as the CPU baseline in bench.py).
 * Every function returns 0 on success.
 */
#ifndef DRTK_ORACLE_H
#define DRTK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRTK_ORACLE_DECL(SFX, REAL)                                                               \
  /* rasterize_kernel_cpu.cpp:54-204 ; depth_img is always float (rasterize_kernel_cpu.cpp:268) */ \
  int drtk_oracle_rasterize_##SFX(                                                                \
      const REAL* v, const int32_t* vi, int64_t N, int64_t V, int64_t F, int64_t vi_sN,           \
      int64_t H, int64_t W, float* depth_img, int32_t* index_img, int nthreads);                  \
  /* wireframe mode, rasterize_kernel.cu:170-400 (CUDA-only; PARITY UNPINNED, see the body) */        \
  int drtk_oracle_rasterize_lines_##SFX(                                                          \
      const REAL* v, const int32_t* vi, int64_t N, int64_t V, int64_t F, int64_t vi_sN,           \
      int64_t H, int64_t W, float* depth_img, int32_t* index_img);                                \
  /* render_kernel_cpu.cpp:18-121 */                                                              \
  int drtk_oracle_render_##SFX(                                                                   \
      const REAL* v, const int32_t* vi, const int32_t* index_img, int64_t N, int64_t V,           \
      int64_t F, int64_t vi_sN, int64_t H, int64_t W, REAL* depth_img, REAL* bary_img,            \
      int nthreads);                                                                              \
  /* render_kernel_cpu.cpp:123-297 ; grad_v must be zero-initialised by the caller */             \
  int drtk_oracle_render_backward_##SFX(                                                          \
      const REAL* v, const int32_t* vi, const int32_t* index_img, const REAL* grad_depth_img,     \
      const REAL* grad_bary_img, int64_t N, int64_t V, int64_t F, int64_t vi_sN, int64_t H,       \
      int64_t W, REAL* grad_v, int nthreads);                                                     \
  /* interpolate_kernel_cpu.cpp:32-110 */                                                         \
  int drtk_oracle_interpolate_##SFX(                                                              \
      const REAL* attrs, const int32_t* vi, const int32_t* index_img, const REAL* bary_img,       \
      int64_t N, int64_t V, int64_t C, int64_t F, int64_t vi_sN, int64_t H, int64_t W, REAL* out, \
      int nthreads);                                                                              \
  /* interpolate_kernel_cpu.cpp:112-228 ; attr_grad (may be NULL) must be zero-initialised,       \
     bary_grad (may be NULL) is fully written */                                                  \
  int drtk_oracle_interpolate_backward_##SFX(                                                     \
      const REAL* grad_out, const REAL* attrs, const int32_t* vi, const int32_t* index_img,       \
      const REAL* bary_img, int64_t N, int64_t V, int64_t C, int64_t F, int64_t vi_sN, int64_t H, \
      int64_t W, REAL* attr_grad, REAL* bary_grad, int nthreads);                                 \
  /* edge_grad_kernel_cpu.cpp:139-359 ; grad_v_pix_img [N,3,H,W] must be zero-initialised */      \
  int drtk_oracle_edge_grad_backward_##SFX(                                                       \
      const REAL* v_pix, const REAL* img, const int32_t* index_img, const int32_t* vi,            \
      const REAL* grad_output, int64_t N, int64_t V, int64_t C, int64_t F, int64_t vi_sN,         \
      int64_t H, int64_t W, double max_dp_dr, REAL* grad_v_pix_img, int nthreads);                 \
  /* sparse interpolation operators, interpolate_kernel_cpu.cpp:411-693 (single-threaded) */      \
  int drtk_oracle_interpolation_matrix_##SFX(                                                     \
      const int32_t* vi, const int32_t* index_img, const REAL* bary_img,                          \
      const int64_t* row_pixels, int64_t R, int64_t N, int64_t F, int64_t vi_sN, int64_t H,       \
      int64_t W, int64_t* col_indices, REAL* values);                                             \
  int drtk_oracle_interpolation_matrix_backward_##SFX(                                            \
      const REAL* grad_values, const int32_t* vi, const int32_t* index_img,                       \
      const int64_t* row_pixels, int64_t R, int64_t N, int64_t F, int64_t vi_sN, int64_t H,       \
      int64_t W, REAL* bary_grad);                                                                \
  int drtk_oracle_normal_matrix_values_##SFX(                                                     \
      const int32_t* pair_indices, const int32_t* index_img, const REAL* bary_img, int64_t N,     \
      int64_t F, int64_t pair_sN, int64_t H, int64_t W, REAL* values);                            \
  int drtk_oracle_normal_matrix_values_backward_##SFX(                                            \
      const REAL* grad_values, const int32_t* pair_indices, const int32_t* index_img,             \
      const REAL* bary_img, int64_t N, int64_t F, int64_t pair_sN, int64_t H, int64_t W,          \
      REAL* bary_grad);                                                                           \
  /* anisotropic mipmap grid sampler, mipmap_grid_sampler_kernel.cu:20-897 (see                   \
     drtk_oracle_mipmap.inc; parity PARTIALLY pinned).  `out` is zero-filled by the call;         \
     grad_levels[l] must be zero-initialised by the caller, grad_grid is fully written. */        \
  int drtk_oracle_mipmap_grid_sampler_2d_##SFX(                                                   \
      const REAL* const* levels, const int64_t* lh, const int64_t* lw, int mipmaps,               \
      const REAL* grid, const REAL* vt_dxdy_img, int64_t N, int64_t C, int64_t H, int64_t W,      \
      int max_aniso, int padding_mode, int interpolation_mode, int align_corners,                 \
      int force_max_aniso, int clip_grad, REAL* out);                                             \
  int drtk_oracle_mipmap_grid_sampler_2d_backward_##SFX(                                          \
      const REAL* grad_out, const REAL* const* levels, const int64_t* lh, const int64_t* lw,      \
      int mipmaps, const REAL* grid, const REAL* vt_dxdy_img, int64_t N, int64_t C, int64_t H,    \
      int64_t W, int max_aniso, int padding_mode, int interpolation_mode, int align_corners,      \
      int force_max_aniso, int clip_grad, REAL* const* grad_levels, REAL* grad_grid);

DRTK_ORACLE_DECL(f32, float)
DRTK_ORACLE_DECL(f64, double)

/* Number of OpenMP threads the library would use for nthreads = 0 ("all"). */
int drtk_oracle_max_threads(void);
/* Test helper: while on, every backward function accumulates |term| instead of term -- its outputs are the magnitudes
 * summed into each element (the per-element scale of float32 rounding error).  Process-wide; off by default. */
void drtk_oracle_set_abs_accumulate(int on);

#ifdef __cplusplus
}
#endif
#endif
