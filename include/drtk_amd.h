/* drtk_amd -- C ABI of the MI355X-native rasterize -> render -> interpolate -> edge_grad hot path.
 *
 * This is the drop-in boundary below the torch-operator layer.  The reference (facebookresearch/
 * DRTK) has no extern "C" surface of its own: its extensions are reached through the torch
 * dispatcher, whose CUDA-key implementations are the C++ host launchers cited on every entry
 * below.  Each function here replaces exactly one of those launchers (same inputs, outputs and
 * semantics, contiguous layouts, raw device pointers, an explicit HIP stream) so that the
 * torch-op shim (drtk_amd/csrc/torch_ops/<op>.cpp), a ctypes caller or a C program can all drive the
 * same kernels.  INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (hipMalloc / torch caching allocator), tensors contiguous:
 *       v, v_pix        [N,V,3]        vi         [N,F,3] int32, batch stride `vi_sN` elements
 *                                                 (0 = one [F,3] topology shared by all views)
 *       index_img       [N,H,W] int32  (-1 = empty)
 *       depth_img       [N,H,W]        bary_img   [N,3,H,W]   (planar)
 *       attrs           [N,V,C]        img / out  [N,C,H,W]   (planar)
 *   - alignment: tensor pointers need the alignment of their ELEMENT type only (4 bytes for float / int32, 8 for
 *     double) -- a contiguous view at any element offset of a larger buffer is fine, for inputs and for outputs.
 *     The kernels take their 16-byte vector paths when a tensor happens to be 16-byte aligned (every hipMalloc /
 *     torch allocation is) and the image width allows it; otherwise they fall back to scalar accesses or rely on
 *     the platform's support for dword-aligned wide accesses (tests/test_gpu_parity.py, ..._odd_element_offsets).
 *     WORKSPACES must be 16-byte aligned (they hold 64-bit counters updated atomically); this is checked:
 *     DRTK_ERR_INVALID_ARGUMENT otherwise.
 *   - `dtype` selects float or double for every floating tensor of the call (the reference
 *     dispatches float/double only: src/include/kernel_utils.h:35-57); indices are int32.
 *   - N (views) is unbounded on the path's operators: a launch takes 65 535 views (the view is blockIdx.y), a larger
 *     batch is executed as consecutive slices by the entry point itself -- views are independent, results identical
 *     (the reference's grid-stride kernels take any N: render_kernel.cu:349-377).  H * W < 2^31 per view (in-plane
 *     offsets are 32-bit); the two normal-matrix operators, which accumulate all views into one array, keep N <= 65 535.
 *   - `stream` is a hipStream_t (NULL = the null stream).  No call synchronises the device or
 *     allocates memory; all work is enqueued on `stream`.
 *   - return value: DRTK_OK or a negative drtk_status_t; drtk_amd_status_string() explains it.
 *     Argument validation mirrors the reference's TORCH_CHECKs where it can be expressed on raw
 *     sizes; tensor-level checks (dtype, device, ndim) live in the torch shim.
 *   - reproducibility: forward outputs and per-pixel gradients are bit-identical from run to run; per-VERTEX
 *     gradients (grad_v, attr_grad, grad_v_pix) are accumulated with float atomics in varying order and agree to
 *     ~1e-7 of their largest value between runs, like the reference's atomicAdd scatter.
 *   - knife-edge decisions: float arithmetic follows the reference's HOST source in its written order under IEEE
 *     rules (no contraction, correctly rounded / and sqrt), i.e. what a strict build of its CPU kernels evaluates.
 *     Where the reference decides on a quantity that is zero up to rounding -- the sign get_dp_dr takes from `d`
 *     where two DIFFERENT surfaces meet in the image with face normals equal to ~1e-8 yet not bit-identical
 *     (edge_grad_kernel_cpu.cpp:113-137; d ~ 1e-9 in exact arithmetic, the output of that pixel is +-max_dp_dr * ...;
 *     exact instanced copies give d == 0 and generic offsets |d| ~ 1e-6: both signed stably) -- its own builds
 *     differ from one another at that pixel: the shipped host
 *     build is -O3 --fast-math (setup.py:23-24), the CUDA build normalises with ::rnorm3df where the host divides by
 *     sqrt (cuda_math_helper.h:173-176).  Expect differences of 2 * max_dp_dr at a few such silhouette pixels when
 *     comparing against those (about 1 in 150 of the small two-object float32 test scenes has one); against the
 *     strict evaluation there are none (profiles/NOTES.md section 3).
 *   - thread-safety: re-entrant; no global mutable state except the rasterizer's depth-order setting (one atomic int).
 */
#ifndef DRTK_AMD_H
#define DRTK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden and a linker version script (drtk_amd/csrc/exports.map): the functions
 * declared between this push and its pop are its ENTIRE dynamic symbol table (tests/test_host_logic.py checks nm -D). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define DRTK_AMD_VERSION_MAJOR 0
#define DRTK_AMD_VERSION_MINOR 4

typedef enum { DRTK_F32 = 0, DRTK_F64 = 1 } drtk_dtype_t;

typedef enum {
  DRTK_OK = 0,
  DRTK_ERR_INVALID_ARGUMENT = -1, /* bad size / null pointer / unknown dtype */
  DRTK_ERR_WORKSPACE_TOO_SMALL = -2,
  DRTK_ERR_LAUNCH = -3,       /* hipGetLastError() after a launch was not hipSuccess */
  DRTK_ERR_UNSUPPORTED = -4,  /* reserved */
  DRTK_ERR_TOO_MANY_VERTICES = -5 /* V >= 2^28, rasterize_kernel.cu:459-462 */
} drtk_status_t;

typedef void* drtk_stream_t; /* hipStream_t */

const char* drtk_amd_status_string(int status);
/* "major.minor (gfx950)" */
const char* drtk_amd_version(void);

/* ------------------------------------------------------------------------------------------
 * rasterize        replaces rasterize_cuda            (src/rasterize/rasterize_kernel.cu:417-563)
 *
 * Z-buffer rasterization of N views.  Writes index_img (triangle id, -1 where empty; lower id
 * wins depth ties) and depth_img (ALWAYS float32, 0 where empty, rasterize_kernel.cu:481) --
 * bit-exact with the reference's arithmetic (rasterize_kernel.cu:69-166).
 * The depth's summation order is the one the reference's source spells, evaluated strictly.  A reference
 * build compiled with -ffast-math (its setup.py:22-24) associates that sum differently: the library's DEPTH-ORDER
 * SETTING (drtk_amd_set_depth_order below, or DRTK_AMD_DEPTH_ORDER=fastmath in the environment) makes every later
 * rasterize call evaluate it in the order such a build was observed to use, for deployments that must match one bit
 * for bit.
 * `workspace` holds the tile bins; query its size first.
 * `wireframe != 0` selects the line mode (rasterize_kernel.cu:170-400: edges whose bit is set in the top
 * nibble of vi[...,0] are drawn by the diamond rule, the triangles themselves only occlude); it needs the
 * workspace of drtk_amd_rasterize_lines_workspace_bytes (a packed [N,H,W] 64-bit buffer) instead.
 */
typedef enum {
  DRTK_DEPTH_ORDER_STRICT = 0,  /* s = dinv0 (e0/|den|) + dinv1 (e1/|den|) + dinv2 (e2/|den|): rasterize_kernel.cu:148-153 as written */
  DRTK_DEPTH_ORDER_FASTMATH = 1 /* s = ((e1 dinv1 + e0 dinv0) + e2 dinv2) (1/|den|): the host build of setup.py:22-24 (-O3 --fast-math) */
} drtk_depth_order_t;
/* The one library-level setting (everything else is re-entrant and stateless): which of the two orders
 * drtk_amd_rasterize launches from now on.  Initial value: DRTK_DEPTH_ORDER_FASTMATH if the environment holds
 * DRTK_AMD_DEPTH_ORDER=fastmath when the setting is first read or written, else DRTK_DEPTH_ORDER_STRICT.  The torch
 * operators (torch.ops.rasterize_ext.rasterize, drtk.rasterize) call the same entry point and follow it.
 * Set it before the first rasterize call of a deployment; a change is atomic and affects later launches only. */
int drtk_amd_set_depth_order(int order); /* DRTK_OK, or DRTK_ERR_INVALID_ARGUMENT for anything but the two values */
int drtk_amd_get_depth_order(void);      /* a drtk_depth_order_t */
int drtk_amd_rasterize_workspace_bytes(int64_t N, int64_t F, int64_t H, int64_t W, size_t* bytes);
int drtk_amd_rasterize_lines_workspace_bytes(int64_t N, int64_t H, int64_t W, size_t* bytes);
int drtk_amd_rasterize(
    drtk_dtype_t dtype, const void* v, const int32_t* vi, int64_t N, int64_t V, int64_t F,
    int64_t vi_sN, int64_t H, int64_t W, int wireframe, float* depth_img, int32_t* index_img,
    void* workspace, size_t workspace_bytes, drtk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * render           replaces render_cuda               (src/render/render_kernel.cu:283-380)
 * Perspective-correct barycentrics + depth per pixel; zeros where index_img == -1.
 */
int drtk_amd_render(
    drtk_dtype_t dtype, const void* v, const int32_t* vi, const int32_t* index_img, int64_t N,
    int64_t V, int64_t F, int64_t vi_sN, int64_t H, int64_t W, void* depth_img, void* bary_img,
    drtk_stream_t stream);

/* render_backward  replaces render_cuda_backward      (src/render/render_kernel.cu:382-436)
 * grad_v [N,V,3] is zero-filled by this call (render_kernel.cu:397) and then accumulated.
 */
int drtk_amd_render_backward(
    drtk_dtype_t dtype, const void* v, const int32_t* vi, const int32_t* index_img,
    const void* grad_depth_img, const void* grad_bary_img, int64_t N, int64_t V, int64_t F,
    int64_t vi_sN, int64_t H, int64_t W, void* grad_v, drtk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * interpolate      replaces interpolate_cuda          (src/interpolate/interpolate_kernel.cu:454-570)
 * out[n,c,y,x] = sum_k attrs[n,vi[t,k],c] * bary[n,k,y,x]; background pixels get the
 * reference's +-1 coordinate sweep (interpolate_kernel.cu:104-108).
 */
int drtk_amd_interpolate(
    drtk_dtype_t dtype, const void* attrs, const int32_t* vi, const int32_t* index_img,
    const void* bary_img, int64_t N, int64_t V, int64_t C, int64_t F, int64_t vi_sN, int64_t H,
    int64_t W, void* out, drtk_stream_t stream);

/* Extension (no reference counterpart): interpolate with the background written as 0 instead of the
 * +-1 coordinate sweep -- the fused form of `interpolate(...) * (index_img != -1)[:, None]`, the way every
 * DRTK pipeline uses the op (test/two_triangles.py:52-58).  Same gradient as drtk_amd_interpolate: the
 * backward never reads the upstream gradient of a background pixel. */
int drtk_amd_interpolate_masked(
    drtk_dtype_t dtype, const void* attrs, const int32_t* vi, const int32_t* index_img,
    const void* bary_img, int64_t N, int64_t V, int64_t C, int64_t F, int64_t vi_sN, int64_t H,
    int64_t W, void* out, drtk_stream_t stream);

/* interpolate_backward  replaces interpolate_cuda_backward (interpolate_kernel.cu:642-697)
 * attr_grad [N,V,C] (NULL = not wanted) is zero-filled then accumulated; bary_grad [N,3,H,W]
 * (NULL = not wanted) is fully written.  An EMPTY output (N*V*C == 0 / N*H*W == 0) has no storage, so NULL for it
 * carries no information: both NULL is DRTK_ERR_INVALID_ARGUMENT only when both gradients would have had elements,
 * otherwise there is nothing to write and the call returns DRTK_OK (e.g. a batch of zero views).
 */
int drtk_amd_interpolate_backward(
    drtk_dtype_t dtype, const void* grad_out, const void* attrs, const int32_t* vi,
    const int32_t* index_img, const void* bary_img, int64_t N, int64_t V, int64_t C, int64_t F,
    int64_t vi_sN, int64_t H, int64_t W, void* attr_grad, void* bary_grad, drtk_stream_t stream);
/* The same with an OPTIONAL scratch buffer (round 6).  For float attributes of 11 ... 15 channels with both gradients requested
 * (rows of attr_grad that are not whole 64-byte segments, one channel chunk) the vertex gradient is accumulated in rows padded
 * to 64 bytes inside `workspace` and compacted into attr_grad afterwards: 4-10 % faster than the unpadded route's vertex table
 * (C = 12 0.617 -> 0.554 ms on 8 x 2048^2; other shapes measured slower padded and do not pad).  `_workspace_bytes` returns 0
 * where no workspace is used; a NULL or too small workspace is never an error -- the call then takes the unpadded route of
 * drtk_amd_interpolate_backward.  The workspace must be 64-byte aligned; nothing in it survives the call.  Results equal the
 * unpadded route's up to the order of float summation. */
int drtk_amd_interpolate_backward_workspace_bytes(drtk_dtype_t dtype, int64_t N, int64_t V, int64_t C, size_t* bytes);
int drtk_amd_interpolate_backward_ws(
    drtk_dtype_t dtype, const void* grad_out, const void* attrs, const int32_t* vi,
    const int32_t* index_img, const void* bary_img, int64_t N, int64_t V, int64_t C, int64_t F,
    int64_t vi_sN, int64_t H, int64_t W, void* attr_grad, void* bary_grad, void* workspace, size_t workspace_bytes,
    drtk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * edge_grad_backward  replaces edge_grad_estimator_cuda_backward (src/edge_grad/edge_grad_kernel.cu:475-506)
 * grad_v_pix_img [N,3,H,W] is fully written (the reference zero-fills and scatters with
 * atomics, edge_grad_kernel.cu:430-445,489; this implementation gathers, same values).
 * `workspace` holds the two per-pixel pair terms (2*N*H*W scalars); query its size first.
 */
int drtk_amd_edge_grad_backward_workspace_bytes(
    drtk_dtype_t dtype, int64_t N, int64_t H, int64_t W, size_t* bytes);
int drtk_amd_edge_grad_backward(
    drtk_dtype_t dtype, const void* v_pix, const void* img, const int32_t* index_img,
    const int32_t* vi, const void* grad_output, int64_t N, int64_t V, int64_t C, int64_t F,
    int64_t vi_sN, int64_t H, int64_t W, double max_dp_dr, void* grad_v_pix_img, void* workspace,
    size_t workspace_bytes, drtk_stream_t stream);

/* edge_grad_backward_fused  =  edge_grad_backward followed by the C=3 interpolate backward that
 * drtk.edge_grad_estimator hangs behind it (drtk/edge_grad_estimator.py:168-176 + interpolate_kernel.cu
 * :642-697), without materialising grad_v_pix_img: writes grad_v_pix [N,V,3] (zero-filled here) directly.
 * Same values as the two calls (within float summation order).  Used by drtk_amd.edge_grad_estimator
 * when no v_pix_img hook is registered.
 */
int drtk_amd_edge_grad_backward_fused_workspace_bytes(
    drtk_dtype_t dtype, int64_t N, int64_t H, int64_t W, size_t* bytes);
int drtk_amd_edge_grad_backward_fused(
    drtk_dtype_t dtype, const void* v_pix, const void* img, const int32_t* index_img,
    const int32_t* vi, const void* bary_img, const void* grad_output, int64_t N, int64_t V, int64_t C,
    int64_t F, int64_t vi_sN, int64_t H, int64_t W, double max_dp_dr, void* grad_v_pix,
    void* workspace, size_t workspace_bytes, drtk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sparse interpolation operators -- the remaining ops of the reference's interpolate_ext:
 *
 * interpolation_matrix           replaces interpolation_matrix_cuda (interpolate_kernel.cu:699-770):
 *   one CSR row per foreground pixel (row_pixels = flat [N*H*W] indices of index_img != -1, ascending,
 *   computed by the caller), 3 entries per row: vertex columns sorted ascending with their
 *   barycentric weights.  crow_indices is arange(0, 3R+1, 3) and is left to the caller.
 * interpolation_matrix_backward  replaces interpolation_matrix_cuda_backward (:772-819): scatters
 *   d values back to bary_grad [N,3,H,W] (zero-filled here).
 * interpolation_normal_matrix_values[_backward]  replace ..._values_cuda[_backward] (:821-907):
 *   values[pair_indices[n,tri,3i+j]] += bary_i * bary_j over foreground pixels (values zero-filled
 *   here); pair_indices [N,F,9] int32 with batch stride pair_sN (0 = shared).  The CSR pattern itself
 *   (crow / col / pair_indices) is topology-only host work done by the torch shim.
 */
int drtk_amd_interpolation_matrix(
    drtk_dtype_t dtype, const int32_t* vi, const int32_t* index_img, const void* bary_img,
    const int64_t* row_pixels, int64_t R, int64_t N, int64_t F, int64_t vi_sN, int64_t H, int64_t W,
    int64_t* col_indices, void* values, drtk_stream_t stream);
int drtk_amd_interpolation_matrix_backward(
    drtk_dtype_t dtype, const void* grad_values, const int32_t* vi, const int32_t* index_img,
    const int64_t* row_pixels, int64_t R, int64_t N, int64_t F, int64_t vi_sN, int64_t H, int64_t W,
    void* bary_grad, drtk_stream_t stream);
int drtk_amd_interpolation_normal_matrix_values(
    drtk_dtype_t dtype, const int32_t* pair_indices, const int32_t* index_img, const void* bary_img,
    int64_t N, int64_t F, int64_t pair_sN, int64_t H, int64_t W, int64_t nnz, void* values,
    drtk_stream_t stream);
int drtk_amd_interpolation_normal_matrix_values_backward(
    drtk_dtype_t dtype, const void* grad_values, const int32_t* pair_indices, const int32_t* index_img,
    const void* bary_img, int64_t N, int64_t F, int64_t pair_sN, int64_t H, int64_t W, void* bary_grad,
    drtk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * mipmap_grid_sampler_2d -- grid_sample with trilinear mip selection and anisotropic taps; replaces
 * mipmap_aniso_grid_sampler_2d_cuda / _cuda_backward (mipmap_grid_sampler_kernel.cu:899-1249).
 *   levels[l]    device pointer of mip level l, [N,C,level_h[l],level_w[l]] with contiguous views, l < mipmaps <= 11
 *                (the arrays levels / level_h / level_w / level_sN / grad_levels themselves are HOST arrays)
 *   level_sN[l]  elements between consecutive views of level l: C*h*w for a contiguous tensor, 0 for ONE texture
 *                shared by all views (a [1,C,h,w] tensor expanded to N: the reference indexes through the tensor's
 *                strides, mipmap_grid_sampler_kernel.cu:40,65 (`input.data + n * inp_sN`), so an expanded pyramid is never copied);
 *                NULL = every level contiguous
 *   grid         [N,H,W,2] uv in [-1,1];  vt_dxdy_img [N,H,W,2,2] = [[du/dx, dv/dx],[du/dy, dv/dy]], contiguous
 *   grid_layout  HOST array {sN, sP, sC}: element (n, pixel y*W+x, c) of grid lies n*sN + pixel*sP + c*sC elements from
 *                `grid` (the reference reads grid through its strides, :430-445).  NULL = contiguous {2HW, 2, 1}; the
 *                channel-first uv image of `interpolate` seen through permute(0,2,3,1) is {2HW, 1, HW} -- no copy.
 *                grad_grid_layout: the same for grad_grid (every element is written)
 *   out          [N,C,H,W]
 *   padding_mode 0 zeros | 1 border | 2 reflection;  interpolation_mode 0 bilinear | 2 bicubic
 *   align_corners is ignored by the forward pass and honoured by the backward pass -- as in the
 *   reference (:423 vs :641-897).
 * Backward: grad_levels[l] (contiguous [N,C,h,w] whatever level_sN is, zero-filled here) and grad_grid [N,H,W,2]
 * (fully written); no gradient is defined for vt_dxdy_img.
 * NO ALIASING: grad_grid and grad_levels[] must not overlap any input (grad_out, levels, grid, vt_dxdy_img) nor each other --
 * textures of more than four channels are processed in several launches of one stream, each adding its part of the grid
 * gradient to what the previous launch stored (the torch operator allocates fresh outputs; tests/test_gpu_mipmap.py covers
 * C = 5 ... 16 in both grid layouts).
 * FINITE TEXELS ASSUMED where a weight is exactly zero: the forward pass skips a mip level whose blend weight is
 * exactly 0, the backward pass skips (pixel, level) pairs whose weighted upstream gradient is 0 in every channel and
 * whole tiles without upstream gradient.  The reference evaluates 0 * texel there, so an Inf / NaN texel in a level
 * that does not contribute (or under a zero upstream gradient) turns its output / gradient into NaN; here the
 * result stays finite.  For finite textures the two agree exactly (tests/test_gpu_mipmap.py pins the difference).
 */
int drtk_amd_mipmap_grid_sampler_2d(
    drtk_dtype_t dtype, const void* const* levels, const int64_t* level_h, const int64_t* level_w,
    const int64_t* level_sN, int mipmaps, const void* grid, const int64_t* grid_layout, const void* vt_dxdy_img, int64_t N,
    int64_t C, int64_t H, int64_t W, int max_aniso, int padding_mode, int interpolation_mode, int align_corners,
    int force_max_aniso, int clip_grad, void* out, drtk_stream_t stream);
int drtk_amd_mipmap_grid_sampler_2d_backward(
    drtk_dtype_t dtype, const void* grad_out, const void* const* levels, const int64_t* level_h,
    const int64_t* level_w, const int64_t* level_sN, int mipmaps, const void* grid, const int64_t* grid_layout,
    const void* vt_dxdy_img, int64_t N, int64_t C, int64_t H, int64_t W, int max_aniso, int padding_mode,
    int interpolation_mode, int align_corners, int force_max_aniso, int clip_grad, void* const* grad_levels,
    void* grad_grid, const int64_t* grad_grid_layout, drtk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * screen_space_uv_derivative -- vt_dxdy_img [N,H,W,2,2] = [[du/dx, dv/dx],[du/dy, dv/dy]] per pixel, the
 * Jacobian input of mipmap_grid_sampler_2d; replaces the PyTorch composite
 * drtk/screen_space_uv_derivative.py:15-80 (face_dpdt + 2x interpolate + project_points_grad + inv_ex +
 * mask) with one kernel.  Pinhole cameras only (as project_points_grad, projection.py:650-709).
 *   v [N,V,3] world-space (v_sN = 3V, or 0 for one shared [V,3]);  vt [N,T,2] (vt_sN = 2T or 0)
 *   vi, vti [F,3] int32;  index_img [N,H,W];  bary_img [N,3,H,W];  mask [N,H,W] uint8 or NULL
 *   campos [N,3], camrot [N,3,3], focal [N,2,2].   Pixels with index -1 or mask 0 are written 0.
 * A face with ZERO UV AREA: the reference inverts every face's UV edge matrix up front (face_dpdt,
 * drtk/utils/geometry.py:71-82, th.inverse) and raises for the whole call, visible face or not.  This kernel works
 * per pixel and never inverts that matrix (J^-1 = (A^-1 G)^-1 is evaluated as G^-1 A, A the UV edge matrix, G the
 * projected position edges): pixels of such a face get the finite limit, every other pixel is unaffected.  The one
 * 2x2 inverse left is ill-conditioned for triangles seen edge-on ON SCREEN, where the derivative itself is large; in
 * float32 the result is as near the float64 composite as the reference's float32 composite is, and 10-30x nearer on
 * faces that are slivers in UV space (tests/fuzz_next_ops.py).
 * Forward only.  (The reference composite looks differentiable but is not: it masks the output of linalg.inv_ex in
 * place, drtk/screen_space_uv_derivative.py:79, and backward() through it raises -- recorded from the reference in
 * tests/golden/refpy_uv_derivative_autograd.npz; its consumer, mipmap_grid_sampler_2d, defines no gradient for this
 * input.  The Python wrapper reproduces exactly that: part of the graph, an error only if a gradient arrives.)
 */
int drtk_amd_screen_space_uv_derivative(
    drtk_dtype_t dtype, const void* v, int64_t v_sN, const void* vt, int64_t vt_sN, const int32_t* vi,
    const int32_t* vti, const int32_t* index_img, const void* bary_img, const uint8_t* mask,
    const void* campos, const void* camrot, const void* focal, int64_t N, int64_t V, int64_t T, int64_t F,
    int64_t H, int64_t W, void* out, drtk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * transform_pinhole  -- the vertex stage in front of the path; replaces the pure-PyTorch pinhole
 * branch of drtk.transform (drtk/transform.py:13-119, drtk/utils/projection.py:33-53,486-540):
 *   v_cam = camrot (v - campos);  v_pix = (focal (v_cam.xy / clamp(v_cam.z)) + princpt, v_cam.z)
 * v: [N,V,3] (v_sN = 3V) or ONE shared [V,3] (v_sN = 0); campos [N,3], camrot [N,3,3] row-major,
 * focal [N,2,2], princpt [N,2]; v_pix [N,V,3]; v_cam [N,V,3] or NULL.
 * The backward gives the gradient wrt v only ([N,V,3], or [V,3] already summed over the views
 * when v is shared); camera parameters are treated as constants.
 */
int drtk_amd_transform_pinhole(
    drtk_dtype_t dtype, const void* v, int64_t v_sN, const void* campos, const void* camrot,
    const void* focal, const void* princpt, int64_t N, int64_t V, void* v_pix, void* v_cam,
    drtk_stream_t stream);
int drtk_amd_transform_pinhole_backward(
    drtk_dtype_t dtype, const void* v, int64_t v_sN, const void* campos, const void* camrot,
    const void* focal, const void* princpt, const void* grad_v_pix, int64_t N, int64_t V,
    void* grad_v, drtk_stream_t stream);

/* Diagnostics: compares the rasterizer's reciprocal-based exact division with the IEEE `/` on `count`
 * pseudo-random operand pairs on the device; *d_mismatches (device memory) receives the number of
 * differing results (must be 0). */
int drtk_amd_selftest_exact_div(
    drtk_dtype_t dtype, uint64_t seed, int64_t count, uint64_t* d_mismatches, drtk_stream_t stream);

/* Per-kernel timing for benchmarks (bench.py's `roofline`): between _begin and _report every kernel the library
 * launches -- from any entry point above, on any stream -- is bracketed by a pair of HIP events recorded on the
 * stream it is launched on.  Results are unaffected; outside a collection a launch pays one atomic load.  Do not
 * open a collection while a stream is being captured into a graph (event records would become graph nodes).
 * _report closes the collection, waits for the recorded events and writes one line per launch site in order of
 * first launch, "<kernel as spelled at the launch site>\t<launches>\t<total milliseconds>\n", NUL-terminated, into
 * buf[0..capacity); *needed (optional) receives the size the whole report takes, and a report that did not fit
 * returns DRTK_ERR_WORKSPACE_TOO_SMALL.  Process-global, serialised by a mutex. */
int drtk_amd_kernel_timing_begin(void);
int drtk_amd_kernel_timing_report(char* buf, size_t capacity, size_t* needed);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* DRTK_AMD_H */
