"""`interpolate`, `interpolation_matrix`, `interpolation_normal_matrix` -- host-side mirror of
drtk/interpolate.py:20-193."""
import torch as th
from drtk_amd.utils import load_torch_ops

load_torch_ops("drtk.interpolate_ext")


@th.compiler.disable
def interpolate(
    vert_attributes: th.Tensor,
    vi: th.Tensor,
    index_img: th.Tensor,
    bary_img: th.Tensor,
) -> th.Tensor:
    """Barycentric interpolation of per-vertex attributes over the rasterized fragments.

    Args:
        vert_attributes: `[N, V, C]`.
        vi: `[F, 3]` or `[N, F, 3]` int32.
        index_img: `[N, H, W]` int32.
        bary_img: `[N, 3, H, W]`, same dtype as `vert_attributes`.

    Returns:
        `[N, C, H, W]`.  Pixels with `index_img == -1` hold a +-1 coordinate sweep (even channels:
        x, odd: y), exactly like the reference -- mask them out before use.  Gradients flow to
        `vert_attributes` and `bary_img`.
    """
    if vi.ndim == 2:
        vi = vi[None].expand(vert_attributes.shape[0], -1, -1)
    return th.ops.interpolate_ext.interpolate(vert_attributes, vi, index_img, bary_img)


@th.compiler.disable
def interpolate_masked(
    vert_attributes: th.Tensor,
    vi: th.Tensor,
    index_img: th.Tensor,
    bary_img: th.Tensor,
) -> th.Tensor:
    """drtk_amd extension: `interpolate(...) * (index_img != -1)[:, None]` in one pass.

    Every DRTK pipeline masks the background of `interpolate`'s output (the reference fills it with
    a coordinate sweep).  Done with torch ops that costs a full read+write of the image forward and
    another one backward; here the kernel writes 0 there itself and the backward is `interpolate`'s
    own (which never reads the upstream gradient of a background pixel).  Values and gradients are
    identical to the two-op form.
    """
    if vi.ndim == 2:
        vi = vi[None].expand(vert_attributes.shape[0], -1, -1)
    return th.ops.drtk_amd_ext.interpolate_masked(vert_attributes, vi, index_img, bary_img)


def _broadcast_vi(vi: th.Tensor, n: int) -> th.Tensor:
    # drtk/interpolate.py:107-110 : [F,3] and [1,F,3] broadcast to the batch with a stride-0 expand
    if vi.ndim == 2:
        return vi[None].expand(n, -1, -1)
    if vi.ndim == 3 and vi.shape[0] == 1 and n != 1:
        return vi.expand(n, -1, -1)
    return vi


@th.compiler.disable
def interpolation_matrix(
    vi: th.Tensor,
    index_img: th.Tensor,
    bary_img: th.Tensor,
    num_vertices: int,
) -> th.Tensor:
    """Sparse pixel-to-vertex interpolation matrix `A` with `pixel_values = A @ X`.

    One CSR row per foreground pixel, in flattened `[N, H, W]` order with `index_img == -1` pixels
    skipped; three entries per row -- the barycentric weights at the triangle's three vertex
    columns, columns sorted ascending within the row.  Faces must have three distinct vertex
    indices and `num_vertices` must exceed every index in `vi` (not validated, as in the
    reference: drtk/interpolate.py:72-82).  Gradients flow from the sparse values to `bary_img`
    only.  The row count is data dependent, so the call synchronises (as the reference's does).

    Args:
        vi: `[F, 3]`, `[1, F, 3]` or `[N, F, 3]` int32.
        index_img: `[N, H, W]` int32.
        bary_img: `[N, 3, H, W]`.
        num_vertices: number of CSR columns.

    Returns:
        sparse CSR tensor `[num_foreground_pixels, num_vertices]` with `3 * rows` values.
    """
    vi = _broadcast_vi(vi, index_img.shape[0])
    crow_indices, col_indices, values, row_pixels = th.ops.interpolate_ext.interpolation_matrix(vi, index_img, bary_img)
    return th.sparse_csr_tensor(
        crow_indices,
        col_indices,
        values,
        size=(int(row_pixels.numel()), int(num_vertices)),
        device=values.device,
        dtype=values.dtype,
        check_invariants=False,
    )


@th.compiler.disable
def interpolation_normal_matrix(
    vi: th.Tensor,
    index_img: th.Tensor,
    bary_img: th.Tensor,
    num_vertices: int,
) -> th.Tensor:
    """Sparse normal matrix `A.T @ A` of :func:`interpolation_matrix`, assembled without forming `A`.

    Every foreground pixel adds its nine `bary_i * bary_j` products into the CSR entries of the
    owning triangle's directed vertex pairs.  The sparsity pattern depends on topology only; it is
    built once per face-index tensor (on the device, keyed on the tensor's identity and version
    counter -- keep `vi` alive and unmodified across iterations to stay on cache hits) and only the
    values are recomputed per call.  Gradients flow to `bary_img` only (product rule).

    Args / shapes as :func:`interpolation_matrix`.

    Returns:
        sparse CSR tensor `[num_vertices, num_vertices]`.
    """
    vi = _broadcast_vi(vi, index_img.shape[0])
    crow_indices, col_indices, values = th.ops.interpolate_ext.interpolation_normal_matrix(
        vi, index_img, bary_img, int(num_vertices)
    )
    return th.sparse_csr_tensor(
        crow_indices,
        col_indices,
        values,
        size=(int(num_vertices), int(num_vertices)),
        device=values.device,
        dtype=values.dtype,
        check_invariants=False,
    )
