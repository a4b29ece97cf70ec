"""Loader for the native torch-operator library.

The reference loads one `drtk/<name>_ext*.so` per extension through `importlib` +
`torch.ops.load_library` (drtk/utils/load_torch_ops.py:14-28).  drtk_amd ships its operator
namespaces (`rasterize_ext`, `render_ext`, `interpolate_ext`, `edge_grad_ext`,
`mipmap_grid_sampler_ext`) in ONE in-tree
library, `drtk_amd/drtk_amd_torch_ops.so`, which links `drtk_amd/libdrtk_amd.so` (HIP kernels +
C ABI).  Loading fails loudly: there is no eager/PyTorch fallback for these ops.
"""
import os
import sys
import threading

import torch as th

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_OPS = os.path.join(_PKG, "drtk_amd_torch_ops.so")
_LIB = os.path.join(_PKG, "libdrtk_amd.so")
_lock = threading.Lock()
_loaded = False

_NAMESPACES = {
    "drtk.rasterize_ext": "rasterize_ext",
    "drtk.render_ext": "render_ext",
    "drtk.interpolate_ext": "interpolate_ext",
    "drtk.edge_grad_ext": "edge_grad_ext",
    "drtk.mipmap_grid_sampler_ext": "mipmap_grid_sampler_ext",
}


def native_library_paths():
    """(path of the C-ABI kernel library, path of the torch-op shim)."""
    return _LIB, _OPS


def load_torch_ops(extension: str = "drtk.rasterize_ext") -> None:
    """Make `torch.ops.<extension>` available.  `extension` uses the reference's dotted names
    ("drtk.rasterize_ext", ...) or the drtk_amd spelling ("drtk_amd.rasterize_ext")."""
    global _loaded
    name = extension.replace("drtk_amd.", "drtk.")
    if name not in _NAMESPACES:
        raise ImportError(f"drtk_amd provides {sorted(_NAMESPACES)}; '{extension}' is not one of them")
    with _lock:
        if _loaded:
            return
        missing = [p for p in (_LIB, _OPS) if not os.path.isfile(p)]
        if missing:
            # Same escape hatch as the reference: documentation builds may import without binaries.
            if "sphinx" in sys.modules:
                return
            raise ImportError(
                "drtk_amd native libraries are not built: missing "
                + ", ".join(missing)
                + f". Run `python {os.path.join(_PKG, 'build.py')}` (needs hipcc for gfx950; run the file -- "
                "`python -m drtk_amd.build` would import this package first and end up here again)."
            )
        th.ops.load_library(_OPS)
        _loaded = True
