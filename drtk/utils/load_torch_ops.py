"""drtk.utils.load_torch_ops of the drop-in (drtk/utils/load_torch_ops.py:14-28): takes the reference's extension
names ("drtk.rasterize_ext", "drtk.render_ext", "drtk.interpolate_ext", "drtk.edge_grad_ext",
"drtk.mipmap_grid_sampler_ext") and makes `torch.ops.<name>` available from drtk_amd's operator library.  Each
name is also importable as a module (`drtk/<name>.so`, exporting `PyInit_<name>` like the reference's extensions:
rasterize_module.cpp:73-75), so the reference's own three-line loader -- import the module, hand its `__file__` to
`torch.ops.load_library` -- works on them as well.  An unknown name raises ImportError, a missing build too."""
from drtk_amd.utils.load_torch_ops import load_torch_ops  # noqa: F401
