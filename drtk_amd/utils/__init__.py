from drtk_amd.utils.load_torch_ops import load_torch_ops, native_library_paths  # noqa: F401
