/* `drtk/<name>_ext.so`: the importable extension module the reference's loader expects for each operator
 * namespace (drtk/utils/load_torch_ops.py:14-20 imports `drtk.<name>_ext` and hands its __file__ to
 * torch.ops.load_library; the reference's modules are an empty PYBIND11_MODULE, rasterize_module.cpp:73-75).
 * Here: an empty CPython module whose shared object NEEDS drtk_amd/drtk_amd_torch_ops.so, so loading it loads
 * -- once per process -- the library whose static initialisers register all operator namespaces.
 * Built once per name with -DDRTK_EXT_NAME=<name>_ext by drtk_amd/build.py. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#define STR2(x) #x
#define STR(x) STR2(x)
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)

static struct PyModuleDef drtk_ext_def = {
    PyModuleDef_HEAD_INIT, STR(DRTK_EXT_NAME),
    "drtk_amd operator namespace " STR(DRTK_EXT_NAME) " (registered with torch by drtk_amd_torch_ops.so)", -1, NULL,
};

PyMODINIT_FUNC CAT(PyInit_, DRTK_EXT_NAME)(void) {
  return PyModule_Create(&drtk_ext_def);
}
