#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- builds the reference's own CPU kernels into oracle/_ref/.

Recipe (runs only where /root/reference exists, i.e. in the build container; the GPU box only
ever sees the resulting .so files, which are git-ignored but travel with the gpurun snapshot):

  * sources, compiled from where they lie, unmodified:
      /root/reference/src/rasterize/rasterize_kernel_cpu.cpp
      /root/reference/src/render/render_kernel_cpu.cpp
      /root/reference/src/interpolate/interpolate_kernel_cpu.cpp
      /root/reference/src/edge_grad/edge_grad_kernel_cpu.cpp
    plus oracle/ref_driver.cpp (ours: a torch-op shim that forwards to the *_cpu entry points).
  * compiler: g++ (the host compiler the reference's setup.py uses for its *.cpp files), with the
    genuine NVIDIA `cuda_runtime.h` that this image ships inside the triton wheel
    (<site-packages>/triton/backends/nvidia/include) on the include path.  That is the header
    src/include/cuda_math_helper.h:11 asks for on its plain-C++ branch (std::min/std::max,
    <cmath>) -- no stand-in headers, no stubs.
  * build_modules(): the reference's OWN torch-op modules (src/<x>/<x>_module.cpp: schemas, the four
    C++ autograd Functions, autocast wrappers) + the same *_kernel_cpu.cpp, one .so per extension
    (oracle/_ref/<x>_ext.so), so that the reference's Python package (drtk/*.py) runs end to end on
    CPU in the build container -- used by oracle/gen_golden_refpy.py only.  The modules also name the
    CUDA launchers (<x>_cuda, defined in *.cu: they need nvcc, which this image lacks, so they are NOT
    built and NOT replaced by any code of ours): at link time each of those names is made an ALIAS of the
    reference's own CPU twin of the same signature (`-Wl,--defsym,<mangled x_cuda>=<mangled x_cpu>`,
    names read from the objects with nm; torch refuses to register a null kernel pointer, so the
    symbols cannot simply stay undefined), i.e. the CUDA dispatch-key slots and the `is_cuda()`
    branches -- which CPU tensors never reach -- point at reference code, and nothing is written in
    their place.  These libraries register the reference's real namespaces
    (rasterize_ext, ...), so they can never be loaded next to drtk_amd's own shim: the generator is a
    process of its own.
    NOTE (measured here): the other route -- hipcc host-only, which makes the header take its
    `<hip/hip_runtime.h>` branch -- compiles but is WRONG on the host: `using ::min/::max`
    (cuda_math_helper.h:114-121) then binds to HIP's host-side `int max(int,int)`, so
    epsclamp(0.3f) == 0 and depths come out inf.  It is not used.
  * two variants, different torch namespaces so both can be loaded in one process:
      strict -> libdrtk_ref_strict.so : -O2 -ffp-contract=off -fno-fast-math   (source-order IEEE)
      fast   -> libdrtk_ref_fast.so   : -O3 --fast-math                  (reference's setup.py:23-24)
    The strict build is what the oracle restatement (oracle/drtk_oracle.c) is pinned against
    bit-for-bit; the fast build documents how far the reference's own flags move the results.

Never run the reference's own setup.py; never copy its sources.
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DRTK_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")
CXX = os.environ.get("CXX", "g++")

SRCS = [
    "rasterize/rasterize_kernel_cpu.cpp",
    "render/render_kernel_cpu.cpp",
    "interpolate/interpolate_kernel_cpu.cpp",
    "edge_grad/edge_grad_kernel_cpu.cpp",
]
VARIANTS = {
    "strict": ["-O2", "-ffp-contract=off", "-fno-fast-math"],
    "fast": ["-O3", "--fast-math"],
}


def cuda_include_dir():
    """Directory of the real cuda_runtime.h shipped with the triton wheel (None if absent)."""
    try:
        import importlib.util

        spec = importlib.util.find_spec("triton")
        if spec is None or not spec.submodule_search_locations:
            return None
        d = os.path.join(list(spec.submodule_search_locations)[0], "backends", "nvidia", "include")
        return d if os.path.isfile(os.path.join(d, "cuda_runtime.h")) else None
    except Exception:
        return None


def available() -> bool:
    return all(os.path.isfile(os.path.join(REF, "src", s)) for s in SRCS) and cuda_include_dir() is not None


def _torch_flags():
    import torch
    from torch.utils import cpp_extension as ce

    inc = [f"-I{p}" for p in ce.include_paths()]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    return inc, libdir, abi


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed:\n" + " ".join(cmd) + "\n" + r.stdout[-4000:])
    return r.stdout


def build(variants=("strict", "fast"), force=False, verbose=True):
    if not available():
        raise RuntimeError(f"reference sources under {REF} or triton's cuda_runtime.h not found; nothing to build")
    os.makedirs(OUT, exist_ok=True)
    inc, libdir, abi = _torch_flags()
    common = [
        CXX, "-std=c++17", "-fPIC", "-w", "-DNO_PYBIND", f"-D_GLIBCXX_USE_CXX11_ABI={abi}",
        f"-I{REF}/src/include", f"-I{cuda_include_dir()}",
    ] + [f"-I{REF}/src/{d}" for d in ("rasterize", "render", "interpolate", "edge_grad")] + inc
    jobs = []
    for var in variants:
        so = os.path.join(OUT, f"libdrtk_ref_{var}.so")
        srcs = [os.path.join(REF, "src", s) for s in SRCS] + [os.path.join(HERE, "ref_driver.cpp")]
        if (not force) and os.path.isfile(so) and all(
            os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs + [__file__]
        ):
            if verbose:
                print(f"[ref_build] {so} up to date")
            continue
        objs = []
        for s in srcs:
            o = os.path.join(OUT, f"{var}_{os.path.basename(s)}.o")
            objs.append(o)
            jobs.append(common + VARIANTS[var] + [f"-DREF_NS=drtk_ref_{var}", "-c", s, "-o", o])
        jobs.append(("link", var, so, objs, libdir))
    compiles = [j for j in jobs if isinstance(j, list)]
    links = [j for j in jobs if isinstance(j, tuple)]
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(compiles)))) as ex:
        list(ex.map(_run, compiles))
    for _, var, so, objs, libdir in links:
        _run([
            "g++", "-shared", "-o", so, *objs, "-Wl,-Bsymbolic", f"-L{libdir}",
            "-ltorch", "-ltorch_cpu", "-lc10", f"-Wl,-rpath,{libdir}",
        ])
        for o in objs:
            os.remove(o)
        if verbose:
            print(f"[ref_build] built {so}")
    return [os.path.join(OUT, f"libdrtk_ref_{v}.so") for v in variants]


MODULES = ("rasterize", "render", "interpolate", "edge_grad")


def module_path(name):
    return os.path.join(OUT, f"{name}_ext.so")


def build_modules(force=False, verbose=True, flags=("-O2", "-ffp-contract=off", "-fno-fast-math")):
    """oracle/_ref/<x>_ext.so = /root/reference/src/<x>/<x>_module.cpp + <x>_kernel_cpu.cpp (strict
    flags, like the fixtures), CUDA launcher symbols left null -- see the module docstring."""
    if not available():
        raise RuntimeError(f"reference sources under {REF} or triton's cuda_runtime.h not found; nothing to build")
    os.makedirs(OUT, exist_ok=True)
    inc, libdir, abi = _torch_flags()
    common = [
        CXX, "-std=c++17", "-fPIC", "-w", "-DNO_PYBIND", f"-D_GLIBCXX_USE_CXX11_ABI={abi}",
        f"-I{REF}/src/include", f"-I{cuda_include_dir()}", *inc, *flags,
    ]

    def one(name):
        so = module_path(name)
        srcs = [os.path.join(REF, "src", name, f"{name}_module.cpp"), os.path.join(REF, "src", name, f"{name}_kernel_cpu.cpp")]
        if (not force) and os.path.isfile(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs + [__file__]):
            return f"[ref_build] {so} up to date"
        objs = []
        for s in srcs:
            o = os.path.join(OUT, f"mod_{os.path.basename(s)}.o")
            _run(common + ["-c", s, "-o", o])
            objs.append(o)
        undefined = {u for u in _run(["nm", "-u", objs[0]]).split() if u.startswith("_Z")}
        alias = []
        for u in sorted(undefined):
            m = re.match(r"_Z(\d+)([A-Za-z0-9_]+?)(RKN2at6Tensor.*)$", u)
            if m is None or "_cuda" not in m.group(2)[: int(m.group(1))]:
                continue
            name = m.group(2)[: int(m.group(1))]
            rest = m.group(2)[int(m.group(1)):] + m.group(3)
            twin = name.replace("_cuda", "_cpu")
            cpu = f"_Z{len(twin)}{twin}{rest}"
            if cpu not in undefined:
                raise RuntimeError(f"{u}: no CPU twin {cpu} among the module's references")
            alias.append(f"-Wl,--defsym,{u}={cpu}")
        _run(["g++", "-shared", "-o", so, *objs, *alias, f"-L{libdir}",
              "-ltorch", "-ltorch_cpu", "-lc10", f"-Wl,-rpath,{libdir}"])
        for o in objs:
            os.remove(o)
        return f"[ref_build] built {so} (CUDA launcher names aliased to their CPU twins: {len(alias)})"

    with ThreadPoolExecutor(max_workers=4) as ex:
        for msg in ex.map(one, MODULES):
            if verbose:
                print(msg)
    return [module_path(m) for m in MODULES]


def load(variant="strict"):
    """Load oracle/_ref/libdrtk_ref_<variant>.so and return torch.ops.drtk_ref_<variant>."""
    import torch

    so = os.path.join(OUT, f"libdrtk_ref_{variant}.so")
    if not os.path.isfile(so):
        raise FileNotFoundError(so)
    torch.ops.load_library(so)
    return getattr(torch.ops, f"drtk_ref_{variant}")


if __name__ == "__main__":
    build(force="--force" in sys.argv)
