#!/usr/bin/env python3
"""Build the drtk_amd native libraries in-tree (gfx950 only).

  drtk_amd/libdrtk_amd.so        HIP kernels + C ABI (include/drtk_amd.h); hipcc, no torch dependency
  drtk_amd/drtk_amd_torch_ops.so torch-op shim (rasterize_ext / render_ext / interpolate_ext /
                                 edge_grad_ext schemas + autograd), g++ against libtorch, links the above
  drtk/<name>_ext.so             importable extension modules of the `import drtk` drop-in (gcc, CPython API)

`python drtk_amd/build.py` builds both (`--force` rebuilds, `--dry-run` only reports what is missing or stale);
__graft_entry__.build() calls build_all().  Run the FILE, not `python -m drtk_amd.build`: `-m` imports the package
first, and the package refuses to import before these libraries exist (no fallback path) -- which is why this
module imports nothing from it.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(ROOT, "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

KERNEL_SRCS = ["rasterize.hip", "rasterize_lines.hip", "render.hip", "interpolate.hip", "edge_grad.hip", "transform.hip", "interp_matrix.hip", "mipmap.hip", "uv_derivative.hip", "capi.hip"]
HEADERS = ["common.hpp", "segscatter.hpp"]
LIB = os.path.join(PKG, "libdrtk_amd.so")
OPS = os.path.join(PKG, "drtk_amd_torch_ops.so")
EXPORTS_MAP = os.path.join(CSRC, "exports.map")
# the C ABI and nothing else in the dynamic symbol table: -fvisibility=hidden (include/drtk_amd.h pushes `default` around
# its declarations) + a version script that also localises what the toolchain adds (__hip_cuid_*)
LINK_FLAGS = [f"--offload-arch={ARCH}", "-shared", "-fPIC", f"-Wl,--version-script={EXPORTS_MAP}"]

# No fast-math, no FMA contraction: coverage/depth/classification are exact float decisions.
HIP_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
    "-fno-gpu-rdc", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical", f"-I{INC}", f"-I{CSRC}",
]


def _kernel_deps():
    return [os.path.join(CSRC, f) for f in KERNEL_SRCS + HEADERS] + [EXPORTS_MAP, os.path.join(INC, "drtk_amd.h"), __file__]


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed:\n" + " ".join(cmd) + "\n" + r.stdout[-6000:])
    return r.stdout


def _digest(deps, extra=""):
    import hashlib

    h = hashlib.sha256(extra.encode())
    for d in sorted(deps):
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _newer(target, deps, extra=""):
    """Is `target` the build of exactly these sources?  Decided by CONTENT, not by time stamps: the libraries travel
    prebuilt (to the GPU box, between checkouts), where modification times mean nothing -- a sidecar `<target>.src`
    holds the SHA-256 of the sources (and flags) the library was built from."""
    side = target + ".src"
    if not (os.path.isfile(target) and os.path.isfile(side) and all(os.path.isfile(d) for d in deps)):
        return False
    with open(side) as f:
        return f.read().strip() == _digest(deps, extra)


def _stamp(target, deps, extra=""):
    with open(target + ".src", "w") as f:
        f.write(_digest(deps, extra) + "\n")


def check_no_vgpr_spills(compiler_output, what):
    """No kernel of the library may spill vector registers.  A spill is not only slow: ROCm 7.2's register allocator was
    seen to place a spill STORE in a block that runs with an empty EXEC mask (straight after a divergent loop's exit,
    before the mask is restored), so the value never reached its slot and the reload returned a stale one -- in
    tile_raster that silently dropped almost half of the work items (round 3).  Kernels are kept below their register
    bound by construction; this check (over -Rpass-analysis=kernel-resource-usage) keeps it that way."""
    import re

    name, bad = "?", []
    for line in compiler_output.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and int(m.group(1)) != 0:
            bad.append(f"{name}: {m.group(1)} VGPRs spilled")
    if bad:
        raise RuntimeError(f"{what}: vector register spills are not allowed (see build.py check_no_vgpr_spills):\n  " + "\n  ".join(bad))


def build_kernels(force=False, verbose=True):
    deps = _kernel_deps()
    if not force and _newer(LIB, deps, " ".join(HIP_FLAGS)):
        return LIB
    objs, cmds = [], []
    for s in KERNEL_SRCS:
        o = os.path.join(CSRC, s + ".o")
        objs.append(o)
        cmds.append([HIPCC, *HIP_FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, s), "-o", o])
    with ThreadPoolExecutor(max_workers=len(cmds)) as ex:
        outs = list(ex.map(_run, cmds))
    for s, out in zip(KERNEL_SRCS, outs):
        check_no_vgpr_spills(out, s)
    outs = ["\n".join(l for l in out.splitlines() if "remark:" not in l and "[-Rpass-analysis" not in l and l.strip() not in ("^",) and not l.lstrip().startswith(("|", "^")) and not l.strip().split(" | ")[0].strip().isdigit()) for out in outs]
    _run([HIPCC, *LINK_FLAGS, "-o", LIB, *objs])
    for o in objs:
        os.remove(o)
    _stamp(LIB, deps, " ".join(HIP_FLAGS))
    if verbose:
        msg = "".join(outs).strip()
        print(f"[drtk_amd] built {LIB}" + (("\n" + msg) if msg else ""))
    return LIB


ABLATE_LIB = os.path.join(ROOT, "profiles", "libdrtk_amd_ablate.so")


def build_ablation(force=False, verbose=True):
    """profiles/libdrtk_amd_ablate.so: the same sources with -DDRTK_AMD_ABLATION, i.e. WITH the phase switches
    and `drtk_amd_debug_set_flags` (csrc/common.hpp).  A profiling tool's library (profiles/kernel_bench.py
    --flags ...); it lives outside the package, the product never loads it and build_all() does not build it."""
    deps = _kernel_deps()
    if not force and _newer(ABLATE_LIB, deps, "ablation"):
        return ABLATE_LIB
    objs, cmds = [], []
    for s in KERNEL_SRCS:
        o = os.path.join(CSRC, s + ".ablate.o")
        objs.append(o)
        cmds.append([HIPCC, *HIP_FLAGS, "-DDRTK_AMD_ABLATION", "-c", os.path.join(CSRC, s), "-o", o])
    with ThreadPoolExecutor(max_workers=len(cmds)) as ex:
        list(ex.map(_run, cmds))
    _run([HIPCC, *LINK_FLAGS, "-o", ABLATE_LIB, *objs])
    for o in objs:
        os.remove(o)
    _stamp(ABLATE_LIB, deps, "ablation")
    if verbose:
        print(f"[drtk_amd] built {ABLATE_LIB}")
    return ABLATE_LIB


def build_variant(out, defines, verbose=True):
    """A/B helper for kernel work: the library compiled with extra -D switches into `out` (profiles/kernel_bench.py --lib)."""
    objs, cmds = [], []
    for s in KERNEL_SRCS:
        o = os.path.join(CSRC, s + "." + os.path.basename(out) + ".o")
        objs.append(o)
        # DEFINE[=value] becomes -DDEFINE[=value]; anything that starts with '-' is passed to hipcc as it is
        cmds.append([HIPCC, *HIP_FLAGS, *[d if d.startswith("-") else f"-D{d}" for d in defines], "-c", os.path.join(CSRC, s), "-o", o])
    with ThreadPoolExecutor(max_workers=len(cmds)) as ex:
        list(ex.map(_run, cmds))
    _run([HIPCC, *LINK_FLAGS, "-o", out, *objs])
    for o in objs:
        os.remove(o)
    if verbose:
        print(f"[drtk_amd] built {out} with {defines}")
    return out


OPS_DIR = os.path.join(CSRC, "torch_ops")
OPS_SRCS = ["rasterize.cpp", "render.cpp", "interpolate.cpp", "interp_matrix.cpp", "mipmap.cpp", "edge_grad.cpp", "transform.cpp"]


def _ops_deps():
    return [os.path.join(OPS_DIR, f) for f in OPS_SRCS + ["common.hpp"]] + [os.path.join(INC, "drtk_amd.h"), LIB + ".src", __file__]


def build_torch_ops(force=False, verbose=True):
    """drtk_amd_torch_ops.so from csrc/torch_ops/*.cpp: one translation unit per operator family, compiled side by side
    (each pulls in the torch headers: ~40 s a piece), linked against libdrtk_amd.so."""
    import torch
    from torch.utils import cpp_extension as ce

    deps = _ops_deps()
    if not force and _newer(OPS, deps):
        return OPS
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{INC}", "-I/opt/rocm/include"]
    cxx = os.environ.get("CXX", "g++")
    flags = [
        "-std=c++17", "-O2", "-fPIC", "-Wall", "-Wno-array-bounds", "-Wno-unknown-pragmas", "-D__HIP_PLATFORM_AMD__=1",
        "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", *inc,
    ]
    objdir = os.path.join(PKG, "_obj")
    os.makedirs(objdir, exist_ok=True)
    objs = [os.path.join(objdir, "torch_ops_" + f[:-4] + ".o") for f in OPS_SRCS]
    jobs = [[cxx, *flags, "-c", os.path.join(OPS_DIR, f), "-o", o] for f, o in zip(OPS_SRCS, objs)]
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
        out = "".join(pool.map(_run, jobs))
    out += _run([
        cxx, "-shared", *objs, "-o", OPS, f"-L{PKG}", "-ldrtk_amd", "-Wl,-rpath,$ORIGIN", f"-L{libdir}", "-ltorch",
        "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip", f"-Wl,-rpath,{libdir}",
    ])
    shutil.rmtree(objdir, ignore_errors=True)
    _stamp(OPS, deps)
    if verbose:
        print(f"[drtk_amd] built {OPS}" + (("\n" + out.strip()) if out.strip() else ""))
    return OPS


EXT_NAMES = ["rasterize_ext", "render_ext", "interpolate_ext", "edge_grad_ext", "mipmap_grid_sampler_ext"]
SHIM = os.path.join(ROOT, "drtk")


def build_ext_modules(force=False, verbose=True):
    """drtk/<name>_ext.so for the `import drtk` drop-in: importable CPython modules (PyInit_<name>_ext) that pull
    in drtk_amd_torch_ops.so -- see csrc/ext_module.c."""
    import sysconfig

    src = os.path.join(CSRC, "ext_module.c")
    outs = []
    for name in EXT_NAMES:
        so = os.path.join(SHIM, name + ".so")
        outs.append(so)
        if not force and _newer(so, [src, OPS + ".src", __file__], name):
            continue
        _run([
            os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-Wall", f"-DDRTK_EXT_NAME={name}",
            f"-I{sysconfig.get_paths()['include']}", src, "-o", so, "-Wl,--no-as-needed", f"-L{PKG}",
            "-l:drtk_amd_torch_ops.so", "-Wl,-rpath,$ORIGIN/../drtk_amd",
        ])
        _stamp(so, [src, OPS + ".src", __file__], name)
        if verbose:
            print(f"[drtk_amd] built {so}")
    return outs


def build_all(force=False, verbose=True):
    """The product: kernels + C ABI, torch-operator shim, `import drtk` modules.  (Until round 5 a second copy of the
    library carried the rasterizer's other depth order; it is a run-time setting of the one library now:
    include/drtk_amd.h drtk_amd_set_depth_order.)"""
    build_kernels(force=force, verbose=verbose)
    stale = os.path.join(PKG, "libdrtk_amd_depth_fastmath.so")  # a leftover of such a build would only mislead
    for f in (stale, stale + ".src"):
        if os.path.isfile(f):
            os.remove(f)
    build_torch_ops(force=force, verbose=verbose)
    build_ext_modules(force=force, verbose=verbose)
    return LIB, OPS


def _state(target, deps, extra=""):
    if not os.path.isfile(target):
        return "missing"
    return "up to date" if _newer(target, deps, extra) else "stale"


def dry_run():
    """What build_all() would do, without compiling anything."""
    kdeps = _kernel_deps()
    odeps = _ops_deps()
    for target, deps, extra in ((LIB, kdeps, " ".join(HIP_FLAGS)), (OPS, odeps, "")):
        print(f"[drtk_amd] {target}: {_state(target, deps, extra)}")


if __name__ == "__main__":
    if "--dry-run" in sys.argv:
        dry_run()
    elif "--variant" in sys.argv:  # python drtk_amd/build.py --variant out.so DEFINE[=value] ...
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    elif "--ablation" in sys.argv:
        build_ablation(force="--force" in sys.argv)
    else:
        build_all(force="--force" in sys.argv)
