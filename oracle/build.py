#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- builds oracle/libdrtk_oracle.so (the CPU restatement) with gcc.

Flags matter: strict IEEE, no contraction, no fast-math -- the restatement's operand order IS the
specification the HIP kernels are compared against.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libdrtk_oracle.so")
SRCS = [os.path.join(HERE, f) for f in ("drtk_oracle.c", "drtk_oracle_body.inc", "drtk_oracle_mipmap.inc", "drtk_oracle.h")]


def build(force=False, verbose=True):
    if (not force) and os.path.isfile(SO) and all(
        os.path.getmtime(SO) >= os.path.getmtime(s) for s in SRCS + [__file__]
    ):
        return SO
    cmd = [
        os.environ.get("CC", "gcc"), "-std=c11", "-O2", "-ffp-contract=off", "-fno-fast-math",
        "-fno-unsafe-math-optimizations", "-fopenmp", "-fPIC", "-shared", "-Wall", "-Wextra",
        "-o", SO, SRCS[0], "-lm",
    ]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout)
    if verbose:
        print(f"[oracle] built {SO}" + (("\n" + r.stdout) if r.stdout.strip() else ""))
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
