"""In-tree build of the gfx950 engine: hipcc -> iteres_amd/libiteres_amd.so (+ the C host CLI when present).

    python -m iteres_amd.build [--force]
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libiteres_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "iteres_amd.h")]
    if not (force or _newer(LIB, deps)):
        return LIB
    objs = []
    for s in srcs:
        o = os.path.join(CSRC, os.path.basename(s)[:-4] + ".o")
        if force or _newer(o, deps):
            cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_host(force=False, verbose=False):
    """The C host program (iteres CLI clone) — plain gcc, links libiteres_amd.so."""
    hdir = os.path.join(HERE, "host")
    mk = os.path.join(hdir, "Makefile")
    if not os.path.exists(mk):
        return None
    args = ["make", "-s", "-C", hdir] + (["-B"] if force else [])
    if verbose:
        print(" ".join(args))
    subprocess.check_call(args)
    return os.path.join(hdir, "iteres")


def build_all(force=False, verbose=False):
    lib = build_lib(force, verbose)
    exe = build_host(force, verbose)
    return lib, exe


if __name__ == "__main__":
    out = build_all(force="--force" in sys.argv, verbose=True)
    print(out)
