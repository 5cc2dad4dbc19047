"""Build libbiem_mi355.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbiem_mi355.so")
SOURCES = ["abi.cpp", "plan.cpp", "kernels_fill.hip", "kernels_uscat.hip", "kernels_lu.hip"]
HEADERS = ["common.hpp", "plan.hpp", "special.hpp", os.path.join("..", "..", "include", "biem_mi355.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit into one shared library; returns its path."""
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB]
    cmd += os.environ.get("BIEM_HIPCC_FLAGS", "").split()      # experiments only
    cmd += [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
