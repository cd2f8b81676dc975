"""Compile libsbbseg.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsbbseg.so")
SOURCES = ["kernels.hip", "api.hip", "loader.cpp"]
HEADERS = [os.path.join(CSRC, "internal.h"), os.path.join(HERE, "..", "include", "sbbseg.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libsbbseg.so cannot be built (and there is no CPU fallback)")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB + ".tmp"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB + ".tmp", LIB)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
