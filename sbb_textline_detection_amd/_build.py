"""Compile libsbbseg.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

Every source is compiled to its own object under ``_obj/`` next to this file (git-ignored, not shipped) and only stale objects are
rebuilt, in parallel; the link step produces ``libsbbseg.so`` next to this file, which is what travels to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(HERE, "_obj")          # package-local (git- and gpurun-ignored): an installed copy never writes outside itself
LIB = os.path.join(HERE, "libsbbseg.so")
SOURCES = ["kernels.hip", "block_x3.hip", "stem_pool_x3.hip", "dec_halo_x3.hip", "dec_halo_f16.hip", "expand_reduce_x3.hip", "conv3_expand_reduce.hip", "region.hip", "api.hip", "loader.cpp"]
HEADERS = [os.path.join(CSRC, "internal.h"), os.path.join(CSRC, "region.h"), os.path.join(HERE, "..", "include", "sbbseg.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _sources():
    paths = [os.path.join(CSRC, s) for s in SOURCES]
    missing = [p for p in paths if not os.path.exists(p)]
    if missing:                                  # a typo here used to surface as an obscure link error / missing symbols
        raise RuntimeError("libsbbseg sources missing: " + ", ".join(missing))
    return paths


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _sources() + HEADERS)


def _stale(src: str, obj: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(p) > t for p in [src] + HEADERS)


# The probe build (`--probes`): the same sources with -DSBBSEG_PROBES, i.e. with the timing probes that make kernels compute WRONG
# results on purpose (internal.h SBBSEG_PROBE) compiled in and their environment switches read.  It is a separate library that nothing
# in the package loads: tools/ scripts pass its path to _capi.load_library() explicitly.
PROBE_OBJ_DIR = os.path.join(HERE, "_obj_probes")
PROBE_LIB = os.path.join(HERE, "..", "tools", "probes", "bin", "libsbbseg_probes.so")


def build(force: bool = False, verbose: bool = False, probes: bool = False) -> str:
    obj_dir, lib, flags = (PROBE_OBJ_DIR, os.path.normpath(PROBE_LIB), FLAGS + ["-DSBBSEG_PROBES"]) if probes else (OBJ_DIR, LIB, FLAGS)
    if not force and not probes and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libsbbseg.so cannot be built (and there is no CPU fallback)")
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    jobs = []
    for src in _sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        if force or _stale(src, obj):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        lang = ["-x", "hip"] if src.endswith(".hip") else []
        res = subprocess.run([hipcc, *flags, *lang, "-c", src, "-o", obj + ".tmp"], capture_output=True, text=True)
        if res.returncode != 0:
            return "hipcc failed on %s:\n%s%s" % (os.path.basename(src), res.stdout, res.stderr)
        os.replace(obj + ".tmp", obj)
        return None

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        errors = [e for e in ex.map(compile_one, jobs) if e]
    if errors:
        raise RuntimeError("\n".join(errors))
    objs = [os.path.join(obj_dir, os.path.basename(s) + ".o") for s in _sources()]
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib + ".tmp"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    os.replace(lib + ".tmp", lib)
    if verbose:
        print("built", lib, "(recompiled: %s)" % (", ".join(os.path.basename(s) for s, _ in jobs) or "nothing, relinked"))
    return lib


if __name__ == "__main__":
    import sys
    build(force="--incremental" not in sys.argv, verbose=True, probes="--probes" in sys.argv)
