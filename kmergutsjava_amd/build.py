"""Build the native pieces in-tree (hipcc cross-compiles gfx950 without a GPU).

    python -m kmergutsjava_amd.build            # libkmerguts_hip.so
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkmerguts_hip.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def _stale(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_native(force: bool = False, verbose: bool = False) -> str:
    # the translation unit first, then everything it includes (a header edit must trigger a rebuild)
    srcs = [os.path.join(CSRC, "kmerguts_hip.hip")] + sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join(ROOT, "include", "kmerguts_hip.h")]
    if force or _stale(LIB, srcs):
        cmd = [_hipcc(), *HIPCC_FLAGS, "-o", LIB, srcs[0], "-lz", "-lpthread"]      # zlib: kmer.table.mem_map.gz
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB


CLI = os.path.join(HERE, "kmer_guts")


def build_cli(force: bool = False, verbose: bool = False) -> str:
    """The native command line (C++ over the C ABI): kmergutsjava_amd/kmer_guts."""
    build_native(False, verbose)
    src = os.path.join(CSRC, "kmer_guts_cli.cpp")
    if force or _stale(CLI, [src, os.path.join(ROOT, "include", "kmerguts_hip.h")]):
        cxx = shutil.which("g++") or shutil.which("hipcc")
        cmd = [cxx, "-O2", "-std=c++17", "-Wall", "-pthread", "-o", CLI, src, "-L" + HERE, "-lkmerguts_hip", "-lz",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return CLI


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
    print(build_cli(force="--force" in sys.argv, verbose=True))
