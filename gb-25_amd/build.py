"""Compile libgb25hip.so in-tree for gfx950 (called by __graft_entry__.build())."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(_HERE, "csrc", "gb25_api.hip")]
HEADERS = [os.path.join(_HERE, "csrc", n) for n in ("kernels.hpp", "device_common.hpp")] + \
          [os.path.join(_HERE, "..", "include", "gb25.h")]
OUTPUT = os.path.join(_HERE, "libgb25hip.so")


def _stale():
    if not os.path.exists(OUTPUT):
        return True
    t = os.path.getmtime(OUTPUT)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared: cross-compiles without a GPU."""
    if not force and not _stale():
        return OUTPUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-o", OUTPUT] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUTPUT
