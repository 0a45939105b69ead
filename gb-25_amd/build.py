"""Compile libgb25hip.so in-tree for gfx950 (called by __graft_entry__.build())."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(_HERE, "csrc", "gb25_api.hip")]
HEADERS = [os.path.join(_HERE, "csrc", n) for n in ("kernels.hpp", "kernels_v2.hpp", "device_common.hpp")] + \
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
    # -fno-slp-vectorize: hipcc otherwise packs neighbouring scalar f32 ops into v_pk_* pairs, which on these
    # stencil kernels costs ~140 v_mov per kernel and 20-30 VGPRs (k_gu: 94 -> 70, tracers: 82 -> 61) for no
    # throughput gain; measured 177 -> 217 steps/s at 1440x720x48 (profiles/r01_tuning_log.md).
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-fno-slp-vectorize", "-o", OUTPUT] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUTPUT
