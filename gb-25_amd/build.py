"""Compile libgb25hip.so (Float32) and libgb25hip_f64.so (Float64) in-tree for gfx950
(called by __graft_entry__.build())."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(_HERE, "csrc", "gb25_api.hip")]
HEADERS = [os.path.join(_HERE, "csrc", n) for n in ("kernels.hpp", "kernels_v2.hpp", "device_common.hpp")] + \
          [os.path.join(_HERE, "..", "include", "gb25.h")]
OUTPUT = os.path.join(_HERE, "libgb25hip.so")
OUTPUTS = {"Float32": (OUTPUT, "float"), "Float64": (os.path.join(_HERE, "libgb25hip_f64.so"), "double")}


def _stale(path):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build_library(force=False, verbose=False, float_types=("Float32", "Float64")):
    """hipcc --offload-arch=gfx950 -shared: cross-compiles without a GPU.  Returns the Float32 library's path."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    procs = []
    for ft in float_types:
        out, ctype = OUTPUTS[ft]
        if not force and not _stale(out):
            continue
        # -fno-slp-vectorize: hipcc otherwise packs neighbouring scalar f32 ops into v_pk_* pairs, which on these
        # stencil kernels costs ~140 v_mov per kernel and 20-30 VGPRs (k_gu: 94 -> 70, tracers: 82 -> 61) for no
        # throughput gain; measured 177 -> 217 steps/s at 1440x720x48 (profiles/r01_tuning_log.md).
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
               "-fno-slp-vectorize", f"-DGB25_REAL={ctype}", "-o", out] + SOURCES
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    return OUTPUT
