"""Compile libgb25hip.so (Float32) and libgb25hip_f64.so (Float64) in-tree for gfx950
(called by __graft_entry__.build())."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(_HERE, "csrc", "gb25_api.hip")]
HEADERS = sorted(os.path.join(_HERE, "csrc", n) for n in os.listdir(os.path.join(_HERE, "csrc")) if n.endswith(".hpp")) + \
          [os.path.join(_HERE, "..", "include", "gb25.h")]
OUTPUT = os.path.join(_HERE, "libgb25hip.so")
OUTPUTS = {"Float32": (OUTPUT, "float"), "Float64": (os.path.join(_HERE, "libgb25hip_f64.so"), "double")}


def _stale(path):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build_library(force=False, verbose=False, float_types=("Float32", "Float64")):
    """hipcc --offload-arch=gfx950 -shared: cross-compiles without a GPU.  Returns the Float32 library's path.
    Each library is written to a temporary file and renamed into place while holding a lock file, so that several
    ranks starting at once on a fresh checkout build it once and never dlopen a half-written file."""
    import fcntl
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    try:   # (the lock next to the outputs; a read-only install still serialises its builders through the temp dir)
        lock = open(os.path.join(_HERE, ".build.lock"), "w")
    except OSError:
        import tempfile
        lock = open(os.path.join(tempfile.gettempdir(), "gb25_amd.build.lock"), "w")
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)          # other ranks wait here, then find the library up to date
        procs = []
        for ft in float_types:
            out, ctype = OUTPUTS[ft]
            if not force and not _stale(out):
                continue
            tmp = f"{out}.tmp{os.getpid()}"
            # -fno-slp-vectorize: hipcc otherwise packs neighbouring scalar f32 ops into v_pk_* pairs, which on these
            # stencil kernels costs ~140 v_mov per kernel and 20-30 VGPRs for no throughput gain; measured 177 -> 217
            # steps/s at 1440x720x48 (profiles/r01_tuning_log.md).  Values that are born as pairs are packed by hand.
            # -amdgpu-use-amdgpu-trackers: the scheduler measures register pressure with the AMDGPU-specific trackers; the
            # schedules it then picks for the two tendency kernels at their register limits are better: momentum launch 1.2144
            # -> 1.1965 ms, 421.9 -> 424.2 steps/s (same box, three alternating triples with the max-ilp strategy as the third
            # variant: 409; profiles/r03_tuning_log.md).  Scheduling only: no arithmetic changes.
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                   "-fno-slp-vectorize", "-mllvm", "-amdgpu-use-amdgpu-trackers", f"-DGB25_REAL={ctype}", "-o", tmp] + SOURCES + ["-ldl"]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd), tmp, out))
        err = None
        for cmd, p, tmp, out in procs:
            if p.wait() != 0:
                err = subprocess.CalledProcessError(p.returncode, cmd)
                if os.path.exists(tmp):
                    os.remove(tmp)
            else:
                os.replace(tmp, out)
        if err:
            raise err
    return OUTPUT
