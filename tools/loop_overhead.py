"""Fixed cost of a synchronised gb25_loop call: wall time of loops of n steps, each bracketed by gb25_synchronize, fitted as
a + b n.  usage: python tools/loop_overhead.py [LIB]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gb25_amd.binding import Config
import torch  # noqa: F401  (one HIP runtime per process: torch's, loaded first)
lib = C.CDLL(os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gb-25_amd/libgb25hip.so"))
P = C.c_void_p
lib.gb25_default_config.argtypes = [C.POINTER(Config), C.c_int32, C.c_int32, C.c_int32]
lib.gb25_create.argtypes = [C.POINTER(Config), C.POINTER(P)]
for f in ("gb25_set_baroclinic_instability", "gb25_first_time_step", "gb25_synchronize"):
    getattr(lib, f).argtypes = [P]
lib.gb25_loop.argtypes = [P, C.c_int32]
cfg = Config()
lib.gb25_default_config(C.byref(cfg), 1440, 720, 48)
cfg.dt = 120.0
h = P()
assert lib.gb25_create(C.byref(cfg), C.byref(h)) == 0
lib.gb25_set_baroclinic_instability(h)
lib.gb25_first_time_step(h)
lib.gb25_synchronize(h)
for c in range(12):           # the first steps of a fresh model, five at a time
    t = time.perf_counter()
    lib.gb25_loop(h, 5)
    lib.gb25_synchronize(h)
    print(f"steps {5 * c + 1:3d}-{5 * c + 5:3d}: {(time.perf_counter() - t) * 200:7.3f} ms/step", flush=True)
for n in (1, 2, 5, 10, 20, 40, 100):
    ts = []
    for rep in range(5):
        t = time.perf_counter()
        lib.gb25_loop(h, n)
        lib.gb25_synchronize(h)
        ts.append(time.perf_counter() - t)
    best = min(ts)
    print(f"n={n:4d}  {best * 1e3:8.3f} ms  {best * 1e3 / n:7.3f} ms/step  {n / best:7.1f} steps/s", flush=True)
