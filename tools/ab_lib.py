"""A/B two builds of libgb25hip.so on the same GPU in one process each: python tools/ab_lib.py LIB [Nx Ny Nz] [steps].
Raw ctypes (only symbols both builds have), so an older build can be timed beside the current one."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gb25_amd.binding import Config
lib = C.CDLL(os.path.abspath(sys.argv[1]))
Nx, Ny, Nz = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (1440, 720, 48)
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 100
P = C.c_void_p
lib.gb25_default_config.argtypes = [C.POINTER(Config), C.c_int32, C.c_int32, C.c_int32]
lib.gb25_create.argtypes = [C.POINTER(Config), C.POINTER(P)]
for f in ("gb25_set_baroclinic_instability", "gb25_first_time_step", "gb25_synchronize"):
    getattr(lib, f).argtypes = [P]
lib.gb25_loop.argtypes = [P, C.c_int64]
cfg = Config()
lib.gb25_default_config(C.byref(cfg), Nx, Ny, Nz)
cfg.dt = 120.0
h = P()
assert lib.gb25_create(C.byref(cfg), C.byref(h)) == 0
lib.gb25_set_baroclinic_instability(h)
lib.gb25_first_time_step(h)
out = []
for rep in range(4):
    lib.gb25_loop(h, 20)
    lib.gb25_synchronize(h)
    t = time.perf_counter()
    lib.gb25_loop(h, steps)
    lib.gb25_synchronize(h)
    out.append(steps / (time.perf_counter() - t))
print(os.path.basename(sys.argv[1]), " ".join(f"{x:.1f}" for x in out), "steps/s", flush=True)
