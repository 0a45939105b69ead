"""Two builds of libgb25hip.so in ONE process on the same GPU, a fresh model per timed loop, alternating A, B, B, A, ...: the boxes of the pool (and
one box from one process to the next) differ by more than most tuning steps gain.  python tools/ab_pair.py LIB_A LIB_B [Nx Ny Nz] [steps] [rounds] [grid_type] [catke]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gb25_amd.binding import Config
paths = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
Nx, Ny, Nz = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (1440, 720, 48)
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 60
rounds = int(sys.argv[7]) if len(sys.argv) > 7 else 8
grid_type = int(sys.argv[8]) if len(sys.argv) > 8 else 0   # gb25_grid_type: 1 lat-lon + islands, 4 tripolar + islands
catke = len(sys.argv) > 9 and sys.argv[9] == "catke"
P = C.c_void_p
libs = []
for p in paths:
    lib = C.CDLL(p)
    lib.gb25_default_config.argtypes = [C.POINTER(Config), C.c_int32, C.c_int32, C.c_int32]
    lib.gb25_create.argtypes = [C.POINTER(Config), C.POINTER(P)]
    for f in ("gb25_set_baroclinic_instability", "gb25_first_time_step", "gb25_synchronize", "gb25_destroy"):
        getattr(lib, f).argtypes = [P]
    lib.gb25_loop.argtypes = [P, C.c_int64]
    libs.append(lib)


def timed(lib):
    """a model of its own, alone on the card (a second resident model steps 4 % slower: where its arrays land), stepped and destroyed"""
    cfg = Config()
    lib.gb25_default_config(C.byref(cfg), Nx, Ny, Nz)
    cfg.dt = 60.0 if grid_type >= 3 else 120.0   # (the tripolar grids leave the finite range before step 200 at 120 s)
    cfg.grid_type = grid_type
    h = P()
    assert lib.gb25_create(C.byref(cfg), C.byref(h)) == 0
    if catke:
        lib.gb25_set_closure_catke.argtypes = [P, C.c_int32]
        assert lib.gb25_set_closure_catke(h, 1) == 0
    lib.gb25_set_baroclinic_instability(h)
    lib.gb25_first_time_step(h)
    lib.gb25_loop(h, 20)
    lib.gb25_synchronize(h)
    best = 0.0
    for _ in range(2):
        t = time.perf_counter()
        lib.gb25_loop(h, steps)
        lib.gb25_synchronize(h)
        best = max(best, steps / (time.perf_counter() - t))
    lib.gb25_destroy(h)
    return best


out = [[], []]
for r in range(rounds):
    for q in ((0, 1) if r % 2 == 0 else (1, 0)):
        out[q].append(timed(libs[q]))
for q in (0, 1):
    v = out[q][1:]   # (the first round warms the clocks)
    print(os.path.basename(paths[q]).ljust(28), " ".join(f"{x:.1f}" for x in out[q]), "| mean %.1f" % (sum(v) / len(v)), flush=True)
a, b = sum(out[0][1:]) / (rounds - 1), sum(out[1][1:]) / (rounds - 1)
print("B / A = %.4f" % (b / a))
