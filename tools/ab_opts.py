#!/usr/bin/env python3
"""A/B option sets of ONE build on the same GPU, alternating: python tools/ab_opts.py [--lib LIB] [--size Nx Ny Nz] [--grid-type G]
[--reps R] "a=1,b=2" "a=0" ...   ("-" = the defaults).  Each set: a fresh model, first_time_step, 20 warm-up steps, then
R timed loops of 100 steps; prints steps/s per loop."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--size", type=int, nargs=3, default=[1440, 720, 48])
ap.add_argument("--grid-type", default="simple_lat_lon")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("sets", nargs="+")
a = ap.parse_args()
if a.lib:
    os.environ["GB25_LIB"] = "1"
    import gb25_amd.binding as _b
    _b.LIB_PATHS["Float32"] = os.path.abspath(a.lib)
import numpy as np
import gb25_amd as gb
for rnd in range(a.rounds):
    for st in a.sets:
        opts = {} if st == "-" else {k: int(v) for k, v in (kv.split("=") for kv in st.split(","))}
        m = gb.baroclinic_instability_model(gb.GPU(), *a.size, dt=120.0, grid_type=a.grid_type, options=opts)
        gb.set_baroclinic_instability(m)
        rng = np.random.default_rng(1)
        m.set(u=(1e-3 * rng.random(m.velocities.u.shape)).astype(np.float32))
        gb.first_time_step(m)
        gb.loop(m, 20)
        m.backend.synchronize()
        out = []
        for _ in range(a.reps):
            t = time.perf_counter()
            gb.loop(m, 100)
            m.backend.synchronize()
            out.append(100 / (time.perf_counter() - t))
        print(f"{st:40s}", " ".join(f"{x:7.1f}" for x in out), flush=True)
        m.backend.close()
