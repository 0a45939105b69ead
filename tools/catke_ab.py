#!/usr/bin/env python3
"""One build of the Float32 library stepping a CATKE model (wind + cooling): hashes of the fields after the run (are two builds
bit for bit each other?) and steps/s of three timed loops.  python tools/catke_ab.py LIB [--size Nx Ny Nz] [--grid-type G] [--steps N]"""
import argparse, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("lib")
ap.add_argument("--size", type=int, nargs=3, default=[1440, 720, 48])
ap.add_argument("--grid-type", default="simple_lat_lon")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--data-free", action="store_true", help="the data-free climate model (bench.py --data-free): islands, drag, WENO7, coupled")
a = ap.parse_args()
os.environ["GB25_ALLOW_STALE"] = "1"
import gb25_amd.binding as _b
_b.LIB_PATHS["Float32"] = os.path.abspath(a.lib)
import numpy as np
import gb25_amd as gb
Nx, Ny, Nz = a.size
rng = np.random.default_rng(1)
if a.data_free:
    m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=120.0, grid_type="gaussian_islands", closure=gb.default_ocean_closure())
    m.grid_type = "gaussian_islands"
    m.backend.set_bottom_drag(0.003)
    m.backend.set_tracer_advection_order(7)
    gb.set_data_free_state(m)
    m.set(u=(1e-3 * rng.random(m.velocities.u.shape)).astype(np.float32))
else:
    m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=120.0, grid_type=a.grid_type, closure=gb.CATKEVerticalDiffusivity())
    gb.set_baroclinic_instability(m)
    m.set(u=(1e-2 * rng.standard_normal(m.velocities.u.shape)).astype(np.float32), e=(1e-4 * rng.random((Nx, Ny, Nz)) + 1e-6).astype(np.float32))
    gb.set_top_flux(m, u=np.full((Nx, Ny), -1e-4, np.float32), T=(1e-5 + 4e-5 * rng.random((Nx, Ny))).astype(np.float32))
gb.first_time_step(m)
gb.loop(m, 10)
m.backend.synchronize()
rates = []
for _ in range(3):
    t = time.perf_counter()
    gb.loop(m, a.steps)
    m.backend.synchronize()
    rates.append(a.steps / (time.perf_counter() - t))
out = []
for f in ("e", "kappa_u", "kappa_c", "kappa_e", "Le", "Jb", "u", "T", "previous_u", "previous_v"):
    x = np.ascontiguousarray(m.backend.get_field(f, True))
    out.append(f"{f}:{hashlib.sha256(x.tobytes()).hexdigest()[:10]}:{float(np.nanmax(np.abs(x))):.6g}")
print(os.path.basename(a.lib), " ".join(f"{r:.1f}" for r in rates), "steps/s |", " ".join(out), flush=True)
m.backend.close()
