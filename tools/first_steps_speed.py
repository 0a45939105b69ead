#!/usr/bin/env python3
"""Is the slower start of a run the data or the clock?  The bench protocol (first_time_step, 5 steps, 20
timed steps, then 100 more and 20 timed again) with the initial velocity noise scaled by --noise (1 = the bench's 1e-3 m/s).
usage: first_steps_speed.py [--noise 1.0]"""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--noise", type=float, default=1.0)
a = ap.parse_args()
import gb25_amd as gb
import bench
Nx, Ny, Nz = 1440, 720, 48
m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=120.0)
gb.set_baroclinic_instability(m)
amp = 1e-3 * a.noise
m.set(u=(amp * bench.counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32), v=(amp * bench.counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(np.float32))
gb.first_time_step(m)
for _ in range(5):
    gb.time_step(m)
m.backend.synchronize()
def timed(n):
    t0 = time.perf_counter()
    gb.loop(m, n)
    m.backend.synchronize()
    return n / (time.perf_counter() - t0)
first = timed(20)
gb.loop(m, 100)
m.backend.synchronize()
later = timed(20)
print(f"noise x{a.noise}: steps 7-26: {first:.1f} steps/s   steps 127-146: {later:.1f} steps/s", flush=True)
