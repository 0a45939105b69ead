#!/usr/bin/env python3
"""How the bench workload evolves: max |u|, |v|, |w|, |eta| and where, every `every` steps, until a field stops being finite.
usage: stability_probe.py [--size 1440 720 48] [--dt 120] [--steps 300] [--every 10] [--noise 1e-2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, nargs=3, default=[1440, 720, 48])
ap.add_argument("--dt", type=float, default=120.0)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--every", type=int, default=10)
ap.add_argument("--noise", type=float, default=None, help="replace the bench's velocity noise amplitude")
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
import gb25_amd as gb
import bench
Nx, Ny, Nz = a.size
m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=a.dt)
for kv in a.opt:
    k, v = kv.split("=")
    m.backend.set_option(k, int(v))
gb.set_baroclinic_instability(m)
amp = 1e-3 if a.noise is None else a.noise
m.set(u=(amp * bench.counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32), v=(amp * bench.counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(np.float32))
gb.first_time_step(m)
done = 1
while done < a.steps:
    gb.loop(m, a.every)
    done += a.every
    line = [f"step {done:4d}"]
    bad = False
    for n in ("u", "v", "w", "eta", "T"):
        x = m.backend.get_field(n, False)
        fin = np.isfinite(x)
        if not fin.all():
            idx = np.argwhere(~fin)
            line.append(f"{n}: {len(idx)} non-finite, first at {idx[0].tolist()}")
            bad = True
        else:
            k = np.unravel_index(np.argmax(np.abs(x)), x.shape)
            line.append(f"{n} max {np.abs(x).max():.3e} at {tuple(int(q) for q in k)}")
    print("  ".join(line), flush=True)
    if bad:
        break
