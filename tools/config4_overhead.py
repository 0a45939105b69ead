"""BASELINE configs[3] (data-free climate model, 1440x720x60 tripolar grid with the islands, CATKE, coupled) as one domain, as eight
x slabs and on the reference's 4 x 2 mesh -- all ranks in lock-step on ONE GPU (local transport): the extra work and launches of
the decompositions, not a speed-up.  Run on the GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gb25_amd as gb
from gb25_amd.data_free import ATMOSPHERE_FIELDS
from gb25_amd.distributed import LocalSlabEnsemble

NX, NY, NZ, DT, H, steps = 1440, 720, 60, 30.0, 8, 10
m = gb.data_free_ocean_climate_model_init(gb.GPU(), Nz=NZ, dt=DT, size=(NX, NY))
init = {n: m.backend.get_field(n, False) for n in ("T", "S")}
gb.first_time_step(m); gb.loop(m, 3); m.synchronize()
t0 = time.perf_counter(); gb.loop(m, steps); m.synchronize()
t1 = (time.perf_counter() - t0) / steps
print(f"single domain: {1e3 * t1:.2f} ms/step", flush=True)
m.backend.close()
atm = gb.analytic_atmosphere()
only = os.environ.get("GB25_C4_ONLY")      # e.g. "4x2": that decomposition alone (for a profiler pass)
for Rx, Ry in ((8, 1), (4, 2), (2, 4)):
    if only and only != f"{Rx}x{Ry}":
        continue
    ens = LocalSlabEnsemble(NX, NY, NZ, Rx * Ry, dt=DT, grid_type=4, ranks_y=Ry)
    for b in ens.backends:
        b.set_catke(True)
        b.set_catke_parameters(**gb.default_ocean_closure().parameters)
        b.set_bottom_drag(0.003)
        b.set_tracer_advection_order(7)
        phi = np.asarray(b.metric2("phicc"))[:, : b.Ny_local + 2 * H]
        for n in ATMOSPHERE_FIELDS:
            b.set_prescribed_atmosphere(n, atm.interpolate(n, np.zeros_like(phi), phi))
    for n, a in init.items():
        ens.scatter(n, a)
    ens.first_time_step(); ens.loop(3); ens.synchronize()
    t0 = time.perf_counter(); ens.loop(steps); ens.synchronize()
    tp = (time.perf_counter() - t0) / steps
    print(f"{Rx} x {Ry} local ranks of {NX // Rx} columns x {NY // Ry} rows: {1e3 * tp:.2f} ms/step for all ranks ({tp / t1:.3f} x single "
          f"domain; {1e3 * tp / (Rx * Ry):.3f} ms per rank-step)", flush=True)
    ens.close()
