"""300 steps of a developed flow on a 2 x 2 mesh (lat-lon; tripolar grid with the islands) and on four lazy x slabs against the
single domain, compared bit for bit every 50 steps (the tests compare after 6): look-ahead chains, adoptions and exchanges
in their steady state.  Run on the GPU box; output kept as profiles/r03_long_run_bitwise.txt."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gb25_amd as gb
from gb25_amd.distributed import LocalSlabEnsemble
from helpers import counter_rng
for gt, name, Rx, Ry in ((0, "simple_lat_lon", 2, 2), (4, "gaussian_islands", 2, 2), (0, "simple_lat_lon", 4, 1)):
    Nx, Ny, Nz, dt = 256, 96, 12, 600.0
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, grid_type=name, options=dict(w_on_the_fly=0))
    gb.set_baroclinic_instability(single)
    vr = Ny if gt >= 3 else Ny + 1
    single.set(u=(1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32), v=(1e-2 * counter_rng((Nx, vr, Nz), 42, 2)).astype(np.float32))
    init = {n: single.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta")}
    ens = LocalSlabEnsemble(Nx, Ny, Nz, Rx * Ry, dt=dt, ranks_y=Ry, options=dict(w_on_the_fly=0), **(dict(grid_type=gt) if gt else {}))
    for n, a in init.items():
        ens.scatter(n, a)
    gb.first_time_step(single); ens.first_time_step()
    for chunk in range(6):
        gb.loop(single, 50); ens.loop(50)
        bad = [n for n in ("u", "v", "w", "T", "S", "eta", "U", "V", "Gn.u", "Gn.T") if not np.array_equal(ens.gather(n), single.backend.get_field(n, False), equal_nan=True)]
        print(name, Rx, Ry, "steps", 1 + 50 * (chunk + 1), "mismatching fields:", bad, "max|u|", float(np.nanmax(np.abs(single.backend.get_field("u", False)))), flush=True)
    ens.close(); single.backend.close()
