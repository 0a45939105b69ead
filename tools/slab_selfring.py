#!/usr/bin/env python3
"""One rank's share of an x-slab decomposition as a proxy on a one-GPU box: a slab of --columns columns whose west / east
neighbour is itself (slab_mode = 1, the RCCL transport: ncclSend / ncclRecv to self on the comm stream).  The staged step,
the pack / unpack kernels, the widened sub-cycle and the exchanges are those of a rank of the 8-GPU run; only the wire is
missing.  Prints the wall time per step; under `rocprofv3 --kernel-trace` the trace is one rank's timeline.
--mesh Rx Ry --rank r: rank r of a 2-D decomposition instead (GB25_REHEARSE_ALONE=1: its southern / northern neighbour is itself
too; --size is then the GLOBAL row count): a timing proxy -- the data that cross the seams are not a simulation's.
usage: slab_selfring.py [--columns 180] [--steps 200] [--mesh 4 2 --rank 5]"""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # as bench.py: one rank per process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--columns", type=int, default=180)
ap.add_argument("--size", type=int, nargs=2, default=[720, 48], metavar=("Ny", "Nz"))
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--dt", type=float, default=120.0)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--lib", default=None, help="another build of libgb25hip.so (A/B on the same box)")
ap.add_argument("--grid-type", type=int, default=0, help="gb25_grid_type: 0 lat-lon, 1 lat-lon + islands, 3 tripolar, 4 tripolar + islands (the rank is then its own fold partner too)")
ap.add_argument("--mesh", type=int, nargs=2, default=None, metavar=("Rx", "Ry"))
ap.add_argument("--rank", type=int, default=0)
a = ap.parse_args()
if a.mesh:
    os.environ["GB25_REHEARSE_ALONE"] = "1"
import numpy as np
if a.lib:
    os.environ["GB25_LIB"] = "1"           # (no rebuild check: the file is what it is)
    import gb25_amd.binding as _bind
    _bind.LIB_PATHS["Float32"] = os.path.abspath(a.lib)
import gb25_amd as gb
from gb25_amd.distributed import SlabModel
# (options go in at creation: some are read when the exchange context is built)
Rx, Ry = a.mesh or (1, 1)
m = SlabModel(a.columns * Rx, a.size[0], a.size[1], dt=a.dt, rank=a.rank, nranks=Rx * Ry, ranks_y=Ry, slab_mode=1, transport="rccl",
              options={kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.opt},
              **(dict(grid_type=a.grid_type) if a.grid_type else {}))
gb.set_baroclinic_instability(m)
gb.first_time_step(m)
gb.loop(m, 20)
m.backend.synchronize()
t0 = time.perf_counter()
gb.loop(m, a.steps)
m.backend.synchronize()
t = (time.perf_counter() - t0) / a.steps
print(f"{a.columns} columns x {m.backend.Ny_local} rows x {a.size[1]} levels" + (f" (rank {a.rank} of {Rx} x {Ry})" if a.mesh else "") + f" grid_type {a.grid_type}: {1e3 * t:.3f} ms per step ({1 / t:.0f} steps/s per rank)", flush=True)
