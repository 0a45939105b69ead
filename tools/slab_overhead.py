"""Cost of the x-slab and 2-D mesh paths relative to the single-domain step, on ONE GPU: the same 1440x720x48 grid stepped (a) as one
domain and (b) as P local slabs of 1440/P columns in lock-step (LocalSlabEnsemble: the library's sequencer, stages,
pack / unpack kernels, two streams and interior/edge split; device-to-device copies instead of RCCL).  One GPU does the
work of all P slabs here, so the ratio shows the EXTRA work and launch overhead of the decomposition (halo columns,
widened barotropic slabs, ragged tiles, ~4x the launches), not the speed-up P GPUs would give.  Run on the GPU box."""
import sys, time
sys.path.insert(0, ".")
import gb25_amd as gb
from gb25_amd.distributed import LocalSlabEnsemble

Nx, Ny, Nz, dt, steps = 1440, 720, 48, 120.0, 20
m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
gb.set_baroclinic_instability(m)
gb.first_time_step(m); gb.loop(m, 5); m.synchronize()
t0 = time.perf_counter(); gb.loop(m, steps); m.synchronize()
t1 = (time.perf_counter() - t0) / steps
print(f"single domain: {1e3 * t1:.3f} ms/step", flush=True)
m.backend.close()
cases = [(2, 1, 1), (4, 1, 1), (8, 1, 1), (8, 1, 0), (4, 2, 0), (2, 4, 0), (2, 2, 0)]   # (Rx, Ry, split_tendencies; no split on a mesh)
for Rx, Ry, split in cases:
    e = LocalSlabEnsemble(Nx, Ny, Nz, Rx * Ry, dt=dt, ranks_y=Ry, options=dict(split_tendencies=split))
    for b in e.backends:
        b.set_baroclinic_instability()
    e.first_time_step(); e.loop(5); e.synchronize()
    t0 = time.perf_counter(); e.loop(steps); e.synchronize()
    tp = (time.perf_counter() - t0) / steps
    print(f"{Rx} x {Ry} local ranks of {Nx // Rx} columns x {Ny // Ry} rows, split_tendencies={split}: {1e3 * tp:.3f} ms/step "
          f"({tp / t1:.3f} x single domain; {1e3 * tp / (Rx * Ry):.3f} ms per rank)", flush=True)
    e.close()
