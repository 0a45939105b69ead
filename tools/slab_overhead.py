"""Cost of the staged x-slab path relative to the single-domain step, on ONE GPU: the same 1440x720x48 grid stepped
(a) as one domain and (b) as P local slabs in lock-step (LocalSlabEnsemble: same kernels, pack/unpack and stage cuts
as the multi-process path, device-to-device copies instead of RCCL).  Run on the GPU box."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import gb25_amd as gb
from gb25_amd.distributed import LocalSlabEnsemble

Nx, Ny, Nz, dt, steps = 1440, 720, 48, 240.0, 20
m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
gb.set_baroclinic_instability(m)
gb.first_time_step(m); gb.loop(m, 5); m.synchronize()
t0 = time.perf_counter(); gb.loop(m, steps); m.synchronize()
t1 = (time.perf_counter() - t0) / steps
print(f"single domain: {1e3 * t1:.3f} ms/step")
T0 = m.tracers.T.interior.copy(); S0 = None
del m
for P, W in ((2, 1440), (2, 720), (4, 360)):
    e = LocalSlabEnsemble(W * P, Ny, Nz, P, dt=dt)
    for b in e.backends:
        b.set_baroclinic_instability()
    e.first_time_step(); e.loop(5); torch.cuda.synchronize()
    for b in e.backends: b.synchronize()
    t0 = time.perf_counter(); e.loop(steps)
    for b in e.backends: b.synchronize()
    torch.cuda.synchronize()
    tp = (time.perf_counter() - t0) / steps
    print(f"{P} local slabs of {W} columns: {1e3 * tp:.3f} ms/step for {W * P} columns = "
          f"{1e3 * tp * Nx / (W * P):.3f} ms per 1440 columns ({tp * Nx / (W * P) / t1:.3f} x single domain)")
    del e
