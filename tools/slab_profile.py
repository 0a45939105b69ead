"""Kernel trace target: 2 local slabs of 1440 columns, 10 steps (see tools/slab_overhead.py)."""
import sys, time
import torch
sys.path.insert(0, "/root/repo")
from gb25_amd.distributed import LocalSlabEnsemble
e = LocalSlabEnsemble(2880, 720, 48, 2, dt=240.0)
for b in e.backends:
    b.set_baroclinic_instability()
e.first_time_step(); e.loop(3); torch.cuda.synchronize()
t0 = time.perf_counter(); e.loop(10); torch.cuda.synchronize()
print("ms per step (2 slabs):", 1e2 * (time.perf_counter() - t0))
