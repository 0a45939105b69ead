"""Kernel trace target: P local slabs of the 1440x720x48 grid, 10 steps (see tools/slab_overhead.py)."""
import sys, time
sys.path.insert(0, "/root/repo")
from gb25_amd.distributed import LocalSlabEnsemble
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
e = LocalSlabEnsemble(1440, 720, 48, P, dt=120.0)
for b in e.backends:
    b.set_baroclinic_instability()
e.first_time_step(); e.loop(3); e.synchronize()
t0 = time.perf_counter(); e.loop(10); e.synchronize()
print(f"ms per step ({P} slabs):", 1e2 * (time.perf_counter() - t0))
