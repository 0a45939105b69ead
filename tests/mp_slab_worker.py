"""Worker of tests/test_gpu_multiprocess.py: one rank of an x-slab run (launched by torch.distributed.run)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import gb25_amd as gb                                   # noqa: E402
from gb25_amd.distributed import SlabModel              # noqa: E402
from helpers import counter_rng                         # noqa: E402

if __name__ == "__main__":
    out_dir, Nx, Ny, Nz, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get("GB25_DIST_BACKEND", "gloo"))
    grid_type = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    kw = dict(grid_type=grid_type) if grid_type else {}
    m = SlabModel(Nx, Ny, Nz, dt=600.0, rank=rank, nranks=world, device=0, **kw)   # gloo => the host-callback transport
    nloc = Nx // world
    gb.set_baroclinic_instability(m)
    u0 = (1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32)
    v0 = (1e-2 * counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(np.float32)
    m.set(u=u0[rank * nloc:(rank + 1) * nloc], v=v0[rank * nloc:(rank + 1) * nloc])
    gb.first_time_step(m)
    gb.loop(m, nsteps - 1)
    m.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"),
             **{n: m.backend.get_field(n, False) for n in ("u", "v", "w", "T", "S", "eta", "Gn.u", "Gn.T")})
    dist.barrier()
    dist.destroy_process_group()
