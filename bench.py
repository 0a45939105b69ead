#!/usr/bin/env python3
"""Headline benchmark: model time-steps/s of the HydrostaticFreeSurfaceModel hot path.

Protocol (SURVEY.md section 8d, GB-25 sharding/sharded_baroclinic_instability_simulation_run.jl:145-164):
build the model, first_time_step!, W untimed warm-up steps, then exactly K time steps bracketed by
barrier + device synchronisation; the metric is K / wall (max over ranks).  A "step" is one time_step! over
the whole grid.  Inputs are synthetic (deterministic baroclinic-instability IC + seeded velocity noise) and are
resident in HBM before the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size Nx Ny Nz] [--weak] [--no-cpu-baseline]
N > 1 is launched by the driver with torch.distributed.run, one rank per GPU: the SAME Nx x Ny x Nz grid (BASELINE.json:
1440x720x48 at 1/2/4/8 GPUs) cut into N x-slabs, i.e. strong scaling; --weak gives every rank its own Nx x Ny x Nz slab of
an (N Nx) x Ny x Nz grid instead (the reference's own scaling protocol, sharding/sharded_..._run.jl:82-88).  Started
WITHOUT a launcher (no WORLD_SIZE in the environment) `--gpus N` starts its own N ranks: a `torch.distributed.run` child
process, before this process has touched the GPU, whose output (rank 0's JSON line) and exit code are relayed.

Time step: 120 s by default (--dt).  The reference's scaling runs step with dt = 1 s (sharding/sharded_..._run.jl:91), its
quarter-degree climate script quotes 240 s; the cost of a step does not depend on dt, but the state does: with this initial
condition (S = -5e-3 z: N = 6e-3 / s, first internal mode 7.8 m/s) and 4.9 km between the cells of the rows at 80 degrees,
AB2 is unstable there from about 200 s on -- at 240 s w grows along the two walls from step 20 and the fields stop being
finite near step 55 (tools/stability_probe.py; 120 s and 150 s: finite and unremarkable through 600 steps).  A state full of
NaNs also steps 2 % FASTER than a finite one (fewer toggling bits, higher clocks), so lines longer than 50 steps at 240 s
flattered the number; `finite` in the output line says which kind of run it was.
"""
import argparse
import json
import os
import sys
import time

# RCCL / device-memory sharing across the ranks of a node needs dmabuf IPC on this driver stack; the variable is
# exported on the build and GPU boxes already, this only covers a bare launch.  Must precede the first HIP call.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The step runs on three streams (main, side, comm) beside RCCL's own; with the runtime's default of four hardware queues
# per process two of them end up sharing one (rocprofv3 timeline of tools/slab_selfring.py: the 10 us pack kernel of the
# 3-D bundle queued behind the 60 us pressure kernel).  Eight queues: 0.77 -> 0.73 ms per step of a 180-column rank; no
# measurable difference on the single GPU.  (One model per process, as here.  P slabs stepped in ONE process -- the
# LocalSlabEnsemble of the tests, 2 P + 2 streams -- run up to 1.6x slower with eight queues than with four: the Python
# package itself leaves the runtime's default alone.)  Read when the runtime starts; an exported value wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic fp32 field accesses per interior cell per launch, x 4 bytes
ALGORITHMIC_BYTES_PER_CELL = {
    "gu": 5 * 4, "gv": 5 * 4, "momentum": 10 * 4, "tracers": 10 * 4, "compute_w": 3 * 4, "compute_p": 3 * 4,
    "ab2_velocities": 12 * 4, "ab2_tracers": 8 * 4, "corrector": 6 * 4,
}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
KERNEL_SYMBOL = {"gu": "k_gu", "gv": "k_gv", "tracers": "k_tracer_tendencies", "compute_w": "k_compute_w",
                 "compute_p": "k_compute_p", "ab2_velocities": "k_ab2_velocities", "ab2_tracers": "k_ab2_tracers4",
                 "corrector": "k_corrector", "momentum": "k_momentum_tendencies"}   # prefixes of the HIP kernel names


def measured_traffic(kernel, size, prefer=None):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r*_hbm_traffic_<size>.json:
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, gfx950 corrections applied as the
    microarchitecture guide prescribes).  None when no measurement of this kernel at this size is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_hbm_traffic_{size[0]}x{size[1]}x{size[2]}.json")))
    for f in reversed(files):
        try:
            prefix = KERNEL_SYMBOL.get(kernel, kernel)
            hits = [(name, k) for name, k in json.load(open(f))["kernels"].items() if name.startswith(prefix)]
            if prefer:       # several template instances of one kernel in the trace: the one the timed loop runs
                hits = [h for h in hits if h[0].endswith(tuple(prefer))] or hits
            if hits:
                return hits[0][1]["hbm_bytes"], os.path.basename(f)
        except Exception:
            pass
    return None, None


def measured_valu_instructions(kernel, size, prefer=None):
    """Wave-level VALU instructions per launch of `kernel` (SQ_INSTS_VALU of the committed `rocprofv3 --pmc` pass,
    profiles/r*_pmc_sq.csv; only files of this grid size are named so) and the shader clock the chip held under that kernel
    in that pass (GRBM_GUI_ACTIVE, summed over the 8 XCDs, / 8 / the kernel's duration).  None when no such measurement is
    committed."""
    import csv
    import glob
    if tuple(size) != (1440, 720, 48):
        return None, None, None
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.csv")))):
        try:
            prefix = KERNEL_SYMBOL.get(kernel, kernel)
            rows = [r for r in csv.DictReader(open(f)) if r["Counter"] == "SQ_INSTS_VALU" and r["Kernel"].startswith(prefix)]
            if prefer:
                rows = [r for r in rows if r["Kernel"].endswith(tuple(prefer))] or rows
            if rows:
                best = max(rows, key=lambda r: int(float(r["Launches"])))
                clock = None
                for r in csv.DictReader(open(f)):
                    if r["Counter"] == "GRBM_GUI_ACTIVE" and r["Kernel"] == best["Kernel"] and float(r["MeanDurationNs"]) > 0:
                        clock = float(r["MeanValue"]) / 8.0 / float(r["MeanDurationNs"]) * 1e9
                return float(best["MeanValue"]), os.path.basename(f), clock
        except Exception:
            pass
    return None, None, None


def counter_rng(shape, seed, salt):
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64(salt) * np.uint64(0xD1B54A32D192ED03))
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return ((x >> np.uint64(11)).astype(np.float64) / float(1 << 53)).reshape(shape, order="F")


GRID_NAMES = {"simple_lat_lon": "LatitudeLongitudeGrid", "gaussian_islands_lat_lon": "LatitudeLongitudeGrid + GridFittedBottom(gaussian_islands)",
              "lat_lon_as_curvilinear": "LatitudeLongitudeGrid (curvilinear kernels)", "tripolar": "TripolarGrid",
              "gaussian_islands": "TripolarGrid + GridFittedBottom(gaussian_islands)"}


def cpu_baseline(Nx, Ny, Nz, dt, budget_s=25.0, max_steps=50, grid_type="simple_lat_lon"):
    """The oracle (fp32 build, OpenMP) timed on this host on a bounded sample of the same workload.
    kind = "port": the reference's Julia CPU path cannot run here (no Julia)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gb25_amd as gb
    from oracle_backend import CPU
    try:
        cores = len(os.sched_getaffinity(0))     # the cores this job may actually use, not the host's count
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("GB25_CPU_BASELINE_CORES", "16")))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    m = gb.baroclinic_instability_model(CPU("f32"), Nx, Ny, Nz, dt=dt, grid_type=grid_type)
    gb.set_baroclinic_instability(m)
    m.set(u=(1e-3 * counter_rng(m.velocities.u.shape, 42, 1)).astype(np.float32),
          v=(1e-3 * counter_rng(m.velocities.v.shape, 42, 2)).astype(np.float32))
    gb.first_time_step(m)
    n, t0 = 0, time.perf_counter()
    while True:
        gb.time_step(m)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= max_steps or el / n * (n + 1) > 1.6 * budget_s:
            break
    m.backend.close()
    return {"value": n / el, "unit": "steps/s", "cores": int(os.environ["OMP_NUM_THREADS"]), "kind": "port",
            "sample": f"{n} time steps of the same {Nx}x{Ny}x{Nz} workload after first_time_step (fp32 C oracle, OpenMP)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 100: a timed region the driver's GPU samples can see)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 5)")
    ap.add_argument("--size", type=int, nargs=3, default=[1440, 720, 48], metavar=("Nx", "Ny", "Nz"))
    ap.add_argument("--dt", type=float, default=None,
                    help="time step in seconds (default: 120 on the lat-lon grids; 60 on the tripolar ones, whose cells next to the poles "
                         "leave the finite range before step 200 at 120 s -- the step's cost does not depend on it)")
    ap.add_argument("--mesh", default=None, metavar="RxxRy",
                    help="N > 1: a 2-D decomposition, Partition(Rx, Ry, 1) with Rx Ry = N (e.g. 4x2; default: N x slabs)")
    ap.add_argument("--weak", action="store_true", help="N > 1: every rank a full --size slab of an N times wider grid")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="library option for A/B runs (gb25_set_option), e.g. --opt subcycle_lookahead=0")
    ap.add_argument("--grid-type", default="simple_lat_lon",
                    help="single GPU: gaussian_islands_lat_lon | tripolar | gaussian_islands (the reference's TripolarGrid + "
                         "GridFittedBottom); the headline line is the default, simple_lat_lon")
    ap.add_argument("--closure", default=None, metavar="NU,KAPPA|catke",
                    help="VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), kappa, nu), e.g. 1e-4,1e-5 "
                         "(src/baroclinic_instability_model.jl:31), or catke = CATKEVerticalDiffusivity() (:30); "
                         "the headline line is closure = nothing")
    ap.add_argument("--data-free", action="store_true",
                    help="BASELINE.json configs[3]: the data-free climate model -- TripolarGrid with the Gaussian islands, "
                         "CATKE, analytic atmosphere + similarity-theory fluxes every step, dt = 30 s (use --size 1440 720 60); "
                         "x slabs where the reference decomposes in 2-D.  Not the headline line")
    ap.add_argument("--burn", type=int, default=0,
                    help="single GPU: run this many steps of a throw-away model first (GPU clocks at load before the timed model starts)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timers")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 100
    if args.warmup is None:
        args.warmup = 5
    if args.data_free:
        args.grid_type, args.closure, args.dt = "gaussian_islands", "catke", 30.0
    if args.dt is None:
        args.dt = 60.0 if args.grid_type in ("tripolar", "gaussian_islands") else 120.0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # No launcher around us: start the N ranks ourselves.  A CHILD process (never a re-exec: this process may not replace
        # itself once anything GPU-side is loaded, and a child keeps the exit code honest), started before torch or the library is
        # imported here, so this process never initialises the GPU.  Rank 0 of the child prints the JSON line on our stdout.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"bench.py: --gpus {args.gpus} without a launcher: starting {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    import gb25_amd as gb
    from gb25_amd.binding import library_stale

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks "
                         "(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N, or no launcher at all)")
    rehearsal = os.environ.get("GB25_ALL_ON_DEVICE0") == "1"
    if world > 1 and not rehearsal and torch.cuda.device_count() < world:     # (device_count does not initialise the GPU)
        raise SystemExit(f"bench.py --gpus {world}: this host shows {torch.cuda.device_count()} GPU(s).  RCCL needs one device per "
                         "rank; to REHEARSE the multi-process path on fewer devices set GB25_DIST_BACKEND=gloo "
                         "GB25_ALL_ON_DEVICE0=1 (host-callback transport; not a measurement)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if os.environ.get("GB25_ALL_ON_DEVICE0") == "1":   # 1-GPU rehearsal of the multi-process path (tests only)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    Nx, Ny, Nz = args.size

    gNx = Nx * world if (args.weak and world > 1) else Nx          # global grid
    if world > 1:
        import torch.distributed as dist
        from gb25_amd.distributed import SlabModel
        # "nccl" IS RCCL on ROCm: torch.distributed only launches the ranks, shares the communicator's unique id and
        # takes the max of the timings; the halo exchanges are the library's own ncclSend/ncclRecv.
        # GB25_DIST_BACKEND=gloo + GB25_ALL_ON_DEVICE0=1 rehearse the multi-process path with every rank on one GPU
        # (RCCL refuses two ranks per device; the exchanges then go through the host-callback transport): tests only.
        backend = os.environ.get("GB25_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        Rx, Ry = (int(t) for t in args.mesh.lower().split("x")) if args.mesh else (world, 1)
        if Rx * Ry != world:
            raise SystemExit(f"--mesh {args.mesh} is not {world} ranks")
        if gNx % Rx or Ny % Ry:
            raise SystemExit(f"Nx = {gNx}, Ny = {Ny} are not divisible by the {Rx} x {Ry} mesh")
        if args.weak:
            # the zonal spacing shrinks with the rank count, so the time step shrinks with it (constant barotropic
            # Courant number, as the reference's resolution-dependent dt: simulations/ocean_climate_simulation.jl:50-51)
            args.dt = args.dt / world
        gt = {"simple_lat_lon": 0, "gaussian_islands_lat_lon": 1, "lat_lon_as_curvilinear": 2, "tripolar": 3,
              "gaussian_islands": 4}[args.grid_type]
        model = SlabModel(gNx, Ny, Nz, dt=args.dt, rank=rank, nranks=world, device=local_rank, ranks_y=Ry,
                          **(dict(grid_type=gt) if gt else {}))
        if args.closure == "catke":
            model.backend.set_catke(True)
            if args.data_free:
                model.backend.set_catke_parameters(**gb.default_ocean_closure().parameters)
                model.backend.set_bottom_drag(0.003)
                model.backend.set_tracer_advection_order(7)
            model.enable_catke_fields()
        elif args.closure:
            model.backend.set_vertical_diffusivity(*map(float, args.closure.split(",")))
        barrier = dist.barrier
    else:
        closure = (gb.default_ocean_closure() if args.data_free else gb.CATKEVerticalDiffusivity() if args.closure == "catke" else
                   gb.VerticalScalarDiffusivity(*map(float, args.closure.split(","))) if args.closure else None)
        model = gb.baroclinic_instability_model(gb.GPU(local_rank), Nx, Ny, Nz, dt=args.dt, grid_type=args.grid_type,
                                                closure=closure)
        barrier = lambda: None
        Rx, Ry = 1, 1
    locNx, locNy = gNx // Rx, Ny // Ry
    b = model.backend
    for kv in args.opt:
        name, val = kv.split("=")
        b.set_option(name, int(val))

    scratch = None
    if args.burn > 0 and world == 1:
        scratch = gb.baroclinic_instability_model(gb.GPU(local_rank), Nx, Ny, Nz, dt=args.dt)
        gb.set_baroclinic_instability(scratch)
        gb.first_time_step(scratch)

    # synthetic inputs, resident in HBM before timing
    if args.data_free:
        model.grid_type = "gaussian_islands"
        if world == 1:
            model.backend.set_bottom_drag(0.003)   # ocean_simulation's defaults
            model.backend.set_tracer_advection_order(7)
        gb.set_data_free_state(model)        # T = Ti, S = Si and the analytic atmosphere: coupled
    else:
        gb.set_baroclinic_instability(model)
    ush, vsh = model.velocities.u.shape, model.velocities.v.shape
    # (the global noise field, cut into the rank's columns: the same initial state whatever the rank count)
    # (a y-face field has one row more where the northern edge is a wall: on the ranks of the top row of a lat-lon mesh)
    i0, j0 = (rank % Rx) * locNx, (rank // Rx) * locNy
    gvy = Ny + (0 if args.grid_type in ("tripolar", "gaussian_islands") else 1)
    u0 = (1e-3 * counter_rng((gNx, Ny, ush[2]), 42, 1)).astype(np.float32)[i0:i0 + ush[0], j0:j0 + ush[1]]
    v0 = (1e-3 * counter_rng((gNx, gvy, vsh[2]), 42, 2)).astype(np.float32)[i0:i0 + vsh[0], j0:j0 + vsh[1]]
    model.set(u=u0, v=v0)
    del u0, v0
    gb.first_time_step(model)
    # (the library times every fourth launch of a kernel that is timed alone: event records on every launch cost ~1.6 %)
    # Per-kernel times of every kernel are taken during the WARM-UP steps; in the timed region only the dominant kernel
    # carries event records (50 event records per step cost ~3 % of the step, and only that kernel's duration is needed
    # live for the roofline).  Without warm-up steps everything is timed inside the timed region.
    names = ("fill_halos", "compute_w", "compute_p", "gu", "gv", "tracers", "ab2_velocities", "ab2_tracers",
             "barotropic", "corrector", "implicit", "closure", "fluxes")

    def collect():
        out = {}
        for k in names:
            n, ms = b.profile_get(k)
            if n:
                out[k] = {"launches": n, "avg_ms": ms / n, "total_ms": ms}
        return out

    if scratch is not None:      # (--burn: the GPU busy right up to the warm-up steps, the host writes above left it idle)
        gb.loop(scratch, args.burn)
        scratch.backend.synchronize()      # (closed after the timed region: freeing its fields would idle the GPU again)
    warm_kernels, dominant = {}, None
    if not args.no_profile and args.warmup > 0:
        b.profile_enable(True)
        b.profile_reset()
    for _ in range(args.warmup):
        gb.time_step(model)
    b.synchronize()
    if not args.no_profile:
        if args.warmup > 0:
            warm_kernels = collect()
            timed_w = {k: v for k, v in warm_kernels.items() if k in ALGORITHMIC_BYTES_PER_CELL or k == "gu"}
            dominant = max(timed_w, key=lambda k: timed_w[k]["total_ms"]) if timed_w else None
        b.profile_enable(True, only=dominant) if dominant else b.profile_enable(True)
        b.profile_reset()

    barrier(); b.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    gb.loop(model, args.steps)
    b.synchronize(); torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    if scratch is not None:
        scratch.backend.close()

    rank_ms = [1e3 * elapsed / args.steps]
    transport, rccl_ranks = (b.comm_info() if world > 1 else ("none", 0))
    if world > 1:
        mine = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [1e3 * float(t.item()) / args.steps for t in every]
        elapsed = max(float(t.item()) for t in every)          # the contract: the MAX over ranks

    finite = bool(np.isfinite(model.free_surface.eta.interior).all())
    kernels = {}
    if not args.no_profile:
        kernels = collect()                       # timed region: the dominant kernel alone (or all, without warm-up)
        for k, v in warm_kernels.items():         # the others: per-kernel means of the warm-up steps
            kernels.setdefault(k, v)

    if rank == 0:
        cells = locNx * locNy * Nz                 # per GPU (= per launch of a kernel)
        steps_per_s = args.steps / elapsed
        out = {
            "metric": "model time-steps/sec", "value": steps_per_s, "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak" if (args.weak and world > 1) else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"baroclinic_instability_model {gNx}x{Ny}x{Nz} {GRID_NAMES[args.grid_type]}, "
                                   f"halo 8, SplitExplicit(30), {'WENO5 momentum / WENO7 tracers' if args.data_free else 'WENO5'}, TEOS10, dt={args.dt:g}s"
                                   + (f", closure {args.closure}" if args.closure else "")
                                   # (the library's built-in islands / tripolar generator: the reference's mtn1 / mtn2 at ITS grid's
                                   # coordinates come through gb25_set_bottom_height / gb25_set_curvilinear_grid from a Julia host)
                                   + (", STAND-IN grid and bathymetry (built-in generator)" if args.grid_type != "simple_lat_lon" else "")
                                   + (", data-free forcing (similarity-theory fluxes every step)" if args.data_free else ""),
                       "grid": [gNx, Ny, Nz], "cells_per_gpu": cells, "local_columns": locNx, "local_rows": locNy,
                       "parallelism": ((f"{Rx} x {Ry} mesh (Partition(Rx, Ry, 1))" if Ry > 1 else f"x-slab x{world}")
                                       + ", RCCL send/recv inside the library" if world > 1 else "single GPU"),
                       "transport": getattr(model, "transport_kind", None),
                       # the size of the communicator AS RCCL REPORTS IT (ncclCommCount): n_gpus ranks really talked to each
                       # other only if this equals n_gpus (0: no RCCL communicator -- single GPU, or the gloo rehearsal)
                       "rccl_ranks": rccl_ranks,
                       "simulated_years_per_day": steps_per_s * args.dt / 365.0},
            "finite": finite,
            "ms_per_step_by_rank": rank_ms,
            "library_stale": bool(library_stale("Float32")),   # True: the loaded binary is older than its sources (binding.load_library)
        }
        if kernels and "gv" not in kernels and "gu" in kernels:
            kernels["momentum"] = kernels.pop("gu")     # the fused G_u + G_v kernel reports under the "gu" timer
        if kernels:
            timed = {k: v for k, v in kernels.items() if k in ALGORITHMIC_BYTES_PER_CELL}
            dom = max(timed, key=lambda k: timed[k]["total_ms"])
            if dominant:                          # chosen from the warm-up steps, the only one timed live
                dom = "momentum" if (dominant == "gu" and "momentum" in timed) else dominant
            launches_per_step = timed[dom]["launches"] / args.steps
            # Algorithmic bytes of a kernel = the SURVEY 8a rows it implements.  With the AB2 look-ahead the
            # momentum kernel also does a2 + a3(u,v) (4R + 2x(3R+1W) = 48 B/cell) and the tracer kernel a3(T,S)
            # (2x(3R+1W) = 32 B/cell); the stand-alone kernels of those rows then do not run at all.
            alg = dict(ALGORITHMIC_BYTES_PER_CELL)
            if "ab2_velocities" not in kernels:
                alg["momentum"] += 12 * 4
            if "ab2_tracers" not in kernels:
                alg["tracers"] += 8 * 4
            # ... and between the steps of the timed loop! call the corrector's sweep over u, v (a6: 2R + 2W) does not run
            # either: the correction is added where u, v are loaded, and the momentum kernel produces the next u, v from
            # the corrected values (option lazy_corrector; flat lat-lon single domain with both look-aheads on)
            lazy = (world == 1 and args.grid_type == "simple_lat_lon" and not args.closure and args.steps > 1
                    and b.get_option("lazy_corrector") and b.get_option("subcycle_lookahead")
                    and b.get_option("ab2_lookahead") == 1 and b.get_option("fold_fills") and b.get_option("two_streams"))
            if lazy and "ab2_velocities" not in kernels:
                alg["momentum"] += 4 * 4
                if b.get_option("w_on_the_fly"):
                    alg["momentum"] -= 4      # ... and w is not read: carried up the chunk from the divergence (a10: 4R + 1W -> 3R + 1W)
                    alg["tracers"] -= 4
            bytes_per_launch = alg[dom] * cells
            achieved = bytes_per_launch / (timed[dom]["avg_ms"] * 1e-3) / 1e9
            # (the template instance the timed loop runs: ..., LAZY, DRAG, WFLY> = "true, false, true>" with w on the fly)
            inst = (", true, false, true>", ", true, false, false>", ", true, false>", ", true>") if (lazy and dom == "momentum") else None
            traffic, traffic_src = measured_traffic(dom, (locNx, locNy, Nz), prefer=inst)
            # What bounds the kernel: the tendency kernels are VALU-issue bound -- a wave64 fp32 VALU instruction occupies its
            # SIMD for 4 cycles (profiles/r03_valu_rate_noslp.txt), so the ceiling is 1024 SIMDs x clock / 4 instructions/s;
            # valu_frac = committed SQ_INSTS_VALU per launch x 4 cycles / (1024 SIMDs x clock x the live launch time), the clock
            # being what the chip held under this kernel in the counter pass (GRBM_GUI_ACTIVE: ~2.1-2.3 GHz, not the 2.4 GHz ceiling).
            # `achieved` / `frac` stay the contract's algorithmic-bytes numbers; `traffic_frac` is what the kernel really pulls.
            insts, insts_src, clock = measured_valu_instructions(dom, (locNx, locNy, Nz), prefer=inst)
            clock = clock or 2.4e9      # (the clock the chip held under this kernel in the counter pass; 2.4 GHz is its ceiling)
            valu_frac = (insts * 4.0 / (1024 * clock) / (timed[dom]["avg_ms"] * 1e-3)) if insts else None
            # (`bound` is the contract's: the roofline the numbers below are priced against -- HBM; `limited_by`
            # says what the kernel actually runs into on this chip)
            limited_by = "valu issue" if (valu_frac is not None and traffic is not None and
                                          valu_frac > traffic / (timed[dom]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) else "hbm"
            out["roofline"] = {"bound": "hbm", "limited_by": limited_by, "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": traffic_src,
                               "valu_frac": valu_frac, "valu_instructions_per_launch": insts, "valu_source": insts_src,
                               "valu_peak": f"1024 SIMDs x {clock / 1e9:.2f} GHz (GRBM_GUI_ACTIVE of the counter pass) / 4 cycles per wave64 instruction",
                               "valu_clock_GHz": clock / 1e9,
                               # what the kernel really pulls from HBM (PMC bytes / live launch time): the tendency
                               # kernels are VALU-issue bound, the fused rows make `achieved` exceed this
                               "traffic_GBps": (traffic / (timed[dom]["avg_ms"] * 1e-3) / 1e9) if traffic else None,
                               "traffic_frac": (traffic / (timed[dom]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                               if traffic else None,
                               "avg_launch_ms": timed[dom]["avg_ms"], "timed_launches_per_step": launches_per_step,
                               "algorithmic_bytes_per_launch": bytes_per_launch,
                               "algorithmic_bytes_per_cell": alg[dom],
                               "algorithmic_rows": ("a10 + a2 + a3(u,v)" + (" + a6(update of u, v)" if lazy else ""))
                                                   if dom == "momentum" else dom,
                               "whole_step_achieved_GBps": 240.0 * cells * steps_per_s / 1e9,
                               "whole_step_frac": 240.0 * cells * steps_per_s / 1e9 / HBM_PEAK_GBS}
            out["kernels_ms_per_launch"] = {k: v["avg_ms"] for k, v in kernels.items()}
            out["kernel_times_from"] = (f"timed region: {dom}; others: warm-up steps" if warm_kernels
                                        else "timed region")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Nx, Ny, Nz, args.dt, grid_type=args.grid_type)
            # BASELINE.json configs[0] (the reference's own CPU-runnable case: 128x64x8, 100 AB2 steps) beside it
            c1 = cpu_baseline(128, 64, 8, 1200.0, budget_s=10.0, max_steps=99)
            c1["sample"] = c1["sample"].replace("the same ", "BASELINE configs[0] ")
            out["cpu_baseline_config1"] = c1
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
