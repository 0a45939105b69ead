"""The multi-process path (one SlabModel per rank, the library's sequencer in every rank) rehearsed on ONE GPU: two
ranks launched by torch.distributed.run, both on device 0.  RCCL refuses two ranks per device, so the exchanges go
through the library's host-callback transport and torch.distributed's gloo ring here; on the 8-GPU node the same ranks
use the library's RCCL communicator.  Result must equal the single-domain run bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import gb25_amd as gb
from helpers import counter_rng

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(args, extra_env, nproc=2):
    env = dict(os.environ, GB25_DIST_BACKEND="gloo", GB25_ALL_ON_DEVICE0="1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", "29531"] + args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=150)


def test_two_rank_slab_run_matches_single_domain(tmp_path):
    Nx, Ny, Nz, nsteps = 128, 48, 8, 5
    res = _launch([os.path.join(ROOT, "tests", "mp_slab_worker.py"), str(tmp_path), str(Nx), str(Ny), str(Nz),
                   str(nsteps)], {})
    assert res.returncode == 0, res.stderr[-3000:]
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=600.0)
    gb.set_baroclinic_instability(single)
    single.set(u=(1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32),
               v=(1e-2 * counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(np.float32))
    gb.first_time_step(single)
    gb.loop(single, nsteps - 1)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    for n in parts[0].files:
        got = np.concatenate([p[n] for p in parts], axis=0)
        assert np.array_equal(got, single.backend.get_field(n, False)), n


def test_two_rank_run_on_the_folded_grid_matches_single_domain(tmp_path):
    """grid_type = :gaussian_islands (tripolar grid + mountains) on two ranks: each is the other's fold partner; the partner
    exchanges (rows next to the fold line once per step, five rows per substep) go through the same host callback."""
    Nx, Ny, Nz, nsteps = 96, 40, 8, 4
    res = _launch([os.path.join(ROOT, "tests", "mp_slab_worker.py"), str(tmp_path), str(Nx), str(Ny), str(Nz),
                   str(nsteps), "4"], {})
    assert res.returncode == 0, res.stderr[-3000:]
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=600.0, grid_type="gaussian_islands")
    gb.set_baroclinic_instability(single)
    single.set(u=(1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32),
               v=(1e-2 * counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(np.float32))
    gb.first_time_step(single)
    gb.loop(single, nsteps - 1)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    for n in parts[0].files:
        got = np.concatenate([p[n] for p in parts], axis=0)
        assert np.array_equal(got, single.backend.get_field(n, False)), n


def test_bench_two_ranks_prints_contract_line():
    res = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "128", "48",
                   "8"], {})
    assert res.returncode == 0, res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "strong" and out["finite"]
    assert out["config"]["grid"] == [128, 48, 8] and out["config"]["local_columns"] == 64 and out["value"] > 0


def test_four_rank_mesh_on_the_folded_grid_matches_single_domain(tmp_path):
    """Partition(2, 2, 1) of grid_type = :gaussian_islands in four processes (the host-callback transport over gloo): the ring
    within each row, the rows to the southern / northern neighbour (buffer sets 5 - 7), the fold partner within the top row."""
    Nx, Ny, Nz, nsteps = 128, 96, 8, 4
    res = _launch([os.path.join(ROOT, "tests", "mp_slab_worker.py"), str(tmp_path), str(Nx), str(Ny), str(Nz),
                   str(nsteps), "4", "2"], {}, nproc=4)
    assert res.returncode == 0, res.stderr[-3000:]
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=600.0, grid_type="gaussian_islands")
    gb.set_baroclinic_instability(single)
    single.set(u=(1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32),
               v=(1e-2 * counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(np.float32))
    gb.first_time_step(single)
    gb.loop(single, nsteps - 1)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(4)]
    for n in parts[0].files:
        got = np.concatenate([np.concatenate([parts[ry * 2 + rx][n] for rx in range(2)], axis=0) for ry in range(2)], axis=1)
        assert np.array_equal(got, single.backend.get_field(n, False)), n


def test_bench_mesh_prints_contract_line():
    res = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "4", "--mesh", "2x2", "--steps", "3", "--warmup", "1", "--size",
                   "128", "96", "8", "--no-cpu-baseline"], {}, nproc=4)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["finite"] and out["config"]["local_columns"] == 64 and out["config"]["local_rows"] == 48
    assert "2 x 2 mesh" in out["config"]["parallelism"] and out["value"] > 0


def test_bench_data_free_on_a_mesh():
    """BASELINE configs[3]'s workload (data-free climate model) in its reference decomposition, as bench.py launches it -- at a
    reduced size, four ranks 2 x 2 on one GPU: the line `bench.py --gpus 8 --mesh 4x2 --data-free --size 1440 720 60` prints."""
    res = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "4", "--mesh", "2x2", "--data-free", "--steps", "3", "--warmup", "1",
                   "--size", "128", "96", "8", "--no-cpu-baseline"], {}, nproc=4)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["finite"] and "data-free" in out["config"]["workload"] and "2 x 2 mesh" in out["config"]["parallelism"]


def test_bench_starts_its_own_ranks_when_nobody_launched_it():
    """`python bench.py --gpus 2` with no launcher around it (no WORLD_SIZE): it starts a torch.distributed.run child with two
    ranks before touching the GPU itself and relays the rank-0 line and the exit code -- what a driver that runs N > 1 the way
    it runs N = 1 gets.  (Rehearsal environment: both ranks on the one device, gloo + host-callback transport: rccl_ranks = 0.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GB25_DIST_BACKEND="gloo", GB25_ALL_ON_DEVICE0="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--size",
                          "128", "48", "8"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=200)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["finite"] and out["value"] > 0
    assert out["config"]["rccl_ranks"] == 0 and out["config"]["transport"] == "host"
    assert len(out["ms_per_step_by_rank"]) == 2 and abs(max(out["ms_per_step_by_rank"]) - out["ms_per_step"]) < 1e-9
    assert out["library_stale"] is False
    # ... and without the rehearsal environment on a one-GPU box it refuses with a message instead of hanging in RCCL
    env.pop("GB25_DIST_BACKEND"); env.pop("GB25_ALL_ON_DEVICE0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--size", "128", "48", "8"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=200)
    import torch
    if torch.cuda.device_count() < 2:
        assert res.returncode != 0 and "REHEARSE" in res.stderr
