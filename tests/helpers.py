"""Shared helpers for the test-suite: deterministic initial conditions and model factories."""
import numpy as np

import gb25_amd as gb
from oracle_backend import CPU


def counter_rng(shape, seed, salt):
    """Build-owned, platform-independent U(0,1) numbers (SplitMix64 on the linear index):
    the Julia RNG stream of the reference (Random.seed!(42)) cannot be reproduced outside Julia."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64(salt) * np.uint64(0xD1B54A32D192ED03))
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    return u.reshape(shape, order="F")


def set_noisy_velocities(model, amplitude=1e-3, seed=42):
    """u, v = 1e-3 * rand (correctness/correctness_baroclinic_instability_simulation_run.jl:40-42)."""
    ui = amplitude * counter_rng(model.velocities.u.shape, seed, 1)
    vi = amplitude * counter_rng(model.velocities.v.shape, seed, 2)
    dt = model.backend.dtype
    model.set(u=ui.astype(dt), v=vi.astype(dt))
    return ui, vi


def make_pair(Nx, Ny, Nz, dt, precision="f64", float_type="Float32", **kw):
    """(HIP model, oracle model) with the same configuration, like rmodel / vmodel of the reference."""
    rmodel = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=dt, **kw)
    vmodel = gb.baroclinic_instability_model(CPU(precision), Nx, Ny, Nz, dt=dt, **kw)
    return rmodel, vmodel


def make_oracle(Nx, Ny, Nz, dt, precision="f64", **kw):
    return gb.baroclinic_instability_model(CPU(precision), Nx, Ny, Nz, dt=dt, **kw)


SQRT_EPS64 = float(np.sqrt(np.finfo(np.float64).eps))   # 1.4901e-8: the reference's rtol for Float64
SQRT_EPS32 = float(np.sqrt(np.finfo(np.float32).eps))   # 3.4527e-4: the reference's rtol for Float32
# One tolerance for every compared field (state AND tendencies), as in the reference.  It is attainable in fp32
# because the hydrostatic pressure (equation of state + vertical integral) is evaluated in fp64 inside the GPU
# kernel; with an fp32 equation of state G.u, G.S and w sit ~1e-3 away from an fp64 run.
TENDENCY_RTOL = SQRT_EPS32
STATE_FIELDS = ("u", "v", "w", "eta", "T", "S", "filtered.U", "filtered.V", "filtered.eta")


def assert_states_close(m1, m2, *, state_rtol=SQRT_EPS32, tendency_rtol=TENDENCY_RTOL, include_halos=True, label=""):
    """compare_states (norm-wise, atol = 0, halos included) at rtol = sqrt(eps(Float32)) for every field of the
    reference's compared set (src/correctness.jl:28-90)."""
    _, report = gb.compare_states(m1, m2, rtol=state_rtol, include_halos=include_halos, verbose=False)
    bad = []
    for r in report:
        tol = state_rtol if r["name"] in STATE_FIELDS else tendency_rtol
        if not (r["rel"] <= tol):
            bad.append((r["name"], r["rel"], tol, r["maxdelta"], r["index"]))
    assert not bad, f"{label}: fields out of tolerance: {bad}"
    return report
