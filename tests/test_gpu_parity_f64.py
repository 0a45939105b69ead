"""Float64 parity: libgb25hip_f64.so (the same kernels compiled with real = double) against the fp64 oracle.

Float64 is the default --float-type of the reference's scripts (src/arg_parsing.jl:12-16) and the one its
correctness run uses unless told otherwise; compare_states then asks for rtol = sqrt(eps(Float64)) = 1.49e-8.
At that tolerance two independent implementations agree only if every stencil, metric, weight and order-reduction
rule is the same: this is the sharp test of the kernels' LOGIC, while the Float32 tests (rtol 3.45e-4) mostly bound
rounding.  Differences that remain are summation order and the algebraically equal forms of the WENO weights.
"""
import numpy as np
import pytest

import gb25_amd as gb
from helpers import SQRT_EPS64, assert_states_close, counter_rng, make_pair, set_noisy_velocities

pytestmark = pytest.mark.gpu

ALL_FIELDS = ["u", "v", "w", "T", "S", "pHY", "Gn.u", "Gn.v", "Gn.T", "Gn.S", "Gm.u", "Gm.v", "Gm.T", "Gm.S",
              "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.U", "Gn.V"]


def pair64(*a, **kw):
    return make_pair(*a, precision="f64", float_type="Float64", **kw)


def sync_all(r, v):
    for n in ALL_FIELDS:
        r.backend.set_field(n, v.backend.get_field(n, True), True)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = max(np.linalg.norm(a.ravel()), np.linalg.norm(b.ravel()))
    return 0.0 if n == 0 else float(np.linalg.norm((a - b).ravel()) / n)


def baroclinic_state(r, v, amplitude=1e-3):
    gb.set_baroclinic_instability(v)
    set_noisy_velocities(v, amplitude)
    sync_all(r, v)


def test_library_is_float64():
    r, v = pair64(24, 16, 6, dt=60.0)
    assert r.backend.dtype == np.float64 and r.backend.lib.gb25_real_bytes() == 8
    a = counter_rng(r.backend.field_dims("T", True), 3, 4)          # full 53-bit mantissas survive the round trip
    r.backend.set_field("T", a, True)
    assert np.array_equal(r.backend.get_field("T", True), a)
    for name in ("dxc", "azf", "fcor"):
        for j in (1, 8, 16):
            assert r.backend.metric(name, j) == v.backend.metric(name, j), (name, j)


def test_phase_by_phase_float64():
    """Every phase of src/precompile.jl:31-42 from identical inputs, at round-off level."""
    r, v = pair64(48, 32, 8, dt=600.0)
    baroclinic_state(r, v, amplitude=1e-2)
    get = lambda m, n: m.backend.get_field(n, True)
    for m in (r, v):
        m.backend.initialize()
    for n in ("U", "V"):
        assert rel(get(r, n), get(v, n)) < 1e-15, n
    sync_all(r, v)
    for m in (r, v):
        m.backend.fill_halo_regions()
    for n in ("u", "v", "T", "S", "eta", "U", "V"):
        assert np.array_equal(get(r, n), get(v, n)), n
    sync_all(r, v)
    for m in (r, v):
        m.backend.compute_auxiliaries()
    H = 8
    core = (slice(H - 1, -(H - 1)), slice(H - 1, -(H - 1)), slice(H, -H))
    assert rel(get(r, "w")[core], get(v, "w")[core]) < 1e-13
    assert rel(get(r, "pHY")[core], get(v, "pHY")[core]) < 1e-13
    sync_all(r, v)
    for m in (r, v):
        m.backend.compute_tendencies()
    for n in ("Gn.T", "Gn.S", "Gn.u", "Gn.v"):
        assert rel(get(r, n), get(v, n)) < 1e-9, (n, rel(get(r, n), get(v, n)))   # divergence forms cancel ~3 digits
    for euler in (True, False):
        sync_all(r, v)
        for m in (r, v):
            m.backend.ab2_step(600.0, euler)
        for n in ("u", "v", "T", "S", "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.U", "Gn.V"):
            assert rel(get(r, n), get(v, n)) < 1e-12, (n, euler, rel(get(r, n), get(v, n)))
    sync_all(r, v)
    for m in (r, v):
        m.backend.fill_halo_regions()
        m.backend.correct_velocities_and_cache_previous_tendencies(600.0)
    for n in ("u", "v", "U_bar", "V_bar"):
        assert rel(get(r, n), get(v, n)) < 1e-14, n


def test_reference_correctness_protocol_float64():
    """The six checkpoints of correctness/correctness_baroclinic_instability_simulation_run.jl:46-102 with the
    script's default float type: rtol = sqrt(eps(Float64)), atol = 0, include_halos = true, throw_error = true."""
    Nx = Ny = 128 - 16
    r, v = pair64(Nx, Ny, 16, dt=1e-9)
    set_noisy_velocities(v)
    gb.sync_states(r, v)
    kw = dict(atol=0.0, include_halos=True, throw_error=True, verbose=False)   # rtol defaults to sqrt(eps(Float64))
    gb.compare_states(r, v, **kw)
    for m in (r, v):
        gb.initialize(m)
        gb.update_state(m)
    gb.compare_states(r, v, **kw)
    gb.sync_states(r, v)
    for m in (r, v):
        gb.first_time_step(m)
    gb.compare_states(r, v, **kw)
    for m in (r, v):
        for _ in range(12):
            gb.time_step(m)
    gb.compare_states(r, v, **kw)
    gb.sync_states(r, v)
    gb.update_state(r)
    gb.compare_states(r, v, **kw)
    for m in (r, v):
        gb.loop(m, 100)
    ok, report = gb.compare_states(r, v, **kw)
    assert ok and r.clock.iteration == v.clock.iteration == 113


def test_config1_baroclinic_run_float64():
    """128x64x8, dt = 1200 s, 100 steps of a developing flow: the two implementations stay within sqrt(eps(Float64))."""
    r, v = pair64(128, 64, 8, dt=1200.0)
    baroclinic_state(r, v)
    for m in (r, v):
        gb.first_time_step(m)
    assert_states_close(r, v, state_rtol=SQRT_EPS64, tendency_rtol=SQRT_EPS64, label="f64 first step")
    for m in (r, v):
        gb.loop(m, 99)
    rep = assert_states_close(r, v, state_rtol=SQRT_EPS64, tendency_rtol=SQRT_EPS64, label="f64 after 100 steps")
    print({x["name"]: "%.2e" % x["rel"] for x in rep})
    assert np.abs(r.velocities.u.interior).max() > 0.1


@pytest.mark.parametrize("shape,halo", [((52, 22, 6), 8), ((24, 9, 5), 5), ((70, 13, 7), 8), ((360, 180, 24), 8)])
def test_shapes_float64(shape, halo):
    """Ragged tiles, minimum sizes/halos and BASELINE configs[1]'s shape."""
    Nx, Ny, Nz = shape
    r, v = pair64(Nx, Ny, Nz, dt=300.0, halo=halo)
    baroclinic_state(r, v, amplitude=1e-2)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 4)
    assert_states_close(r, v, state_rtol=SQRT_EPS64, tendency_rtol=SQRT_EPS64, label=f"f64 {shape}")


def test_float32_and_float64_libraries_coexist():
    """Both libraries in one process (the reference builds Float32 and Float64 models side by side in sweeps)."""
    a = gb.baroclinic_instability_model(gb.GPU(float_type="Float32"), 48, 32, 8, dt=600.0)
    b = gb.baroclinic_instability_model(gb.GPU(float_type="Float64"), 48, 32, 8, dt=600.0)
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        gb.first_time_step(m)
        gb.loop(m, 3)
    assert a.backend.dtype == np.float32 and b.backend.dtype == np.float64
    assert rel(a.tracers.T.interior, b.tracers.T.interior) < 1e-5
