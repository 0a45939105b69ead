"""Parity of the HIP path (through the C ABI) against the CPU oracle -- the tests proper.

Model of the reference's own test (correctness/correctness_baroclinic_instability_simulation_run.jl):
two models with one configuration, `rmodel` on the accelerator and `vmodel` on the CPU, the same
calls on both, compare_states at checkpoints with rtol = sqrt(eps(Float32)), atol = 0, halos included.
"""
import numpy as np
import pytest

import gb25_amd as gb
from helpers import (SQRT_EPS32, assert_states_close, counter_rng, make_pair, set_noisy_velocities)

pytestmark = pytest.mark.gpu

ALL_FIELDS = ["u", "v", "w", "T", "S", "pHY", "Gn.u", "Gn.v", "Gn.T", "Gn.S", "Gm.u", "Gm.v", "Gm.T", "Gm.S",
              "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.U", "Gn.V"]


def sync_all(rmodel, vmodel, names=ALL_FIELDS):
    """Copy every parent array of the CPU model into the GPU model (rounded to fp32) and back,
    so that both start a phase from bit-identical fp32-representable inputs."""
    for n in names:
        a = vmodel.backend.get_field(n, True).astype(np.float32)
        rmodel.backend.set_field(n, a, True)
        vmodel.backend.set_field(n, a.astype(vmodel.backend.dtype), True)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = max(np.linalg.norm(a.ravel()), np.linalg.norm(b.ravel()))
    return 0.0 if n == 0 else float(np.linalg.norm((a - b).ravel()) / n)


def baroclinic_state(rmodel, vmodel, amplitude=1e-3):
    gb.set_baroclinic_instability(vmodel)
    set_noisy_velocities(vmodel, amplitude)
    sync_all(rmodel, vmodel)


# ------------------------------------------------------------------------------------------------------
def test_field_roundtrip_and_layout():
    r, v = make_pair(24, 16, 6, dt=60.0)
    for name in ("u", "v", "w", "T", "eta", "V"):
        dims = r.backend.field_dims(name, True)
        assert dims == v.backend.field_dims(name, True), name
        a = counter_rng(dims, 7, hash(name) % 97).astype(np.float32)
        r.backend.set_field(name, a, True)
        assert np.array_equal(r.backend.get_field(name, True), a)
        H = 8
        inner = a[H:-H, H:-H, H:-H] if dims[2] > 1 else a[H:-H, H:-H, :]
        assert np.array_equal(r.backend.get_field(name, False), inner)
        b = counter_rng(inner.shape, 9, 3).astype(np.float32)
        r.backend.set_field(name, b, False)
        assert np.array_equal(r.backend.get_field(name, False), b)
        full = r.backend.get_field(name, True)
        mask = np.ones(dims, bool)
        if dims[2] > 1:
            mask[H:-H, H:-H, H:-H] = False
        else:
            mask[H:-H, H:-H, :] = False
        assert np.array_equal(full[mask], a[mask])          # halos untouched by an interior set


def test_grid_metrics_and_substepping_match_oracle():
    r, v = make_pair(128, 64, 8, dt=60.0)
    for name, idxs in (("dxc", range(-5, 72)), ("dxf", range(-5, 72)), ("azc", range(-5, 71)), ("azf", range(-4, 72)),
                       ("fcor", range(1, 66)), ("zc", range(-3, 13)), ("dzc", range(-3, 13)), ("dzf", range(-2, 13))):
        for i in idxs:
            want = np.float32(v.backend.metric(name, i))
            # (the vertical faces are Float32 numbers here and the centres / spacings derive from THOSE: within an ulp of a
            # FACE -- 5e-4 m at 4 km -- of the Float64 oracle's numbers)
            tol = 2 * float(np.spacing(np.float32(6000.0))) if name in ("zc", "dzc", "dzf") else 0.0
            assert abs(r.backend.metric(name, i) - want) <= tol, (name, i)
    nr, fr, wr = r.backend.substepping()
    nv, fv, wv = v.backend.substepping()
    assert nr == nv == 21 and fr == fv
    assert np.array_equal(wr, wv.astype(np.float32).astype(np.float64))


def test_set_baroclinic_instability_kernel():
    r, v = make_pair(48, 32, 8, dt=60.0)
    gb.set_baroclinic_instability(r)
    gb.set_baroclinic_instability(v)
    assert rel(r.tracers.T.interior, v.tracers.T.interior) < 2e-7
    assert rel(r.tracers.S.interior, v.tracers.S.interior) < 2e-7


def test_phase_by_phase_against_oracle():
    """Every phase of src/precompile.jl:31-42, each started from identical inputs."""
    r, v = make_pair(48, 32, 8, dt=600.0)
    baroclinic_state(r, v, amplitude=1e-2)
    get = lambda m, n: m.backend.get_field(n, True)

    # initialize!: barotropic velocities
    for m in (r, v):
        m.backend.initialize()
    for n in ("U", "V"):
        assert rel(get(r, n), get(v, n)) < 5e-7, n

    # halo filling is pure data movement: bit-identical parents
    sync_all(r, v)
    for m in (r, v):
        m.backend.fill_halo_regions()
    for n in ("u", "v", "T", "S", "eta", "U", "V"):
        assert np.array_equal(get(r, n), get(v, n).astype(np.float32)), n

    # auxiliaries on the extended range
    sync_all(r, v)
    for m in (r, v):
        m.backend.compute_auxiliaries()
    H = 8
    # kernels cover -H+2..N+H-1 in x and y; stencils only ever read the first halo ring, and on coarse
    # test grids the deeper latitude halos lie beyond the pole (meaningless metrics), so compare ring 1
    core = (slice(H - 1, -(H - 1)), slice(H - 1, -(H - 1)), slice(H, -H))
    assert rel(get(r, "w")[core], get(v, "w")[core]) < 1e-5
    assert rel(get(r, "pHY")[core], get(v, "pHY")[core]) < 5e-7   # fp64 EOS + integral, stored as fp32
    assert np.isfinite(get(r, "w")[1:-1, 1:-1]).all() and np.isfinite(get(r, "pHY")[1:-1, 1:-1]).all()
    assert np.array_equal(get(r, "w")[:, :, H], np.zeros_like(get(r, "w")[:, :, H]))

    # tendencies
    sync_all(r, v)
    for m in (r, v):
        m.backend.compute_tendencies()
    for n, tol in (("Gn.T", 2e-4), ("Gn.S", 2e-4), ("Gn.u", 2e-4), ("Gn.v", 2e-4)):
        assert rel(get(r, n), get(v, n)) < tol, (n, rel(get(r, n), get(v, n)))

    # ab2_step! incl. the split-explicit sub-cycle (AB2 and Euler variants)
    for euler in (True, False):
        sync_all(r, v)
        for m in (r, v):
            m.backend.ab2_step(600.0, euler)
        for n in ("u", "v", "T", "S", "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.U", "Gn.V"):
            assert rel(get(r, n), get(v, n)) < 2e-5, (n, euler, rel(get(r, n), get(v, n)))

    # corrector + cache
    sync_all(r, v)
    for m in (r, v):
        m.backend.fill_halo_regions()
        m.backend.correct_velocities_and_cache_previous_tendencies(600.0)
    for n in ("u", "v", "U_bar", "V_bar"):
        assert rel(get(r, n), get(v, n)) < 2e-6, n
    for n in ("Gm.u", "Gm.v", "Gm.T", "Gm.S"):
        assert np.array_equal(get(r, n), get(v, n).astype(np.float32)), n


def test_momentum_tendencies_without_pressure_noise():
    """T = S = 0 removes the hydrostatic-pressure round-off: the WENO vector-invariant advection,
    Coriolis and metric arithmetic must then agree with fp64 to ~1e-5."""
    r, v = make_pair(64, 48, 12, dt=60.0)
    set_noisy_velocities(v, 0.1)
    sync_all(r, v)
    for m in (r, v):
        gb.update_state(m)
    for n in ("Gn.u", "Gn.v"):
        a, b = r.backend.get_field(n, False), v.backend.get_field(n, False)
        assert rel(a, b) < 3e-5, (n, rel(a, b))
        # and element-wise, away from round-off of near-cancelling terms
        assert np.max(np.abs(a - b)) < 2e-4 * np.max(np.abs(b)), n
    assert rel(r.velocities.w.interior, v.velocities.w.interior) < 1e-5


def test_smooth_flow_tendencies_elementwise():
    """Smooth large-scale flow (all WENO stencils near their linear weights): element-wise agreement."""
    Nx, Ny, Nz = 64, 48, 8
    r, v = make_pair(Nx, Ny, Nz, dt=60.0)
    lam = (np.arange(Nx) + 0.5) * 2 * np.pi / Nx
    phi = np.linspace(-1, 1, Ny)
    u0 = 0.5 * np.cos(lam)[:, None, None] * np.cos(phi * 1.3)[None, :, None] * np.linspace(0.2, 1, Nz)[None, None, :]
    v0 = 0.3 * np.sin(2 * lam)[:, None, None] * np.sin(np.linspace(0, np.pi, Ny + 1))[None, :, None] * np.ones(Nz)
    T0 = 10 + 5 * np.cos(lam)[:, None, None] * np.cos(phi)[None, :, None] * np.ones(Nz)
    v.set(u=u0, v=v0, T=T0, S=35 + 0 * T0)
    sync_all(r, v)
    for m in (r, v):
        gb.update_state(m)
    for n, tol in (("Gn.T", 5e-5), ("Gn.u", 2e-4), ("Gn.v", 2e-4)):
        a, b = r.backend.get_field(n, False), v.backend.get_field(n, False)
        assert rel(a, b) < tol, (n, rel(a, b))


def test_reference_correctness_protocol():
    """The six checkpoints of correctness/correctness_baroclinic_instability_simulation_run.jl:46-102:
    Nx = Ny = 128-16, Nz = 16, halo 8, dt = 1e-9, u,v = 1e-3 rand, T = S = 0, rtol = sqrt(eps(Float32)),
    atol = 0, include_halos = true, throw_error = true."""
    Nx = Ny = 128 - 16
    r, v = make_pair(Nx, Ny, 16, dt=1e-9)
    set_noisy_velocities(v)
    gb.sync_states(r, v)
    kw = dict(rtol=SQRT_EPS32, atol=0.0, include_halos=True, throw_error=True, verbose=False)
    gb.compare_states(r, v, **kw)                       # at the beginning
    for m in (r, v):
        gb.initialize(m)
        gb.update_state(m)
    gb.compare_states(r, v, **kw)                       # after initialization and update state
    gb.sync_states(r, v)
    for m in (r, v):
        gb.first_time_step(m)
    gb.compare_states(r, v, **kw)                       # after first time step
    for m in (r, v):
        for _ in range(12):                             # 2 warm-up + 10 steps
            gb.time_step(m)
    gb.compare_states(r, v, **kw)                       # after 10 steps
    gb.sync_states(r, v)
    gb.update_state(r)
    gb.compare_states(r, v, **kw)                       # after syncing and updating state again
    for m in (r, v):
        gb.loop(m, 100)
    ok, report = gb.compare_states(r, v, **kw)          # after a loop of 100 steps
    assert ok
    assert r.clock.iteration == v.clock.iteration == 113


def test_config1_baroclinic_run_100_steps():
    """BASELINE.json configs[0]: 128x64x8, fp32, 100 AB2 steps, deterministic IC + velocity noise."""
    r, v = make_pair(128, 64, 8, dt=1200.0)
    baroclinic_state(r, v)
    for m in (r, v):
        gb.first_time_step(m)
    assert_states_close(r, v, label="after first_time_step")
    for m in (r, v):
        gb.loop(m, 99)
    rep = assert_states_close(r, v, label="after 100 steps")
    assert abs(r.clock.time - 100 * 1200.0) < 1e-6 and r.clock.iteration == 100
    assert np.isfinite(r.velocities.u.parent).all()
    # the flow must have developed (this is not a trivial comparison of zeros)
    assert np.abs(r.velocities.u.interior).max() > 0.1 and np.abs(r.free_surface.eta.interior).max() > 0.1


def test_config2_shape_short_run():
    """BASELINE.json configs[1] shape (360x180x24) for a few steps."""
    r, v = make_pair(360, 180, 24, dt=600.0)
    baroclinic_state(r, v)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 4)
    assert_states_close(r, v, label="360x180x24 after 5 steps")


def test_ragged_sizes_not_multiples_of_the_tile():
    """Nx, Ny that are not multiples of the 64x4 tile, minimum Nz for WENO5 order reduction."""
    r, v = make_pair(52, 22, 6, dt=300.0)
    baroclinic_state(r, v, amplitude=1e-2)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 5)
    assert_states_close(r, v, label="52x22x6")


@pytest.mark.parametrize("Nz", [13, 25, 37, 60, 100])
def test_vertical_extents_with_ragged_level_chunks(Nz):
    """The tendency kernels split a column into max(1, Nz // 12) chunks of ceil(Nz / chunks) levels, the last one
    shorter (Nz = 100, the 1/12-degree configuration, gives 8 chunks of 13, 13, ... 9; Nz = 60 gives 5 x 12).  Parity
    with the oracle, and the look-ahead's chunked column sums against the stand-alone kernels bit for bit."""
    r, v = make_pair(70, 22, Nz, dt=300.0)
    baroclinic_state(r, v, amplitude=1e-2)
    plain = gb.baroclinic_instability_model(gb.GPU(), 70, 22, Nz, dt=300.0, options=dict(ab2_lookahead=0))
    for n in ALL_FIELDS:
        plain.backend.set_field(n, r.backend.get_field(n, True), True)
    for m in (r, v, plain):
        gb.first_time_step(m)
        gb.loop(m, 4)
    assert_states_close(r, v, label=f"Nz={Nz}")
    for n in ALL_FIELDS:
        assert np.array_equal(r.backend.get_field(n, True), plain.backend.get_field(n, True)), (Nz, n)


@pytest.mark.parametrize("substeps", [8, 31, 60])
def test_other_substep_counts(substeps):
    """SplitExplicitFreeSurface(substeps=N): the averaging weights, their truncation and the blocked sub-cycle
    (7 substeps per launch, remainder in the last launch) for counts other than GB-25's 30."""
    r, v = make_pair(96, 36, 10, dt=600.0, substeps=substeps)   # (Ny = 40 would put a halo row centre exactly on the pole)
    nr, fr, wr = r.backend.substepping()
    nv, fv, wv = v.backend.substepping()
    assert nr == nv and fr == fv
    baroclinic_state(r, v, amplitude=1e-2)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 5)
    assert_states_close(r, v, label=f"substeps={substeps}")


def test_error_paths():
    from gb25_amd.binding import GB25Error
    with pytest.raises(GB25Error):
        gb.baroclinic_instability_model(gb.GPU(), 4, 4, 2, dt=1.0)          # too small
    with pytest.raises(GB25Error):
        gb.baroclinic_instability_model(gb.GPU(device=99), 32, 16, 8, dt=1.0)
    with pytest.raises(GB25Error, match="decompose in x"):                 # 32-bit element indices: 2^31 elements per array
        gb.baroclinic_instability_model(gb.GPU(), 8640, 2160, 100, dt=1.0)
    m = gb.baroclinic_instability_model(gb.GPU(), 32, 16, 8, dt=1.0)
    with pytest.raises(ValueError):
        m.velocities.u.set(np.zeros((3, 3, 3)))
    with pytest.raises(ValueError):
        gb.baroclinic_instability_model(gb.GPU(), 32, 16, 8, dt=1.0, grid_type="cubed_sphere")
    with pytest.raises(GB25Error, match="even Nx"):                        # the fold maps columns onto columns
        gb.baroclinic_instability_model(gb.GPU(), 33, 16, 8, dt=1.0, grid_type="gaussian_islands")
    with pytest.raises(GB25Error, match="barotropic halo"):                # a slab must hold the widened sub-cycle halo
        gb.baroclinic_instability_model(gb.GPU(), 16, 16, 8, dt=1.0, grid_type="tripolar", slab_mode=1)
    with pytest.raises(GB25Error, match="phase-by-phase"):                 # a slab is driven by the composites only
        s = gb.baroclinic_instability_model(gb.GPU(), 64, 16, 8, dt=1.0, slab_mode=1)
        s.backend.update_state()
    with pytest.raises(GB25Error, match="exchange context"):
        s.backend.time_step()
    with pytest.raises(GB25Error, match="SUBCYCLE_BLOCK"):
        m.backend.set_option("subcycle_block", 4)


def test_lds_kernels_match_direct_stencil_kernels():
    """The flux-sharing / LDS-staged tendency kernels (tendency_kernels.hpp, the default) evaluate the same expressions
    as the direct-stencil kernels (kernels.hpp, option kernels = 1): results agree to the last few bits."""
    m1 = gb.baroclinic_instability_model(gb.GPU(), 150, 70, 20, dt=600.0, options=dict(kernels=1))   # ragged tiles
    m2 = gb.baroclinic_instability_model(gb.GPU(), 150, 70, 20, dt=600.0)
    assert (m1.backend.get_option("kernels"), m2.backend.get_option("kernels")) == (1, 2)
    gb.set_baroclinic_instability(m1)
    set_noisy_velocities(m1, 0.05)
    m1.set(eta=(1e-2 * counter_rng((150, 70, 1), 3, 3)).astype(np.float32))
    gb.sync_states(m2, m1)
    for m in (m1, m2):
        gb.first_time_step(m)
        gb.loop(m, 3)
    for n in ("Gn.u", "Gn.v", "Gn.T", "Gn.S", "u", "v", "T", "S", "eta", "w"):
        a, b = m1.backend.get_field(n, False), m2.backend.get_field(n, False)
        assert np.isfinite(b).all(), n
        assert rel(a, b) < 2e-6, (n, rel(a, b))


def test_ab2_lookahead_is_bitwise_neutral():
    """The tracer tendency kernel writes T, S of the next time level ahead of ab2_step! (Ab2Ahead); the step then
    adopts them by pointer exchange.  Same bits as the stand-alone AXPY kernel (option ab2_lookahead = 0), halos included,
    across everything that must invalidate the look-ahead: a changed dt, host writes into T / G, an Euler restart,
    phase-by-phase driving, and a handed-out device pointer."""
    a = gb.baroclinic_instability_model(gb.GPU(), 150, 70, 12, dt=600.0, options=dict(ab2_lookahead=0))
    # (the sub-cycle look-ahead is off by default on grids this small; w on the fly changes the last bits by design)
    on = dict(ab2_lookahead=1, subcycle_lookahead=1, w_on_the_fly=0)
    b = gb.baroclinic_instability_model(gb.GPU(), 150, 70, 12, dt=600.0, options=on)
    c = gb.baroclinic_instability_model(gb.GPU(), 150, 70, 12, dt=600.0, options=on)   # will hand out its T pointer
    names = ALL_FIELDS

    def same(label):
        for n in names:
            x = a.backend.get_field(n, True)
            assert np.array_equal(x, b.backend.get_field(n, True)), (label, n)
            assert np.array_equal(x, c.backend.get_field(n, True)), (label, n, "exposed pointer")

    for m in (a, b, c):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        # values in the halo layers that no kernel ever rewrites must survive the buffer alternation
        T = m.backend.get_field("T", True)
        T[:, :3, :] = 1.25
        T[:, :, -3:] = -2.5
        m.backend.set_field("T", T, True)
        gb.first_time_step(m)
        gb.loop(m, 3)
    same("plain steps")
    assert c.backend.field_device_ptr("T")
    for m in (a, b, c):
        m.backend.set_dt(450.0)                       # dt differs from the one the look-ahead assumed
        gb.time_step(m)
        gb.time_step(m)
    same("after set_dt")
    S = a.backend.get_field("S", False) + np.float32(0.125)
    g = a.backend.get_field("Gn.T", True) * np.float32(1.5)
    for m in (a, b, c):
        m.backend.set_field("S", S, False)            # host writes into a tracer and into a tendency
        m.backend.set_field("Gn.T", g, True)
        gb.time_step(m)
        gb.time_step(m)
    same("after host writes")
    for m in (a, b, c):
        gb.first_time_step(m)                         # Euler step: chi differs from the look-ahead's
        m.backend.ab2_step(450.0, False)              # phase by phase
        m.backend.fill_halo_regions()
        m.backend.correct_velocities_and_cache_previous_tendencies(450.0)
        m.backend.update_state()
        m.backend.compute_tracer_tendencies()         # a second evaluation must not advance twice
        gb.time_step(m)
    same("after Euler restart and phase-by-phase driving")
    p0 = c.backend.field_device_ptr("T")
    gb.time_step(c)
    assert c.backend.field_device_ptr("T") == p0      # pinned once handed out


def test_lookaheads_are_bitwise_neutral_over_a_longer_run():
    """All look-aheads (tracers, velocities, sub-cycle) against none, 40 steps at BASELINE configs[1]'s size: the
    partner buffers, the pointer exchanges and the work that runs beside the tendency kernels on the side stream must
    not change a bit (a missing stream dependency shows up here as a difference that comes and goes)."""
    a = gb.baroclinic_instability_model(gb.GPU(), 360, 180, 24, dt=600.0, options=dict(ab2_lookahead=0))
    b = gb.baroclinic_instability_model(gb.GPU(), 360, 180, 24, dt=600.0, options=dict(subcycle_lookahead=1, w_on_the_fly=0))
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        gb.first_time_step(m)
    for chunk in (1, 2, 17, 20):
        for m in (a, b):
            gb.loop(m, chunk)
        for n in ALL_FIELDS:
            assert np.array_equal(a.backend.get_field(n, True), b.backend.get_field(n, True)), (chunk, n)
    assert a.clock.iteration == b.clock.iteration == 41
    assert np.abs(b.velocities.u.interior).max() > 0.05


@pytest.mark.parametrize("shape,halo", [((150, 70, 12), 8), ((40, 21, 6), 4)])
def test_fused_halo_fill_is_bitwise_neutral(shape, halo):
    """One launch for the y, z and periodic-x fills (the x copy reads through the other two) against the sequence
    y+z, then x: every cell of every parent array, including halo values the host planted in layers that no fill
    rewrites."""
    Nx, Ny, Nz = shape
    models = []
    for fused in (0, 1):
        m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=300.0, halo=(halo,) * 3,
                                            options=dict(fill_fused=fused))
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        for n, seed in (("T", 5), ("u", 6), ("v", 7), ("eta", 8), ("V", 9)):
            a = m.backend.get_field(n, True)
            a += (1e-3 * counter_rng(a.shape, seed, 1)).astype(np.float32)      # noise in EVERY halo layer
            m.backend.set_field(n, a, True)
        m.backend.fill_halo_regions()
        models.append(m)
    a, b = models
    for n in ALL_FIELDS:
        assert np.array_equal(a.backend.get_field(n, True), b.backend.get_field(n, True)), ("fill", n)
    for m in (a, b):
        gb.first_time_step(m)
        gb.loop(m, 4)
    for n in ALL_FIELDS:
        assert np.array_equal(a.backend.get_field(n, True), b.backend.get_field(n, True)), ("steps", n)


@pytest.mark.parametrize("shape,halo", [((150, 70, 12), 8), ((40, 21, 6), 4), ((1440, 90, 14), 8)])
def test_fills_folded_into_their_producers_are_bitwise_neutral(shape, halo):
    """The corrector (u, v), the tracer look-ahead (T, S) and the last barotropic launch (eta, U, V) write the halo cells
    of what they produce, and the fill launches leave the step (option fold_fills, the default) -- against the explicit
    fills: every cell of every parent array, including halo values the host planted in layers no fill rewrites, through
    a host write and a changed dt (which bring the explicit fills back for two steps)."""
    Nx, Ny, Nz = shape
    models = []
    for fold in (0, 1):
        m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=300.0, halo=(halo,) * 3,
                                            options=dict(fold_fills=fold, subcycle_lookahead=1, w_on_the_fly=0))
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        for n, seed in (("T", 5), ("u", 6), ("v", 7), ("eta", 8), ("V", 9)):
            a = m.backend.get_field(n, True)
            a += (1e-3 * counter_rng(a.shape, seed, 1)).astype(np.float32)      # noise in EVERY halo layer
            m.backend.set_field(n, a, True)
        models.append(m)
    a, b = models

    def same(label):
        for n in ALL_FIELDS:
            assert np.array_equal(a.backend.get_field(n, True), b.backend.get_field(n, True)), (label, n)

    for m in (a, b):
        gb.first_time_step(m)
        gb.loop(m, 7)
    same("8 steps")
    assert b.backend.lookahead_state()[0]
    S = a.backend.get_field("S", False) + np.float32(0.125)
    for m in (a, b):
        m.backend.set_field("S", S, False)
        gb.loop(m, 3)
        m.backend.set_dt(240.0)
        gb.loop(m, 4)
    same("after a host write and a changed dt")


@pytest.mark.parametrize("opts", [dict(two_streams=0), dict(subcycle_block=3), dict(subcycle_block=1),
                                  dict(store_pressure=1), dict(ab2_lookahead=2, subcycle_lookahead=1),
                                  dict(subcycle_lookahead=2), dict(fold_fills=0), dict(tracers_first=0),
                                  dict(tracers_first=0, lazy_corrector=0)])
def test_schedule_options_are_bitwise_neutral(opts):
    """Every schedule switch of gb25_set_option (single stream, 3 substeps per launch, pHY' stored every step, tracer
    look-ahead only) gives the bits of the default schedule, through a changed dt and an option flipped mid-run.  One
    launch per substep (subcycle_block = 1) is a different kernel that divides by the metrics where the blocked one
    multiplies by their reciprocals: equal to round-off."""
    a = gb.baroclinic_instability_model(gb.GPU(), 150, 70, 12, dt=600.0, options=dict(w_on_the_fly=0))
    b = gb.baroclinic_instability_model(gb.GPU(), 150, 70, 12, dt=600.0, options=dict(opts, w_on_the_fly=0))
    for k, v in opts.items():
        assert b.backend.get_option(k) == v
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        gb.first_time_step(m)
        gb.loop(m, 4)
        m.backend.set_dt(450.0)
        gb.loop(m, 3)
    b.backend.set_option("subcycle_lookahead", 0 if b.backend.get_option("subcycle_lookahead") else 1)
    for m in (a, b):
        gb.loop(m, 3)
    for n in ALL_FIELDS:
        x, y = a.backend.get_field(n, True), b.backend.get_field(n, True)
        if opts.get("subcycle_block") == 1:
            assert rel(x, y) < 2e-5, (opts, n, rel(x, y))
        else:
            assert np.array_equal(x, y), (opts, n)
    assert a.clock.iteration == b.clock.iteration == 11


@pytest.mark.parametrize("float_type", ["Float32", "Float64"])
@pytest.mark.parametrize("shape,halo", [((150, 70, 12), 8), ((40, 21, 6), 4), ((1440, 90, 14), 8)])
def test_the_corrector_inside_its_consumers_is_bitwise_neutral(shape, halo, float_type):
    """Between the steps of one loop! call the barotropic correction of u, v is not swept over the fields: a 2-D kernel
    leaves du, dv and the w, momentum and tracer kernels add them as they load (option lazy_corrector, the default where
    both look-aheads run) -- against the sweep: every cell of every parent array when the call returns, loops of
    different lengths (the last step of a call always sweeps), single steps in between, a changed dt."""
    Nx, Ny, Nz = shape
    dtype = np.float32 if float_type == "Float32" else np.float64
    models = []
    for lazy in (0, 1):
        m = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=300.0, halo=(halo,) * 3,
                                            options=dict(lazy_corrector=lazy, subcycle_lookahead=1, w_on_the_fly=0))
        assert m.backend.get_option("lazy_corrector") == lazy
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        for n, seed in (("T", 5), ("u", 6), ("v", 7), ("eta", 8), ("V", 9)):
            a = m.backend.get_field(n, True)
            a += (1e-3 * counter_rng(a.shape, seed, 1)).astype(dtype)          # noise in EVERY halo layer
            m.backend.set_field(n, a, True)
        models.append(m)
    a, b = models

    def same(label):
        for n in ALL_FIELDS:
            x, y = a.backend.get_field(n, True), b.backend.get_field(n, True)
            assert np.array_equal(x, y), (label, n, float(np.abs(x - y).max()))

    for m in (a, b):
        gb.first_time_step(m)
        gb.loop(m, 7)
    same("loop of 7")
    for m in (a, b):
        gb.loop(m, 2)
        gb.time_step(m)
        gb.loop(m, 5)
    same("loops of 2, 1, 5")
    for m in (a, b):
        m.backend.set_dt(240.0)
        gb.loop(m, 6)
    same("after a changed dt")
    assert np.abs(a.velocities.u.interior).max() > 0.01


@pytest.mark.parametrize("shape,halo", [((16, 9, 4), 4), ((24, 9, 5), 5), ((70, 13, 7), 8), ((8, 10, 4), 8)])
def test_minimum_sizes_and_halos(shape, halo):
    """Smallest useful extents (Ny = 8 is excluded: with 20-degree rows the first latitude halo mirrors exactly
    through the pole, Az vanishes there and the never-used wall row of G.v is NaN in the oracle as well).  Every
    wall-adjacent order reduction (WENO5 -> WENO3 -> upwind) overlaps its
    opposite wall's, the x tile is mostly empty, and the halo is narrower than the reference's 8."""
    Nx, Ny, Nz = shape
    r, v = make_pair(Nx, Ny, Nz, dt=200.0, halo=(halo, halo, halo))
    gb.set_baroclinic_instability(v)
    set_noisy_velocities(v, 1e-2)
    for n in ALL_FIELDS:
        a = v.backend.get_field(n, True).astype(np.float32)
        r.backend.set_field(n, a, True)
        v.backend.set_field(n, a.astype(np.float64), True)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 4)
    assert np.isfinite(r.velocities.u.interior).all()
    _, report = gb.compare_states(r, v, rtol=SQRT_EPS32, include_halos=False, verbose=False)
    bad = [(q["name"], q["rel"]) for q in report if not q["rel"] <= SQRT_EPS32]
    assert not bad, bad



@pytest.mark.parametrize("float_type", ["Float32", "Float64"])
@pytest.mark.parametrize("shape", [(150, 70, 24), (1440, 90, 48)])
def test_w_on_the_fly_agrees_to_round_off(shape, float_type):
    """Option w_on_the_fly (the default beside the lazy corrector): between the steps of one loop! call the tendency kernels
    carry w up their chunks of levels themselves -- from the divergence of the transports they hold, starting from 2-D bases
    made of the look-ahead's column integrals -- instead of reading the field a k_compute_w launch left.  Another association
    of the same vertical sum: round-off, not bits.  After 25 steps every field agrees with the stand-alone path to a few
    hundred ulps of its norm, the field w the call leaves behind included (it is recomputed from the final velocities), and
    a constant tracer stays constant."""
    Nx, Ny, Nz = shape
    eps = float(np.finfo(np.float32 if float_type == "Float32" else np.float64).eps)
    models = []
    for fly in (0, 1):
        m = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=300.0,
                                            options=dict(w_on_the_fly=fly, subcycle_lookahead=1))
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        m.set(S=np.full((Nx, Ny, Nz), 35.0, m.backend.dtype))       # a constant tracer
        gb.first_time_step(m)
        gb.loop(m, 24)
        models.append(m)
    a, b = models
    assert b.backend.lookahead_state()[0]
    for n in ALL_FIELDS:
        if n in ("Gn.S", "Gm.S"):     # (round-off noise around zero on one side, exactly zero on the other: below)
            continue
        x, y = a.backend.get_field(n, True), b.backend.get_field(n, True)
        assert rel(x, y) < 2000 * eps, (n, rel(x, y))
    assert rel(a.backend.get_field("w", True), b.backend.get_field("w", True)) < 200 * eps
    for m in (a, b):   # a constant tracer stays constant: continuity and advection see the same transports either way
        assert np.abs(m.backend.get_field("Gn.S", False)).max() < 1e3 * eps * 35.0 * 1e-3
        assert np.abs(m.backend.get_field("S", False) - 35.0).max() < 50 * eps * 35.0


@pytest.mark.parametrize("float_type", ["Float32", "Float64"])
@pytest.mark.parametrize("grid_type,shape", [("gaussian_islands_lat_lon", (150, 70, 24)), ("lat_lon_as_curvilinear", (150, 70, 24)),
                                             ("gaussian_islands", (144, 64, 24)),   # (the bare tripolar grid is singular at its poles)
                                             ("simple_lat_lon", (150, 70, 24))])
def test_w_on_the_fly_beside_the_sweeping_corrector(grid_type, shape, float_type):
    """Round 4: the grids whose kernel instances keep the corrector's sweep -- a GridFittedBottom, the curvilinear grids, the
    zipper fold -- drop the k_compute_w launch all the same: the sweep leaves du, dv as 2-D fields, k_w_bases makes w at the chunk
    boundaries from the look-ahead's chunk integrals (masked thicknesses next to the bottom, image rows beyond the fold), the
    tendency kernels carry w up their chunks.  (The flat grid takes this path with lazy_corrector = 0.)  Same criterion as the
    lazy path's test: every field within a few hundred ulps of its norm after 25 steps, the field w the call leaves behind
    included, and a constant tracer stays constant where nothing is immersed."""
    Nx, Ny, Nz = shape
    eps = float(np.finfo(np.float32 if float_type == "Float32" else np.float64).eps)
    models = []
    for fly in (0, 1):
        m = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=300.0, grid_type=grid_type,
                                            options=dict(w_on_the_fly=fly, subcycle_lookahead=1, lazy_corrector=0))
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        m.set(S=np.full((Nx, Ny, Nz), 35.0, m.backend.dtype))       # a constant tracer
        gb.first_time_step(m)
        m.backend.profile_enable(True)
        m.backend.profile_reset()
        gb.loop(m, 24)
        models.append(m)
    a, b = models
    assert b.backend.lookahead_state()[0]
    # (the stand-alone path launches k_compute_w every step; on the fly: once, when the call returns)
    assert a.backend.profile_get("compute_w")[0] >= 24 and b.backend.profile_get("compute_w")[0] <= 3
    for n in ALL_FIELDS:
        if n in ("Gn.S", "Gm.S"):
            continue
        x, y = a.backend.get_field(n, True), b.backend.get_field(n, True)
        assert np.isfinite(y).all() and rel(x, y) < 4000 * eps, (n, rel(x, y))
    assert rel(a.backend.get_field("w", True), b.backend.get_field("w", True)) < 400 * eps
    if "islands" not in grid_type:
        for m in (a, b):
            assert np.abs(m.backend.get_field("S", False) - 35.0).max() < 50 * eps * 35.0


@pytest.mark.parametrize("float_type,tol", [("Float32", SQRT_EPS32), ("Float64", 1e-9)])
@pytest.mark.parametrize("grid_type", ["simple_lat_lon", "gaussian_islands_lat_lon"])
def test_substep_order_option_matches_the_oracle(grid_type, float_type, tol):
    """Option substep_order = 1 (U, V from the old eta, then eta from the new U, V -- the other order of the two halves of a
    split-explicit substep) in the library and in the oracle: same criterion as the default order.  And the option
    does change the answer (the default is untouched: every other test runs on it)."""
    r, v = make_pair(64, 32, 8, dt=600.0, float_type=float_type, grid_type=grid_type)
    for m in (r, v):
        m.backend.set_option("substep_order", 1)
    gb.set_baroclinic_instability(v)
    set_noisy_velocities(v, 1e-2)
    gb.sync_states(r, v)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 5)
    assert_states_close(r, v, state_rtol=tol, tendency_rtol=tol, label=f"substep_order = 1, {grid_type}")
    d, _ = make_pair(64, 32, 8, dt=600.0, float_type=float_type, grid_type=grid_type)
    gb.sync_states(d, v)      # (any state would do: one step from it in the default order)
    e = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), 64, 32, 8, dt=600.0, grid_type=grid_type,
                                        options=dict(substep_order=1))
    gb.sync_states(e, v)
    for m in (d, e):
        gb.first_time_step(m)
    assert rel(d.free_surface.eta.interior, e.free_surface.eta.interior) > 1e-5
    with pytest.raises(gb.GB25Error):
        gb.baroclinic_instability_model(gb.GPU(), 48, 24, 6, dt=600.0, grid_type="gaussian_islands", options=dict(substep_order=1))


@pytest.mark.parametrize("float_type", ["Float32", "Float64"])
@pytest.mark.parametrize("grid_type,shape", [("gaussian_islands_lat_lon", (150, 70, 24)), ("lat_lon_as_curvilinear", (150, 70, 24)),
                                             ("gaussian_islands", (144, 64, 24)), ("gaussian_islands", (192, 96, 37))])
def test_the_corrector_through_the_tracer_kernel_is_bitwise_neutral(grid_type, shape, float_type):
    """Round 4, the grids with a bottom / curvilinear / folded: between the steps of one loop! call the barotropic correction is
    not a sweep over u and v -- the tracer kernel (first in the step; every u, v is some cell's west / south face) adds du, dv as
    it loads and writes the corrected velocities into a second pair of arrays, which become u, v and get their halo cells from the
    ordinary fill.  Same operands, same additions, masked at the same faces: the same bits as the sweep (lazy_corrector = 0, which
    keeps w on the fly), every parent cell of every compared field, after loops of different lengths and single steps in between."""
    Nx, Ny, Nz = shape
    models = []
    for lazy in (0, 1):
        m = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=300.0, grid_type=grid_type,
                                            options=dict(lazy_corrector=lazy, subcycle_lookahead=1))
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        gb.first_time_step(m)
        m.backend.profile_enable(True)
        m.backend.profile_reset()
        gb.loop(m, 7)
        gb.time_step(m)
        gb.loop(m, 4)
        models.append(m)
    a, b = models
    # (the sweep ran every step on one side, only in the steps that could not adopt a look-ahead on the other)
    assert a.backend.profile_get("corrector")[0] >= 12
    for n in ALL_FIELDS:
        x, y = a.backend.get_field(n, True), b.backend.get_field(n, True)
        assert np.isfinite(x).all() and np.array_equal(x, y), (n, float(np.abs(x - y).max()))
    assert np.abs(a.backend.get_field("u", False)).max() > 1e-3
