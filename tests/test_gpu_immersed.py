"""Immersed boundary on the HIP path: ImmersedBoundaryGrid(lat-lon grid,
GridFittedBottom(gaussian_islands)) -- src/model_utils.jl:67-80,134-146 -- against the oracle, which states the same
rules cell by cell (inactive_cell / stencil_active) where the kernels read per-column tables folded on the host."""
import numpy as np
import pytest

import gb25_amd as gb
from helpers import SQRT_EPS32, assert_states_close, counter_rng, make_pair, set_noisy_velocities

pytestmark = pytest.mark.gpu
ISLANDS = dict(grid_type="gaussian_islands_lat_lon")
ALL_FIELDS = ["u", "v", "w", "T", "S", "pHY", "Gn.u", "Gn.v", "Gn.T", "Gn.S", "Gm.u", "Gm.v", "Gm.T", "Gm.S",
              "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.U", "Gn.V"]


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = max(np.linalg.norm(a.ravel()), np.linalg.norm(b.ravel()))
    return 0.0 if n == 0 else float(np.linalg.norm((a - b).ravel()) / n)


def start(r, v, amplitude=1e-2):
    gb.set_baroclinic_instability(v)
    set_noisy_velocities(v, amplitude)
    for n in ALL_FIELDS:
        a = v.backend.get_field(n, True).astype(np.float32)
        r.backend.set_field(n, a, True)
        v.backend.set_field(n, a.astype(v.backend.dtype), True)


def test_bottom_at_the_grid_depth_is_bitwise_the_plain_model():
    """GridFittedBottom at -4000 m immerses nothing: the model is the flat-bottom model, bit for bit."""
    Nx, Ny, Nz, dt = 150, 70, 24, 600.0
    a = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
    b = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
    b.backend.set_bottom_height(np.full((Nx, Ny), -4000.0))
    assert b.backend.get_option("immersed_kernels") == 0
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        gb.first_time_step(m)
        gb.loop(m, 6)
    for n in ALL_FIELDS:
        assert np.array_equal(a.backend.get_field(n, True), b.backend.get_field(n, True)), n


@pytest.mark.parametrize("float_type,tol", [("Float64", 1e-12), ("Float32", 5e-6)])
def test_flat_bottom_through_the_immersed_kernels(float_type, tol):
    """The same flat bottom FORCED through the immersed-boundary kernel variants (per-column order tables, depth arrays,
    masks): the logic must be the plain kernels' -- agreement to round-off of the Float64 build (1e-12; the two
    template instances contract different multiply-adds into FMAs, so the last bit may differ) and of Float32."""
    Nx, Ny, Nz, dt = 150, 70, 24, 600.0
    dtype = np.float64 if float_type == "Float64" else np.float32
    a = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=dt)
    b = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=dt)
    b.backend.set_bottom_height(np.full((Nx, Ny), -4000.0))
    b.backend.set_option("immersed_kernels", 1)
    assert b.backend.get_option("immersed_kernels") == 1
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        m.set(eta=(1e-2 * counter_rng((Nx, Ny, 1), 3, 3)).astype(dtype))
        gb.first_time_step(m)
        gb.loop(m, 6)
    for n in ALL_FIELDS:
        x, y = a.backend.get_field(n, False), b.backend.get_field(n, False)
        assert rel(x, y) < tol, (n, rel(x, y))
    assert np.abs(a.velocities.u.interior).max() > 0.05


def test_bottom_tables_match_the_oracle():
    r, v = make_pair(180, 80, 10, dt=600.0, **ISLANDS)
    for i in range(1, 181):
        for j in range(1, 81):
            assert r.backend.bottom_info("kbot", i, j) == v.backend.bottom_info("kbot", i, j), (i, j)
            assert r.backend.bottom_info("Hfc", i, j) == pytest.approx(v.backend.bottom_info("Hfc", i, j), rel=1e-6)
            assert r.backend.bottom_info("Hcf", i, j) == pytest.approx(v.backend.bottom_info("Hcf", i, j), rel=1e-6)
    assert r.backend.get_option("immersed_kernels") == 1
    assert max(v.backend.bottom_info("kbot", i, 68) for i in range(1, 181)) == 10      # land at the peaks


def test_phase_by_phase_with_islands():
    r, v = make_pair(180, 80, 10, dt=600.0, **ISLANDS)
    start(r, v)
    get = lambda m, n: m.backend.get_field(n, True)
    sync = lambda: [r.backend.set_field(n, get(v, n).astype(np.float32), True) or
                    v.backend.set_field(n, get(v, n).astype(np.float32).astype(np.float64), True) for n in ALL_FIELDS]
    # masking is data movement: identical parents
    for m in (r, v):
        m.backend.mask_immersed_fields()
    for n in ("u", "v", "T", "S", "U", "V"):
        assert np.array_equal(get(r, n), get(v, n).astype(np.float32)), n
    sync()
    for m in (r, v):
        m.backend.initialize()
        m.backend.update_state()
    H = 8
    core = (slice(H - 1, -(H - 1)), slice(H - 1, -(H - 1)), slice(H, -H))
    assert rel(get(r, "w")[core], get(v, "w")[core]) < 1e-5
    for n, tol in (("Gn.T", 2e-4), ("Gn.S", 2e-4), ("Gn.u", 2e-4), ("Gn.v", 2e-4)):
        assert rel(get(r, n), get(v, n)) < tol, (n, rel(get(r, n), get(v, n)))
    # exactly the same cells carry no tendency (faces that touch the solid)
    for n in ("Gn.u", "Gn.v"):
        assert np.array_equal(get(r, n) == 0, get(v, n) == 0), n
    assert np.all(get(r, "Gn.T")[(get(v, "T") == 0) & (get(v, "Gn.T") == 0)] == 0)      # nothing flows into the solid
    for euler in (True, False):
        sync()
        for m in (r, v):
            m.backend.ab2_step(600.0, euler)
        for n in ("u", "v", "T", "S", "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.U", "Gn.V"):
            assert rel(get(r, n), get(v, n)) < 2e-5, (n, euler, rel(get(r, n), get(v, n)))
    sync()
    for m in (r, v):
        m.backend.fill_halo_regions()
        m.backend.correct_velocities_and_cache_previous_tendencies(600.0)
    for n in ("u", "v", "U_bar", "V_bar"):
        assert rel(get(r, n), get(v, n)) < 2e-6, n
        assert np.array_equal(get(r, n)[H:-H, H:-H] == 0, get(v, n)[H:-H, H:-H] == 0), n


def test_config2_size_with_the_gaussian_mountains():
    """360x180x24 (BASELINE configs[1]'s grid) with the two Gaussian mountains: first_time_step! + 4 steps against the
    oracle at the reference's tolerance, every compared field, halos included; immersed cells exactly zero."""
    r, v = make_pair(360, 180, 24, dt=600.0, **ISLANDS)
    start(r, v, 1e-3)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 4)
    assert_states_close(r, v, label="360x180x24 islands after 5 steps")
    kb = np.array([[r.backend.bottom_info("kbot", i, j) for j in range(1, 181)] for i in range(1, 361)], int)
    assert kb.max() == 24 and (kb > 0).sum() > 500
    active = np.arange(24)[None, None, :] >= kb[:, :, None]
    for n in ("T", "S", "Gn.T"):
        assert np.all(r.backend.get_field(n, False)[~active] == 0.0), n
    u = r.backend.get_field("u", False)
    au = active & np.roll(active, 1, axis=0)
    assert np.all(u[~au] == 0.0) and np.abs(u[au]).max() > 1e-3
    assert np.isfinite(r.backend.get_field("eta", False)).all()


def test_random_bathymetry_matches_the_oracle():
    """An arbitrary bottom (set_bottom_height): columns raised to random levels, land patches, single-cell pits --
    every combination of the order-reduction tables next to each other."""
    Nx, Ny, Nz = 70, 44, 12
    r, v = make_pair(Nx, Ny, Nz, dt=300.0)
    zf = np.array([v.backend.metric("zf", k) for k in range(1, Nz + 2)])
    rng = np.random.default_rng(7)
    level = np.where(rng.random((Nx, Ny)) < 0.35, rng.integers(1, Nz + 1, (Nx, Ny)), 0)
    level[10:14, 20:24] = Nz                                          # an island
    zb = np.where(level > 0, zf[level] - 1e-3, -5000.0)               # just below the top face of cell `level`
    for m in (r, v):
        m.backend.set_bottom_height(zb)
    assert np.array_equal(np.array([[r.backend.bottom_info("kbot", i + 1, j + 1) for j in range(Ny)] for i in range(Nx)]),
                          level)
    start(r, v)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 4)
    assert np.isfinite(r.velocities.u.interior).all()
    _, report = gb.compare_states(r, v, rtol=SQRT_EPS32, include_halos=False, verbose=False)
    bad = [(q["name"], q["rel"]) for q in report if not q["rel"] <= SQRT_EPS32]
    assert not bad, bad


@pytest.mark.parametrize("P", [3, 5])
def test_slabs_with_islands_reproduce_the_single_domain_bitwise(P):
    """Decomposition invariance with bathymetry: 180 columns in 3 / 5 slabs (the first mountain, centred at 70 E,
    straddles the slab edge at 72 E when P = 5); halo columns see the neighbour's bottom without any exchange."""
    from gb25_amd.distributed import LocalSlabEnsemble
    Nx, Ny, Nz, dt = 180, 80, 12, 600.0
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, **ISLANDS)
    gb.set_baroclinic_instability(single)
    single.set(u=(1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32),
               v=(1e-2 * counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(np.float32),
               eta=(1e-2 * counter_rng((Nx, Ny, 1), 42, 3)).astype(np.float32))
    init = {n: single.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta")}
    ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, grid_type=1)
    for n, a in init.items():
        ens.scatter(n, a)
    gb.first_time_step(single)
    ens.first_time_step()
    gb.loop(single, 6)
    ens.loop(6)
    for n in ALL_FIELDS:
        a, b = ens.gather(n), single.backend.get_field(n, False)
        assert np.array_equal(a, b), (P, n, float(np.abs(a - b).max()))
    assert all(b.lookahead_state() == (True, True) for b in ens.backends)
    ens.close()
