"""2-D (x, y) decomposition -- Partition(Rx, Ry, 1) of the reference (sharding/sharded_baroclinic_instability_simulation_run.jl:
65-72): Rx x Ry ranks stepped in lock-step on ONE GPU by the library's own sequencer (local
transport), every rank a window of columns AND rows of the global grid with walls only where the global grid has them, must
reproduce the single-domain run BIT FOR BIT."""
import numpy as np
import pytest

import gb25_amd as gb
from gb25_amd.distributed import LocalSlabEnsemble
from helpers import counter_rng

pytestmark = pytest.mark.gpu

FIELDS = ["u", "v", "w", "T", "S", "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.u", "Gn.v", "Gn.T", "Gn.S",
          "Gm.u", "Gm.v", "pHY"]


# Bit-for-bit comparisons run the ranks with w from the stand-alone kernel, as the small single domains of these tests compute it
# (w carried inside the tendency kernels -- the default of a flat lat-lon rank in steady state -- is another association of the
# vertical sum: test_w_on_the_fly_on_a_mesh).
EXACT = dict(w_on_the_fly=0)
GRID_NAMES = {0: "simple_lat_lon", 1: "gaussian_islands_lat_lon", 2: "lat_lon_as_curvilinear", 3: "tripolar", 4: "gaussian_islands"}


def _initial(Nx, Ny, Nz, single, vrows):
    gb.set_baroclinic_instability(single)
    u0 = (1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32)
    v0 = (1e-2 * counter_rng((Nx, vrows, Nz), 42, 2)).astype(np.float32)
    e0 = (1e-2 * counter_rng((Nx, Ny, 1), 42, 3)).astype(np.float32)
    single.set(u=u0, v=v0, eta=e0)
    return {n: single.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta")}


def _compare(ens, single, what):
    for n in FIELDS:
        a, b = ens.gather(n), single.backend.get_field(n, False)
        assert a.shape == b.shape, (what, n, a.shape, b.shape)
        if not np.array_equal(a, b, equal_nan=True):
            bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
            raise AssertionError((what, n, float(np.nanmax(np.abs(a - b))), len(bad), bad[:4].tolist(), bad[-2:].tolist()))


@pytest.mark.parametrize("Rx,Ry,Nz,grid_type", [(2, 2, 8, 0), (1, 2, 8, 0), (2, 3, 24, 0), (2, 2, 8, 1), (2, 2, 8, 2), (2, 2, 8, 3),
                                                (2, 2, 8, 4), (3, 2, 24, 4), (1, 3, 8, 4)])
def test_mesh_reproduces_single_domain_bitwise(Rx, Ry, Nz, grid_type):
    Nx, Ny, dt = (192 if Rx == 3 else 128), 48 * Ry, 600.0
    kw = dict(grid_type=grid_type)
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, grid_type=GRID_NAMES[grid_type])
    init = _initial(Nx, Ny, Nz, single, Ny if grid_type >= 3 else Ny + 1)   # (a folded grid has Ny rows of y faces)
    ens = LocalSlabEnsemble(Nx, Ny, Nz, Rx * Ry, dt=dt, ranks_y=Ry, slab_mode=1, options=EXACT, **kw)
    for n, a in init.items():
        ens.scatter(n, a)
    gb.first_time_step(single)
    ens.first_time_step()
    _compare(ens, single, "first step")
    if grid_type == 3:       # (the bare tripolar grid is singular at its poles: NaNs spread from there; compared all the same)
        gb.time_step(single)
        ens.time_step()
        _compare(ens, single, "second step")
        return
    gb.loop(single, 6)
    ens.loop(6)
    _compare(ens, single, "6 steps")
    assert np.abs(single.velocities.u.interior).max() > 1e-2      # a developed, non-trivial flow
    # the staged path took the look-ahead route: the last stage 0 adopted the sub-cycle prepared beside the previous tracer
    # kernel (its wide halos in x AND y exchanged on the second stream), and the next one is already prepared
    assert all(b.lookahead_state() == (True, True) for b in ens.backends)


def test_mesh_with_bottom_drag_and_weno7_tracers():
    """What ocean_simulation adds to the momentum and tracer kernels (quadratic bottom drag, WENO(order = 7) tracer advection:
    a stencil of four rows either side) on a 2 x 2 mesh of the tripolar grid with the islands."""
    Nx, Ny, Nz, dt = 128, 96, 8, 600.0
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, grid_type="gaussian_islands")
    init = _initial(Nx, Ny, Nz, single, Ny)
    ens = LocalSlabEnsemble(Nx, Ny, Nz, 4, dt=dt, ranks_y=2, slab_mode=1, grid_type=4)
    for b in [single.backend] + ens.backends:
        b.set_bottom_drag(0.003)
        b.set_tracer_advection_order(7)
    for n, a in init.items():
        ens.scatter(n, a)
    gb.first_time_step(single)
    ens.first_time_step()
    gb.loop(single, 5)
    ens.loop(5)
    _compare(ens, single, "5 steps")


def test_config4_physics_on_a_mesh():
    """BASELINE.json configs[3] in the decomposition the reference runs it in -- Partition(Rx, Ry, 1) -- at a reduced size: the
    data-free climate model (tripolar grid with the islands, CATKE with the TKE tracer and J^b in the bundles, the analytic
    atmosphere with similarity-theory fluxes after every step, quadratic bottom drag, WENO(order = 7) tracers) on a 2 x 2 mesh
    against the single domain, bit for bit.  kappa and the fluxes of the first halo row / column are computed locally from
    exchanged halos, never exchanged."""
    from gb25_amd.data_free import ATMOSPHERE_FIELDS
    Nx, Ny, Nz, dt, Rx, Ry, H = 128, 96, 8, 30.0, 2, 2, 8
    m = gb.data_free_ocean_climate_model_init(gb.GPU(), Nz=Nz, dt=dt, size=(Nx, Ny))
    rng = np.random.default_rng(9)
    m.set(u=(0.3 * rng.standard_normal((Nx, Ny, Nz))).astype(np.float32), v=(0.3 * rng.standard_normal((Nx, Ny, Nz))).astype(np.float32))
    names = ("u", "v", "T", "S", "e", "eta")
    init = {n: m.backend.get_field(n, False) for n in names}
    atm = gb.analytic_atmosphere()
    gb.first_time_step(m)
    gb.loop(m, 5)
    out = names + ("U", "V", "kappa_u", "kappa_c", "Gn.T", "Gn.u", "Gn.e", "w")
    ref = {n: m.backend.get_field(n, False) for n in out}
    flux = {n: m.backend.top_flux(n) for n in ("u", "v", "T", "S")}
    m.backend.close()
    assert all(np.isfinite(a).all() for a in ref.values()) and ref["kappa_u"].max() > 0 and np.abs(flux["u"]).max() > 1e-6
    ens = LocalSlabEnsemble(Nx, Ny, Nz, Rx * Ry, dt=dt, ranks_y=Ry, grid_type=4)
    for b in ens.backends:
        b.set_catke(True)
        b.set_catke_parameters(**gb.default_ocean_closure().parameters)
        b.set_bottom_drag(0.003)
        b.set_tracer_advection_order(7)
        lp = np.asarray(b.metric2("phicc"))[:, : b.Ny_local + 2 * H]
        for n in ATMOSPHERE_FIELDS:
            b.set_prescribed_atmosphere(n, atm.interpolate(n, np.zeros_like(lp), lp))
    for n, a in init.items():
        ens.scatter(n, a)
    ens.first_time_step()
    ens.loop(5)
    for n, a in ref.items():
        got = ens.gather(n)
        assert got.shape == a.shape and np.array_equal(got, a), (n, float(np.abs(got - a).max()), np.argwhere(got != a)[:3].tolist())
    for n, a in flux.items():
        rows = [np.concatenate([ens.backends[ry * Rx + rx].top_flux(n) for rx in range(Rx)], axis=0) for ry in range(Ry)]
        assert np.array_equal(np.concatenate(rows, axis=1), a), n
    ens.close()


def test_state_dump_of_a_mesh_and_offline_gather(tmp_path):
    """save_model_state / load_all_fields (src/sharded_io.jl:122-138,198-213) on a 2 x 2 mesh: every rank writes its window --
    columns AND rows -- with its slice of the global array; the offline gather is the single domain."""
    Nx, Ny, Nz, dt = 128, 96, 8, 600.0
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
    init = _initial(Nx, Ny, Nz, single, Ny + 1)
    ens = LocalSlabEnsemble(Nx, Ny, Nz, 4, dt=dt, ranks_y=2, options=EXACT)
    for n, a in init.items():
        ens.scatter(n, a)
    gb.first_time_step(single)
    ens.first_time_step()
    gb.loop(single, 2)
    ens.loop(2)
    for b in ens.backends:
        b.save_state(str(tmp_path), "mesh")
    got = gb.load_all_fields(str(tmp_path / "mesh"))
    assert got["iteration"] == 3
    for n in ("u", "v", "w", "eta", "T", "S"):
        ref = single.backend.get_field(n, False)
        assert got[n].shape == ref.shape and np.array_equal(got[n], ref), n


def test_mesh_in_float64_with_implicit_vertical_diffusion():
    """The Float64 library and the closure the reference keeps beside `closure = nothing`
    (VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), kappa, nu), src/baroclinic_instability_model.jl:31) on a
    2 x 2 mesh: the implicit solves are column-local, the wall face of v is where the GLOBAL grid has it."""
    Nx, Ny, Nz, dt = 128, 96, 12, 600.0
    single = gb.baroclinic_instability_model(gb.GPU(float_type="Float64"), Nx, Ny, Nz, dt=dt)
    gb.set_baroclinic_instability(single)
    u0 = 1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)
    v0 = 1e-2 * counter_rng((Nx, Ny + 1, Nz), 42, 2)
    single.set(u=u0.astype(np.float32).astype(np.float64), v=v0.astype(np.float32).astype(np.float64))
    init = {n: single.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta")}
    ens = LocalSlabEnsemble(Nx, Ny, Nz, 4, dt=dt, ranks_y=2, float_type="Float64", options=EXACT)
    for b in [single.backend] + ens.backends:
        b.set_vertical_diffusivity(1e-2, 1e-3)
    for n, a in init.items():
        ens.scatter(n, a)
    gb.first_time_step(single)
    ens.first_time_step()
    gb.loop(single, 5)
    ens.loop(5)
    for n in FIELDS:
        a, b = ens.gather(n), single.backend.get_field(n, False)
        assert a.dtype == np.float64 and a.shape == b.shape and np.array_equal(a, b), (n, float(np.abs(a - b).max()))


@pytest.mark.parametrize("grid_type", [0, 4])
def test_mesh_with_ragged_tiles_and_the_narrowest_bands(grid_type):
    """Ranks of 90 columns x 33 rows (ragged 64-wide tiles, bands barely wider than the sub-cycle's halo of 30 rows) on a 2 x 2
    mesh, 3 x 2 on the tripolar grid with its fold partners (1, 3) <-> (3, 3) and the self-partner in the middle."""
    Rx, Ry = (2, 2) if grid_type == 0 else (3, 2)
    Nx, Ny, Nz, dt = 90 * Rx, 66, 10, 600.0
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, grid_type=GRID_NAMES[grid_type])
    init = _initial(Nx, Ny, Nz, single, Ny if grid_type >= 3 else Ny + 1)
    ens = LocalSlabEnsemble(Nx, Ny, Nz, Rx * Ry, dt=dt, ranks_y=Ry, options=EXACT, **(dict(grid_type=grid_type) if grid_type else {}))
    for n, a in init.items():
        ens.scatter(n, a)
    gb.first_time_step(single)
    ens.first_time_step()
    gb.loop(single, 7)
    ens.loop(7)
    _compare(ens, single, "8 steps")


def test_w_on_the_fly_on_a_mesh():
    """The steady-state schedule of a flat lat-lon rank of a 2-D decomposition: the corrector inside its consumers (du, dv over the
    rank's whole extended range -- halo columns AND rows -- from the column integrals the bundles carry) and w carried inside the
    tendency kernels.  Against the same mesh with w from the stand-alone kernel: the same numbers to round-off."""
    Nx, Ny, Nz, dt = 256, 96, 36, 600.0      # three chunks of levels
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
    init = _initial(Nx, Ny, Nz, single, Ny + 1)
    single.backend.close()
    out = {}
    for fly in (1, 0):
        ens = LocalSlabEnsemble(Nx, Ny, Nz, 4, dt=dt, ranks_y=2, options=dict(w_on_the_fly=fly))
        for n, a in init.items():
            ens.scatter(n, a)
        ens.first_time_step()
        ens.loop(12)
        assert all(b.lookahead_state() == (True, True) for b in ens.backends)
        out[fly] = {n: ens.gather(n).astype(np.float64) for n in FIELDS}
        ens.close()
    worst = {n: float(np.linalg.norm((out[1][n] - out[0][n]).ravel()) / max(np.linalg.norm(out[0][n].ravel()), 1e-300)) for n in FIELDS}
    assert max(worst.values()) < 2e-5, worst
    assert any(v > 0 for v in worst.values()), "w on the fly did not run on the mesh"
