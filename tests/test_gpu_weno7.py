"""tracer_advection = WENO(order = 7) (ClimaOcean's ocean_simulation) on the HIP path against the oracle
(tests/test_oracle_weno7.py pins that one): tendencies and stepping on the three kinds of grid, CATKE's e, slabs."""
import numpy as np
import pytest

import gb25_amd as gb
from helpers import SQRT_EPS32, make_pair, set_noisy_velocities

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = max(np.linalg.norm(a.ravel()), np.linalg.norm(b.ravel()))
    return 0.0 if n == 0 else float(np.linalg.norm((a - b).ravel()) / n)


@pytest.mark.parametrize("float_type", ["Float64", "Float32"])
@pytest.mark.parametrize("grid_type", ["simple_lat_lon", "gaussian_islands_lat_lon", "gaussian_islands"])
def test_order_seven_matches_the_oracle(grid_type, float_type):
    r, v = make_pair(96, 44, 12, dt=300.0, float_type=float_type, grid_type=grid_type)
    for m in (r, v):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.1)
        m.backend.set_tracer_advection_order(7)
        gb.update_state(m)
    w = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), 96, 44, 12, dt=300.0, grid_type=grid_type)
    gb.set_baroclinic_instability(w)
    set_noisy_velocities(w, 0.1)
    gb.update_state(w)
    assert rel(w.backend.get_field("Gn.T", False), r.backend.get_field("Gn.T", False)) > 1e-4     # (it is not order 5)
    w.backend.close()
    tol = 1e-9 if float_type == "Float64" else SQRT_EPS32
    for n in ("Gn.T", "Gn.S"):
        assert rel(r.backend.get_field(n, False), v.backend.get_field(n, False)) < tol, n
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 6)
    _, report = gb.compare_states(r, v, rtol=SQRT_EPS32, include_halos=True, verbose=False)
    bad = [(q["name"], q["rel"]) for q in report if not q["rel"] <= (1e-8 if float_type == "Float64" else SQRT_EPS32)]
    assert not bad, bad
    assert r.backend.tracer_advection_order() == 7


def test_order_seven_with_catke_and_on_slabs():
    from gb25_amd.distributed import LocalSlabEnsemble
    Nx, Ny, Nz, dt, P = 192, 44, 12, 120.0, 3
    m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, grid_type="gaussian_islands", closure=gb.CATKEVerticalDiffusivity())
    gb.set_baroclinic_instability(m)
    set_noisy_velocities(m, 0.1)
    m.backend.set_tracer_advection_order(7)
    rng = np.random.default_rng(2)
    m.set(e=(1e-5 * rng.random((Nx, Ny, Nz)) + 1e-7).astype(np.float32))
    init = {n: m.backend.get_field(n, False) for n in ("u", "v", "T", "S", "e", "eta")}
    gb.first_time_step(m)
    gb.loop(m, 5)
    ref = {n: m.backend.get_field(n, False) for n in ("u", "v", "T", "S", "e", "eta", "Gn.T", "Gn.e")}
    m.backend.close()
    ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, grid_type=4)
    for b in ens.backends:
        b.set_catke(True)
        b.set_tracer_advection_order(7)
    for n, a in init.items():
        ens.scatter(n, a)
    ens.first_time_step()
    ens.loop(5)
    bad = [n for n, a in ref.items() if not np.array_equal(ens.gather(n), a)]
    assert not bad, [(n, rel(ens.gather(n), ref[n])) for n in bad]
    ens.close()


def test_order_seven_on_a_shallow_grid_and_its_refusals():
    """Five levels: every vertical stencil is cut by the bottom or the surface (orders 1, 3, 5 only in z); Float64 so that the
    comparison is about logic."""
    from gb25_amd.binding import GB25Error
    r, v = make_pair(64, 44, 5, dt=300.0, float_type="Float64", grid_type="gaussian_islands_lat_lon")
    for m in (r, v):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.1)
        m.backend.set_tracer_advection_order(7)
        gb.first_time_step(m)
        gb.loop(m, 3)
    for n in ("T", "S", "Gn.T", "Gn.S"):
        assert rel(r.backend.get_field(n, True), v.backend.get_field(n, True)) < 1e-9, n
    with pytest.raises(GB25Error):
        r.backend.set_tracer_advection_order(6)
    # (a halo of 3 could not carry the eight-point stencil: gb25_create asks for at least 4 anyway)
