"""Flux boundary conditions on the HIP path: compute_hydrostatic_boundary_tendency_contributions!
(src/precompile.jl:25,52-61) with top FluxBoundaryConditions on u, v, T, S -- what ClimaOcean's
ocean_simulation hands the ocean model (src/data_free_ocean_climate_model.jl:26) -- against the oracle."""
import numpy as np
import pytest

import gb25_amd as gb
from helpers import assert_states_close, counter_rng, make_pair, set_noisy_velocities

pytestmark = pytest.mark.gpu
ALL_FIELDS = ["u", "v", "w", "T", "S", "pHY", "Gn.u", "Gn.v", "Gn.T", "Gn.S", "Gm.u", "Gm.v", "Gm.T", "Gm.S",
              "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.U", "Gn.V"]


def fluxes(Nx, Ny):
    lam = (np.arange(Nx) + 0.5) * 2 * np.pi / Nx
    phi = np.linspace(-1, 1, Ny)
    return dict(T=1e-4 * (1 + np.cos(lam)[:, None] * np.cos(phi)[None, :]),                 # heat
                S=-2e-5 * np.sin(2 * lam)[:, None] * np.ones(Ny)[None, :],                  # fresh water
                u=-1e-4 * (4 * np.sin(2 * phi * 1.4) ** 2)[None, :] * np.ones(Nx)[:, None],  # zonal wind stress
                v=2e-5 * counter_rng((Nx, Ny + 1), 11, 1))


@pytest.mark.parametrize("grid", [{}, dict(grid_type="gaussian_islands_lat_lon")])
def test_top_fluxes_match_the_oracle(grid):
    Nx, Ny, Nz = 180, 80, 10
    r, v = make_pair(Nx, Ny, Nz, dt=600.0, **grid)
    gb.set_baroclinic_instability(v)
    set_noisy_velocities(v, 1e-3)
    for n in ALL_FIELDS:
        a = v.backend.get_field(n, True).astype(np.float32)
        r.backend.set_field(n, a, True)
        v.backend.set_field(n, a.astype(np.float64), True)
    J = fluxes(Nx, Ny)
    for m in (r, v):
        gb.set_top_flux(m, **J)
        gb.first_time_step(m)
        gb.loop(m, 5)
    assert_states_close(r, v, label=f"top fluxes {grid}")
    # the forcing is felt: the top level of T differs from an unforced twin by about 6 steps of J dt / dz
    twin = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=600.0, **grid)
    gb.set_baroclinic_instability(twin)
    set_noisy_velocities(twin, 1e-3)
    gb.first_time_step(twin)
    gb.loop(twin, 5)
    dT = (r.tracers.T.interior - twin.tracers.T.interior)[:, :, -1]
    dz = r.grid.metric("dzc", Nz)
    active = r.tracers.T.interior[:, :, -1] != 0
    expected = (-6 * 600.0 * J["T"] / dz)[active]
    big = np.abs(expected) > 0.2 * np.abs(expected).max()
    assert np.median(np.abs(dT[active][big] - expected[big]) / np.abs(expected[big])) < 0.05


def test_fluxes_keep_the_lookaheads_bitwise_neutral_and_can_be_removed():
    Nx, Ny, Nz = 150, 70, 12
    J = fluxes(Nx, Ny)
    a = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=600.0, options=dict(ab2_lookahead=0, fold_fills=0))
    b = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=600.0, options=dict(subcycle_lookahead=1, w_on_the_fly=0))
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        gb.set_top_flux(m, **J)
        gb.first_time_step(m)
        gb.loop(m, 6)
    for n in ALL_FIELDS:
        assert np.array_equal(a.backend.get_field(n, True), b.backend.get_field(n, True)), n
    # removing the fluxes mid-run (a setter: voids the look-aheads) and going on
    for m in (a, b):
        gb.set_top_flux(m, T=None, S=None, u=None, v=None)
        gb.loop(m, 3)
    for n in ALL_FIELDS:
        assert np.array_equal(a.backend.get_field(n, True), b.backend.get_field(n, True)), ("removed", n)
    with pytest.raises(gb.GB25Error, match="no flux boundary"):
        c = gb.baroclinic_instability_model(gb.GPU(), 64, 32, 8, dt=60.0, options=dict(kernels=1))
        gb.set_top_flux(c, T=np.zeros((64, 32)))
        gb.update_state(c)


def test_slabs_with_fluxes_bitwise():
    from gb25_amd.distributed import LocalSlabEnsemble
    Nx, Ny, Nz, P, dt = 256, 48, 12, 4, 600.0
    J = fluxes(Nx, Ny)
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
    gb.set_baroclinic_instability(single)
    single.set(u=(1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32))
    init = {n: single.backend.get_field(n, False) for n in ("u", "T", "S")}
    gb.set_top_flux(single, **J)
    ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, options=dict(w_on_the_fly=0))   # (bit for bit: w from the stand-alone kernel)
    for n, a in init.items():
        ens.scatter(n, a)
    for r, b in enumerate(ens.backends):
        for n, a in J.items():
            b.set_top_flux(n, a[r * (Nx // P):(r + 1) * (Nx // P)])
    gb.first_time_step(single)
    ens.first_time_step()
    gb.loop(single, 5)
    ens.loop(5)
    for n in ALL_FIELDS:
        assert np.array_equal(ens.gather(n), single.backend.get_field(n, False)), n
    ens.close()
