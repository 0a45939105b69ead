"""Quadratic bottom drag in the oracle (what ClimaOcean's ocean_simulation gives u and v at the bottom, also the immersed one;
GB-25 src/data_free_ocean_climate_model.jl:26) [UPSTREAM-UNVERIFIED]: known answers."""
import numpy as np
import pytest

import gb25_amd as gb
from oracle_backend import CPU


def tendencies(grid_type, Cd, u0=0.3, v0=-0.2):
    m = gb.baroclinic_instability_model(CPU("f64"), 48, 44, 6, dt=60.0, grid_type=grid_type)
    Nx, Ny, Nz = m.grid.size
    m.set(u=np.full((Nx, Ny, Nz), u0), v=np.full((Nx, Ny + 1, Nz), v0))
    m.backend.set_bottom_drag(Cd)
    gb.update_state(m)
    return m, m.backend.get_field("Gn.u", False), m.backend.get_field("Gn.v", False), m.backend.get_field("u", False), m.backend.get_field("v", False)


def test_drag_enters_the_bottom_level_only_with_the_right_size():
    Cd = 0.003
    m, Gu1, Gv1, u, v = tendencies("simple_lat_lon", Cd)
    _, Gu0, Gv0, _, _ = tendencies("simple_lat_lon", 0.0)
    dz = m.backend.metric("dzc", 1)
    dGu, dGv = Gu1 - Gu0, Gv1 - Gv0
    assert np.abs(dGu[:, :, 1:]).max() < 1e-18 and np.abs(dGv[:, :, 1:]).max() < 1e-18      # upper levels untouched
    # away from the walls (v is masked to zero on them) the speed is sqrt(u0^2 + v0^2) everywhere
    want_u = -Cd * 0.3 * np.hypot(0.3, 0.2) / dz
    want_v = -Cd * (-0.2) * np.hypot(0.3, 0.2) / dz
    assert np.allclose(dGu[:, 2:-2, 0], want_u, rtol=1e-9)
    assert np.allclose(dGv[:, 2:-2, 0], want_v, rtol=1e-9)
    assert want_u < 0 < want_v                                                               # it opposes the flow


def test_drag_sits_on_the_immersed_bottom():
    Cd = 0.003
    m, Gu1, _, u, _ = tendencies("gaussian_islands_lat_lon", Cd, v0=0.0)
    _, Gu0, _, _, _ = tendencies("gaussian_islands_lat_lon", 0.0, v0=0.0)
    d = Gu1 - Gu0
    Nx, Ny, Nz = m.grid.size
    seen = 0
    for i in range(2, Nx - 1, 5):
        for j in range(3, Ny - 3, 4):
            col = d[i, j]
            nz = np.nonzero(col)[0]
            wet = np.nonzero(u[i, j])[0]                       # (set! masked the faces that touch the solid)
            if wet.size == 0:
                assert nz.size == 0
                continue
            assert list(nz) == [wet[0]], (i, j, nz, wet)       # exactly the first free level
            seen += 1
    assert seen > 20


def test_drag_spins_a_barotropic_flow_down():
    m = gb.baroclinic_instability_model(CPU("f64"), 32, 24, 4, dt=600.0, depth=40.0)
    Nx, Ny, Nz = m.grid.size
    m.set(u=np.full((Nx, Ny, Nz), 0.5))
    m.backend.set_bottom_drag(0.01)
    ke0 = (m.backend.get_field("u", False) ** 2).sum()
    gb.first_time_step(m)
    gb.loop(m, 20)
    n = gb.baroclinic_instability_model(CPU("f64"), 32, 24, 4, dt=600.0, depth=40.0)
    n.set(u=np.full((Nx, Ny, Nz), 0.5))
    gb.first_time_step(n)
    gb.loop(n, 20)
    ke_drag, ke_free = (m.backend.get_field("u", False) ** 2).sum(), (n.backend.get_field("u", False) ** 2).sum()
    assert np.isfinite(ke_drag) and ke_drag < 0.97 * ke_free
