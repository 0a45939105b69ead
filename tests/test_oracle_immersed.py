"""Known-answer pins of the immersed-boundary restatement (oracle):
ImmersedBoundaryGrid(grid, GridFittedBottom(gaussian_islands)) of src/model_utils.jl:67-80,134-146 on the lat-lon grid.
 * a bottom below every cell centre reproduces the flat-bottom model bit for bit;
 * bottom-height materialisation, first active level and static column depths;
 * immersed cells and peripheral faces stay exactly zero; w vanishes at and below the bottom;
 * the tracer budget closes with islands (volume integral changes only through the free surface);
 * a state of rest over topography stays at rest;
 * a barotropic gravity wave feels the local depth (sqrt(g H)) on a shallow flat shelf."""
import math

import numpy as np
import pytest

import gb25_amd as gb
from helpers import make_oracle, set_noisy_velocities

FIELDS = ["u", "v", "w", "T", "S", "pHY", "Gn.u", "Gn.v", "Gn.T", "Gn.S", "Gm.u", "Gm.v", "eta", "U", "V", "eta_bar",
          "U_bar", "V_bar", "Gn.U", "Gn.V"]


def islands(Nx, Ny, Nz, dt=600.0, **kw):
    return make_oracle(Nx, Ny, Nz, dt, grid_type="gaussian_islands_lat_lon", **kw)


def activity(m):
    """(active cells, non-peripheral u faces, non-peripheral v faces) as boolean arrays of the interior shapes."""
    Nx, Ny, Nz = m.grid.size
    kb = np.array([[m.backend.bottom_info("kbot", i, j) for j in range(1, Ny + 1)] for i in range(1, Nx + 1)], int)
    k = np.arange(Nz)[None, None, :]
    active = k >= kb[:, :, None]
    au = active & np.roll(active, 1, axis=0)
    av = np.zeros((Nx, Ny + 1, Nz), bool)
    av[:, 1:Ny] = active[:, 1:] & active[:, :-1]
    return active, au, av


def test_deep_bottom_is_the_flat_model_bit_for_bit():
    a = make_oracle(32, 20, 8, 600.0)
    b = make_oracle(32, 20, 8, 600.0)
    b.backend.set_bottom_height(np.full((32, 20), -4000.0))      # GridFittedBottom at the depth of the grid
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 1e-2)
        gb.first_time_step(m)
        gb.loop(m, 5)
    for n in FIELDS:
        assert np.array_equal(a.backend.get_field(n, True), b.backend.get_field(n, True)), n


def test_bottom_materialisation_and_column_depths():
    m = make_oracle(16, 12, 6, 60.0)
    zf = np.array([m.backend.metric("zf", k) for k in range(1, 8)])
    zc = np.array([m.backend.metric("zc", k) for k in range(1, 7)])
    zb = np.full((16, 12), -1e9)
    zb[3, 4] = zc[2]                   # exactly a cell centre: z_center <= bottom => that cell is immersed too
    zb[4, 4] = zc[2] - 1e-6            # just below it: the cell stays active
    zb[5, 4] = 50.0                    # above the surface: land
    m.backend.set_bottom_height(zb)
    info = lambda w, i, j: m.backend.bottom_info(w, i + 1, j + 1)
    assert [info("kbot", i, 4) for i in (2, 3, 4, 5)] == [0, 3, 2, 6]
    assert info("Hcc", 3, 4) == pytest.approx(-zf[3]) and info("Hcc", 4, 4) == pytest.approx(-zf[2])
    assert info("Hcc", 5, 4) == 0.0 and info("Hcc", 2, 4) == pytest.approx(4000.0)
    # static depth at a face = min of the two columns it separates
    assert info("Hfc", 4, 4) == info("Hcc", 3, 4) and info("Hfc", 5, 4) == 0.0 and info("Hfc", 6, 4) == 0.0
    assert info("Hcf", 3, 5) == info("Hcc", 3, 4) and info("Hcf", 3, 4) == info("Hcc", 3, 4)


def test_gaussian_islands_geometry():
    m = islands(180, 90, 12)
    kb = np.array([[m.backend.bottom_info("kbot", i, j) for j in range(1, 91)] for i in range(1, 181)])
    # two mountains at (70E, 55N) and (250E, 55N), 5 degrees wide, peaking 100 m above the surface: land at the peaks
    i1, i2, j0 = int(70 / 2), int(250 / 2), int((55 + 80) / (160 / 90))
    assert kb[i1, j0] == 12 and kb[i2, j0] == 12
    assert kb[:, :40].max() == 0 and kb[90, j0] == 0              # far from the mountains nothing is immersed
    assert np.array_equal(kb[i1 - 8:i1 + 8], kb[i2 - 8:i2 + 8])   # the second mountain is the first one shifted by 180 degrees
    assert 0 < (kb > 0).mean() < 0.1


def test_immersed_cells_stay_zero_and_w_vanishes_in_the_solid():
    m = islands(180, 80, 10, dt=600.0)       # 2-degree cells with centres at (69|71 E, 55 N): the peaks are land
    gb.set_baroclinic_instability(m)
    set_noisy_velocities(m, 1e-2)
    gb.first_time_step(m)
    gb.loop(m, 8)
    active, au, av = activity(m)
    assert (~active).sum() > 50 and (active[:, :, -1]).mean() > 0.9
    for name, ok in (("u", au), ("v", av), ("T", active), ("S", active), ("Gn.u", au), ("Gn.T", active)):
        a = m.backend.get_field(name, False)
        assert np.all(a[~ok] == 0.0), name
        assert np.abs(a[ok]).max() > 0, name
    gv = m.backend.get_field("Gn.v", False)
    assert np.all(gv[:, 1:80][~av[:, 1:80]] == 0.0)
    w = m.backend.get_field("w", False)                # faces 0..Nz; face k is the bottom of cell k
    below = np.concatenate([~active, np.zeros((180, 80, 1), bool)], axis=2)
    assert np.all(w[below] == 0.0) and np.isfinite(w).all()
    U, V = m.backend.get_field("U", False)[:, :, 0], m.backend.get_field("V", False)[:, :, 0]
    land_u = ~au[:, :, -1]
    assert land_u.any() and np.all(U[land_u] == 0.0) and np.all(V[~av[:, :, -1]] == 0.0)
    assert np.abs(m.velocities.u.interior).max() < 1.0 and np.isfinite(m.free_surface.eta.interior).all()


def test_tracer_budget_closes_with_islands():
    m = islands(72, 40, 10, dt=10.0)
    set_noisy_velocities(m, amplitude=0.1)
    Nx, Ny, Nz = m.grid.size
    rng = np.random.default_rng(1)
    m.set(T=10 + rng.random((Nx, Ny, Nz)), S=35 + 0 * rng.random((Nx, Ny, Nz)))
    gb.update_state(m)                                   # masks, fills, w, tendencies
    active, _, _ = activity(m)
    b = m.backend
    az = np.array([b.metric("azc", j) for j in range(1, Ny + 1)])
    dz = np.array([b.metric("dzc", k) for k in range(1, Nz + 1)])
    V = az[None, :, None] * dz[None, None, :]
    G = m.timestepper.Gn.T.interior
    assert np.all(G[~active] == 0.0)
    total = (V * G).sum()
    wtop = m.velocities.w.interior[:, :, Nz]
    Tp = m.tracers.T.parent
    H = 8
    c_in, c_halo = Tp[H:-H, H:-H, H + Nz - 1], Tp[H:-H, H:-H, H + Nz]
    top_flux = (az[None, :] * wtop * np.where(wtop > 0, c_in, c_halo)).sum()
    assert abs(total + top_flux) < 1e-12 * np.abs(V * G).sum()
    # a constant tracer has zero tendency in every active cell (continuity + advection stay consistent next to the solid)
    m.set(T=np.where(active, 7.0, 0.0))
    gb.update_state(m)
    assert np.abs(m.timestepper.Gn.T.interior).max() < 1e-17 * 7 * 1e6


def test_rest_state_over_topography_stays_at_rest():
    m = islands(72, 40, 10, dt=600.0)
    Nx, Ny, Nz = m.grid.size
    zc = np.array([m.backend.metric("zc", k) for k in range(1, Nz + 1)])
    m.set(T=np.broadcast_to(10 + 5e-3 * zc, (Nx, Ny, Nz)), S=np.broadcast_to(35 - 1e-3 * zc, (Nx, Ny, Nz)))
    gb.first_time_step(m)
    gb.loop(m, 3)
    for name in ("u", "v", "w", "eta", "U", "V"):
        assert np.abs(m.backend.get_field(name, False)).max() == 0.0, name
    assert np.abs(m.timestepper.Gn.u.interior).max() == 0.0


def test_barotropic_wave_feels_the_local_depth():
    """A zonal free-surface wave in a channel whose whole floor is raised to a shelf: phase speed sqrt(g H_shelf)."""
    Nx, Ny, Nz, nsteps, mode = 64, 16, 8, 120, 2
    m = make_oracle(Nx, Ny, Nz, 200.0, lat_south=-2.0, lat_north=2.0, Omega=0.0)
    zf = np.array([m.backend.metric("zf", k) for k in range(1, Nz + 2)])
    m.backend.set_bottom_height(np.full((Nx, Ny), zf[4] + 0.25 * (zf[5] - zf[4])))   # four levels immersed
    Hd = -zf[4]
    assert m.backend.bottom_info("kbot", 3, 3) == 4 and m.backend.bottom_info("Hcc", 3, 3) == pytest.approx(Hd)
    lam = (np.arange(Nx) + 0.5) * 2 * np.pi / Nx
    A = 1e-3
    m.set(eta=np.broadcast_to((A * np.cos(mode * lam))[:, None, None], (Nx, Ny, 1)))
    gb.first_time_step(m)
    gb.loop(m, nsteps - 1)
    t = m.clock.time
    omega = math.sqrt(9.80665 * Hd) * mode / 6371e3
    eta = m.free_surface.eta.interior[:, Ny // 2, 0]
    amp = 2 * (eta * np.cos(mode * lam)).mean() / A
    assert abs(amp - math.cos(omega * t)) < 0.03
    deep = math.cos(math.sqrt(9.80665 * 4000.0) * mode / 6371e3 * t)
    assert abs(amp - deep) > 0.1                                                  # (and not the full depth's speed)
