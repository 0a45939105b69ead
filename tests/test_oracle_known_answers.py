"""Pins for the CPU oracle (oracle/gb25_oracle.c).

The reference commits no golden vectors and Julia is absent ("parity
unpinned"), so the oracle is pinned by (a) the one published known-answer value on this path
(the TEOS-10 polynomial check value of Roquet et al. 2015), (b) analytic properties of each
operator, and (c) the committed fp64 fixtures in tests/golden/ (test_golden.py).
"""
import math

import numpy as np
import pytest

import gb25_amd as gb
from helpers import make_oracle, set_noisy_velocities
from oracle_backend import OracleBackend


@pytest.fixture(scope="module")
def ob():
    return OracleBackend(16, 16, 4, dt=1.0)


# ------------------------------------------------------------------------------------ TEOS-10
def test_teos10_published_check_value(ob):
    # Roquet et al. (2015), polyTEOS10-bsq check value: rho(SA=30 g/kg, CT=10 C, Z=-1000 m) = 1027.45140 kg/m3
    assert abs(ob.teos10_rho(10.0, 30.0, -1000.0) - 1027.45140) < 6e-6


def test_teos10_physical_sanity(ob):
    assert abs(ob.teos10_rho(0.0, 35.16504, 0.0) - 1028.106) < 2e-3      # standard ocean reference density
    assert abs(ob.teos10_rho(0.0, 0.0, 0.0) - 999.843) < 2e-3            # fresh water at 0 C
    r = [ob.teos10_rho(t, 0.0, 0.0) for t in (0.0, 2.0, 4.0, 6.0, 8.0)]
    assert int(np.argmax(r)) == 2                                      # fresh-water density maximum near 4 C
    assert ob.teos10_rho(10, 36, 0) > ob.teos10_rho(10, 35, 0)            # haline contraction
    assert ob.teos10_rho(20, 35, 0) < ob.teos10_rho(10, 35, 0)            # thermal expansion
    assert ob.teos10_rho(2, 35, -4000) - ob.teos10_rho(2, 35, 0) > 15     # compressibility


# ------------------------------------------------------------------------------------ WENO
def _cell_averages(f_antiderivative, x_faces):
    return np.diff(f_antiderivative(x_faces)) / np.diff(x_faces)


def test_weno5_reproduces_quadratics(ob):
    # cell averages of q(x) = 3x^2 - 2x + 1 on unit cells centred at -2..2; face at x = 0.5
    F = lambda x: x**3 - x**2 + x
    faces = np.arange(-2.5, 3.0, 1.0)
    a = _cell_averages(F, faces)
    exact = 3 * 0.25 - 1 + 1
    assert abs(ob.weno5(*a) - exact) < 1e-13


def test_weno5_linear_weights_on_smooth_data(ob):
    # linear data: all smoothness indicators equal -> tau = 0 -> the optimal 5th-order combination
    a = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    assert abs(ob.weno5(*a) - 3.5) < 1e-14
    # optimal combination = (2a - 13b + 47c + 27d - 3e)/60
    rng = np.random.default_rng(0)
    base = 10 + 1e-6 * rng.standard_normal(5)   # tiny variations: betas ~ 1e-12 << eps
    opt = (2 * base[0] - 13 * base[1] + 47 * base[2] + 27 * base[3] - 3 * base[4]) / 60
    assert abs(ob.weno5(*base) - opt) < 1e-9


def test_weno5_fifth_order_convergence(ob):
    F = lambda x: -np.cos(x)   # antiderivative of sin

    def err(n):
        h = 1.0 / n
        x0 = 0.3
        faces = x0 + h * np.arange(-2.5, 3.0, 1.0)
        a = _cell_averages(F, faces)
        return abs(ob.weno5(*a) - math.sin(x0 + 0.5 * h))

    e1, e2 = err(8), err(16)
    assert 20 < e1 / e2 < 45   # ~2^5


def test_weno5_avoids_discontinuity(ob):
    # step between d and e: the most-downwind stencil must be switched off, result stays ~bounded
    v = ob.weno5(1.0, 1.0, 1.0, 1.0, 100.0)
    assert abs(v - 1.0) < 1e-3


def test_weno3_linear_and_upwind(ob):
    assert abs(ob.weno3(1.0, 2.0, 3.0) - 2.5) < 1e-14
    assert abs(ob.weno3(1.0, 1.0, 50.0) - 1.0) < 1e-3


# ------------------------------------------------------------------------------------ grid / substeps
def test_substep_weights():
    b = OracleBackend(16, 16, 4, dt=1.0, substeps=30)
    n, frac, w = b.substepping()
    assert n == 21 and abs(frac - 2 / 30) < 1e-15
    assert abs(w.sum() - 1) < 1e-14
    assert (w[:4] < 0).all() and (w[4:] > 0).all()
    # centroid of the averaging kernel sits at the new time level (tau = 1 in units of dt)
    tau = frac * np.arange(1, n + 1)
    assert abs((w * tau).sum() - 1.0) < 0.02


def test_grid_metrics():
    Nx, Ny, Nz = 128, 64, 8
    b = OracleBackend(Nx, Ny, Nz, dt=1.0)
    zf = np.array([b.metric("zf", k) for k in range(1, Nz + 2)])
    assert abs(zf[0] + 4000.0) < 1e-9 and zf[-1] == 0.0 and (np.diff(zf) > 0).all()
    dz = np.array([b.metric("dzc", k) for k in range(1, Nz + 1)])
    assert abs(dz.sum() - 4000) < 1e-9 and dz[0] > dz[-1]       # finest at the surface
    az = np.array([b.metric("azc", j) for j in range(1, Ny + 1)])
    R = 6371e3
    band = 2 * math.pi * R * R * (math.sin(math.radians(80)) - math.sin(math.radians(-80)))
    assert abs(Nx * az.sum() / band - 1) < 1e-12
    # dx at the equator-most centre vs R cos(phi) dlambda
    assert abs(b.metric("dxc", Ny // 2) - R * math.cos(math.radians(b.metric("phic", Ny // 2))) * 2 * math.pi / Nx) < 1e-6
    assert abs(b.metric("fcor", Ny + 1) - 2 * 7.292115e-5 * math.sin(math.radians(80))) < 1e-18


# ------------------------------------------------------------------------------------ operators
def _column_profiles(m):
    Nx, Ny, Nz = m.grid.size
    zc = np.array([m.grid.metric("zc", k) for k in range(1, Nz + 1)])
    T = np.broadcast_to(10 + 5e-3 * zc, (Nx, Ny, Nz)).copy()
    S = np.broadcast_to(35 - 1e-3 * zc, (Nx, Ny, Nz)).copy()
    return T, S


def test_state_of_rest_stays_at_rest():
    m = make_oracle(32, 24, 6, dt=600.0)
    T, S = _column_profiles(m)
    m.set(T=T, S=S)
    gb.first_time_step(m)
    gb.loop(m, 5)
    for name in ("u", "v", "w", "eta"):
        assert np.abs(m.fields()[name].interior).max() == 0.0, name
    assert np.abs(m.timestepper.Gn.u.interior).max() == 0.0
    # row j=1 of G.v sits on the southern wall: its pressure difference reaches into the (y,z) corner
    # halo that no boundary fill ever writes (as in the reference); v there is reset by the wall condition
    assert np.abs(m.timestepper.Gn.v.interior[:, 1:, :]).max() == 0.0
    assert np.array_equal(m.tracers.T.interior, T) and np.array_equal(m.tracers.S.interior, S)


def test_hydrostatic_pressure_matches_buoyancy_integral():
    m = make_oracle(16, 16, 8, dt=1.0)
    T, S = _column_profiles(m)
    m.set(T=T, S=S)
    gb.update_state(m)
    b = m.backend
    Nz = 8
    zc = [b.metric("zc", k) for k in range(1, Nz + 1)]
    dzf = [b.metric("dzf", k) for k in range(1, Nz + 2)]
    buoy = [-9.80665 * (b.teos10_rho(T[0, 0, k], S[0, 0, k], zc[k]) - 1020.0) / 1020.0 for k in range(Nz)]
    # halo cell above the surface: T,S copied, geopotential height mirrored (Oceananigans Z^ccc)
    b_top = -9.80665 * (b.teos10_rho(T[0, 0, -1], S[0, 0, -1], zc[-1] - dzf[Nz - 1]) - 1020.0) / 1020.0
    p = np.zeros(Nz)
    p[Nz - 1] = -0.5 * (buoy[Nz - 1] + b_top) * dzf[Nz]
    for k in range(Nz - 2, -1, -1):
        p[k] = p[k + 1] - 0.5 * (buoy[k] + buoy[k + 1]) * dzf[k + 1]
    got = m.pressure.pHY.interior
    assert np.allclose(got[3, 5, :], p, rtol=1e-13, atol=0)
    assert np.ptp(got, axis=(0, 1)).max() == 0.0      # horizontally uniform -> no pressure force


def test_constant_tracer_has_zero_tendency():
    m = make_oracle(32, 24, 6, dt=10.0)
    set_noisy_velocities(m, amplitude=0.1)
    Nx, Ny, Nz = m.grid.size
    m.set(T=np.full((Nx, Ny, Nz), 7.0), S=np.full((Nx, Ny, Nz), 35.0))
    gb.update_state(m)   # halos, w from continuity, tendencies
    w = np.abs(m.velocities.w.interior).max()
    assert w > 0
    # |G| is round-off of c * (sum of six fluxes)/V, with flux/V ~ u/dx ~ 1e-6 s^-1
    assert np.abs(m.timestepper.Gn.T.interior).max() < 1e-17 * 7 * 1e6
    assert np.abs(m.timestepper.Gn.S.interior).max() < 1e-17 * 35 * 1e6


def test_tracer_budget_closes_through_the_surface():
    m = make_oracle(32, 24, 6, dt=10.0)
    set_noisy_velocities(m, amplitude=0.1)
    Nx, Ny, Nz = m.grid.size
    rng = np.random.default_rng(1)
    m.set(T=10 + rng.random((Nx, Ny, Nz)), S=35 + 0 * rng.random((Nx, Ny, Nz)))
    gb.update_state(m)
    b = m.backend
    az = np.array([b.metric("azc", j) for j in range(1, Ny + 1)])
    dz = np.array([b.metric("dzc", k) for k in range(1, Nz + 1)])
    V = az[None, :, None] * dz[None, None, :]
    total = (V * m.timestepper.Gn.T.interior).sum()
    # only the (linear) free-surface face exchanges tracer: first-order upwind value at the top face
    wtop = m.velocities.w.interior[:, :, Nz]
    Tp = m.tracers.T.parent
    H = 8
    c_in, c_halo = Tp[H:-H, H:-H, H + Nz - 1], Tp[H:-H, H:-H, H + Nz]
    top_flux = (az[None, :] * wtop * np.where(wtop > 0, c_in, c_halo)).sum()
    scale = np.abs(V * m.timestepper.Gn.T.interior).sum()
    assert abs(total + top_flux) < 1e-12 * scale


def test_coriolis_sign_and_magnitude():
    m = make_oracle(64, 32, 4, dt=1.0)
    Nx, Ny, Nz = m.grid.size
    U0 = 0.1
    m.set(u=np.full((Nx, Ny, Nz), U0))
    gb.update_state(m)
    j = 3 * Ny // 4                     # northern mid-latitudes, v-face index (1-based j+1)
    f = m.backend.metric("fcor", j + 1)
    Gv = m.timestepper.Gn.v.interior[5, j, 1]
    assert f > 0 and Gv < 0
    assert abs(Gv / (-f * U0) - 1) < 0.05


def test_halo_fill_periodic_and_idempotent():
    m = make_oracle(16, 12, 4, dt=1.0)
    set_noisy_velocities(m)
    rng = np.random.default_rng(3)
    m.set(T=rng.random((16, 12, 4)), S=rng.random((16, 12, 4)), eta=rng.random((16, 12, 1)))
    m.backend.fill_halo_regions()
    first = {n: f.parent.copy() for n, f in m.prognostic_fields().items()}
    m.backend.fill_halo_regions()
    H, Nx, Ny = 8, 16, 12
    for n, f in m.prognostic_fields().items():
        p = f.parent
        assert np.array_equal(p, first[n]), n
        assert np.array_equal(p[:H], p[Nx:Nx + H]) and np.array_equal(p[Nx + H:], p[H:2 * H]), n
    T = first["T"]
    assert np.array_equal(T[:, H - 1, H:-H], T[:, H, H:-H])           # zero-gradient south layer
    assert np.array_equal(T[:, H:-H, H - 1][H:-H], T[:, H:-H, H][H:-H])  # bottom layer
    assert (T[H:-H, :H - 1, H:-H] == 0).all()                          # deeper y halos are never written
    v = first["v"]
    assert (v[:, H, :] == 0).all() and (v[:, H + Ny, :] == 0).all()    # wall-normal velocity


def test_barotropic_gravity_wave_phase_speed():
    # equatorial channel, no rotation, uniform density: a zonal standing wave eta = A cos(m lambda) cos(omega t)
    Nx, Ny, Nz = 64, 16, 4
    m = make_oracle(Nx, Ny, Nz, dt=200.0, lat_south=-8.0, lat_north=8.0, Omega=0.0)
    lam = (np.arange(Nx) + 0.5) * 2 * math.pi / Nx
    A, mode = 1e-3, 2
    m.set(eta=np.broadcast_to(A * np.cos(mode * lam)[:, None, None], (Nx, Ny, 1)).copy())
    gb.first_time_step(m)
    nsteps = 100
    gb.loop(m, nsteps - 1)
    t = m.clock.time
    assert abs(t - 200.0 * nsteps) < 1e-9
    R, g, Hd = 6371e3, 9.80665, 4000.0
    k = mode / R                       # wavenumber at the equator
    omega = math.sqrt(g * Hd) * k
    eta = m.free_surface.eta.interior[:, Ny // 2, 0]
    amp = 2 * (eta * np.cos(mode * lam)).mean() / A
    assert abs(amp - math.cos(omega * t)) < 0.03


def test_fp32_state_with_fp64_pressure_tracks_fp64_oracle():
    """Why the GPU pressure kernel runs its equation of state in fp64: rho(T,S,z) ~ 1e3 kg/m3 has an fp32 ulp of
    1.2e-4 kg/m3, which the hydrostatic integral turns into ~1e-3 of the pressure-gradient signal.  The all-fp32
    oracle build (f32) misses rtol = sqrt(eps(Float32)) on G.u after one step; with rho and the integral in fp64
    (build f32p64, state still fp32) the same field is >10x closer and every compared state field passes."""
    from helpers import SQRT_EPS32, assert_states_close
    cfg = dict(Nx=48, Ny=32, Nz=8, dt=600.0)
    models = {}
    for prec in ("f64", "f32", "f32p64"):
        m = models[prec] = make_oracle(precision=prec, **cfg)
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m)
        gb.first_time_step(m)
    rel = {}
    for prec in ("f32", "f32p64"):
        _, rep = gb.compare_states(models[prec], models["f64"], include_halos=True, verbose=False)
        rel[prec] = {r["name"]: r["rel"] for r in rep}
    assert rel["f32"]["Gn.u"] > SQRT_EPS32 and rel["f32"]["Gn.u"] < 1e-2
    assert rel["f32p64"]["Gn.u"] < 0.1 * rel["f32"]["Gn.u"] and rel["f32p64"]["Gn.u"] < SQRT_EPS32
    for m in models.values():
        gb.loop(m, 10)
    # (the fp32 oracle keeps Oceananigans' expanded smoothness indicators, whose cancellation noise alone costs
    #  ~4e-4 on G.S; the GPU kernels use the factored form and are held to sqrt(eps) on every field)
    assert_states_close(models["f32p64"], models["f64"], tendency_rtol=1e-3,
                        label="fp32 state + fp64 pressure vs fp64, 11 steps")


def test_top_flux_boundary_conditions():
    """compute_hydrostatic_boundary_tendency_contributions! (src/precompile.jl:52-61) with FluxBoundaryConditions at the
    top: a uniform upward heat flux J cools the top cell at J/dz and nothing else; a wind stress accelerates the top
    layer; the default (no flux) leaves the tendencies untouched."""
    m = make_oracle(32, 24, 6, 60.0)
    Nx, Ny, Nz = m.grid.size
    dz_top = m.backend.metric("dzc", Nz)
    gb.update_state(m)
    assert np.abs(m.timestepper.Gn.T.interior).max() == 0.0
    JT, taux = 2.5e-4, -1.0e-4                      # K m/s upward (cooling); m2/s2, negative upward = eastward push
    gb.set_top_flux(m, T=np.full((Nx, Ny), JT), u=np.full((Nx, Ny), taux))
    gb.update_state(m)
    GT, Gu = m.timestepper.Gn.T.interior, m.timestepper.Gn.u.interior
    assert np.allclose(GT[:, :, -1], -JT / dz_top, rtol=1e-14) and np.abs(GT[:, :, :-1]).max() == 0.0
    assert np.allclose(Gu[:, :, -1], -taux / dz_top, rtol=1e-14) and np.abs(Gu[:, :, :-1]).max() == 0.0
    assert np.abs(m.timestepper.Gn.S.interior).max() == 0.0
    gb.first_time_step(m)
    assert np.allclose(m.tracers.T.interior[:, :, -1], -60.0 * JT / dz_top, rtol=1e-12)      # Euler step from T = 0
    gb.set_top_flux(m, T=None, u=None)
    gb.update_state(m)
    assert np.abs(m.timestepper.Gn.T.interior[:, :, -1] - 0.0).max() < 1e-12    # only advection of the tiny state left


def test_substep_order_is_a_second_order_difference():
    """Option substep_order (the order of the two halves of a forward-backward substep changed between upstream
    releases; a Julia dump decides): 0 = eta from the old U, V, then U, V from the new eta; 1 = U, V first.  Both are the same
    forward-backward scheme started half a substep apart: the barotropic gravity wave keeps its speed, and the two answers differ
    by O(dtau) of the signal, not more."""
    import gb25_amd as gb
    from helpers import make_oracle, set_noisy_velocities
    out = []
    for order in (0, 1):
        m = make_oracle(48, 32, 6, 600.0)
        m.backend.set_option("substep_order", order)
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 1e-2)
        Nx, Ny, _ = m.grid.size
        x, y = np.meshgrid(np.arange(Nx), np.arange(Ny), indexing="ij")
        m.set(eta=0.1 * np.exp(-((x - 24.0) ** 2 + (y - 16.0) ** 2) / 18.0))
        gb.first_time_step(m)
        gb.loop(m, 3)
        out.append({n: m.backend.get_field(n, False).copy() for n in ("eta", "U", "V", "u")})
    for n in out[0]:
        a, b = out[0][n], out[1][n]
        d = np.linalg.norm(a - b) / np.linalg.norm(a)
        assert 1e-6 < d < 0.2, (n, d)            # different, but the same wave
