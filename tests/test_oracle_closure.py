"""Known-answer pins of the vertically implicit diffusion restatement (oracle): closure =
VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), κ, ν) of src/baroclinic_instability_model.jl:31 --
SURVEY.md section 8f.2, the "implicit vertical solve" half (batched tridiagonal per column).
 * the solve inverts the operator an independent numpy construction builds (I - dt d/dz K d/dz on the stretched grid);
 * column integrals are conserved (no flux through bottom and top), extrema do not grow, a uniform profile is a fixed
   point (to round-off), diffusion smooths;
 * closure = nothing is the default and is bitwise untouched;
 * immersed columns: only the free levels are solved, solid cells stay zero."""
import numpy as np
import pytest

import gb25_amd as gb
from helpers import make_oracle, set_noisy_velocities


def operator(dzc, dzf, K, dt, kfirst=0):
    """(I - dt d/dz K d/dz) on levels kfirst.. (0-based) of a column: dzc[k] cell thickness, dzf[k] centre spacing k-1..k."""
    n = len(dzc)
    A = np.eye(n)
    for k in range(kfirst, n):
        lo = 0.0 if k == kfirst else -dt * K / (dzc[k] * dzf[k])
        up = 0.0 if k == n - 1 else -dt * K / (dzc[k] * dzf[k + 1])
        A[k, k] = 1 - lo - up
        if k > kfirst:
            A[k, k - 1] = lo
        if k < n - 1:
            A[k, k + 1] = up
    return A


def grid_spacings(m, Nz):
    dzc = np.array([m.backend.metric("dzc", k) for k in range(1, Nz + 1)])
    dzf = np.array([m.backend.metric("dzf", k) for k in range(1, Nz + 2)])
    return dzc, dzf


def test_the_solve_inverts_the_diffusion_operator():
    Nx, Ny, Nz, dt = 16, 12, 20, 1800.0
    K = 10.0                                             # large on purpose: dt K / dz^2 ~ 20 near the surface
    m = make_oracle(Nx, Ny, Nz, dt, closure=gb.VerticalScalarDiffusivity(nu=0.0, kappa=K))
    rng = np.random.default_rng(0)
    T0 = 10 + rng.standard_normal((Nx, Ny, Nz))
    m.set(T=T0, S=np.full((Nx, Ny, Nz), 35.0))
    m.backend.ab2_step(dt, True)                         # tendencies are zero: the step is the implicit solve alone
    T1 = m.tracers.T.interior
    dzc, dzf = grid_spacings(m, Nz)
    A = operator(dzc, dzf, K, dt)
    assert np.abs(np.einsum("kl,ijl->ijk", A, T1) - T0).max() < 1e-12
    assert np.abs((T1 * dzc).sum(-1) - (T0 * dzc).sum(-1)).max() < 1e-10      # column integral
    assert T1.max() <= T0.max() and T1.min() >= T0.min()                      # M-matrix: no new extrema
    assert np.abs(np.diff(T1[..., -6:], axis=-1)).mean() < 0.5 * np.abs(np.diff(T0[..., -6:], axis=-1)).mean()
    assert np.abs(m.tracers.S.interior - 35.0).max() < 1e-12                 # a uniform profile is a fixed point


def test_viscosity_acts_on_u_and_v_and_leaves_the_wall_face_alone():
    Nx, Ny, Nz, dt = 16, 12, 10, 600.0
    m = make_oracle(Nx, Ny, Nz, dt, closure=gb.VerticalScalarDiffusivity(nu=1e-2, kappa=0.0))
    rng = np.random.default_rng(1)
    u0, v0 = rng.standard_normal((Nx, Ny, Nz)), rng.standard_normal((Nx, Ny + 1, Nz))
    v0[:, 0] = v0[:, -1] = 0
    T0 = rng.standard_normal((Nx, Ny, Nz))
    m.set(u=u0, v=v0, T=T0)
    GU0 = m.backend.get_field("U", False).copy()
    m.backend.ab2_step(dt, True)
    dzc, dzf = grid_spacings(m, Nz)
    A = operator(dzc, dzf, 1e-2, dt)
    assert np.abs(np.einsum("kl,ijl->ijk", A, m.velocities.u.interior) - u0).max() < 1e-12
    assert np.abs(np.einsum("kl,ijl->ijk", A, m.velocities.v.interior[:, 1:-1]) - v0[:, 1:-1]).max() < 1e-12
    assert np.array_equal(m.tracers.T.interior, T0)                            # kappa = 0: tracers untouched
    assert np.all(m.velocities.v.interior[:, 0] == 0)


def test_no_closure_is_the_default_and_bitwise_the_same():
    a = make_oracle(32, 20, 8, 600.0)
    b = make_oracle(32, 20, 8, 600.0, closure=gb.VerticalScalarDiffusivity(nu=0.0, kappa=0.0))
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 1e-2)
        gb.first_time_step(m)
        gb.loop(m, 3)
    for n in ("u", "v", "T", "S", "eta"):
        assert np.array_equal(a.backend.get_field(n, True), b.backend.get_field(n, True)), n


def test_diffusion_changes_the_run_and_damps_the_shear():
    out = []
    for closure in (None, gb.VerticalScalarDiffusivity(nu=50.0, kappa=1e-3)):
        m = make_oracle(32, 20, 8, 600.0, closure=closure)
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 1e-2)
        gb.first_time_step(m)
        gb.loop(m, 5)
        out.append(m.velocities.u.interior.copy())
    shear = [np.abs(np.diff(u, axis=-1)).mean() for u in out]
    assert shear[1] < 0.8 * shear[0]


def test_immersed_columns_solve_their_free_levels_only():
    Nx, Ny, Nz, dt = 16, 12, 10, 600.0
    K = 1e-2
    m = make_oracle(Nx, Ny, Nz, dt, closure=gb.VerticalScalarDiffusivity(nu=K, kappa=K))
    zf = np.array([m.backend.metric("zf", k) for k in range(1, Nz + 2)])
    level = np.zeros((Nx, Ny), int)
    level[4:8, 3:7] = 4
    level[10, 5] = Nz
    m.backend.set_bottom_height(np.where(level > 0, zf[level] - 1e-3, -5000.0))
    rng = np.random.default_rng(2)
    T0 = np.where(np.arange(Nz)[None, None, :] >= level[:, :, None], 10 + rng.standard_normal((Nx, Ny, Nz)), 0.0)
    u0 = rng.standard_normal((Nx, Ny, Nz))
    m.set(T=T0, u=u0)
    u0 = m.velocities.u.interior.copy()                   # masked by set!
    m.backend.ab2_step(dt, True)
    T1, u1 = m.tracers.T.interior, m.velocities.u.interior
    dzc, dzf = grid_spacings(m, Nz)
    for (i, j) in ((5, 4), (0, 0), (10, 5)):
        A = operator(dzc, dzf, K, dt, kfirst=level[i, j])
        assert np.abs(A @ T1[i, j] - T0[i, j]).max() < 1e-12, (i, j)
        assert np.all(T1[i, j, :level[i, j]] == 0)
    # a u face next to the raised block is free from the higher of its two columns on
    ku = max(level[3, 4], level[4, 4])
    A = operator(dzc, dzf, K, dt, kfirst=ku)
    assert np.abs(A @ u1[4, 4] - u0[4, 4]).max() < 1e-12 and np.all(u1[4, 4, :ku] == 0)
