"""closure = VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), κ, ν) (src/baroclinic_instability_model.jl:31)
on the HIP path -- SURVEY.md section 8f.2, the implicit vertical solve: `k_implicit_vertical` (one tridiagonal solve per
column and field after the AB2 update, the column in LDS) against the oracle and against the operator it inverts."""
import numpy as np
import pytest

import gb25_amd as gb
from helpers import SQRT_EPS32, assert_states_close, counter_rng, make_pair, set_noisy_velocities
from test_oracle_closure import grid_spacings, operator

pytestmark = pytest.mark.gpu
VSD = gb.VerticalScalarDiffusivity


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = max(np.linalg.norm(a.ravel()), np.linalg.norm(b.ravel()))
    return 0.0 if n == 0 else float(np.linalg.norm((a - b).ravel()) / n)


@pytest.mark.parametrize("Nz", [20, 48, 100, 130])     # register-resident columns (<= 32, 48, 128 levels) and the LDS route
@pytest.mark.parametrize("float_type,tol", [("Float64", 1e-12), ("Float32", 2e-6)])
def test_the_kernel_inverts_the_diffusion_operator(float_type, tol, Nz):
    Nx, Ny, dt, K = 70, 24, 1800.0, 10.0
    dtype = np.float64 if float_type == "Float64" else np.float32
    m = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=dt, closure=VSD(nu=K, kappa=K))
    assert m.backend.vertical_diffusivity() == (K, K)
    rng = np.random.default_rng(0)
    T0 = (10 + rng.standard_normal((Nx, Ny, Nz))).astype(dtype)
    u0 = rng.standard_normal((Nx, Ny, Nz)).astype(dtype)
    v0 = rng.standard_normal((Nx, Ny + 1, Nz)).astype(dtype)
    v0[:, 0] = v0[:, -1] = 0
    m.set(T=T0, S=np.full((Nx, Ny, Nz), 35.0, dtype), u=u0, v=v0)
    m.backend.ab2_step(dt, True)                       # zero tendencies: the step is the implicit solve alone
    dzc, dzf = grid_spacings(m, Nz)
    A = operator(dzc, dzf, K, dt)
    for new, old in ((m.tracers.T.interior, T0), (m.velocities.u.interior, u0), (m.velocities.v.interior[:, 1:-1], v0[:, 1:-1])):
        back = np.einsum("kl,ijl->ijk", A, new.astype(np.float64))
        # backward-stable solve: the residual is a few eps |A| |x| (the thin surface cells of deep grids make
        # dt K / dz^2, hence |A|, large)
        bound = 8 * np.finfo(dtype).eps * np.abs(A).sum(1).max() * np.abs(new).max()
        assert np.abs(back - old).max() < bound, (np.abs(back - old).max(), bound)
    # a uniform profile is a fixed point up to the forward error of the solve, cond(A) eps with cond(A) <= |A| (the
    # inverse of this M-matrix has unit row sums)
    assert np.abs(m.tracers.S.interior - 35.0).max() < 35 * np.finfo(dtype).eps * np.abs(A).sum(1).max()
    assert np.all(m.velocities.v.interior[:, 0] == 0) and np.all(m.velocities.v.interior[:, -1] == 0)
    T1 = m.tracers.T.interior.astype(np.float64)
    assert np.abs((T1 * dzc).sum(-1) - (T0.astype(np.float64) * dzc).sum(-1)).max() < tol * 4000 * 12


@pytest.mark.parametrize("grid_type", ["simple_lat_lon", "gaussian_islands_lat_lon", "gaussian_islands"])
def test_stepping_with_the_closure_matches_the_oracle(grid_type):
    Nx, Ny, Nz, dt = (72, 36, 12, 600.0) if grid_type == "gaussian_islands" else (90, 44, 12, 600.0)
    closure = VSD(nu=5.0, kappa=0.5)
    r, v = make_pair(Nx, Ny, Nz, dt=dt, grid_type=grid_type, closure=closure)
    gb.set_baroclinic_instability(v)
    set_noisy_velocities(v, 1e-2)
    for n in ("u", "v", "T", "S"):
        a = v.backend.get_field(n, True).astype(np.float32)
        r.backend.set_field(n, a, True)
        v.backend.set_field(n, a.astype(np.float64), True)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 5)
    assert_states_close(r, v, label=f"{grid_type} with VerticalScalarDiffusivity after 6 steps")
    # and the closure did something: the same run without it differs
    r0 = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, grid_type=grid_type)
    gb.set_baroclinic_instability(r0)
    set_noisy_velocities(r0, 1e-2)
    gb.first_time_step(r0)
    gb.loop(r0, 5)
    assert rel(r0.velocities.u.interior, r.velocities.u.interior) > 1e-3


def test_schedules_and_slabs_with_the_closure():
    """The implicit solve rides behind every route of the AB2 update (adopted look-ahead buffers or the stand-alone
    kernels), and inside a decomposed step: slabs bit for bit."""
    from gb25_amd.distributed import LocalSlabEnsemble
    Nx, Ny, Nz, dt = 192, 44, 12, 600.0
    nu, kappa = 5.0, 0.5
    outs = []
    init = None
    for opts in (dict(), dict(ab2_lookahead=0), dict(two_streams=0)):
        m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, closure=VSD(nu, kappa), options=opts)
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 1e-2)
        if init is None:
            init = {n: m.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta")}
        gb.first_time_step(m)
        gb.loop(m, 5)
        outs.append({n: m.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta", "U", "V")})
        m.backend.close()
    for o in outs[1:]:
        for n, a in outs[0].items():
            assert rel(a, o[n]) < 2e-6, (n, rel(a, o[n]))
    ens = LocalSlabEnsemble(Nx, Ny, Nz, 3, dt=dt)
    for b in ens.backends:
        b.set_vertical_diffusivity(nu, kappa)
    for n, a in init.items():
        ens.scatter(n, a)
    ens.first_time_step()
    ens.loop(5)
    for n, a in outs[0].items():
        assert np.array_equal(ens.gather(n), a), n
    ens.close()
