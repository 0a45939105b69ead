"""Known-answer pins of the CATKE restatement (oracle): closure = CATKEVerticalDiffusivity() of
src/baroclinic_instability_model.jl:30 / sharding/less_simple_sharding_problem.jl:84-93 -- SURVEY.md section 8f.2.
[UPSTREAM-UNVERIFIED: formulas and calibrated constants after Wagner et al. (2025); no Oceananigans here to compare with.]
 * the diffusivities against an independent numpy statement of the mixing-length formulas;
 * with no TKE, no shear and stable stratification nothing mixes; TKE decays at the dissipation rate;
 * wind stress: surface TKE flux -> TKE, diffusivities, a deepening mixed layer; tracer column integrals conserved;
 * surface cooling: convective branch (J^b > 0, N^2 < 0) switches on;
 * closure = nothing is untouched."""
import numpy as np
import pytest

import gb25_amd as gb
from helpers import make_oracle, set_noisy_velocities

P = dict(Cs=1.131, Cb=0.28, Csp=0.505, CRid=1.02, CRi0=0.254,
         Chi=(0.242, 0.098, 0.548, 0.579), Clo=(0.361, 0.198, 7.863, 1.604), Cun=(0.370, 0.369, 1.447, 0.923),
         Cc=(3.705, 4.793, 3.642, 3.254), Ce=(0.0, 0.112, 0.0, 0.0), Jbmin=1e-11)


def sigma(p, Ri):
    if Ri < 0:
        return P["Cun"][p]
    return P["Clo"][p] + (P["Chi"][p] - P["Clo"][p]) * min(1.0, max(0.0, (Ri - P["CRi0"]) / P["CRid"]))


def catke_model(Nx=16, Ny=12, Nz=16, dt=60.0, **kw):
    return make_oracle(Nx, Ny, Nz, dt, closure=gb.CATKEVerticalDiffusivity(), **kw)


def stratified(m, N2=1e-5):
    """T linear in z so that the buoyancy frequency is roughly N2 (alpha ~ 2e-4/K), S uniform."""
    Nx, Ny, Nz = m.grid.size
    zc = np.array([m.backend.metric("zc", k) for k in range(1, Nz + 1)])
    T = 20.0 + (N2 / (9.80665 * 2e-4)) * zc
    m.set(T=np.broadcast_to(T, (Nx, Ny, Nz)).copy(), S=np.full((Nx, Ny, Nz), 35.0))
    return zc


def test_diffusivities_match_an_independent_statement_of_the_formulas():
    Nx, Ny, Nz = 16, 12, 16
    m = catke_model(Nx, Ny, Nz)
    zc = stratified(m, 2e-5)
    rng = np.random.default_rng(0)
    e = 1e-4 * rng.random((Nx, Ny, Nz)) + 1e-6
    u = 0.05 * rng.standard_normal((Nx, Ny, Nz))
    m.set(e=e, u=u)
    gb.update_state(m)
    zf = np.array([m.backend.metric("zf", k) for k in range(1, Nz + 2)])
    dzf = np.array([m.backend.metric("dzf", k) for k in range(1, Nz + 2)])
    ku, kc, ke = (getattr(m.diffusivity_fields, n).interior for n in ("kappa_u", "kappa_c", "kappa_e"))
    T = m.tracers.T.parent
    from oracle_backend import oracle_lib
    H = 8
    up = m.velocities.u.parent
    checked = 0
    for (i, j) in ((3, 4), (10, 7), (0, 0)):
        for k in range(1, Nz):                      # interior faces (0-based face k between cells k-1 and k)
            # buoyancy through the oracle's own equation of state (pinned separately by the TEOS-10 check value)
            rho = lambda kk: m.backend.teos10_rho(T[H + i, H + j, H + kk], 35.0, zc[kk])
            b = lambda kk: -9.80665 * (rho(kk) - 1020.0) / 1020.0
            N2 = (b(k) - b(k - 1)) / dzf[k]
            duw = (up[H + i, H + j, H + k] - up[H + i, H + j, H + k - 1]) / dzf[k]
            due = (up[H + i + 1, H + j, H + k] - up[H + i + 1, H + j, H + k - 1]) / dzf[k]
            S2 = 0.5 * (duw ** 2 + due ** 2)
            ef = 0.5 * (e[i, j, k - 1] + e[i, j, k])
            ws = np.sqrt(max(ef, 0.0))
            Ri = 0.0 if N2 == 0 else N2 / S2
            ls = min(P["Cs"] * (zf[Nz] - zf[k]), P["Cb"] * (zf[k] - zf[0]))
            if N2 > 0:
                ls = min(ls, ws / np.sqrt(N2))
            for p, arr in ((0, ku), (1, kc), (2, ke)):
                assert arr[i, j, k] == pytest.approx(sigma(p, Ri) * ls * ws, rel=1e-9), (i, j, k, p)
            checked += 1
    assert checked == 3 * (Nz - 1)
    for arr in (ku, kc, ke):
        assert np.all(arr[:, :, 0] == 0) and np.all(arr[:, :, Nz] == 0) and np.all(arr >= 0)
    assert np.all(m.diffusivity_fields.Le.interior < 0) and np.all(m.diffusivity_fields.Jb.interior == 0)


def test_quiescent_stratified_fluid_does_not_mix_and_tke_decays():
    m = catke_model()
    stratified(m)
    Nx, Ny, Nz = m.grid.size
    T0 = m.tracers.T.interior.copy()
    m.set(e=np.full((Nx, Ny, Nz), 1e-4))
    gb.first_time_step(m)
    gb.loop(m, 20)
    e = m.tracers.e.interior
    assert e.max() < 2e-5 and e.min() > -1e-12                    # dissipation, no source
    assert np.abs(m.tracers.T.interior - T0).max() < 2e-3        # a little mixing while the TKE lasted, then none
    assert np.abs(m.velocities.u.interior).max() < 1e-6


def test_wind_stress_deepens_a_mixed_layer_and_conserves_heat():
    m = catke_model(Nz=24, dt=120.0, depth=200.0)                # 200 m deep: 5-6 m cells under the surface
    Nx, Ny, Nz = m.grid.size
    zc = stratified(m, 1e-5)
    T0 = m.tracers.T.interior.copy()
    gb.set_top_flux(m, u=np.full((Nx, Ny), -1e-4))               # tau_x / rho0 = 1e-4 m2/s2 into the ocean (0.1 N/m2)
    gb.first_time_step(m)
    gb.loop(m, 120)
    e, T = m.tracers.e.interior, m.tracers.T.interior
    ku = m.diffusivity_fields.kappa_u.interior
    assert e[:, :, -1].min() > 1e-5 and e[:, :, 0].max() < 1e-12  # TKE near the surface, none at depth
    assert e[:, :, -3].min() > 1e-6                               # ... and it has worked its way down three cells
    assert ku[:, :, Nz - 1].min() > 5e-4 and ku[:, :, 2].max() < 1e-9
    # the top levels have been stirred: the stratification there is weaker than it was
    top = slice(Nz - 2, Nz)
    assert np.abs(np.diff(T[:, :, top], axis=-1)).mean() < 0.75 * np.abs(np.diff(T0[:, :, top], axis=-1)).mean()
    dz = np.array([m.backend.metric("dzc", k) for k in range(1, Nz + 1)])
    assert np.abs(((T - T0) * dz).sum(-1)).max() < 1e-6 * np.abs((T0 * dz).sum(-1)).max()   # mixing moves heat, makes none
    assert np.abs(m.velocities.u.interior[:, :, -1]).mean() > 1e-2                           # and the wind drives a current


def test_surface_cooling_switches_the_convective_length_on():
    m = catke_model(Nz=24, dt=120.0, depth=200.0)
    Nx, Ny, Nz = m.grid.size
    stratified(m, 5e-6)
    gb.set_top_flux(m, T=np.full((Nx, Ny), 1e-4))                # upward heat flux: cooling
    gb.first_time_step(m)
    gb.loop(m, 40)
    Jb = m.diffusivity_fields.Jb.interior
    assert 1e-7 < Jb.min() and Jb.max() < 5e-7                    # g alpha J^T ~ 2.5e-7
    kc = m.diffusivity_fields.kappa_c.interior
    assert kc[:, :, Nz - 1].min() > 1e-4                          # convective mixing under the cooled surface
    assert m.tracers.e.interior[:, :, -1].min() > 1e-6


def test_no_closure_is_untouched():
    a = make_oracle(32, 20, 8, 600.0)
    b = make_oracle(32, 20, 8, 600.0)
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 1e-2)
        gb.first_time_step(m)
        gb.loop(m, 2)
    assert np.array_equal(a.velocities.u.interior, b.velocities.u.interior)
    assert np.all(a.backend.get_field("e", False) == 0) and np.all(a.backend.get_field("kappa_u", False) == 0)
