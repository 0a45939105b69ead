"""Known-answer pins of the CATKE restatement (oracle): closure = CATKEVerticalDiffusivity() of
src/baroclinic_instability_model.jl:30 / sharding/less_simple_sharding_problem.jl:84-93 -- SURVEY.md section 8f.2.
[UPSTREAM-UNVERIFIED: formulas and calibrated constants after Wagner et al. (2025); no Oceananigans here to compare with.]
 * the diffusivities, the TKE step inside compute_diffusivities! and the filtered surface buoyancy flux against an independent
   numpy statement of the formulas (tests/catke_spec.py); alpha and beta against finite differences of the polynomial;
 * with no TKE, no shear and stable stratification nothing mixes; TKE decays at the dissipation rate;
 * wind stress: surface TKE flux -> TKE, diffusivities, a deepening mixed layer; tracer column integrals conserved;
 * surface cooling: convective branch (J^b > 0, N^2 < 0) switches on;
 * closure = nothing is untouched."""
import numpy as np
import pytest

import gb25_amd as gb
from helpers import make_oracle, set_noisy_velocities

import catke_spec as spec


def catke_model(Nx=16, Ny=12, Nz=16, dt=60.0, **kw):
    return make_oracle(Nx, Ny, Nz, dt, closure=gb.CATKEVerticalDiffusivity(), **kw)


def stratified(m, N2=1e-5):
    """T linear in z so that the buoyancy frequency is roughly N2 (alpha ~ 2e-4/K), S uniform."""
    Nx, Ny, Nz = m.grid.size
    zc = np.array([m.backend.metric("zc", k) for k in range(1, Nz + 1)])
    T = 20.0 + (N2 / (9.80665 * 2e-4)) * zc
    m.set(T=np.broadcast_to(T, (Nx, Ny, Nz)).copy(), S=np.full((Nx, Ny, Nz), 35.0))
    return zc


def developed(Nx=12, Ny=10, Nz=14, dt=120.0, steps=4):
    """a model a few steps into a wind- and cooling-driven run: every term of the TKE equation is alive"""
    m = catke_model(Nx, Ny, Nz, dt=dt, depth=150.0)
    zc = stratified(m, 2e-5)
    rng = np.random.default_rng(0)
    T = m.tracers.T.interior.copy()
    T[:, :, -3:] = T[:, :, -4:-3] - 0.05 * (1 + np.arange(3))   # a statically unstable skin under the cooled surface: N^2 < 0 on the top faces
    m.set(e=1e-4 * rng.random((Nx, Ny, Nz)) + 1e-6, u=0.05 * rng.standard_normal((Nx, Ny, Nz)), T=T,
          S=np.broadcast_to(35.0 - 2e-3 * zc, (Nx, Ny, Nz)).copy())
    gb.set_top_flux(m, u=np.full((Nx, Ny), -1e-4), T=1e-5 + 4e-5 * rng.random((Nx, Ny)))
    gb.first_time_step(m)
    gb.loop(m, steps)
    return m


def test_sensitivities_are_the_derivatives_of_the_polynomial():
    m = catke_model()
    for (T, S, Z) in ((10.0, 35.0, -1000.0), (25.0, 33.0, -5.0), (2.0, 36.5, -3500.0)):
        a, b = m.backend.teos10_sensitivities(T, S, Z)
        h = 1e-4
        fa = -(m.backend.teos10_rho(T + h, S, Z) - m.backend.teos10_rho(T - h, S, Z)) / (2 * h)
        fb = (m.backend.teos10_rho(T, S + h, Z) - m.backend.teos10_rho(T, S - h, Z)) / (2 * h)
        assert a == pytest.approx(fa, rel=1e-6) and b == pytest.approx(fb, rel=1e-6)
    a, b = m.backend.teos10_sensitivities(10.0, 35.0, -1000.0)
    assert 1.5e-4 < a / 1020.0 < 2.0e-4 and 7.0e-4 < b / 1020.0 < 8.0e-4      # thermal expansion / haline contraction of sea water


def test_diffusivities_match_an_independent_statement_of_the_formulas():
    m = developed()
    st = spec.State(m)                     # kappa of the last compute_diffusivities! were made from exactly this e, J^b, u, v, T, S
    F = spec.face_quantities(st)
    for name, key in (("ku", "kappa_u"), ("kc", "kappa_c"), ("ke", "kappa_e")):
        got = getattr(m.diffusivity_fields, key).interior
        assert np.allclose(got, F[name], rtol=2e-6, atol=1e-14), (name, np.abs(got - F[name]).max())
        assert np.all(got[:, :, 0] == 0) and np.all(got[:, :, -1] == 0) and np.all(got >= 0) and got.max() > 1e-4
    Jb = m.diffusivity_fields.Jb.interior.reshape(F["N2"].shape[:2])
    assert Jb.min() > 1e-9                  # the cooling has been felt
    conv = (F["N2"] < 0) & (Jb[:, :, None] > 1e-11)
    assert conv.any()                       # ... and the convective branch is exercised by this state


def test_the_tke_step_matches_an_independent_statement():
    """time_step_catke_equation!: the e step of the NEXT update_state! from the state as it is (old kappa_u, kappa_c, J^b,
    previous velocities), redone in numpy: shear production between the previous and the current velocities, buoyancy flux
    split into its explicit and implicit parts, dissipation and the bottom flux on the diagonal, AB2 with chi = 0.1, one
    tridiagonal solve with the new kappa_e."""
    m = developed()
    st = spec.State(m)
    assert np.abs(st.um - st.u).max() == 0          # compute_diffusivities! left u- = u behind ...
    dt = 120.0
    # ... so move the velocities on, as a time step would between two computes
    rng = np.random.default_rng(5)
    Nx, Ny, Nz = m.grid.size
    m.set(u=m.velocities.u.interior + 0.01 * rng.standard_normal((Nx, Ny, Nz)))
    m.backend.fill_halo_regions()
    st = spec.State(m)
    want = spec.tke_step(st, dt)
    gb.update_state(m)
    e = m.tracers.e.interior
    assert np.allclose(e, want["e"], rtol=3e-6, atol=1e-13), np.abs(e - want["e"]).max()
    assert np.allclose(m.backend.get_field("Gm.e", False), want["Gm"], rtol=3e-6, atol=1e-16)
    assert np.allclose(m.diffusivity_fields.Le.interior, want["Le"], rtol=3e-6, atol=1e-14)
    P = spec.shear_production(st)
    assert P.max() > 1e-9 and np.abs(want["e"] - st.cells(st.e)).max() > 1e-7     # the test moved something
    # G^n.e holds the SLOW tendency only: advection and the surface TKE flux (positive in the top cell, wind + cooling)
    Gn = m.backend.get_field("Gn.e", False)
    assert Gn[:, :, -1].min() > 0 and np.abs(Gn[:, :, 2:-2]).max() < 1e-2 * Gn[:, :, -1].max()


def test_the_surface_buoyancy_flux_is_filtered_over_the_convective_time_scale():
    m = developed(steps=2)
    Jb_old = np.array(m.backend.get_field("Jb", True), dtype=np.float64)
    gb.time_step(m)
    st = spec.State(m)
    st.Jb = Jb_old
    Nx, Ny, Nz = m.grid.size
    T, S = st.cells(st.T)[:, :, -1], st.cells(st.S)[:, :, -1]
    al, _ = spec.sensitivities(st, T, S, st.zc[-1])
    JT = m.backend.get_top_flux("T")
    want = spec.filtered_surface_flux(st, st.cells(st.e), spec.G * al * JT, 120.0)
    got = m.diffusivity_fields.Jb.interior.reshape(want.shape)
    assert np.allclose(got, want, rtol=1e-6), np.abs(got / want - 1).max()
    inst = spec.G * al * JT
    assert np.all(got < inst) and np.all(got > 0.02 * inst)      # still catching up with the instantaneous flux


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
    assert ku[:, :, Nz - 1].min() > 5e-4 and ku[:, :, 2].max() < 5e-7     # (at depth: the floor sqrt(e_min) N^-1 of the stable length)
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
