"""The data-free forcing of the oracle (GB-25 src/data_free_ocean_climate_model.jl:12-70): the
similarity-theory flux solve restated in oracle/gb25_oracle.c ("data-free forcing") against an independent numpy statement
of the same published formulas (COARE 3.5: Edson et al. 2013), known answers (drag coefficient at 10 m/s, signs, land), and
the coupled model stepping.  ClimaOcean is not in /root/reference: [UPSTREAM-UNVERIFIED], parity unpinned for this row."""
import ctypes as C

import numpy as np
import pytest

import gb25_amd as gb
from oracle_backend import CPU


def point(b, **kw):
    d = dict(ua=8.0, va=0.0, Ta=288.15, qa=0.008, pa=101325.0, Qsw=0.0, Qlw=0.0, uo=0.0, vo=0.0, To=16.0, So=35.0,
             g=9.80665, rho0=1020.0)
    d.update(kw)
    f = b._fn("similarity_fluxes_point")
    f.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
    f.restype = None
    a = (C.c_double * 13)(*[d[k] for k in ("ua", "va", "Ta", "qa", "pa", "Qsw", "Qlw", "uo", "vo", "To", "So", "g", "rho0")])
    out = (C.c_double * 4)()
    f(a, int(kw.get("iterations", 5)) if "iterations" in kw else 5, out)
    return np.array(out[:]), d


@pytest.fixture(scope="module")
def backend():
    m = gb.baroclinic_instability_model(CPU("f64"), 16, 8, 4, dt=60.0)
    return m.backend


def psi_u(z):
    if z < 0:
        x = (1 - 15 * z) ** 0.25
        pk = 2 * np.log((1 + x) / 2) + np.log((1 + x * x) / 2) - 2 * np.arctan(x) + np.pi / 2
        y = np.cbrt(1 - 10.15 * z)
        pc = 1.5 * np.log((1 + y + y * y) / 3) - np.sqrt(3) * np.arctan((1 + 2 * y) / np.sqrt(3)) + np.pi / np.sqrt(3)
        f = z * z / (1 + z * z)
        return (1 - f) * pk + f * pc
    return -(0.7 * z + 0.75 * (z - 5 / 0.35) * np.exp(-min(50, 0.35 * z)) + 0.75 * 5 / 0.35)


def psi_q(z):
    if z < 0:
        x = (1 - 15 * z) ** 0.5
        pk = 2 * np.log((1 + x) / 2)
        y = np.cbrt(1 - 34.15 * z)
        pc = 1.5 * np.log((1 + y + y * y) / 3) - np.sqrt(3) * np.arctan((1 + 2 * y) / np.sqrt(3)) + np.pi / np.sqrt(3)
        f = z * z / (1 + z * z)
        return (1 - f) * pk + f * pc
    return -((1 + 2 / 3 * z) ** 1.5 + 2 / 3 * (z - 14.28) * np.exp(-min(50, 0.35 * z)) + 8.525)


def fluxes_numpy(d, iterations=5):
    """Independent statement: the bulk algorithm written from the formulas of the oracle's header comment."""
    k, h, g = 0.4, 10.0, d["g"]
    Rd, Rv, cpd, cpv, cpl, Lv0, T0 = 287.0, 461.5, 1005.0, 1859.0, 4181.0, 2.5008e6, 273.16
    Ts = d["To"] + 273.15
    psat = 611.657 * (Ts / T0) ** ((cpv - cpl) / Rv) * np.exp((Lv0 - (cpv - cpl) * T0) / Rv * (1 / T0 - 1 / Ts))
    qs = 0.98 * (Rd / Rv) * psat / (d["pa"] - (1 - Rd / Rv) * psat)
    qa, Ta = d["qa"], d["Ta"]
    rho = d["pa"] / ((Rd * (1 - qa) + Rv * qa) * Ta)
    cpm = cpd * (1 - qa) + cpv * qa
    Lv = Lv0 + (cpv - cpl) * (Ts - T0)
    du, dv = d["ua"] - d["uo"], d["va"] - d["vo"]
    dth, dq = Ta + g / cpm * h - Ts, qa - qs
    Tv = Ta * (1 + 0.608 * qa)
    U = np.sqrt(du ** 2 + dv ** 2 + 0.04)
    us, ths, qst = (k * x / np.log(h / 1e-4) for x in (U, dth, dq))
    for _ in range(iterations):
        bs = g / Tv * (ths * (1 + 0.608 * qa) + 0.608 * Ta * qst)
        Ug = max(0.2, 1.2 * np.cbrt(max(-us * bs, 0.0) * 600.0))
        U = np.sqrt(du ** 2 + dv ** 2 + Ug ** 2)
        lu = 0.011 * us ** 2 / g + 0.11 * 1.5e-5 / us
        lq = min(1.6e-4, 5.8e-5 / (lu * us / 1.5e-5) ** 0.72)
        z = float(np.clip(k * h * bs / us ** 2, -50, 50))
        us = k * U / (np.log(h / lu) - psi_u(z) + psi_u(z * lu / h))
        cq = np.log(h / lq) - psi_q(z) + psi_q(z * lq / h)
        ths, qst = k * dth / cq, k * dq / cq
    Q = (-rho * cpm * us * ths - rho * Lv * us * qst + 0.97 * (5.670374419e-8 * Ts ** 4 - d["Qlw"]) - 0.95 * d["Qsw"])
    return np.array([rho * us ** 2 * du / U, rho * us ** 2 * dv / U, Q / (d["rho0"] * 3991.86795711963),
                     -d["So"] * (-rho * us * qst) / 1000.0]), rho, U


@pytest.mark.parametrize("kw", [dict(), dict(ua=-3.0, va=4.0, Ta=280.0, To=18.0), dict(ua=1.0, Ta=300.0, To=5.0, qa=0.0),
                                dict(ua=0.0, va=0.0, Ta=283.0, To=25.0), dict(ua=20.0, uo=1.0, vo=-0.5, Qsw=300.0, Qlw=350.0)])
def test_the_c_restatement_is_the_formulas(backend, kw):
    got, d = point(backend, **kw)
    want, _, _ = fluxes_numpy(d)
    assert np.allclose(got, want, rtol=1e-10, atol=1e-18), (got, want)


def test_known_answers(backend):
    # the neutral 10 m drag coefficient of COARE 3.5 at 10 m/s is 1.2e-3 .. 1.4e-3 (Edson et al. 2013, fig. 6)
    got, d = point(backend, ua=10.0, Ta=289.05, To=16.0, qa=0.0111)      # air-sea differences near zero: near neutral
    _, rho, U = fluxes_numpy(d)
    cd = got[0] / (rho * 10.0 * U)
    assert 1.1e-3 < cd < 1.5e-3, cd
    # signs (fluxes positive upward, the stress as it acts on the ocean)
    got, _ = point(backend, ua=8.0, Ta=278.0, To=20.0, qa=0.002)
    assert got[0] > 0 and got[1] == 0 and got[2] > 0 and got[3] < 0       # eastward stress, ocean loses heat, evaporation salts
    hot, _ = point(backend, ua=8.0, Ta=278.0, To=20.0, qa=0.002, Qsw=800.0)
    assert hot[2] < got[2]                                                # downwelling shortwave heats
    calm, _ = point(backend, ua=0.0, Ta=278.0, To=20.0, qa=0.002)
    assert calm[0] == 0 and calm[2] > 0                                   # free convection: gustiness keeps the exchange going
    # more iterations change little after five
    a, _ = point(backend, ua=6.0, Ta=285.0, To=18.0)
    f = backend._fn("similarity_fluxes_point")
    d = [6.0, 0.0, 285.0, 0.008, 101325.0, 0.0, 0.0, 0.0, 0.0, 18.0, 35.0, 9.80665, 1020.0]
    out = (C.c_double * 4)()
    f((C.c_double * 13)(*d), 30, out)
    assert np.allclose(a, np.array(out[:]), rtol=2e-3)


def test_the_atmosphere_of_the_reference():
    atm = gb.analytic_atmosphere()
    phi = np.array([-60.0, -12.0, 0.0, 33.3, 79.9])
    # linear functions between grid rows are reproduced; the analytic fields to the interpolation error of a 1-degree grid
    assert np.allclose(atm.interpolate("T", 0 * phi, phi), gb.Tatm(0, phi) + 273.15, atol=5e-3)
    assert np.allclose(atm.interpolate("u", 0 * phi, phi), gb.zonal_wind(0, phi), atol=6e-2)   # (the kink of |phi| at the equator)
    assert np.allclose(atm.interpolate("shortwave", 0 * phi, phi), gb.sunlight(0, phi), atol=0.1)
    assert (atm.fields["q"] == 0).all() and (atm.fields["p"] == 101325.0).all() and (atm.fields["longwave"] == 0).all()


def test_the_coupled_model_steps():
    """data_free_ocean_climate_model_init on the oracle: fluxes of the initial state, then steps that see them."""
    m = gb.data_free_ocean_climate_model_init(CPU("f64"), resolution=4, Nz=8, dt=30.0)
    gb.first_time_step(m)
    gb.loop(m, 3)
    b = m.backend
    Ju, JT, JS = b.top_flux("u"), b.top_flux("T"), b.top_flux("S")
    assert np.isfinite(Ju).all() and np.isfinite(JT).all() and np.isfinite(JS).all()
    wet = b.get_field("T", False)[:, :, -1] != 0
    assert 1e-6 < np.abs(Ju[wet]).max() < 1e-3                   # |tau| / rho0: up to ~0.1 N/m2 / 1020
    assert np.abs(JT[wet]).max() < 1e-3 and np.abs(JT[wet]).max() > 1e-6
    assert (JS[wet] <= 0).all()                                  # dry air: evaporation everywhere
    T = b.get_field("T", False)
    assert np.isfinite(T).all()
    # land columns (the two mountains reach the surface) carry no flux
    land = np.array([[b.bottom_info("kbot", i, j) for j in range(1, m.grid.size[1] + 1)] for i in range(1, m.grid.size[0] + 1)]) >= m.grid.size[2]
    if land.any():
        assert (JT[land] == 0).all()
