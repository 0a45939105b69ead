"""WENO(order = 7) of the oracle (ClimaOcean's ocean_simulation: tracer_advection = WENO(order = 7)) [UPSTREAM-UNVERIFIED:
Balsara & Shu (2000) coefficients recalled]: the reconstruction is exact for polynomials of degree 6, converges at 7th
order, its smoothness indicators are the Jiang-Shu integrals of the candidate polynomials, and it degrades next to walls."""
import ctypes as C

import numpy as np
import pytest

import gb25_amd as gb
from oracle_backend import CPU


@pytest.fixture(scope="module")
def weno7():
    b = gb.baroclinic_instability_model(CPU("f64"), 16, 8, 4, dt=60.0).backend
    f = b._fn("weno7")
    f.restype = C.c_double
    f.argtypes = [C.POINTER(C.c_double)]
    return lambda v: f((C.c_double * 7)(*[float(x) for x in v]))


def cell_averages(poly, edges):
    P = np.polynomial.Polynomial(poly).integ()
    return (P(edges[1:]) - P(edges[:-1])) / np.diff(edges)


def test_linear_part_is_exact_for_degree_six():
    """The candidate polynomials and linear weights typed into the oracle, restated here: each p_r reproduces the point value
    of a cubic from its four cell averages, their d-weighted sum that of a polynomial of degree six."""
    P = [np.array([0, 0, 0, 3, 13, -5, 1]) / 12, np.array([0, 0, -1, 7, 7, -1, 0]) / 12, np.array([0, 1, -5, 13, 3, 0, 0]) / 12,
         np.array([-3, 13, -23, 25, 0, 0, 0]) / 12]
    d = np.array([4, 18, 12, 1]) / 35
    rng = np.random.default_rng(1)
    edges = np.arange(-4.0, 4.0)
    for _ in range(10):
        cubic = np.concatenate([rng.standard_normal(4), np.zeros(3)])
        v = cell_averages(cubic, edges)
        for p in P:
            assert abs(p @ v - cubic[0]) < 1e-12
        poly = rng.standard_normal(7)
        v = cell_averages(poly, edges)
        assert abs(sum(w * (p @ v) for w, p in zip(d, P)) - poly[0]) < 1e-11


def test_reduces_to_the_linear_scheme_on_smooth_data(weno7):
    rng = np.random.default_rng(2)
    edges = 0.01 * np.arange(-4.0, 4.0)
    lin = (4 * np.array([0, 0, 0, 3, 13, -5, 1]) + 18 * np.array([0, 0, -1, 7, 7, -1, 0]) + 12 * np.array([0, 1, -5, 13, 3, 0, 0])
           + np.array([-3, 13, -23, 25, 0, 0, 0])) / (12 * 35)
    for _ in range(10):
        poly = rng.standard_normal(7)
        v = cell_averages(poly, edges)
        assert abs(weno7(v) - lin @ v) < 1e-9


def test_seventh_order_convergence(weno7):
    f = lambda x: np.sin(x + 0.3)
    F = lambda x: -np.cos(x + 0.3)
    errs = []
    for h in (0.2, 0.1, 0.05):
        edges = h * np.arange(-4.0, 4.0)
        v = (F(edges[1:]) - F(edges[:-1])) / h
        errs.append(abs(weno7(v) - f(0.0)))
    rates = [np.log2(errs[q] / errs[q + 1]) for q in range(2)]
    assert min(rates) > 6.5, (errs, rates)


def test_does_not_oscillate_at_a_step(weno7):
    v = np.array([1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0])       # the face sits at the discontinuity, upwind side = 1
    assert abs(weno7(v) - 1.0) < 1e-6
    v = np.array([1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0])
    assert abs(weno7(v) - 0.0) < 1e-6


def test_tendencies_change_and_stay_conservative():
    """order 7 in the tracer tendencies: differs from order 5 by the truncation error, keeps a constant tracer at rest, and
    keeps the volume integral of the tendency at zero (flux form) on the immersed tripolar grid."""
    out = {}
    for order in (5, 7):
        m = gb.baroclinic_instability_model(CPU("f64"), 48, 44, 8, dt=60.0, grid_type="gaussian_islands")
        f = m.backend._fn("set_tracer_advection_order")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int]
        f(m.backend.h, order)
        gb.set_baroclinic_instability(m)
        Nx, Ny, Nz = m.grid.size
        rng = np.random.default_rng(3)
        m.set(u=0.1 * rng.standard_normal((Nx, Ny, Nz)), v=0.1 * rng.standard_normal((Nx, Ny + 1, Nz)))
        S0 = m.backend.get_field("S", False)
        m.set(S=np.where(S0 != 0, 35.0, 0.0))
        gb.update_state(m)
        out[order] = m.backend.get_field("Gn.T", False)
        GS = m.backend.get_field("Gn.S", False)
        assert np.abs(GS).max() < 1e-9 * 35.0                 # constant tracer: no tendency (continuity), order 7 too
    d = np.linalg.norm(out[7] - out[5]) / np.linalg.norm(out[5])
    assert 1e-4 < d < 0.5, d
