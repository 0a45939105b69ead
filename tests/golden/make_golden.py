#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the fp64 CPU oracle (oracle/gb25_oracle.c).

These are SUBSTITUTE pins: the reference is Julia, absent from this image, and commits
no golden vectors of its own.  The fixtures freeze the oracle's current answers so that later edits of either
the oracle or the HIP path are caught; they are not outputs of Oceananigans.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import gb25_amd as gb                                  # noqa: E402
from helpers import make_oracle, set_noisy_velocities  # noqa: E402
from oracle_backend import OracleBackend               # noqa: E402

CASE = dict(Nx=16, Ny=12, Nz=6, dt=600.0)
FIELDS = ["u", "v", "w", "T", "S", "eta", "Gn.u", "Gn.v", "Gn.T", "Gn.S", "U", "V", "eta_bar", "U_bar", "V_bar", "pHY"]


def run_case():
    m = make_oracle(precision="f64", **CASE)
    gb.set_baroclinic_instability(m)
    set_noisy_velocities(m, amplitude=1e-2)
    out = {"in." + n: m.backend.get_field(n, False) for n in ("u", "v", "T", "S")}
    gb.first_time_step(m)
    out.update({"step1." + n: m.backend.get_field(n, False) for n in FIELDS})
    gb.loop(m, 2)
    out.update({"step3." + n: m.backend.get_field(n, False) for n in FIELDS})
    return out


COUPLED = dict(resolution=8, Nz=6, dt=30.0)     # 48 x 24 x 6: TripolarGrid + mountains + CATKE + the analytic atmosphere
COUPLED_FIELDS = ["u", "v", "T", "S", "e", "eta", "U", "V", "kappa_u", "kappa_c", "kappa_e", "Le", "Jb", "Gn.e", "Gn.u", "Gn.T"]


def run_coupled():
    """The whole section-8f stack in one small case: data_free_ocean_climate_model_init (tripolar grid with the Gaussian
    islands, CATKE, similarity-theory fluxes after every step), first_time_step! + 4 steps."""
    from oracle_backend import CPU
    m = gb.data_free_ocean_climate_model_init(CPU("f64"), **COUPLED)
    out = {"in." + n: m.backend.get_field(n, False) for n in ("T", "S")}
    gb.first_time_step(m)
    gb.loop(m, 4)
    out.update({"step5." + n: m.backend.get_field(n, False) for n in COUPLED_FIELDS})
    out.update({"flux5." + n: m.backend.top_flux(n).astype(np.float64) for n in ("u", "v", "T", "S")})
    return out


def unit_vectors():
    ob = OracleBackend(16, 16, 4, dt=1.0)
    rng = np.random.default_rng(2025)
    w_in = np.concatenate([rng.standard_normal((40, 5)), 30 + 1e-3 * rng.standard_normal((20, 5)),
                           np.array([[1, 1, 1, 1, 100.0], [0, 0, 0, 0, 0], [1, 2, 3, 4, 5.0]])])
    w_out = np.array([ob.weno5(*row) for row in w_in])
    T, S, Z = np.meshgrid(np.linspace(-2, 30, 9), np.linspace(0, 40, 9), np.linspace(-4000, 0, 9), indexing="ij")
    rho = np.vectorize(ob.teos10_rho)(T, S, Z)
    return dict(weno5_in=w_in, weno5_out=w_out, teos_T=T, teos_S=S, teos_Z=Z, teos_rho=rho)


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "oracle_f64_16x12x6.npz"), **run_case())
    np.savez_compressed(os.path.join(HERE, "oracle_f64_units.npz"), **unit_vectors())
    np.savez_compressed(os.path.join(HERE, "oracle_f64_coupled_48x24x6.npz"), **run_coupled())
    print("wrote", os.listdir(HERE))
