"""The data-free forcing on the HIP path (GB-25 src/data_free_ocean_climate_model.jl:12-70) against
the oracle's restatement (tests/test_oracle_data_free.py checks that one against an independent statement of the formulas):
the flux solve alone, the coupled model stepping, and x slabs."""
import numpy as np
import pytest

import gb25_amd as gb
from helpers import SQRT_EPS32
from oracle_backend import CPU

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = max(np.linalg.norm(a.ravel()), np.linalg.norm(b.ravel()))
    return 0.0 if n == 0 else float(np.linalg.norm((a - b).ravel()) / n)


def stir(m, seed=5):
    Nx, Ny, Nz = m.grid.size
    rng = np.random.default_rng(seed)
    dt = m.backend.dtype
    m.set(u=(0.3 * rng.standard_normal((Nx, Ny, Nz))).astype(np.float32).astype(dt),
          v=(0.3 * rng.standard_normal((Nx, Ny + 1, Nz))).astype(np.float32).astype(dt))


@pytest.mark.parametrize("float_type,tol", [("Float64", 1e-12), ("Float32", 2e-6)])
@pytest.mark.parametrize("grid_type", ["simple_lat_lon", "gaussian_islands"])
def test_the_flux_solve_matches_the_oracle(grid_type, float_type, tol):
    """compute_atmosphere_ocean_fluxes! alone: J^u, J^v, J^T, J^S from the same state (the solve runs in fp64 on both sides;
    with a Float32 state the difference is the rounding of the fluxes themselves)."""
    models = []
    for arch in (gb.GPU(float_type=float_type), CPU("f64" if float_type == "Float64" else "f32")):
        m = gb.baroclinic_instability_model(arch, 96, 48, 6, dt=30.0, grid_type=grid_type)
        gb.set_baroclinic_instability(m)
        stir(m)
        gb.set_prescribed_atmosphere(m, gb.analytic_atmosphere())
        gb.update_state(m)
        m.backend.compute_atmosphere_ocean_fluxes()
        models.append(m)
    r, v = models
    for n in ("u", "v", "T", "S"):
        a, b = r.backend.top_flux(n), v.backend.top_flux(n)
        assert np.isfinite(a).all() and np.abs(b).max() > 0
        assert rel(a, b) < tol, (n, rel(a, b))


@pytest.mark.parametrize("float_type", ["Float64", "Float32"])
def test_the_coupled_model_steps_like_the_oracle(float_type):
    """data_free_ocean_climate_model_init: TripolarGrid with the Gaussian islands, CATKE, the analytic atmosphere;
    first_time_step! + 20 steps of 30 s."""
    kw = dict(resolution=4, Nz=10, dt=30.0)
    r = gb.data_free_ocean_climate_model_init(gb.GPU(float_type=float_type), **kw)
    v = gb.data_free_ocean_climate_model_init(CPU("f64"), **kw)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 20)
    _, report = gb.compare_states(r, v, rtol=SQRT_EPS32, include_halos=True, verbose=False)
    if float_type == "Float64":
        # (L^e = wb-/e [e > e_min] next to e = 0, where this run starts: its switches amplify round-off most; 1e-6 for it)
        bad = [(q["name"], q["rel"]) for q in report if not q["rel"] <= (1e-6 if q["name"] == "Le" else 1e-7)]
    else:
        w = gb.data_free_ocean_climate_model_init(CPU("f32"), **kw)      # the yardstick of tests/test_gpu_catke.py
        gb.first_time_step(w)
        gb.loop(w, 20)
        _, own = gb.compare_states(w, v, rtol=SQRT_EPS32, include_halos=True, verbose=False)
        own = {q["name"]: q["rel"] for q in own}
        bad = [(q["name"], q["rel"], own[q["name"]]) for q in report if not q["rel"] <= max(SQRT_EPS32, 2.0 * own[q["name"]])]
    assert not bad, bad
    for n in ("u", "v", "T", "S"):
        a, b = r.backend.top_flux(n), v.backend.top_flux(n)
        assert rel(a, b) < (1e-9 if float_type == "Float64" else 1e-4), (n, rel(a, b))
    assert np.abs(r.backend.top_flux("u")).max() > 1e-6


@pytest.mark.parametrize("float_type", ["Float64", "Float32"])
def test_w_on_the_fly_in_the_coupled_model(float_type):
    """The data-free climate model with w on the fly beside the corrector's sweep (what a single domain of 8 M cells and more
    runs between the steps of a loop): the instances of the three tendency kernels with the quadratic bottom drag, WENO(order = 7)
    tracers and the closure's e carry w up their chunks, no k_compute_w launch.  Against the stand-alone w: round-off, with the
    closure's switches on top (the limits of tests/test_gpu_catke.py::test_w_on_the_fly_with_catke)."""
    eps = float(np.finfo(np.float32 if float_type == "Float32" else np.float64).eps)
    models = []
    for fly in (0, 1):
        m = gb.data_free_ocean_climate_model_init(gb.GPU(float_type=float_type), resolution=4, Nz=24, dt=30.0,
                                                  options=dict(w_on_the_fly=fly, subcycle_lookahead=1))
        gb.first_time_step(m)
        m.backend.profile_enable(True)
        m.backend.profile_reset()
        gb.loop(m, 12)
        models.append(m)
    a, b = models
    assert a.backend.profile_get("compute_w")[0] >= 12 and b.backend.profile_get("compute_w")[0] <= 3
    got = {n: rel(a.backend.get_field(n, True), b.backend.get_field(n, True))
           for n in ("u", "v", "w", "T", "S", "eta", "Gn.u", "Gn.T", "Gn.e", "e", "kappa_u", "kappa_c", "Le")}
    print(got)
    loose = dict(e=2e-3, kappa_u=3e-2, kappa_c=3e-2, Le=0.3) if float_type == "Float32" else dict.fromkeys(("e", "kappa_u", "kappa_c", "Le"), 1e-8)
    for n, r_ in got.items():
        assert np.isfinite(b.backend.get_field(n, True)).all()
        assert r_ < loose.get(n, 4000 * eps if float_type == "Float32" else 1e-11), (n, got)
    for n in ("u", "T"):
        assert rel(a.backend.top_flux(n), b.backend.top_flux(n)) < (1e-4 if float_type == "Float32" else 1e-10)
    for m in models:
        m.backend.close()


def test_coupled_slabs_are_the_single_domain_bit_for_bit():
    """The fluxes of a slab's first halo column and fold row are computed from exchanged halos, never exchanged."""
    from gb25_amd.distributed import LocalSlabEnsemble
    from gb25_amd.data_free import ATMOSPHERE_FIELDS
    Nx, Ny, Nz, dt, P = 96, 48, 8, 30.0, 2
    m = gb.data_free_ocean_climate_model_init(gb.GPU(), resolution=4, Nz=Nz, dt=dt)
    stir(m, 9)
    names = ("u", "v", "T", "S", "e", "eta")
    init = {n: m.backend.get_field(n, False) for n in names}
    H = 8
    phi = np.asarray(m.backend.metric2("phicc"))[:, : Ny + 2 * H]
    atm = gb.analytic_atmosphere()
    gb.first_time_step(m)
    gb.loop(m, 5)
    out = names + ("U", "V", "kappa_u", "Gn.T", "Gn.u")
    ref = {n: m.backend.get_field(n, False) for n in out}
    flux = {n: m.backend.top_flux(n) for n in ("u", "v", "T", "S")}
    m.backend.close()
    ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, grid_type=4)
    w = Nx // P
    for r, b in enumerate(ens.backends):
        b.set_catke(True)
        b.set_catke_parameters(**gb.default_ocean_closure().parameters)
        b.set_bottom_drag(0.003)
        b.set_tracer_advection_order(7)
        lp = np.asarray(b.metric2("phicc"))[:, : Ny + 2 * H]
        for n in ATMOSPHERE_FIELDS:
            b.set_prescribed_atmosphere(n, atm.interpolate(n, np.zeros_like(lp), lp))
    for n, a in init.items():
        ens.scatter(n, a)
    ens.first_time_step()
    ens.loop(5)
    bad = [n for n, a in ref.items() if not np.array_equal(ens.gather(n), a)]
    assert not bad, [(n, rel(ens.gather(n), ref[n])) for n in bad]
    for n, a in flux.items():
        got = np.concatenate([b.top_flux(n) for b in ens.backends], axis=0)
        assert np.array_equal(got, a), n
    ens.close()


def test_rccl_self_ring_with_the_coupled_model():
    """The RCCL transport with the larger bundles of a coupled CATKE model (e and J^b in group 0 and in the fold rows, buffers
    re-sized after the closure was switched on): ONE rank that is its own neighbour and fold partner; bit for bit the single
    domain."""
    from gb25_amd.distributed import SlabModel
    Nx, Ny, Nz, dt = 96, 48, 8, 30.0
    single = gb.data_free_ocean_climate_model_init(gb.GPU(), resolution=4, Nz=Nz, dt=dt)
    stir(single, 4)
    init = {n: single.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta")}
    ring = SlabModel(Nx, Ny, Nz, dt=dt, rank=0, nranks=1, slab_mode=1, transport="rccl", grid_type=4)
    ring.grid_type = "gaussian_islands"
    ring.backend.set_catke(True)
    ring.backend.set_catke_parameters(**gb.default_ocean_closure().parameters)
    ring.backend.set_bottom_drag(0.003)
    ring.backend.set_tracer_advection_order(7)
    ring.enable_catke_fields()
    gb.set_prescribed_atmosphere(ring, gb.analytic_atmosphere())
    for n, a in init.items():
        ring.backend.set_field(n, a, False)
    for m in (single, ring):
        gb.first_time_step(m)
        gb.loop(m, 4)
    for n in ("u", "v", "T", "S", "e", "eta", "U", "V", "kappa_u", "Gn.e", "Gn.u"):
        a, b = ring.backend.get_field(n, False), single.backend.get_field(n, False)
        assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
    for n in ("u", "v", "T", "S"):
        assert np.array_equal(ring.backend.top_flux(n), single.backend.top_flux(n)), n
    ring.backend.close()
    single.backend.close()
