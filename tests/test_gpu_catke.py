"""closure = CATKEVerticalDiffusivity() on the HIP path (GB-25 src/baroclinic_instability_model.jl:
30,50-51, sharding/less_simple_sharding_problem.jl:84-93) against the oracle's restatement (tests/test_oracle_catke.py
pins that one): the diffusivity fields the reference compares (src/correctness.jl:60-67), the TKE tracer, the implicit
solves with diffusivity fields, wind- and cooling-driven mixing."""
import numpy as np
import pytest

import gb25_amd as gb
from gb25_amd.binding import GB25Error
from helpers import SQRT_EPS32, counter_rng, make_pair

pytestmark = pytest.mark.gpu
CATKE = gb.CATKEVerticalDiffusivity


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = max(np.linalg.norm(a.ravel()), np.linalg.norm(b.ravel()))
    return 0.0 if n == 0 else float(np.linalg.norm((a - b).ravel()) / n)


def start(r, v, N2=1e-5, wind=None, heat=None):
    Nx, Ny, Nz = v.grid.size
    zc = np.array([v.backend.metric("zc", k) for k in range(1, Nz + 1)])
    T = np.broadcast_to(20.0 + (N2 / (9.80665 * 2e-4)) * zc, (Nx, Ny, Nz))
    rng = np.random.default_rng(3)
    e = 1e-5 * rng.random((Nx, Ny, Nz)) + 1e-7
    u = 0.02 * rng.standard_normal((Nx, Ny, Nz))
    for m in (r, v):
        dt = m.backend.dtype
        S = np.broadcast_to(35.0 - 1e-3 * zc, (Nx, Ny, Nz))
        m.set(T=T.astype(np.float32).astype(dt), S=S.astype(np.float32).astype(dt), e=e.astype(np.float32).astype(dt),
              u=u.astype(np.float32).astype(dt))
        if wind is not None:
            gb.set_top_flux(m, u=np.full((Nx, Ny), wind, dt))
        if heat is not None:
            gb.set_top_flux(m, T=np.full((Nx, Ny), heat, dt))


@pytest.mark.parametrize("float_type,tol", [("Float64", 1e-10), ("Float32", 2e-5)])
def test_diffusivity_fields_match_the_oracle(float_type, tol):
    """compute_diffusivities! alone (update_state!): kappa_u, kappa_c, kappa_e, L^e, J^b and the TKE tendency, halos
    included (a14: the fill of the diffusivity fields)."""
    r, v = make_pair(40, 44, 16, dt=120.0, float_type=float_type, depth=200.0, closure=CATKE())
    start(r, v, wind=-1e-4, heat=5e-5)
    names = ("kappa_u", "kappa_c", "kappa_e", "Le", "Jb", "e", "Gn.e", "Gm.e", "Gn.T", "Gn.u", "previous_u", "previous_v")
    for m in (r, v):
        gb.update_state(m)
    for n in names:
        a, b = r.backend.get_field(n, True), v.backend.get_field(n, True)
        assert rel(a, b) < tol, (n, rel(a, b))
    # (the first compute_diffusivities! has stepped e with nothing to produce it yet and left J^b alone: no time has passed)
    assert r.diffusivity_fields.kappa_u.interior.max() > 1e-5 and r.diffusivity_fields.Jb.interior.max() == 0
    # a second one right away steps e again, now with shear production and buoyancy flux from the diffusivities of the first ...
    for m in (r, v):
        gb.update_state(m)
    for n in names:
        a, b = r.backend.get_field(n, True), v.backend.get_field(n, True)
        assert rel(a, b) < 4 * tol, (n, rel(a, b))
    # ... and after two time steps the surface buoyancy flux has come through its filter
    for m in (r, v):
        gb.first_time_step(m)
        gb.time_step(m)
    for n in names:
        a, b = r.backend.get_field(n, True), v.backend.get_field(n, True)
        assert rel(a, b) < (1e-8 if float_type == "Float64" else 5e-4), (n, rel(a, b))
    assert r.diffusivity_fields.Jb.interior.min() > 1e-9


@pytest.mark.parametrize("float_type", ["Float64", "Float32"])
@pytest.mark.parametrize("case", ["wind", "cooling", "islands", "deep", "tripolar"])
def test_stepping_with_catke_matches_the_oracle(case, float_type):
    """first_time_step! + 30 steps.  The Float64 build is the logic check (every compared field, halos included, to
    1e-7: the switches of the mixing lengths -- min / max / step of Ri -- amplify round-off a little).  Float32 against
    the Float64 oracle: sqrt(eps) where a Float32 run can reach it; the diffusivity fields, e and what they feed are
    products of square roots, clipped ratios and switches on quantities near zero (L^e = -sqrt|e|/l_D + wb-/e [e > e_min])
    -- there the yardstick is the Float32 ORACLE's own distance from the Float64 one (as close to a
    Float32 reference run as that run is to the truth)."""
    # tripolar: the reference's grid_type = :gaussian_islands (TripolarGrid + mountains) -- the diffusivity fields cross the
    # zipper fold (kappa_u is averaged onto the v faces of the fold line)
    kw = dict(grid_type="gaussian_islands_lat_lon") if case == "islands" else dict(grid_type="gaussian_islands") if case == "tripolar" else {}
    big = case in ("islands", "tripolar")
    # deep: 72 levels -- past the register-resident implicit solve (64 levels in Float32, 32 in Float64): the streamed one
    size = (90, 44, 16) if case == "islands" else (72, 44, 16) if case == "tripolar" else (40, 44, 72) if case == "deep" else (40, 44, 24)
    r, v = make_pair(*size, dt=120.0, float_type=float_type, depth=4000.0 if big else 200.0, closure=CATKE(), **kw)
    start(r, v, wind=-1e-4 if case != "cooling" else None, heat=1e-4 if case == "cooling" else None)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 30)
    ok, report = gb.compare_states(r, v, rtol=SQRT_EPS32, include_halos=True, verbose=False)
    if float_type == "Float64":
        # (deep: 72 levels over 200 m -- the differences of S across a cell are 4e-5 of S, its tendency loses those digits)
        bad = [(q["name"], q["rel"]) for q in report if not q["rel"] <= (5e-7 if case == "deep" else 1e-7)]
        assert not bad, bad
    else:
        # what a Float32 run of the reference itself is away from the Float64 truth: the oracle in Float32
        from oracle_backend import CPU
        w = gb.baroclinic_instability_model(CPU("f32"), *size, dt=120.0, depth=4000.0 if big else 200.0,
                                            closure=CATKE(), **kw)
        start(w, w, wind=-1e-4 if case != "cooling" else None, heat=1e-4 if case == "cooling" else None)
        gb.first_time_step(w)
        gb.loop(w, 30)
        _, own = gb.compare_states(w, v, rtol=SQRT_EPS32, include_halos=True, verbose=False)
        own = {q["name"]: q["rel"] for q in own}
        bad = [(q["name"], q["rel"], own[q["name"]]) for q in report if not q["rel"] <= max(SQRT_EPS32, 2.0 * own[q["name"]])]
        print({q["name"]: (round(q["rel"], 6), round(own[q["name"]], 6)) for q in report if q["rel"] > SQRT_EPS32})
        assert not bad, bad
        # The switches of the closure must fall the same way as in the Float64 oracle almost everywhere: a field that
        # is within tolerance in norm but has its masks elsewhere would be a different model.  Cells, halos excluded.
        emin = r.backend.catke_parameters().minimum_tke
        switches = {"e > e_min": lambda m: m.tracers.e.interior > emin,
                    "kappa_u > 0": lambda m: m.diffusivity_fields.kappa_u.interior > 0,
                    "kappa_c > kappa_u (Ri-dependent Prandtl number below one)":
                        lambda m: m.diffusivity_fields.kappa_c.interior > m.diffusivity_fields.kappa_u.interior,
                    "Le < 0 (dissipation wins)": lambda m: m.diffusivity_fields.Le.interior < 0,
                    "Jb > 0 (cooled)": lambda m: m.diffusivity_fields.Jb.interior > 0}
        agree = {k: float(np.mean(f(r) == f(v))) for k, f in switches.items()}
        # (e > e_min: under pure cooling the TKE below the convecting layer decays to the floor and sits within round-off of it)
        assert all(a >= (0.998 if k.startswith("e >") else 0.999) for k, a in agree.items()), agree
        # (the measured numbers behind DESIGN.md section 0's table of the closure fields)
        import json, os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        path = os.path.join(root, "gpurun_out", "r04_catke_fp32.json")
        doc = json.load(open(path)) if os.path.exists(path) else {}
        doc[case] = {"size": list(size), "steps": 31, "rtol_reference": SQRT_EPS32,
                     "hip_f32_vs_oracle_f64": {q["name"]: q["rel"] for q in report},
                     "oracle_f32_vs_oracle_f64": own, "switch_agreement_with_oracle_f64": agree}
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    names = [q["name"] for q in report]
    for n in ("e", "Gn.e", "kappa_u", "kappa_c", "kappa_e", "Le", "Jb"):
        assert n in names
    e = r.tracers.e.interior
    assert e[:, :, -1].max() > 1e-6 and np.isfinite(e).all()


def test_catke_fields_exist_only_with_the_closure():
    m = gb.baroclinic_instability_model(gb.GPU(), 32, 16, 8, dt=60.0)
    with pytest.raises(GB25Error):                                # closure = nothing: no TKE tracer
        m.backend.get_field("e", False)


@pytest.mark.parametrize("grid_type,P", [("simple_lat_lon", 3), ("gaussian_islands_lat_lon", 2), ("gaussian_islands", 2),
                                         ("gaussian_islands", 4)])
def test_catke_on_slabs_is_the_single_domain_bit_for_bit(grid_type, P):
    """x slabs with CATKE: e is stepped and J^b filtered on the own columns inside compute_diffusivities!, their halos travel
    (one exchange per update_state!: columns, and the rows beyond a zipper fold), kappa in the first halo column on either side
    / the fold row is computed locally from the exchanged halos."""
    from gb25_amd.distributed import LocalSlabEnsemble
    Nx, Ny, Nz, dt = 192, 44, 12, 300.0
    depth = 4000.0 if "islands" in grid_type else 200.0
    m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, depth=depth, grid_type=grid_type, closure=CATKE())
    start(m, m, wind=-1e-4, heat=5e-5)
    rng = np.random.default_rng(11)
    Ju = (-1e-4 * (1.0 + 0.3 * rng.random((Nx, Ny)))).astype(np.float32)     # (fluxes that differ from column to column)
    JT = (5e-5 * (1.0 + 0.3 * rng.random((Nx, Ny)))).astype(np.float32)
    gb.set_top_flux(m, u=Ju, T=JT)
    names = ("u", "v", "T", "S", "e", "eta")
    init = {n: m.backend.get_field(n, False) for n in names}
    gb.first_time_step(m)
    gb.loop(m, 6)
    out = ("u", "v", "T", "S", "e", "eta", "U", "V", "kappa_u", "kappa_c", "kappa_e", "Le", "Jb", "Gn.e", "previous_u", "previous_v")
    ref = {n: m.backend.get_field(n, False) for n in out}
    assert np.isfinite(ref["e"]).all() and ref["kappa_u"].max() > 1e-6
    m.backend.close()
    gt = {"simple_lat_lon": 0, "gaussian_islands_lat_lon": 1, "gaussian_islands": 4}[grid_type]
    ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, depth=depth, **(dict(grid_type=gt) if gt else {}))
    w = Nx // P
    for r, b in enumerate(ens.backends):
        b.set_catke(True)
        b.set_top_flux("u", np.ascontiguousarray(Ju[r * w:(r + 1) * w]))
        b.set_top_flux("T", np.ascontiguousarray(JT[r * w:(r + 1) * w]))
    for n, a in init.items():
        ens.scatter(n, a)
    ens.first_time_step()
    ens.loop(6)
    bad = [n for n, a in ref.items() if not np.array_equal(ens.gather(n), a)]
    assert not bad, [(n, rel(ens.gather(n), ref[n])) for n in bad]
    ens.close()


@pytest.mark.parametrize("grid_type", ["simple_lat_lon", "gaussian_islands"])
def test_previous_velocities_move_by_pointer_exchange(grid_type):
    """diffusivity_fields.previous_velocities on a single domain: no copies in the steady state -- the AB2 step adopts the
    look-ahead's buffers and the ones it gives up ARE the velocities of the last compute_diffusivities! (gb25_api.hip,
    prev_uv_src).  A loop of steps (the fields brought home once, when the host looks) against single steps with the host
    looking after every one (a copy per step), bit for bit; u- = u after compute_diffusivities!, halos included."""
    Nx, Ny, Nz, dt = 96, 44, 16, 300.0
    depth = 4000.0 if "islands" in grid_type else 200.0
    names = ("u", "v", "T", "e", "kappa_u", "kappa_e", "Le", "Jb", "Gm.e", "previous_u", "previous_v")
    outs = []
    for single_steps in (False, True):
        m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, depth=depth, grid_type=grid_type, closure=CATKE())
        start(m, m, wind=-1e-4, heat=5e-5)
        gb.first_time_step(m)
        if single_steps:
            for _ in range(7):
                gb.time_step(m)
                assert np.array_equal(m.backend.get_field("previous_u", True), m.backend.get_field("u", True))
            gb.update_state(m)          # (twice in a row: the second e step sees u- = u)
        else:
            gb.loop(m, 7)
            gb.update_state(m)
        outs.append({n: m.backend.get_field(n, True) for n in names})
        m.backend.close()
    bad = [n for n in names if not np.array_equal(outs[0][n], outs[1][n])]
    assert not bad, [(n, rel(outs[0][n], outs[1][n])) for n in bad]
    assert np.array_equal(outs[0]["previous_v"], outs[0]["v"]) and np.abs(outs[0]["u"]).max() > 1e-3


@pytest.mark.parametrize("float_type", ["Float32", "Float64"])
@pytest.mark.parametrize("grid_type,shape", [("simple_lat_lon", (150, 70, 24)), ("gaussian_islands_lat_lon", (150, 70, 24)),
                                             ("gaussian_islands", (144, 64, 24))])
def test_w_on_the_fly_with_catke(grid_type, shape, float_type):
    """w on the fly beside the corrector's sweep with the closure on: the implicit solve of u, v rewrites the look-ahead's chunk
    sums with those of the velocities it leaves (k_w_bases takes w at the chunk boundaries from them) and the advection of e
    carries w like the two other tendency kernels -- no k_compute_w launch.  Against the stand-alone w: round-off (the vertical
    sum is associated by chunks), amplified by the closure's switches to what the fp32 story of the closure fields states."""
    Nx, Ny, Nz = shape
    eps = float(np.finfo(np.float32 if float_type == "Float32" else np.float64).eps)
    depth = 4000.0 if "islands" in grid_type else 400.0
    models = []
    for fly in (0, 1):
        m = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=300.0, depth=depth, grid_type=grid_type,
                                            closure=CATKE(), options=dict(w_on_the_fly=fly, subcycle_lookahead=1))
        start(m, m, wind=-1e-4, heat=5e-5)
        gb.first_time_step(m)
        m.backend.profile_enable(True)
        m.backend.profile_reset()
        gb.loop(m, 12)
        models.append(m)
    a, b = models
    assert a.backend.profile_get("compute_w")[0] >= 12 and b.backend.profile_get("compute_w")[0] <= 3
    for n in ("u", "v", "T", "S", "eta", "U", "V", "Gn.u", "Gn.T", "Gn.e"):
        x, y = a.backend.get_field(n, True), b.backend.get_field(n, True)
        # (Float64: 1e-12 measured on the grids with mountains -- the closure's switches sit on top of the round-off of w)
        assert np.isfinite(y).all() and rel(x, y) < (4000 * eps if float_type == "Float32" else 1e-11), (n, rel(x, y))
    assert rel(a.backend.get_field("w", True), b.backend.get_field("w", True)) < (400 * eps if float_type == "Float32" else 1e-11)
    # (Float32: N^2 of a mixed layer is a difference of temperatures a few ulps apart, and the stratification-limited lengths go
    # with N^-1 -- the distances are those of the Float32 story of the closure fields, profiles/r04_catke_fp32.json)
    got = {n: rel(a.backend.get_field(n, True), b.backend.get_field(n, True)) for n in ("e", "kappa_u", "kappa_c", "kappa_e", "Le")}
    print(got)
    lim = dict(e=2e-3, kappa_u=3e-2, kappa_c=3e-2, kappa_e=3e-2, Le=0.3) if float_type == "Float32" else dict.fromkeys(got, 1e-8)
    assert all(np.isfinite(b.backend.get_field(n, True)).all() and got[n] < lim[n] for n in got), got
    for m in models:
        m.backend.close()


def test_catke_schedules_agree():
    outs = []
    for opts in (dict(), dict(two_streams=0), dict(ab2_lookahead=0)):
        m = gb.baroclinic_instability_model(gb.GPU(), 40, 44, 24, dt=120.0, depth=200.0, closure=CATKE(), options=opts)
        v = m
        start(m, v, wind=-1e-4)
        gb.first_time_step(m)
        gb.loop(m, 10)
        outs.append({n: m.backend.get_field(n, False) for n in ("u", "T", "e", "kappa_c", "eta")})
        m.backend.close()
    for o in outs[1:]:
        for n, a in outs[0].items():
            assert rel(a, o[n]) < 2e-6, (n, rel(a, o[n]))


@pytest.mark.parametrize("float_type,tol", [("Float64", 1e-10), ("Float32", 2e-5)])
def test_catke_parameters_reach_the_kernels(float_type, tol):
    """gb25_set_catke_parameters (ClimaOcean's default_ocean_closure changes C^b; any other parameter likewise): the
    diffusivity fields follow, HIP and oracle alike."""
    fields = {}
    for par in (dict(), dict(Cb=0.01, Cs=0.8, Cun=(0.5, 0.4, 1.0, 0.9), CWu=2.0)):
        r, v = make_pair(40, 44, 16, dt=120.0, float_type=float_type, depth=200.0, closure=CATKE(**par))
        start(r, v, wind=-1e-4, heat=5e-5)
        for m in (r, v):
            gb.update_state(m)
        for n in ("kappa_u", "kappa_c", "kappa_e", "Le", "Gn.e"):
            a, b = r.backend.get_field(n, True), v.backend.get_field(n, True)
            assert rel(a, b) < tol, (par, n, rel(a, b))
        fields[bool(par)] = r.backend.get_field("kappa_c", False)
        p = r.backend.catke_parameters()
        assert p.Cb == par.get("Cb", 0.28) and tuple(p.Cun) == tuple(par.get("Cun", (0.370, 0.369, 1.447, 0.923)))
        r.backend.close()
    assert rel(fields[True], fields[False]) > 1e-2          # (they do change the answer)
