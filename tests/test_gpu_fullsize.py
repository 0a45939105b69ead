"""Size-independent properties at BASELINE.json's full single-GPU size (1440x720x48), where the oracle
is too slow to run: state of rest, discrete consistency of continuity + tracer advection, zonal symmetry
and exact x-translation invariance of the periodic direction, finiteness."""
import numpy as np
import pytest

import gb25_amd as gb
from helpers import counter_rng

pytestmark = pytest.mark.gpu
NX, NY, NZ = 1440, 720, 48
DT = 120.0   # the bench line's time step (240 s is beyond the internal-wave stability limit of this grid: tools/stability_probe.py)


def fresh_model(**options):
    """A new all-zero model (device allocation + memset: far cheaper than uploading 22 zero arrays)."""
    return gb.baroclinic_instability_model(gb.GPU(), NX, NY, NZ, dt=DT, options=options)


@pytest.fixture()
def model():
    m = fresh_model()
    yield m
    m.backend.close()


def test_state_of_rest(model):
    m = model
    zc = np.array([m.grid.metric("zc", k) for k in range(1, NZ + 1)], np.float32)
    T = np.broadcast_to(10 + 5e-3 * zc, (NX, NY, NZ)).astype(np.float32)
    S = np.broadcast_to(35 - 1e-3 * zc, (NX, NY, NZ)).astype(np.float32)
    m.set(T=T, S=S)
    gb.first_time_step(m)
    gb.loop(m, 2)
    for name in ("u", "v", "w", "eta"):
        assert np.abs(m.fields()[name].interior).max() == 0.0, name
    assert np.abs(m.timestepper.Gn.u.interior).max() == 0.0
    assert np.abs(m.timestepper.Gn.v.interior[:, 1:, :]).max() == 0.0
    assert np.array_equal(m.tracers.T.interior, T)


def test_constant_tracer_and_budget(model):
    m = model
    u0 = (0.2 * (counter_rng((NX, NY, NZ), 1, 1) - 0.5)).astype(np.float32)
    v0 = (0.2 * (counter_rng((NX, NY + 1, NZ), 1, 2) - 0.5)).astype(np.float32)
    m.set(u=u0, v=v0, T=np.full((NX, NY, NZ), 7.0, np.float32),
          S=(35 + counter_rng((NX, NY, NZ), 1, 3)).astype(np.float32))
    gb.update_state(m)
    w = m.velocities.w.interior
    assert np.isfinite(w).all() and np.abs(w).max() > 0
    GT = m.timestepper.Gn.T.interior
    # fp32 round-off of a constant field: c * ulp * (|u|/dx + |w|/dz); the random velocities make w/dz dominate
    dzmin = min(m.grid.metric("dzc", k) for k in range(1, NZ + 1))
    dxmin = m.grid.metric("dxc", 1)
    assert np.abs(GT).max() < 7.0 * 5e-7 * (np.abs(w).max() / dzmin + 0.2 / dxmin)
    az = np.array([m.grid.metric("azc", j) for j in range(1, NY + 1)])
    dz = np.array([m.grid.metric("dzc", k) for k in range(1, NZ + 1)])
    GS = m.timestepper.Gn.S.interior.astype(np.float64)
    vol = az[None, :, None] * dz[None, None, :]
    total = (vol * GS).sum()
    Sp = m.tracers.S.parent
    H = 8
    wtop = w[:, :, NZ].astype(np.float64)
    c_in, c_halo = Sp[H:-H, H:-H, H + NZ - 1], Sp[H:-H, H:-H, H + NZ]
    top = (az[None, :] * wtop * np.where(wtop > 0, c_in, c_halo)).sum()
    assert abs(total + top) < 2e-6 * np.abs(vol * GS).sum()


def test_zonal_symmetry_and_translation_invariance(model):
    m = model
    gb.set_baroclinic_instability(m)          # zonally symmetric
    gb.first_time_step(m)
    gb.loop(m, 3)
    for name in ("v", "T", "eta", "w"):
        a = m.fields()[name].interior
        assert np.isfinite(a).all(), name
        assert np.array_equal(a, np.broadcast_to(a[:1], a.shape)), name      # every longitude identical
    assert np.abs(m.velocities.v.interior).max() > 1e-3

    # periodic x: shifting the initial state by s columns shifts the result by s columns, bit for bit
    s = 517
    u0 = (1e-2 * counter_rng((NX, NY, NZ), 42, 1)).astype(np.float32)
    results = []
    for shift in (0, s):
        m.backend.close()
        m = fresh_model()
        gb.set_baroclinic_instability(m)
        m.set(u=np.roll(u0, shift, axis=0))
        gb.first_time_step(m)
        gb.loop(m, 2)
        results.append({n: m.fields()[n].interior for n in ("u", "v", "T", "eta")})
    m.backend.close()
    for n in results[0]:
        assert np.array_equal(np.roll(results[0][n], s, axis=0), results[1][n]), n


def test_full_size_parity_against_oracle():
    """BASELINE.json's full single-GPU size, 1440x720x48, against the fp64 oracle itself (about 3.5 s per oracle
    step on 16 cores): first_time_step! + 1 step from the deterministic baroclinic state with seeded velocity noise."""
    from helpers import assert_states_close
    from oracle_backend import CPU
    r = fresh_model()
    v = gb.baroclinic_instability_model(CPU("f64"), NX, NY, NZ, dt=DT)
    gb.set_baroclinic_instability(v)
    v.set(u=1e-3 * counter_rng((NX, NY, NZ), 42, 1), v=1e-3 * counter_rng((NX, NY + 1, NZ), 42, 2))
    for n in ("u", "v", "T", "S"):
        a = v.backend.get_field(n, False).astype(np.float32)
        r.backend.set_field(n, a, False)
        v.backend.set_field(n, a.astype(np.float64), False)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 1)
    rep = assert_states_close(r, v, include_halos=False, label="1440x720x48 after 2 steps")
    assert max(q["rel"] for q in rep) < 3.4527e-4
    r.backend.close()
    v.backend.close()


def test_lookaheads_bitwise_at_full_size():
    """The look-aheads (tracers, velocities, sub-cycle beside the tracer kernel) against the stand-alone kernels at the
    benchmark size, where the concurrent kernels really do overlap for milliseconds: 8 steps, every prognostic field and
    tendency bit for bit."""
    a = fresh_model(ab2_lookahead=0)
    b = fresh_model(w_on_the_fly=0)        # (w on the fly changes the last bits by design)
    u0 = (1e-2 * counter_rng((NX, NY, NZ), 42, 1)).astype(np.float32)
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        m.set(u=u0)
        gb.first_time_step(m)
        gb.loop(m, 7)
    for n in ("u", "v", "w", "T", "S", "eta", "U", "V", "eta_bar", "U_bar", "Gn.u", "Gn.v", "Gn.T", "Gn.S", "Gm.u",
              "Gm.T", "Gn.U", "Gn.V"):
        x = a.backend.get_field(n, True)
        assert np.array_equal(x, b.backend.get_field(n, True)), n
        assert np.isfinite(x).all(), n
    a.backend.close()
    b.backend.close()


def test_slabs_at_the_benchmark_size_bitwise():
    """BASELINE config 3 as the multi-GPU run cuts it: 1440x720x48 as P = 2, 4, 8 x-slabs (720, 360, 180 columns; the
    8-way slab has a ragged 52-column edge tile and a 22-column wide barotropic halo on 180 columns), stepped by the
    library's sequencer with the local transport on this one GPU.  first_time_step! + 4 steps, every compared field bit
    for bit against the single domain; the look-ahead route was taken."""
    from gb25_amd.distributed import LocalSlabEnsemble
    names = ("u", "v", "w", "T", "S", "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.u", "Gn.v", "Gn.T", "Gn.S",
             "Gm.u", "Gm.T", "Gn.U", "Gn.V")
    # (bit for bit: the slabs compute w with the stand-alone kernel, and their narrower tendency launches march chunks of 12
    # levels where the wide single domain's default is 24: the association of the column integrals of u, v)
    single = fresh_model(w_on_the_fly=0, momentum_chunk_levels=12, tracer_chunk_levels=12)
    gb.set_baroclinic_instability(single)
    u0 = (1e-2 * counter_rng((NX, NY, NZ), 42, 1)).astype(np.float32)
    v0 = (1e-2 * counter_rng((NX, NY + 1, NZ), 42, 2)).astype(np.float32)
    single.set(u=u0, v=v0)
    T0, S0 = single.tracers.T.interior, single.tracers.S.interior
    gb.first_time_step(single)
    gb.loop(single, 4)
    ref = {n: single.backend.get_field(n, False) for n in names}
    single.backend.close()
    assert np.abs(ref["u"]).max() > 1e-2 and np.isfinite(ref["Gn.u"]).all()
    for P in (2, 4, 8):
        ens = LocalSlabEnsemble(NX, NY, NZ, P, dt=DT, options=dict(w_on_the_fly=0, momentum_chunk_levels=12, tracer_chunk_levels=12))
        for n, a in (("u", u0), ("v", v0), ("T", T0), ("S", S0)):
            ens.scatter(n, a)
        ens.first_time_step()
        ens.loop(4)
        assert all(b.lookahead_state() == (True, True) for b in ens.backends), P
        assert all(b.get_option("split_tendencies") == 1 for b in ens.backends)
        for n in names:
            got = ens.gather(n)
            assert np.array_equal(got, ref[n]), (P, n, float(np.abs(got - ref[n]).max()))
        ens.close()


def test_slabs_with_their_default_options_at_the_benchmark_size():
    """What `bench.py --gpus N` runs: the ranks with their DEFAULT options -- the corrector inside its consumers, w on the fly,
    chunks of 24 levels on the 720-column ranks of N = 2 and 12 on the narrower ones -- against the single domain with its
    defaults.  Round-off, not bits (w on the fly associates the vertical sum by chunks, and the ranks' chunk bases come from
    exchanged sums): every field within 2e-5 of its norm after first_time_step! + 5 steps, finite, the look-ahead route taken."""
    from gb25_amd.distributed import LocalSlabEnsemble
    names = ("u", "v", "w", "T", "S", "eta", "U", "V", "Gn.u", "Gn.T")
    single = fresh_model()
    gb.set_baroclinic_instability(single)
    u0 = (1e-2 * counter_rng((NX, NY, NZ), 42, 1)).astype(np.float32)
    v0 = (1e-2 * counter_rng((NX, NY + 1, NZ), 42, 2)).astype(np.float32)
    single.set(u=u0, v=v0)
    T0, S0 = single.tracers.T.interior, single.tracers.S.interior
    gb.first_time_step(single)
    gb.loop(single, 5)
    assert single.backend.get_option("momentum_chunk_levels") == 24
    ref = {n: single.backend.get_field(n, False) for n in names}
    single.backend.close()
    for P, chunk in ((2, 24), (8, 12)):
        ens = LocalSlabEnsemble(NX, NY, NZ, P, dt=DT)
        for n, a in (("u", u0), ("v", v0), ("T", T0), ("S", S0)):
            ens.scatter(n, a)
        ens.first_time_step()
        ens.loop(5)
        assert all(b.get_option("momentum_chunk_levels") == chunk and b.get_option("tracer_chunk_levels") == chunk for b in ens.backends)
        assert all(b.lookahead_state() == (True, True) for b in ens.backends), P
        for n in names:
            got = ens.gather(n).astype(np.float64)
            assert np.isfinite(got).all(), (P, n)
            d = np.linalg.norm((got - ref[n]).ravel()) / np.linalg.norm(ref[n].astype(np.float64).ravel())
            assert d < 2e-5, (P, n, d)
        ens.close()
