"""Decomposition invariance: P x-slabs stepped in lock-step on ONE GPU by the library's
own sequencer -- the same stages, pack / unpack kernels, two streams and interior/edge split of the momentum tendencies
as the multi-process path; only the transport differs (device-to-device copies, or RCCL on the one-rank self-ring
instead of RCCL between ranks) -- must reproduce the single-domain run BIT FOR BIT."""
import numpy as np
import pytest

import gb25_amd as gb
from gb25_amd.distributed import LocalSlabEnsemble
from helpers import counter_rng

pytestmark = pytest.mark.gpu

FIELDS = ["u", "v", "w", "T", "S", "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.u", "Gn.v", "Gn.T", "Gn.S",
          "Gm.u", "Gm.v", "pHY"]
# Bit-for-bit comparisons run the slabs with w from the stand-alone kernel, as the small single domains of these tests compute
# it: w carried inside the tendency kernels (the default beside the corrector inside its consumers) is another association of
# the vertical sum -- the same numbers to round-off, test_w_on_the_fly_on_slabs.
EXACT = dict(w_on_the_fly=0)


def _initial(Nx, Ny, Nz, single):
    gb.set_baroclinic_instability(single)
    u0 = (1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32)
    v0 = (1e-2 * counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(np.float32)
    e0 = (1e-2 * counter_rng((Nx, Ny, 1), 42, 3)).astype(np.float32)
    single.set(u=u0, v=v0, eta=e0)
    return {n: single.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta")}


@pytest.mark.parametrize("P,Nz", [(2, 8), (4, 8), (2, 24), (4, 36)])
def test_slabs_reproduce_single_domain_bitwise(P, Nz):
    # Nz >= 24: the column integrals are summed in chunks of levels (the corrector of a halo column must use the
    # owner's association)
    Nx, Ny, dt = 128, 48, 600.0
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
    init = _initial(Nx, Ny, Nz, single)
    ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, options=EXACT)
    for n, a in init.items():
        ens.scatter(n, a)
    gb.first_time_step(single)
    ens.first_time_step()
    for n in FIELDS:
        assert np.array_equal(ens.gather(n), single.backend.get_field(n, False)), ("first step", n)
    gb.loop(single, 6)
    ens.loop(6)
    for n in FIELDS:
        a, b = ens.gather(n), single.backend.get_field(n, False)
        assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
    assert np.abs(single.velocities.u.interior).max() > 1e-2      # a developed, non-trivial flow
    # the staged path really took the look-ahead route: the last stage 0 adopted the sub-cycle prepared beside the
    # previous tracer kernel, and the next one is already prepared
    assert all(b.lookahead_state() == (True, True) for b in ens.backends)
    # halo columns of a slab equal the neighbour's interior columns (what the exchange + extended corrector produce)
    H = 8
    left, right = ens.backends[0], ens.backends[1]
    for n in ("u", "v", "T", "w", "pHY"):
        a, b = left.get_field(n, True), right.get_field(n, True)
        assert np.array_equal(a[-H:-1, H:-H, H:-H], b[H:2 * H - 1, H:-H, H:-H]), n


@pytest.mark.parametrize("float_type", ["Float32", "Float64"])
def test_slabs_fall_back_when_a_lookahead_is_not_adopted(float_type):
    """A changed dt (and a host write) voids the look-aheads that are already in flight on the second stream: the next
    step must take the in-step route (groups 1, 2 and stage 1) and later return to the look-ahead route, bit for bit
    like the single domain.  Also run through the Float64 library."""
    Nx, Ny, Nz, P, dt = 128, 48, 24, 2, 600.0
    dtype = np.float64 if float_type == "Float64" else np.float32
    single = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=dt)
    init = _initial(Nx, Ny, Nz, single)
    ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, float_type=float_type, options=EXACT)
    for n, a in init.items():
        ens.scatter(n, a.astype(dtype))
    gb.first_time_step(single)
    ens.first_time_step()
    gb.loop(single, 3)
    ens.loop(3)
    assert all(b.lookahead_state() == (True, True) for b in ens.backends)
    single.backend.set_dt(450.0)
    for b in ens.backends:
        b.set_dt(450.0)
    gb.time_step(single)
    ens.time_step()
    assert all(b.lookahead_state()[1] is False for b in ens.backends)      # this step ran its sub-cycle itself
    gb.loop(single, 2)
    ens.loop(2)
    assert all(b.lookahead_state() == (True, True) for b in ens.backends)   # and the look-ahead route is back
    S = single.backend.get_field("S", False) + dtype(0.125)
    single.backend.set_field("S", S, False)
    ens.scatter("S", S)
    gb.loop(single, 3)
    ens.loop(3)
    for n in FIELDS:
        a, b = ens.gather(n), single.backend.get_field(n, False)
        assert a.dtype == dtype and np.array_equal(a, b), (n, float(np.abs(a - b).max()))


@pytest.mark.parametrize("split", [1, 0])
def test_rccl_self_ring_equals_periodic_domain(split):
    """The RCCL transport on the one-GPU box: ONE rank whose west and east neighbour is itself (slab_mode = 1,
    ncclCommInitRank with nranks = 1, ncclSend/ncclRecv to self inside a group on the library's two streams).  The slab
    path with its exchanges must equal the periodic single domain bit for bit -- with and without the interior/edge
    split of the momentum tendencies -- through the look-ahead route and the in-step fallback (changed dt)."""
    from gb25_amd.distributed import SlabModel
    Nx, Ny, Nz, dt = 256, 48, 24, 600.0
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
    init = _initial(Nx, Ny, Nz, single)
    ring = SlabModel(Nx, Ny, Nz, dt=dt, rank=0, nranks=1, slab_mode=1, transport="rccl",
                     options=dict(EXACT, split_tendencies=split))
    for n, a in init.items():
        ring.backend.set_field(n, a, False)
    for m in (single, ring):
        gb.first_time_step(m)
        gb.loop(m, 5)
    assert ring.backend.lookahead_state() == (True, True)
    for m in (single, ring):
        m.backend.set_dt(450.0)
        gb.loop(m, 3)
    for n in FIELDS:
        a, b = ring.backend.get_field(n, False), single.backend.get_field(n, False)
        assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
    # the x halos of the self-ring hold what the periodic copy of the single domain holds
    for n in ("u", "v", "T", "S", "eta", "U", "V"):
        a, b = ring.backend.get_field(n, True), single.backend.get_field(n, True)
        assert np.array_equal(a[:, 8:-8], b[:, 8:-8]), n
    assert np.abs(single.velocities.u.interior).max() > 1e-2
    ring.backend.close()
    single.backend.close()


def test_split_tendencies_is_bitwise_neutral_on_ragged_slabs():
    """Interior tile columns first, edge tile columns after the halos arrived against the unsplit kernels:
    slabs of 160 columns (2.5 tiles), 129 (the interior shrinks to one tile) and 66 (no interior at all)."""
    for Nx, P in ((480, 3), (258, 2), (132, 2)):
        Ny, Nz, dt = 40, 12, 600.0
        ens = [LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, options=dict(EXACT, split_tendencies=sp)) for sp in (0, 1)]
        single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
        init = _initial(Nx, Ny, Nz, single)
        for e in ens:
            for n, a in init.items():
                e.scatter(n, a)
        for m in ens + [single]:
            m.first_time_step() if m is not single else gb.first_time_step(single)
            m.loop(4) if m is not single else gb.loop(single, 4)
        for n in FIELDS:
            ref = single.backend.get_field(n, False)
            for e in ens:
                assert np.array_equal(e.gather(n), ref), (Nx, P, n)
        for e in ens:
            e.close()
        single.backend.close()


def test_state_dump_through_the_abi_and_offline_gather(tmp_path):
    """save_model_state / load_all_fields (src/sharded_io.jl:122-138,198-213): every slab writes its own
    fields_rank<R>.npz through gb25_save_state (no communication); the offline gather reproduces the single domain."""
    Nx, Ny, Nz, P, dt = 128, 48, 8, 4, 600.0
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
    init = _initial(Nx, Ny, Nz, single)
    ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, options=EXACT)
    for n, a in init.items():
        ens.scatter(n, a)
    gb.first_time_step(single)
    ens.first_time_step()
    gb.loop(single, 3)
    ens.loop(3)
    for b in ens.backends:
        path = b.save_state(str(tmp_path), "after_loop")
        assert path.endswith(f"fields_rank{b.cfg.rank}.npz")
    got = gb.load_all_fields(str(tmp_path / "after_loop"))
    assert got["iteration"] == 4 and got["time"] == 4 * dt
    for n in ("u", "v", "w", "eta", "T", "S"):
        assert np.array_equal(got[n], single.backend.get_field(n, False)), n
    # the single domain through the same entry point
    gb.save_model_state(str(tmp_path), single, label="single")
    one = gb.load_all_fields(str(tmp_path / "single"))
    assert np.array_equal(one["T"], got["T"]) and one["iteration"] == 4
    ens.close()


def test_w_on_the_fly_on_slabs():
    """The default schedule of a slab in steady state: the corrector inside its consumers and w carried inside the tendency
    kernels (no k_corrector sweep, no k_compute_w launch; the chunk bases of w next to the x halos from the chunk sums the bundle
    carries).  Against the same slabs with w from the stand-alone kernel: the same numbers to round-off (another association of
    the vertical sum), every field -- and the field w itself, recomputed when the call returns, from velocities that agree."""
    Nx, Ny, Nz, P, dt = 384, 48, 36, 2, 600.0      # three chunks of levels, three tile columns per slab (interior + edges)
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt)
    init = _initial(Nx, Ny, Nz, single)
    single.backend.close()
    out = {}
    for fly in (1, 0):
        ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, options=dict(w_on_the_fly=fly))
        for n, a in init.items():
            ens.scatter(n, a)
        ens.first_time_step()
        ens.loop(12)
        assert all(b.lookahead_state() == (True, True) for b in ens.backends)
        out[fly] = {n: ens.gather(n).astype(np.float64) for n in FIELDS}
        ens.close()
    worst = {}
    for n in FIELDS:
        a, b = out[1][n], out[0][n]
        worst[n] = float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
    assert max(worst.values()) < 2e-5, worst
    assert any(v > 0 for v in worst.values()), "w on the fly did not run on the slabs"   # (else the two runs are the same bits)
