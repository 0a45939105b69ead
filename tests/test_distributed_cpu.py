"""The N>1 path on CPU: the ring-exchange protocol under torch.distributed (gloo, world_size 2 and 3) and the
order in which the LIBRARY sequences one time step of a decomposition into stages, packs, exchanges, unpacks and stream
dependencies (a dry run of csrc/slab_step.hpp's sequencer through the C ABI: no GPU needed)."""
import ctypes
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gb25_amd.binding import load_library
from gb25_amd.distributed import TorchDistributedTransport
from gb25_amd.sharding import slab_neighbours


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ring_worker(rank, world, port, n, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = TorchDistributedTransport(rank, world)
        for rep in range(3):     # repeated exchanges must keep matching
            sw = torch.full((n,), rank * 100 + 10 + rep, dtype=torch.float32)
            se = torch.full((n,), rank * 100 + 20 + rep, dtype=torch.float32)
            rw, re = torch.empty(n), torch.empty(n)
            t.exchange(sw, se, rw, re)
            west, east = slab_neighbours(rank, world)
            assert torch.all(rw == west * 100 + 20 + rep), (rank, rep, rw[0].item())   # west halo <- west nbr's EAST pack
            assert torch.all(re == east * 100 + 10 + rep), (rank, rep, re[0].item())   # east halo <- east nbr's WEST pack
        results[rank] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ring_exchange_gloo(world):
    """world_size 2 is the degenerate ring: both neighbours are the same peer, so matching relies on the
    posting order documented in TorchDistributedTransport."""
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, 1000, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(results.get(r) for r in range(world))


def _sequence(nslabs, first=False, adopted=False, ready=False):
    """The library's own sequencing of a time step as a dry run (gb25_debug_sequence: no GPU is touched)."""
    lib = load_library("Float32")
    need = lib.gb25_debug_sequence(nslabs, int(first), int(adopted), int(ready), None, 0)
    buf = ctypes.create_string_buffer(need)
    assert lib.gb25_debug_sequence(nslabs, int(first), int(adopted), int(ready), buf, need) == need
    return [tuple(int(t) if t.lstrip("-").isdigit() else t for t in line.split()) for line in buf.value.decode().splitlines()]


def _ops_of_slab(log, slab):
    """(op, number, stream) of one slab, exchanges included (they involve every slab)."""
    out = []
    for e in log:
        if e[0] in ("stage", "pack", "unpack") and e[3] == slab:
            out.append((e[0], e[1], e[-1]))
        elif e[0] == "exchange":
            out.append(e)
    return out


def test_time_step_sequencing_in_step_subcycle():
    """No look-ahead is valid (first steps, changed dt, host writes): the sub-cycle and its exchange run inside the step.
    The slab is widened by Ns + 1 + H columns, so the sub-cycle leaves the x halo columns of the new eta, U, V behind as
    well: nothing is exchanged after it (no group 2)."""
    log = _sequence(4)
    mine = _ops_of_slab(log, 2)
    assert mine == [("stage", 0, "main"),
                    ("pack", 1, "main"), ("exchange", 1, "main"),          # small barotropic exchange FIRST (critical path)
                    ("pack", 0, "comm"), ("exchange", 0, "comm"),          # the 3-D bundle leaves on the second stream
                    ("unpack", 1, "main"), ("stage", 1, "main"),
                    ("stage", 2, "main"),                                   # own columns + interior tendencies meanwhile
                    ("unpack", 0, "main"), ("stage", 3, "main"), ("stage", 4, "main")]
    idx = lambda *e: log.index(e)
    # the comm stream starts packing the bundle only after stage 0 of every slab (event 0), the corrector of stage 2
    # waits for the pack (event 1), the unpack of group 0 for the arrival (event 3 recorded on comm)
    assert idx("record", 0, "main") > idx("stage", 0, "slab", 3, "euler", 0, "main")
    assert idx("wait", 0, "comm") < idx("pack", 0, "slab", 0, "comm")
    assert idx("record", 1, "comm") < idx("exchange", 0, "comm")
    assert idx("wait", 1, "main") < idx("stage", 2, "slab", 0, "euler", 0, "main")
    last_stage2 = idx("stage", 2, "slab", 3, "euler", 0, "main")
    assert last_stage2 < idx("record", 3, "comm") < idx("wait", 3, "main") < idx("unpack", 0, "slab", 0, "main")
    assert log[-1] == ("lookahead_in_flight", 0)
    assert not any(e[:2] in (("exchange", 2), ("exchange", 4)) for e in log)
    # every slab packs a group before its exchange and unpacks after
    for grp in (0, 1):
        ex = [i for i, e in enumerate(log) if e[:2] == ("exchange", grp)]
        assert len(ex) == 1
        assert all(i < ex[0] for i, e in enumerate(log) if e[:2] == ("pack", grp))
        assert all(i > ex[0] for i, e in enumerate(log) if e[:2] == ("unpack", grp))


def test_time_step_sequencing_on_a_folded_grid():
    """Tripolar grid in slabs: the rows beyond a slab's pivot row belong to the mirrored rank.  The sub-cycle's work arrays are
    tall as well as wide: after the wide-halo exchange and the interior copy (stage 1) the rows south of the pivot row go to
    the partner ONCE ("exchange 8", buffer set 4), then all substeps run without any exchange (stage 16); the rows next to the
    pivot row of u, v, T, S, eta, U, V travel once per step ("exchange 6") after the x halos and y/z layers of EVERY slab are
    in place (the partner sends its halo columns too) and before w, pressure and the tendencies."""
    lib = load_library("Float32")
    need = lib.gb25_debug_sequence(3, 2, 0, 0, None, 0)
    buf = ctypes.create_string_buffer(need)
    lib.gb25_debug_sequence(3, 2, 0, 0, buf, need)
    log = [tuple(int(t) if t.lstrip("-").isdigit() else t for t in line.split()) for line in buf.value.decode().splitlines()]
    mine = _ops_of_slab(log, 1)
    assert mine == [("stage", 0, "main"), ("pack", 1, "main"), ("exchange", 1, "main"),
                    ("pack", 0, "comm"), ("exchange", 0, "comm"),
                    ("unpack", 1, "main"), ("stage", 1, "main"), ("pack", 8, "main"), ("exchange", 8, "main"),
                    ("unpack", 8, "main"), ("stage", 16, "main"),
                    ("stage", 2, "main"),
                    ("unpack", 0, "main"), ("stage", 30, "main"), ("pack", 6, "main"),
                    ("exchange", 6, "main"), ("unpack", 6, "main"), ("stage", 31, "main"), ("stage", 4, "main")]
    assert not any(e[:2] in (("exchange", 7), ("exchange", 2)) for e in log)
    # ... and with the look-aheads: the NEXT step's sub-cycle on the second stream, its two exchanges included
    need = lib.gb25_debug_sequence(3, 2, 1, 1, None, 0)
    buf = ctypes.create_string_buffer(need)
    lib.gb25_debug_sequence(3, 2, 1, 1, buf, need)
    ahead = [tuple(int(t) if t.lstrip("-").isdigit() else t for t in line.split()) for line in buf.value.decode().splitlines()]
    mine = _ops_of_slab(ahead, 1)
    assert mine == [("stage", 0, "main"), ("pack", 0, "comm"), ("exchange", 0, "comm"), ("stage", 2, "main"),
                    ("unpack", 0, "main"), ("stage", 30, "main"), ("pack", 6, "main"), ("exchange", 6, "main"),
                    ("unpack", 6, "main"), ("stage", 31, "main"),
                    ("pack", 3, "comm"), ("exchange", 3, "comm"), ("unpack", 3, "comm"), ("stage", 5, "comm"),
                    ("pack", 8, "comm"), ("exchange", 8, "comm"), ("unpack", 8, "comm"), ("stage", 56, "sub"),
                    ("stage", 4, "main")]
    assert ahead[-1] == ("lookahead_in_flight", 1)
    # every slab has packed its rows before the partner exchange and no slab unpacks before it
    ex = [i for i, e in enumerate(log) if e[:2] == ("exchange", 6)]
    assert len(ex) == 1
    assert all(i < ex[0] for i, e in enumerate(log) if e[:2] == ("pack", 6))
    assert all(i > ex[0] for i, e in enumerate(log) if e[:2] == ("unpack", 6))
    assert not any(e[:2] == ("stage", 5) for e in log) and log[-1] == ("lookahead_in_flight", 0)
    # first_time_step!: mask + y/z layers of every slab, the fold exchange, then auxiliaries and tendencies
    need = lib.gb25_debug_sequence(2, 3, 0, 0, None, 0)
    buf = ctypes.create_string_buffer(need)
    lib.gb25_debug_sequence(2, 3, 0, 0, buf, need)
    names = [line.split()[0] for line in buf.value.decode().splitlines()]
    first = names[:names.index("stage")]
    assert first.index("mask_fill_local") < first.index("auxiliaries_tendencies_local")
    assert "update_state_local" not in first


def test_time_step_sequencing_with_the_subcycle_lookahead():
    """When the previous step left a valid look-ahead, stage 0 adopts the sub-cycle: groups 1, 2 and stage 1 vanish
    from the step; after the momentum tendencies (stage 3) the NEXT sub-cycle is prepared beside the tracer
    tendencies: group 3 on the second stream, then stage 5 -- the substeps -- on a third one, so that the next step's bundle,
    posted on the second stream right behind stage 0, does not queue behind five sub-cycle launches."""
    log = _sequence(3, adopted=True, ready=True)
    mine = _ops_of_slab(log, 1)
    assert mine == [("stage", 0, "main"), ("pack", 0, "comm"), ("exchange", 0, "comm"), ("stage", 2, "main"),
                    ("unpack", 0, "main"), ("stage", 3, "main"),
                    ("pack", 3, "comm"), ("exchange", 3, "comm"), ("unpack", 3, "comm"), ("stage", 5, "sub"),
                    ("stage", 4, "main")]      # (no group 4: the widened sub-cycle leaves the x halo columns behind too)
    assert [e[1] for e in log if e[0] == "exchange"] == [0, 3]
    assert log[-1] == ("lookahead_in_flight", 1)
    # the look-ahead starts after the momentum tendencies of every slab (event 2 recorded on main, awaited by comm)
    i_mom = max(i for i, e in enumerate(log) if e[:2] == ("stage", 3))
    i_rec = max(i for i, e in enumerate(log) if e == ("record", 2, "main"))
    i_wait = max(i for i, e in enumerate(log) if e == ("wait", 2, "comm"))
    i_pack3 = min(i for i, e in enumerate(log) if e[:2] == ("pack", 3))
    assert i_mom < i_rec < i_wait < i_pack3
    # ... the substeps behind the unpacked wide halos (event 5: comm -> sub), and the chain's end is event 4 on the sub stream
    i_un3 = max(i for i, e in enumerate(log) if e[:2] == ("unpack", 3))
    i_st5 = min(i for i, e in enumerate(log) if e[:2] == ("stage", 5))
    assert i_un3 < log.index(("record", 5, "comm")) < log.index(("wait", 5, "sub")) < i_st5 < log.index(("record", 4, "sub"))


def test_first_time_step_sequencing():
    log = _sequence(2, first=True)
    mine = [e for e in log if (len(e) > 3 and e[2] == "slab" and e[3] == 0) or e[0] == "exchange"
            or (e[0] in ("initialize", "fill_local", "update_state_local") and e[2] == 0)]
    assert [e[:2] for e in mine[:9]] == [("initialize", "slab"), ("fill_local", "slab"), ("pack", 0), ("pack", 2),
                                          ("exchange", 0), ("exchange", 2), ("unpack", 0), ("unpack", 2),
                                          ("update_state_local", "slab")]
    stages = [e for e in log if e[0] == "stage" and e[3] == 0]
    assert [e[1] for e in stages] == [0, 1, 2, 3, 4] and all(e[5] == 1 for e in stages)      # Euler first step


def _raw_sequence(nslabs, first, adopted=0, ready=0):
    lib = load_library("Float32")
    need = lib.gb25_debug_sequence(nslabs, first, adopted, ready, None, 0)
    buf = ctypes.create_string_buffer(need)
    lib.gb25_debug_sequence(nslabs, first, adopted, ready, buf, need)
    return [tuple(int(t) if t.lstrip("-").isdigit() else t for t in line.split()) for line in buf.value.decode().splitlines()]


def test_first_time_step_of_a_coupled_model():
    """Data-free forcing on slabs: after the ordinary update_state! every slab computes the atmosphere-ocean fluxes of the
    initial state (the first halo column's are COMPUTED, nothing travels), then the tendencies that see them; only then the
    Euler step.  With closure = CATKE every update_state! has an exchange of its own inside: e is stepped and J^b filtered on
    the own columns (compute_diffusivities!), their halos travel (group 20; rows 21 on a mesh, the fold partner 22), then the
    diffusivities of the first halo column and the slow tendency of e."""
    log = _raw_sequence(2, 1 | 4)
    names = [e[0] for e in log]
    i_state = names.index("update_state_local")
    i_flux = names.index("first_fluxes_local")
    i_tend = names.index("tendencies_local")
    ex0 = [i for i, e in enumerate(log) if e[:2] == ("exchange", 0)]
    assert i_state < i_flux < i_tend and len(ex0) == 2                    # initial halos, the Euler step's bundle
    assert i_tend < ex0[1] and not [e for e in log if e[:2] == ("exchange", 20)]
    plain = _raw_sequence(2, 1)
    assert "first_fluxes_local" not in [e[0] for e in plain]
    # the same with CATKE (bit 7): three update_state! (initial, coupled iteration 0, the Euler step), each with its exchange
    log = _raw_sequence(2, 1 | 4 | 128)
    ex20 = [i for i, e in enumerate(log) if e[:2] == ("exchange", 20)]
    fin = [i for i, e in enumerate(log) if e[0] == "catke_finish_local" and e[2] == 0] + \
          [i for i, e in enumerate(log) if e[:4] == ("stage", 41, "slab", 0)]
    assert len(ex20) == 3 and len(fin) == 3 and all(a < b for a, b in zip(ex20, fin))
    assert log.index(("stage", 4, "slab", 1, "euler", 1, "main")) < ex20[2] < log.index(("stage", 41, "slab", 0, "euler", 1, "main"))
    # a folded 2-D decomposition: columns, then whole rows (the corners ride along), then the fold partner
    log = _raw_sequence(2, 2 | 16 | 128, adopted=1, ready=1)
    order = [e[1] for e in log if e[0] == "exchange" and e[1] >= 20]
    assert order == [20, 21, 22]


def test_a_step_behind_a_look_ahead_chain_in_flight():
    """The previous step left its look-ahead chain (group 3, stage 5) on the second stream: stage 0 does not wait for it;
    the corrector (stage 2) does, through the event recorded behind the chain -- or, when the look-ahead is not adopted after
    all, everything that reuses its buffers does."""
    log = _raw_sequence(3, 8, adopted=1, ready=1)
    idx = lambda *e: log.index(e)
    w = idx("wait", 4, "main")
    assert idx("stage", 0, "slab", 2, "euler", 0, "main") < w < idx("stage", 2, "slab", 0, "euler", 0, "main")
    assert idx("exchange", 0, "comm") < w                                  # the bundle is on its way by then
    assert ("record", 4, "sub") in log and log[-1] == ("lookahead_in_flight", 1)
    log = _raw_sequence(3, 8, adopted=0, ready=0)
    idx = lambda *e: log.index(e)
    assert idx("stage", 0, "slab", 2, "euler", 0, "main") < idx("wait", 4, "main") < idx("pack", 1, "slab", 0, "main")


def _mesh_worker(rank, Rx, Ry, port, n, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    world = Rx * Ry
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gb25_amd.sharding import mesh_neighbours
        t = TorchDistributedTransport(rank, world, ranks_y=Ry)
        nb = mesh_neighbours(rank, Rx, Ry)
        for rep in range(2):
            # x: the ring within the row
            sw = torch.full((n,), rank * 100 + 10 + rep, dtype=torch.float32)
            se = torch.full((n,), rank * 100 + 20 + rep, dtype=torch.float32)
            rw, re = torch.empty(n), torch.empty(n)
            t.exchange(sw, se, rw, re)
            assert torch.all(rw == nb["west"] * 100 + 20 + rep) and torch.all(re == nb["east"] * 100 + 10 + rep), rank
            # y: south / north where they exist (no wrap: walls / the fold)
            ss = torch.full((n,), rank * 100 + 30 + rep, dtype=torch.float32) if nb["south"] is not None else None
            sn = torch.full((n,), rank * 100 + 40 + rep, dtype=torch.float32) if nb["north"] is not None else None
            rs = torch.empty(n) if nb["south"] is not None else None
            rn = torch.empty(n) if nb["north"] is not None else None
            t.exchange_y(ss, sn, rs, rn)
            if rs is not None:
                assert torch.all(rs == nb["south"] * 100 + 40 + rep), (rank, rs[0].item())   # southern halo <- the NORTHERN pack of the rank below
            if rn is not None:
                assert torch.all(rn == nb["north"] * 100 + 30 + rep), (rank, rn[0].item())
            # the fold partner: mirrored in x within the row
            sp = torch.full((n,), rank * 100 + 50 + rep, dtype=torch.float32)
            rp = torch.empty(n)
            t.exchange_partner(sp, rp)
            assert torch.all(rp == nb["partner"] * 100 + 50 + rep), rank
        results[rank] = True
    finally:
        dist.destroy_process_group()


def test_mesh_exchange_gloo_world_size_4():
    """Partition(2, 2, 1): rank = ry Rx + rx; the x ring closes within a row, y neighbours exist only inside the domain, the
    fold partner is the mirrored rank of the same row."""
    from gb25_amd.sharding import mesh_neighbours
    assert mesh_neighbours(0, 2, 2) == dict(west=1, east=1, south=None, north=2, partner=1)
    assert mesh_neighbours(5, 4, 2) == dict(west=4, east=6, south=1, north=None, partner=6)
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_mesh_worker, args=(r, 2, 2, port, 500, results)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(results.get(r) for r in range(4))


def test_time_step_sequencing_of_a_2d_decomposition():
    """Partition(Rx, Ry, 1): every exchange in x is followed by the exchange of whole ROWS with the southern / northern
    neighbour, which then carry the x halo columns just received (the corners of the diagonal neighbours): the work arrays of
    the sub-cycle after their wide halo columns and the interior copy ("exchange 11" after group 1 and stage 1, the substeps in
    stage 16), the 3-D bundle after group 0 AND the barotropic corrector of the own rows' x halo columns (stage 32), so that
    the rows arrive corrected ("exchange 10"); on the top row of a tripolar mesh the fold partner's groups 8 and 6 ride along."""
    log = _raw_sequence(4, 16)
    mine = _ops_of_slab(log, 2)
    assert mine == [("stage", 0, "main"), ("pack", 1, "main"), ("exchange", 1, "main"),
                    ("pack", 0, "comm"), ("exchange", 0, "comm"),
                    ("unpack", 1, "main"), ("stage", 1, "main"), ("pack", 11, "main"), ("exchange", 11, "main"),
                    ("unpack", 11, "main"), ("stage", 16, "main"),
                    ("stage", 2, "main"),
                    ("unpack", 0, "main"), ("stage", 32, "main"), ("pack", 10, "main"), ("exchange", 10, "main"),
                    ("unpack", 10, "main"), ("stage", 3, "main"), ("stage", 4, "main")]
    for grp in (10, 11):     # every slab has packed before the exchange, none unpacks before it
        ex = [i for i, e in enumerate(log) if e[:2] == ("exchange", grp)]
        assert len(ex) == 1
        assert all(i < ex[0] for i, e in enumerate(log) if e[:2] == ("pack", grp))
        assert all(i > ex[0] for i, e in enumerate(log) if e[:2] == ("unpack", grp))
    # tripolar mesh: the partner's groups between the y exchange and the rest of update_state!
    fold = _ops_of_slab(_raw_sequence(4, 16 | 2), 3)
    assert fold == [("stage", 0, "main"), ("pack", 1, "main"), ("exchange", 1, "main"),
                    ("pack", 0, "comm"), ("exchange", 0, "comm"),
                    ("unpack", 1, "main"), ("stage", 1, "main"), ("pack", 11, "main"), ("pack", 8, "main"),
                    ("exchange", 11, "main"), ("exchange", 8, "main"), ("unpack", 11, "main"), ("unpack", 8, "main"),
                    ("stage", 16, "main"), ("stage", 2, "main"),
                    ("unpack", 0, "main"), ("stage", 32, "main"), ("pack", 10, "main"), ("exchange", 10, "main"),
                    ("unpack", 10, "main"), ("stage", 30, "main"), ("pack", 6, "main"), ("exchange", 6, "main"),
                    ("unpack", 6, "main"), ("stage", 31, "main"), ("stage", 4, "main")]
    # first_time_step!: the initial state's rows follow its columns
    first = _raw_sequence(4, 16 | 1)
    names = [(e[0], e[1]) for e in first if e[0] in ("exchange",)]
    assert names[:4] == [("exchange", 0), ("exchange", 2), ("exchange", 10), ("exchange", 12)]


def test_time_step_sequencing_of_a_lazy_step():
    """A steady-state slab step keeps the barotropic corrector inside its consumers: du, dv and the chunk bases of w of the own
    columns (stage 20) write nothing the bundle is packed from, so they run as soon as the adopted sub-cycle is complete -- ahead
    of the wait for the packed bundle (event 1), which the interior momentum pass (stage 2) still honours."""
    log = _raw_sequence(3, 8 | 32, adopted=1, ready=1)
    idx = lambda *e: log.index(e)
    assert idx("wait", 4, "main") < idx("stage", 20, "slab", 0, "euler", 0, "main") < idx("stage", 20, "slab", 2, "euler", 0, "main") \
        < idx("wait", 1, "main") < idx("stage", 2, "slab", 0, "euler", 0, "main")
    plain = _raw_sequence(3, 8, adopted=1, ready=1)
    assert not any(e[:2] == ("stage", 20) for e in plain)
    assert plain.index(("wait", 1, "main")) < plain.index(("wait", 4, "main"))


def test_time_step_sequencing_with_the_bundle_unpacked_on_the_exchange_stream():
    """Plain x slabs: the halo columns are unpacked on the exchange stream right behind the transfer of the bundle and the two
    pressure strips next to the x halos (stage 33) follow them there -- beside the own-column work of stage 2 on the main stream,
    which waits for both (event 3) only in front of stage 3."""
    log = _raw_sequence(2, 8 | 32 | 64, adopted=1, ready=1)
    mine = _ops_of_slab(log, 1)
    i = mine.index(("exchange", 0, "comm"))
    assert mine[i + 1:i + 3] == [("unpack", 0, "comm"), ("stage", 33, "comm")]
    assert mine.index(("stage", 33, "comm")) < mine.index(("stage", 20, "main")) < mine.index(("stage", 2, "main")) < mine.index(("stage", 3, "main"))
    assert ("unpack", 0, "main") not in mine
    assert log.index(("record", 3, "comm")) < log.index(("wait", 3, "main")) < log.index(("stage", 3, "slab", 0, "euler", 0, "main"))


def test_time_step_sequencing_of_a_lazy_step_of_a_2d_decomposition():
    """A rank of a 2-D decomposition in its steady state (corrector inside its consumers, bundle unpacked on the exchange stream):
    the rows leave as they are -- stage 32 has nothing to correct -- so their pack, transfer and unpack follow the unpacked bundle on
    the exchange stream; the main stream meets them once (event 3) in front of stage 3, and nothing of group 10 is left on it."""
    log = _raw_sequence(4, 8 | 16 | 32 | 64, adopted=1, ready=1)
    mine = _ops_of_slab(log, 2)
    i = mine.index(("exchange", 0, "comm"))
    assert mine[i + 1:i + 5] == [("unpack", 0, "comm"), ("pack", 10, "comm"), ("exchange", 10, "comm"), ("unpack", 10, "comm")]
    assert not any(e in mine for e in (("pack", 10, "main"), ("unpack", 10, "main"), ("unpack", 0, "main"), ("stage", 32, "main")))
    ex = [k for k, e in enumerate(log) if e[:2] == ("exchange", 10)]
    assert len(ex) == 1
    assert all(k < ex[0] for k, e in enumerate(log) if e[:2] == ("pack", 10))       # every rank packs before the transfer
    assert all(k > ex[0] for k, e in enumerate(log) if e[:2] == ("unpack", 10))
    assert all(k < ex[0] for k, e in enumerate(log) if e[:2] == ("unpack", 0))      # ... with its corners in
    assert ex[0] < log.index(("record", 3, "comm")) < log.index(("wait", 3, "main")) < log.index(("stage", 3, "slab", 0, "euler", 0, "main"))
    # a step with the stand-alone corrector: the rows wait for stage 32 on the main stream, as before
    plain = _ops_of_slab(_raw_sequence(4, 8 | 16 | 64, adopted=1, ready=1), 2)
    j = plain.index(("stage", 32, "main"))
    assert plain[j + 1:j + 4] == [("pack", 10, "main"), ("exchange", 10, "main"), ("unpack", 10, "main")]


def _exchange_plan(Rx, Ry, rank, folded_grid, group):
    lib = load_library("Float32")
    need = lib.gb25_debug_exchange_plan(Rx, Ry, rank, int(folded_grid), group, None, 0)
    assert need > 0
    buf = ctypes.create_string_buffer(need)
    lib.gb25_debug_exchange_plan(Rx, Ry, rank, int(folded_grid), group, buf, need)
    return [(op, int(peer), int(side)) for op, peer, side in (line.split() for line in buf.value.decode().splitlines() if line != "copy")]


@pytest.mark.parametrize("Rx,Ry", [(8, 1), (4, 2), (2, 4), (2, 1), (3, 1), (1, 2), (3, 2)])
@pytest.mark.parametrize("folded_grid", [False, True])
def test_every_rank_sends_what_its_peer_receives(Rx, Ry, folded_grid):
    """The RCCL transport has never run on more than one GPU (no such box): what CAN be proved without one is that the
    point-to-point protocol is consistent.  For every rank of the decomposition the library prints the order of operations of a
    step (gb25_debug_sequence: the same sequencer that drives the GPU) and, per exchange group, the sends and receives that rank
    posts in posting order (gb25_debug_exchange_plan: RcclTransport::exchange's own plan).  NCCL / RCCL match the k-th send of a
    to b with the k-th receive b posts from a.  So for every ordered pair of ranks, over the whole step and stream by stream: as
    many sends as receives, the same exchange groups in the same order, and the packed side of the sender landing in the halo of
    the facing side (my west pack -> your east halo; my southern rows -> your northern halo; fold partners: pack -> image rows).
    Covers x slabs (P = 8, 2 and an odd 3: the middle slab is its own fold partner), the 4 x 2 and 2 x 4 meshes of config 4, a
    1 x 2 mesh (its own west and east neighbour) and 3 x 2, with and without the zipper fold (only the top row of ranks folds),
    first steps, steady steps with both look-aheads, lazy steps, closure = CATKE, the coupled model."""
    P = Rx * Ry
    variants = [dict(first=1), dict(first=0), dict(first=0, adopted=1, ready=1), dict(first=0, adopted=1, ready=1, extra=32 | 64),
                dict(first=0, adopted=1, ready=1, extra=128), dict(first=1, extra=4 | 128), dict(first=0, adopted=0, ready=1, extra=8)]
    for var in variants:
        sends, recvs = {}, {}     # (a, b, stream) -> [(group, side)] in posting order; stream "*" = the host's issue order
        for r in range(P):
            top = r // Rx == Ry - 1
            flags = var["first"] | var.get("extra", 0) | (2 if (folded_grid and top) else 0) | (16 if Ry > 1 else 0)
            log = _raw_sequence(1, flags, var.get("adopted", 0), var.get("ready", 0))
            for e in log:
                if e[0] != "exchange":
                    continue
                group, stream = e[1], e[2]
                for op, peer, side in _exchange_plan(Rx, Ry, r, folded_grid, group):
                    book = sends if op == "send" else recvs
                    key = (r, peer) if op == "send" else (peer, r)          # always (sender, receiver)
                    for st in (stream, "*"):
                        book.setdefault(key + (st,), []).append((group, side))
        assert set(sends) == set(recvs), (var, sorted(set(sends) ^ set(recvs))[:4])
        for key, sent in sends.items():
            got = recvs[key]
            assert [g for g, _ in sent] == [g for g, _ in got], (var, key, sent, got)      # same groups, same order
            for (g, s_side), (_, r_side) in zip(sent, got):
                partner_kind = g in (6, 8, 22)
                assert r_side == (s_side if partner_kind else 1 - s_side), (var, key, g, s_side, r_side)
        # every rank takes part in the ring exchanges; only the top row of a folded grid talks to a fold partner
        assert any(g == 0 for k, v in sends.items() if k[2] == "*" for g, _ in v)
        folds = {k[0] for k, v in sends.items() for g, _ in v if g in (6, 8, 22)}
        assert all(a // Rx == Ry - 1 for a in folds) and (not folds or folded_grid)
