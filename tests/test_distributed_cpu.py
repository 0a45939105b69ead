"""The N>1 path on CPU: the ring-exchange protocol under torch.distributed (gloo, world_size 2 and 3) and the
order in which one slab's time step cuts into stages, packs, exchanges and unpacks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gb25_amd.distributed import (EAST, WEST, LocalRingTransport, SlabStepper, TorchDistributedTransport,
                                  first_step_slabs, step_slabs)
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


class _RecordingBackend:
    """Stands in for HipBackend: records the call order and moves tagged 'columns' through the buffers."""

    def __init__(self, rank, log):
        self.rank, self.log = rank, log
        self.unpacked = {}

    ready = False        # velocity look-ahead of the next step exists (stage 5 may run)
    adopted = False      # stage 0 adopted the sub-cycle look-ahead

    def halo_buffer_elems(self, group): return 4
    def set_stream(self, s): pass
    def lookahead_state(self): return (self.ready, self.adopted)
    def _rec(self, *a): self.log.append((self.rank,) + a)
    def time_step_stage(self, stage, euler=False): self._rec("stage", stage, bool(euler))
    def initialize(self): self._rec("initialize")
    def fill_halo_regions_local(self): self._rec("fill_local")
    def update_state_local(self): self._rec("update_state_local")

    def halo_pack(self, group, side, ptr):
        self._rec("pack", group, side)
        self.bufs_send[group][side].fill_(self.rank * 100 + group * 10 + side)

    def halo_unpack(self, group, side, ptr):
        self._rec("unpack", group, side)
        self.unpacked[(group, side)] = float(self.bufs_recv[group][side][0])

    def halo_pack_both(self, group, west_ptr, east_ptr):        # one launch for both sides in the product
        for side in (WEST, EAST):
            self.halo_pack(group, side, None)

    def halo_unpack_both(self, group, west_ptr, east_ptr):
        for side in (WEST, EAST):
            self.halo_unpack(group, side, None)


def _make_local_ring(P):
    log = []
    backs = [_RecordingBackend(r, log) for r in range(P)]
    steppers = [SlabStepper(b, torch.device("cpu")) for b in backs]
    for b, s in zip(backs, steppers):
        b.bufs_send, b.bufs_recv = s.send, s.recv
    exchange = lambda group: (log.append(("exchange", group)), LocalRingTransport.exchange_all(steppers, group))
    return backs, steppers, exchange, log


def test_time_step_sequencing_and_local_ring():
    P = 4
    backs, steppers, exchange, log = _make_local_ring(P)
    step_slabs(steppers, exchange, euler=False)
    mine = [e[1:] for e in log if e[0] == 2]
    assert mine == [("stage", 0, False), ("pack", 1, WEST), ("pack", 1, EAST), ("pack", 0, WEST), ("pack", 0, EAST),
                    ("unpack", 1, WEST), ("unpack", 1, EAST), ("stage", 1, False), ("pack", 2, WEST), ("pack", 2, EAST),
                    ("stage", 2, False),          # own-column corrector while group 2 (and 0) are in flight
                    ("unpack", 2, WEST), ("unpack", 2, EAST), ("unpack", 0, WEST), ("unpack", 0, EAST),
                    ("stage", 3, False), ("stage", 4, False)]
    # exchanges happen once per group, between the pack of every slab and the unpack of any slab;
    # the small barotropic exchange (group 1, critical path) is posted first, then the 3-D bundle (group 0),
    # which stays in flight during the sub-cycle (stage 1)
    ex = [i for i, e in enumerate(log) if e[0] == "exchange"]
    assert [log[i][1] for i in ex] == [1, 0, 2]
    stage1 = min(i for i, e in enumerate(log) if e[0] != "exchange" and e[1:3] == ("stage", 1))
    unpack0 = min(i for i, e in enumerate(log) if e[0] != "exchange" and e[1:3] == ("unpack", 0))
    assert ex[1] < stage1 < unpack0
    stage2 = min(i for i, e in enumerate(log) if e[0] != "exchange" and e[1:3] == ("stage", 2))
    unpack2 = min(i for i, e in enumerate(log) if e[0] != "exchange" and e[1:3] == ("unpack", 2))
    assert ex[2] < stage2 < unpack2               # group 2 is posted before the own-column corrector starts
    for i, grp in zip(ex, (1, 0, 2)):
        assert all(not (e[1] == "unpack" and e[2] == grp) for e in log[:i] if e[0] != "exchange")
        assert all(not (e[1] == "pack" and e[2] == grp) for e in log[i:] if e[0] != "exchange")
    # data: my west halo holds the west neighbour's EAST pack, my east halo the east neighbour's WEST pack
    for r, b in enumerate(backs):
        west, east = slab_neighbours(r, P)
        for grp in (0, 1, 2):
            assert b.unpacked[(grp, WEST)] == west * 100 + grp * 10 + EAST
            assert b.unpacked[(grp, EAST)] == east * 100 + grp * 10 + WEST


def test_time_step_sequencing_with_the_subcycle_lookahead():
    """When the previous step left a valid look-ahead, stage 0 adopts the sub-cycle: groups 1, 2 and stage 1 vanish
    from the step; after the momentum tendencies (stage 3) the NEXT sub-cycle is prepared beside the tracer
    tendencies: group 3 -> stage 5 -> group 4."""
    P = 3
    backs, steppers, exchange, log = _make_local_ring(P)
    for b in backs:
        b.ready, b.adopted = True, True
    step_slabs(steppers, exchange, euler=False)
    mine = [e[1:] for e in log if e[0] == 1]
    assert mine == [("stage", 0, False), ("pack", 0, WEST), ("pack", 0, EAST), ("stage", 2, False),
                    ("unpack", 0, WEST), ("unpack", 0, EAST), ("stage", 3, False),
                    ("pack", 3, WEST), ("pack", 3, EAST), ("unpack", 3, WEST), ("unpack", 3, EAST),
                    ("stage", 5, False), ("pack", 4, WEST), ("pack", 4, EAST), ("unpack", 4, WEST), ("unpack", 4, EAST),
                    ("stage", 4, False)]
    assert [e[1] for e in log if e[0] == "exchange"] == [0, 3, 4]
    assert steppers[0].lookahead_in_flight
    for r, b in enumerate(backs):
        west, east = slab_neighbours(r, P)
        for grp in (0, 3, 4):
            assert b.unpacked[(grp, WEST)] == west * 100 + grp * 10 + EAST
            assert b.unpacked[(grp, EAST)] == east * 100 + grp * 10 + WEST


def test_first_time_step_sequencing():
    backs, steppers, exchange, log = _make_local_ring(2)
    first_step_slabs(steppers, exchange)
    mine = [e[1:] for e in log if e[0] == 0]
    assert mine[:10] == [("initialize",), ("fill_local",), ("pack", 0, WEST), ("pack", 0, EAST), ("pack", 2, WEST),
                         ("pack", 2, EAST), ("unpack", 0, WEST), ("unpack", 0, EAST), ("unpack", 2, WEST),
                         ("unpack", 2, EAST)]
    assert mine[10] == ("update_state_local",)
    assert mine[11] == ("stage", 0, True) and mine[-2:] == [("stage", 3, True), ("stage", 4, True)]      # Euler first step
