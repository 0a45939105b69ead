"""x-slab multi-GPU driver: one process (rank) per GPU, packed halo exchange between ring neighbours.

Replaces what the reference gets from `Oceananigans.Distributed(arch; partition=Partition(Rx, Ry, 1))` + XLA's SPMD
partitioner (GB-25 sharding/sharded_baroclinic_instability_simulation_run.jl:65-72): there the halos travel as
XLA collective-permutes; here each time step has three explicit point-to-point exchanges,
the large one hidden behind the barotropic sub-cycle on a second HIP stream:

    stage 0   AB2 update of u,v,T,S (adoption of the look-aheads), y/z layers of the 3-D bundle; pressure of the own
              columns starts on the model's side stream                                            (compute stream)
    group 0   H columns of u,v,T,S          -> x halos        packed + sent on the COMM stream, in flight during stage 2
    stage 2   barotropic corrector on the slab's own columns                                      (compute stream)
    stage 3   [wait for group 0] corrector in the halo columns, w, p' strips, momentum tendencies (compute stream)
    stage 4   tracer tendencies                                                                   (compute stream)
      beside stage 4, on the COMM stream, the sub-cycle of the NEXT step (its G.U, G.V exist since stage 3):
    group 3   W = Ns+1 columns of eta,U,V and of the next G.U,G.V -> wide barotropic halos
    stage 5   Ns split-explicit substeps on the widened slab into the partner buffers of eta,U,V, filtered state
    group 4   H columns of the new eta,U,V  -> x halos of the partner buffers
    The next stage 0 adopts them.  When a look-ahead is not valid (first step, changed dt, host writes) the same work
    runs inside the step instead: group 1 (= 3), stage 1 (= 5), group 2 (= 4), on the critical path.

No collective is needed: every rank talks to its west and east neighbour only (send/recv over xGMI via
torch.distributed, backend "nccl" = RCCL).  The transport is injected so that the same sequencing code runs
(a) across processes and (b) over several slabs inside one process (tests: decomposition invariance on one GPU).
"""
import numpy as np
import torch

from .binding import HipBackend
from .model import HydrostaticFreeSurfaceModel
from .sharding import slab_neighbours

WEST, EAST = 0, 1


class TorchDistributedTransport:
    """Ring exchange with torch.distributed point-to-point ops.

    Posting order is part of the protocol: sends [west pack, east pack], receives [east halo, west halo].
    With two ranks both neighbours are the same peer and messages between one pair match in posting order,
    so the peer's FIRST send (its west pack) must meet our FIRST receive (our east halo)."""

    def __init__(self, rank, nranks, dist=None):
        import torch.distributed as tdist
        self.dist = dist or tdist
        self.rank, self.nranks = rank, nranks
        self.west, self.east = slab_neighbours(rank, nranks)

    def exchange(self, send_west, send_east, recv_west, recv_east):
        d = self.dist
        stage = send_west.is_cuda and d.get_backend() != "nccl"
        if stage:
            # rehearsal transport (gloo has no device-memory point-to-point): bounce through host buffers
            dev = (recv_west, recv_east)
            send_west, send_east = send_west.cpu(), send_east.cpu()
            recv_west, recv_east = torch.empty_like(send_west), torch.empty_like(send_east)
        ops = [d.P2POp(d.isend, send_west, self.west), d.P2POp(d.isend, send_east, self.east),
               d.P2POp(d.irecv, recv_east, self.east), d.P2POp(d.irecv, recv_west, self.west)]
        for req in d.batch_isend_irecv(ops):
            req.wait()
        if stage:
            dev[0].copy_(recv_west)
            dev[1].copy_(recv_east)


class LocalRingTransport:
    """All slabs live in one process: `exchange_all` moves every slab's packs to its neighbours' receive buffers."""

    @staticmethod
    def exchange_all(steppers, group):
        P = len(steppers)
        for r, s in enumerate(steppers):
            west, east = slab_neighbours(r, P)
            steppers[west].recv[group][EAST].copy_(s.send[group][WEST])   # my west pack -> west nbr's east halo
            steppers[east].recv[group][WEST].copy_(s.send[group][EAST])   # my east pack -> east nbr's west halo


GROUPS = (0, 1, 2)


class SlabStepper:
    """Sequencing of one slab: owns the pack buffers and cuts the step at its exchange points.

    The model's kernels run on torch's current stream of the device (the compute stream), so pack kernels, the
    transport's send/recv (or copies) and unpack kernels are ordered by the streams themselves: no host
    synchronisation.  `comm` is the second stream on which the 3-D halo bundle travels during the sub-cycle."""

    def __init__(self, backend, device, comm_stream=None):
        self.b = backend
        self.send, self.recv = {}, {}
        for group in GROUPS:
            n = backend.halo_buffer_elems(group)
            dt = torch.float64 if np.dtype(getattr(backend, "dtype", np.float32)).itemsize == 8 else torch.float32
            self.send[group] = [torch.empty(n, dtype=dt, device=device) for _ in range(2)]
            self.recv[group] = [torch.empty(n, dtype=dt, device=device) for _ in range(2)]
        # groups 3 and 4 (sub-cycle look-ahead) reuse the buffers of groups 1 and 2: never in use at the same time
        self.send[3], self.recv[3] = self.send[1], self.recv[1]
        self.send[4], self.recv[4] = self.send[2], self.recv[2]
        self.lookahead_in_flight = False
        self.cuda = device.type == "cuda"
        if self.cuda:
            self.main = torch.cuda.current_stream(device)
            self.comm = comm_stream or torch.cuda.Stream(device)
            backend.set_stream(self.main.cuda_stream)

    def on_comm(self, fn):
        """Run model calls with the model's kernels on the comm stream."""
        if self.cuda:
            self.b.set_stream(self.comm.cuda_stream)
        try:
            fn()
        finally:
            if self.cuda:
                self.b.set_stream(self.main.cuda_stream)

    def pack(self, group, on_comm=False):
        if on_comm and self.cuda:
            self.b.set_stream(self.comm.cuda_stream)
        self.b.halo_pack_both(group, self.send[group][WEST].data_ptr(), self.send[group][EAST].data_ptr())
        if on_comm and self.cuda:
            self.b.set_stream(self.main.cuda_stream)

    def unpack(self, group):
        self.b.halo_unpack_both(group, self.recv[group][WEST].data_ptr(), self.recv[group][EAST].data_ptr())


class _OnComm:
    """Context: torch's current stream = the comm stream of the steppers (no-op on CPU test doubles).
    `after` is an event recorded on the compute stream: the comm stream starts once it has completed."""

    def __init__(self, steppers, after=None):
        self.s = steppers[0]
        self.after = after
        self.ctx = None

    def __enter__(self):
        if self.s.cuda:
            if self.after is not None:
                self.s.comm.wait_event(self.after)    # everything stage 0 wrote is visible to the comm stream
            else:
                self.s.comm.wait_stream(self.s.main)
            self.ctx = torch.cuda.stream(self.s.comm)
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _run_stage(steppers, fn):
    for s in steppers:
        fn(s)


def step_slabs(steppers, exchange, euler=False):
    """One time step of a list of slabs (a single one in the multi-process case)."""
    s0 = steppers[0]
    if s0.cuda and s0.lookahead_in_flight:
        s0.main.wait_stream(s0.comm)              # stage 5 and groups 3, 4 of the previous step have finished
    s0.lookahead_in_flight = False
    _run_stage(steppers, lambda s: s.b.time_step_stage(0, euler))
    adopted = all(s.b.lookahead_state()[1] for s in steppers)   # the sub-cycle of this step is already done
    stage0_done = s0.main.record_event() if s0.cuda else None
    if not adopted:
        # The small barotropic exchange is on the critical path and is posted FIRST: a process group's
        # point-to-point transfers share one RCCL stream and run in posting order, so the 6 MB bundle must not be
        # queued ahead of it.
        _run_stage(steppers, lambda s: s.pack(1))
        exchange(1)
    with _OnComm(steppers, stage0_done):          # the 3-D bundle leaves on the second stream ...
        _run_stage(steppers, lambda s: s.pack(0, on_comm=True))
        packed0 = s0.comm.record_event() if s0.cuda else None
        exchange(0)
    if not adopted:
        # ... and is in flight while the sub-cycle runs here
        _run_stage(steppers, lambda s: (s.unpack(1), s.b.time_step_stage(1, euler), s.pack(2)))
        packed2 = s0.main.record_event() if s0.cuda else None
        with _OnComm(steppers, packed2):          # eta, U, V columns leave behind the bundle on the second stream
            exchange(2)
    if packed0 is not None:
        s0.main.wait_event(packed0)               # the corrector rewrites the columns the bundle was packed from
    _run_stage(steppers, lambda s: s.b.time_step_stage(2, euler))   # own columns, while the exchanges are in flight
    if s0.cuda:
        s0.main.wait_stream(s0.comm)
    if not adopted:
        _run_stage(steppers, lambda s: s.unpack(2))
    _run_stage(steppers, lambda s: (s.unpack(0), s.b.time_step_stage(3, euler)))
    # the next step's G.U, G.V exist now: its wide-halo exchange, sub-cycle and eta,U,V exchange run on the second
    # stream beside the tracer tendencies
    if all(s.b.lookahead_state()[0] for s in steppers):
        mom_done = s0.main.record_event() if s0.cuda else None
        with _OnComm(steppers, mom_done):
            _run_stage(steppers, lambda s: s.on_comm(lambda: s.pack(3)))
            exchange(3)
            _run_stage(steppers, lambda s: s.on_comm(lambda: (s.unpack(3), s.b.time_step_stage(5, euler), s.pack(4))))
            exchange(4)
            _run_stage(steppers, lambda s: s.on_comm(lambda: s.unpack(4)))
        s0.lookahead_in_flight = True
    _run_stage(steppers, lambda s: s.b.time_step_stage(4, euler))


def first_step_slabs(steppers, exchange):
    """first_time_step!: initialize!, update_state!, then an Euler step (src/timestepping_utils.jl:21-27)."""
    _run_stage(steppers, lambda s: (s.b.initialize(), s.b.fill_halo_regions_local(), s.pack(0), s.pack(2)))
    exchange(0)
    exchange(2)
    _run_stage(steppers, lambda s: (s.unpack(0), s.unpack(2), s.b.update_state_local()))
    step_slabs(steppers, exchange, euler=True)


class SlabModel(HydrostaticFreeSurfaceModel):
    """One rank's slab of a (Nx_global x Ny x Nz) model; same API as the single-GPU model.
    Fields are the LOCAL slab (Nx_global / nranks columns)."""

    def __init__(self, Nx_global, Ny, Nz, *, dt, rank, nranks, device=0, halo=8, substeps=30, transport=None, **kw):
        backend = HipBackend(Nx_global, Ny, Nz, dt=dt, halo=halo, substeps=substeps, device=device, rank=rank,
                             nranks=nranks, **kw)
        super().__init__(_SlabBackendFacade(backend, self), Nx_global // nranks, Ny, Nz, halo)
        self.rank, self.nranks = rank, nranks
        self.stepper = SlabStepper(backend, torch.device("cuda", device))
        self.transport = transport or TorchDistributedTransport(rank, nranks)

    def _exchange(self, group):
        s = self.stepper
        self.transport.exchange(s.send[group][WEST], s.send[group][EAST], s.recv[group][WEST], s.recv[group][EAST])


class _SlabBackendFacade:
    """Gives model.first_time_step / time_step / loop (which call backend.*) the staged implementation,
    and forwards everything else to the HipBackend."""

    def __init__(self, backend, owner):
        self._b, self._owner = backend, owner

    def __getattr__(self, name):
        return getattr(self._b, name)

    def first_time_step(self):
        first_step_slabs([self._owner.stepper], self._owner._exchange)

    def time_step(self):
        step_slabs([self._owner.stepper], self._owner._exchange)

    def loop(self, n):
        for _ in range(int(n)):
            step_slabs([self._owner.stepper], self._owner._exchange)


class LocalSlabEnsemble:
    """P slabs of one global model stepped in lock-step inside ONE process on one GPU (tests)."""

    def __init__(self, Nx_global, Ny, Nz, P, *, dt, device=0, **kw):
        self.P, self.Nx_loc = P, Nx_global // P
        self.backends = [HipBackend(Nx_global, Ny, Nz, dt=dt, device=device, rank=r, nranks=P, **kw) for r in range(P)]
        dev = torch.device("cuda", device)
        comm = torch.cuda.Stream(dev)
        self.steppers = [SlabStepper(b, dev, comm_stream=comm) for b in self.backends]
        self._exchange = lambda group: LocalRingTransport.exchange_all(self.steppers, group)

    def scatter(self, name, global_interior):
        for r, b in enumerate(self.backends):
            b.set_field(name, np.ascontiguousarray(global_interior[r * self.Nx_loc:(r + 1) * self.Nx_loc]), False)

    def gather(self, name):
        return np.concatenate([b.get_field(name, False) for b in self.backends], axis=0)

    def first_time_step(self):
        first_step_slabs(self.steppers, self._exchange)

    def time_step(self):
        step_slabs(self.steppers, self._exchange)

    def loop(self, n):
        for _ in range(n):
            self.time_step()
