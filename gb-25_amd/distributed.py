"""x-slab multi-GPU models: one process (rank) per GPU, packed halo exchange between ring neighbours.

Replaces what the reference gets from `Oceananigans.Distributed(arch; partition=Partition(Rx, Ry, 1))` + XLA's SPMD
partitioner (GB-25 sharding/sharded_baroclinic_instability_simulation_run.jl:65-72).  The sequencing of a slab's time
step, the pack / unpack kernels and the exchanges all live INSIDE the library (csrc/slab_step.hpp): after the exchange
context exists, `first_time_step`, `time_step` and `loop` are one ABI call each, exactly as on a single GPU.  This
module only

  * bootstraps the RCCL communicator: rank 0 asks the library for the 128-byte unique id (ncclGetUniqueId) and hands it
    to the other ranks through torch.distributed's rendezvous store (any broadcast would do: MPI in a Julia host);
  * offers the two other transports the library knows: all slabs in one process (`LocalSlabEnsemble`: decomposition
    invariance tests on a one-GPU box) and a host-side exchange through torch.distributed point-to-point operations
    (`TorchDistributedTransport`: rehearsal of the multi-process path with gloo where RCCL cannot run, e.g. two ranks
    on one device).
No collective is needed in a time step: every rank talks to its west and east neighbour only.
"""
import os
import numpy as np
import torch

from .binding import HipBackend
from .model import HydrostaticFreeSurfaceModel
from .sharding import mesh_neighbours, slab_neighbours

WEST, EAST = 0, 1


class TorchDistributedTransport:
    """Ring exchange with torch.distributed point-to-point ops (the host-callback transport of the library).

    Posting order is part of the protocol: sends [west pack, east pack], receives [east halo, west halo].
    With two ranks both neighbours are the same peer and messages between one pair match in posting order,
    so the peer's FIRST send (its west pack) must meet our FIRST receive (our east halo)."""

    def __init__(self, rank, nranks, dist=None, ranks_y=1):
        import torch.distributed as tdist
        self.dist = dist or tdist
        self.rank, self.nranks = rank, nranks
        # Partition(Rx, Ry, 1): the ring runs within the rank's row; south / north are None beyond the walls
        nb = mesh_neighbours(rank, nranks // max(1, ranks_y), max(1, ranks_y))
        self.west, self.east, self.south, self.north, self.partner = (nb[k] for k in ("west", "east", "south", "north", "partner"))

    def exchange(self, send_west, send_east, recv_west, recv_east):
        d = self.dist
        stage = send_west.is_cuda and d.get_backend() != "nccl"
        if stage:
            # gloo has no device-memory point-to-point: bounce through host buffers
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
            torch.cuda.current_stream().synchronize()     # the library's streams know nothing of torch's


    def exchange_partner(self, send, recv):
        """Zipper fold of a decomposed tripolar grid: slab r <-> slab P-1-r (buffer sets 3 and 4 of the library)."""
        d = self.dist
        partner = self.partner
        if partner == self.rank:
            recv.copy_(send)
            return
        stage = send.is_cuda and d.get_backend() != "nccl"
        dev = recv
        if stage:
            send, recv = send.cpu(), torch.empty_like(send, device="cpu")
        for req in d.batch_isend_irecv([d.P2POp(d.isend, send, partner), d.P2POp(d.irecv, recv, partner)]):
            req.wait()
        if stage:
            dev.copy_(recv)
            torch.cuda.current_stream().synchronize()


    def exchange_y(self, send_south, send_north, recv_south, recv_north):
        """y halos of a 2-D decomposition (buffer sets 5 - 7 of the library): my southern pack -> the southern neighbour's
        northern halo, my northern pack -> the northern neighbour's southern halo; None where there is no neighbour."""
        d = self.dist
        pairs = [(send_south, recv_south, self.south), (send_north, recv_north, self.north)]
        pairs = [(s_, r_, peer) for s_, r_, peer in pairs if peer is not None]
        if not pairs:
            return
        stage = pairs[0][0].is_cuda and d.get_backend() != "nccl"
        dev = [r_ for _, r_, _ in pairs]
        if stage:
            pairs = [(s_.cpu(), torch.empty_like(s_, device="cpu"), peer) for s_, _, peer in pairs]
        ops = [d.P2POp(d.isend, s_, peer) for s_, _, peer in pairs] + [d.P2POp(d.irecv, r_, peer) for _, r_, peer in reversed(pairs)]
        for req in d.batch_isend_irecv(ops):
            req.wait()
        if stage:
            for t, (_, r_, _) in zip(dev, pairs):
                t.copy_(r_)
            torch.cuda.current_stream().synchronize()


class _DevicePointer:
    """A device buffer of the library seen by torch (zero-copy) through __cuda_array_interface__."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _as_tensor(ptr, nbytes, device):
    return torch.as_tensor(_DevicePointer(ptr, nbytes), device=device)


def share_unique_id(backend, rank, dist=None):
    """ncclGetUniqueId on rank 0, handed to every rank through torch.distributed (object broadcast: works with any
    backend).  A Julia host would MPI.Bcast the same 128 bytes."""
    import torch.distributed as tdist
    d = dist or tdist
    box = [backend.comm_unique_id() if rank == 0 else None]
    d.broadcast_object_list(box, src=0)
    return box[0]


class SlabModel(HydrostaticFreeSurfaceModel):
    """One rank's slab of a (Nx_global x Ny x Nz) model; same API as the single-GPU model: first_time_step / time_step
    / loop are ONE library call each.  Fields are the LOCAL slab (Nx_global / nranks columns).

    transport: "rccl" (default when torch.distributed runs on the nccl backend) -- the library's own communicator,
               ncclSend/ncclRecv on its second HIP stream;  "host" -- torch.distributed point-to-point through the
               library's callback transport (gloo rehearsal)."""

    def __init__(self, Nx_global, Ny, Nz, *, dt, rank, nranks, device=0, halo=8, substeps=30, transport=None, ranks_y=1, **kw):
        import torch.distributed as dist
        # ranks_y > 1: Partition(nranks / ranks_y, ranks_y, 1), rank = ry Rx + rx; the fields are the rank's window of columns
        # AND rows
        backend = HipBackend(Nx_global, Ny, Nz, dt=dt, halo=halo, substeps=substeps, device=device, rank=rank,
                             nranks=nranks, ranks_y=ranks_y, **kw)
        super().__init__(backend, backend.Nx_local, backend.Ny_local, Nz, halo)
        self.rank, self.nranks, self.ranks_y = rank, nranks, ranks_y
        if transport is None:
            transport = "rccl" if (nranks == 1 or os.environ.get("GB25_REHEARSE_ALONE") == "1" or dist.get_backend() == "nccl") else "host"
        self.transport_kind = transport
        if transport == "rccl":
            # (GB25_REHEARSE_ALONE=1: one rank of a decomposition alone in its process, its own neighbour on every side)
            alone = nranks == 1 or os.environ.get("GB25_REHEARSE_ALONE") == "1"
            uid = backend.comm_unique_id() if alone else share_unique_id(backend, rank)
            backend.comm_init_rccl(uid)
        elif transport == "host":
            self._ring = TorchDistributedTransport(rank, nranks, ranks_y=ranks_y)
            dev = torch.device("cuda", device)

            def exchange(buffer_set, sw, se, rw, re, nbytes):
                # whom a buffer set travels to (set_kind in csrc/slab_step.hpp): 0 - 2, 8: the west / east ring neighbours;
                # 3, 4, 10: the fold partner; 5 - 7, 9: the southern / northern neighbour (8 - 10: CATKE's e and J^b)
                kind = 1 if buffer_set in (3, 4, 10) else 2 if buffer_set in (5, 6, 7, 9) else 0
                if kind == 2:            # y halos: (south, north) in the place of (west, east); null where there is no neighbour
                    t = lambda p: _as_tensor(p, nbytes, dev) if p else None
                    self._ring.exchange_y(t(sw), t(se), t(rw), t(re))
                    if torch.cuda.is_available():
                        torch.cuda.current_stream().synchronize()
                    return
                if kind == 1:            # to and from the fold partner (the east pointers are null)
                    self._ring.exchange_partner(_as_tensor(sw, nbytes, dev), _as_tensor(rw, nbytes, dev))
                    if torch.cuda.is_available():
                        torch.cuda.current_stream().synchronize()
                    return
                self._ring.exchange(_as_tensor(sw, nbytes, dev), _as_tensor(se, nbytes, dev),
                                    _as_tensor(rw, nbytes, dev), _as_tensor(re, nbytes, dev))
            backend.comm_init_callback(exchange)
        else:
            raise ValueError(f"transport must be 'rccl' or 'host', got {transport!r}")


class LocalSlabEnsemble:
    """P slabs of one global model stepped in lock-step inside ONE process on one GPU: the library's local transport
    (ring of device-to-device copies), the same stages, pack / unpack kernels and two streams as the RCCL path."""

    def __init__(self, Nx_global, Ny, Nz, P, *, dt, device=0, ranks_y=1, **kw):
        # ranks_y > 1: a 2-D decomposition, Partition(P / ranks_y, ranks_y, 1); rank = ry Rx + rx
        self.P, self.Ry, self.Rx = P, ranks_y, P // ranks_y
        self.Nx_loc, self.Ny_loc = Nx_global // self.Rx, Ny // ranks_y
        self.backends = [HipBackend(Nx_global, Ny, Nz, dt=dt, device=device, rank=r, nranks=P, ranks_y=ranks_y, **kw)
                         for r in range(P)]
        HipBackend.comm_init_local(self.backends)

    def scatter(self, name, global_interior):
        for b in self.backends:
            d = b.field_dims(name, False)      # (a y-face field has one row more on the ranks that hold the northern wall)
            i0, j0 = b.rx * self.Nx_loc, b.ry * self.Ny_loc
            b.set_field(name, np.ascontiguousarray(global_interior[i0:i0 + d[0], j0:j0 + d[1]]), False)

    def gather(self, name):
        rows = [np.concatenate([self.backends[ry * self.Rx + rx].get_field(name, False) for rx in range(self.Rx)], axis=0)
                for ry in range(self.Ry)]
        return rows[0] if self.Ry == 1 else np.concatenate(rows, axis=1)

    def set_option(self, name, value):
        for b in self.backends:
            b.set_option(name, value)

    def synchronize(self):
        for b in self.backends:
            b.synchronize()

    # the composites of any member step every slab of the exchange context
    def first_time_step(self):
        self.backends[0].first_time_step()

    def time_step(self):
        self.backends[0].time_step()

    def loop(self, n):
        self.backends[0].loop(int(n))

    def close(self):
        for b in self.backends:
            b.close()
