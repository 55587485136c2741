"""Row-slab decomposition across GPUs: one process per GPU, one halo exchange per step.

The reference is single-GPU; this is the multi-GPU driver on top of the C-ABI's
slab entry points (include/fluid.h: fluid_slab_plan / fluid_halo_pack / fluid_halo_unpack).

Design: the whole step's dependency cone (advection reach A, D diffusion sweeps, divergence, J
Jacobi sweeps, gradient subtract) is covered by ONE exchange of J+1+D+A velocity rows and A
density rows per neighbour before the step; each slab recomputes the shrinking cone in its
halo.  On an xGMI node a halo row is microseconds of wire time but every exchange costs tens of
microseconds of launch/latency, so fewer, deeper exchanges win (J=40: 1 exchange instead of 52
per step).  Messages move with torch.distributed P2P — backend "nccl" is RCCL over xGMI on GPUs,
"gloo" on CPU for tests — batched so each rank posts all four transfers at once (no ordering
deadlock on the open chain of slabs).
"""
from __future__ import annotations

from typing import Callable, Optional


class HaloExchanger:
    """Moves one step's halo messages between neighbouring ranks.

    ``pack(direction, ptr)`` must fill the send buffer at device/host address ``ptr`` with the
    message for the neighbour in ``direction`` (-1: rank-1, +1: rank+1); ``unpack(direction,
    ptr)`` consumes the message received from that neighbour.  Buffers are uint8 torch tensors on
    ``device``; on CUDA the calling code must run on torch's current stream so that the NCCL
    transfers are ordered after pack and before unpack.
    """

    def __init__(self, rank: int, world: int, bytes_lo: int, bytes_hi: int, device, group=None):
        import torch
        self.rank, self.world, self.group = rank, world, group
        self.has_lo, self.has_hi = rank > 0 and bytes_lo > 0, rank < world - 1 and bytes_hi > 0
        mk = lambda nbytes: torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.send_lo = mk(bytes_lo) if self.has_lo else None
        self.recv_lo = mk(bytes_lo) if self.has_lo else None
        self.send_hi = mk(bytes_hi) if self.has_hi else None
        self.recv_hi = mk(bytes_hi) if self.has_hi else None
        self.bytes_per_step = (bytes_lo if self.has_lo else 0) + (bytes_hi if self.has_hi else 0)
        # NCCL/RCCL transfers are stream-ordered after the packs; any other backend reads the
        # buffers from the host side, so the packs must have landed first.
        import torch.distributed as dist
        self.host_ordered = torch.device(device).type == "cuda" and dist.get_backend(group) != "nccl"

    def exchange(self, pack: Callable[[int, int], None], unpack: Callable[[int, int], None]):
        import torch.distributed as dist
        if not (self.has_lo or self.has_hi):
            return
        ops = []
        if self.has_lo:
            pack(-1, self.send_lo.data_ptr())
        if self.has_hi:
            pack(+1, self.send_hi.data_ptr())
        if self.host_ordered:
            import torch
            torch.cuda.current_stream().synchronize()
        if self.has_lo:
            ops.append(dist.P2POp(dist.isend, self.send_lo, self.rank - 1, self.group))
            ops.append(dist.P2POp(dist.irecv, self.recv_lo, self.rank - 1, self.group))
        if self.has_hi:
            ops.append(dist.P2POp(dist.isend, self.send_hi, self.rank + 1, self.group))
            ops.append(dist.P2POp(dist.irecv, self.recv_hi, self.rank + 1, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()  # CUDA: the current stream waits for the transfer; no host block
        if self.has_lo:
            unpack(-1, self.recv_lo.data_ptr())
        if self.has_hi:
            unpack(+1, self.recv_hi.data_ptr())


class SlabSim:
    """One rank's slab of a domain decomposed over ``world`` processes (one GPU each).

    Mirrors :class:`FluidSim` (add_source / step / read); ``read`` returns the owned rows,
    ``gather`` assembles the whole field on rank 0.
    """

    def __init__(self, pkg, n: int, *, rank: int, world: int, device_index: int = 0, group=None,
                 **params):
        import torch
        self.pkg, self.rank, self.world = pkg, rank, world
        self.torch = torch
        torch.cuda.set_device(device_index)
        self.stream = torch.cuda.Stream(device=device_index)
        self.sim = pkg.FluidSim(n, device=device_index, slab_rank=rank, slab_count=world, **params)
        # kernels, pack/unpack copies and the RCCL transfers are all ordered on this stream
        self.sim.set_stream(self.stream.cuda_stream)
        self.n, self.row0, self.rows = self.sim.n, self.sim.row0, self.sim.rows
        self.ex = HaloExchanger(rank, world, self.sim.halo_bytes(-1), self.sim.halo_bytes(+1),
                                torch.device("cuda", device_index), group)

    def add_source(self, *a, **k):
        self.sim.add_source(*a, **k)

    def step(self, nsteps: int = 1):
        with self.torch.cuda.stream(self.stream):
            for _ in range(nsteps):
                self.ex.exchange(self.sim.halo_pack, self.sim.halo_unpack)
                self.sim.step(1)

    def sync(self):
        self.sim.sync()

    def read(self, field):
        return self.sim.read(field)

    def gather(self, field):
        """Whole field on rank 0 (None elsewhere); slabs in y order."""
        import numpy as np
        import torch.distributed as dist
        mine = self.read(field)
        parts = [None] * self.world if self.rank == 0 else None
        dist.gather_object(mine, parts, dst=0)
        return np.concatenate(parts, axis=0) if self.rank == 0 else None

    def close(self):
        self.sim.close()


class LocalSlabGroup:
    """All slabs of one domain as separate contexts on ONE device, halos moved by handing each
    slab its neighbour's packed send buffer.  Exercises exactly the library's slab path (plan,
    margins, pack/unpack) without a second GPU; used by the GPU tests to demand bit-identical
    results to the whole-domain context."""

    def __init__(self, pkg, n: int, count: int, device_index: int = 0, **params):
        import torch
        self.pkg, self.count = pkg, count
        self.sims = [pkg.FluidSim(n, device=device_index, slab_rank=r, slab_count=count, **params)
                     for r in range(count)]
        dev = torch.device("cuda", device_index)
        mk = lambda nbytes: torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        self.send = [{-1: mk(s.halo_bytes(-1)), +1: mk(s.halo_bytes(+1))} for s in self.sims]

    def add_source(self, *a, **k):
        for s in self.sims:
            s.add_source(*a, **k)

    def step(self, nsteps: int = 1):
        for _ in range(nsteps):
            for r, s in enumerate(self.sims):
                for d in (-1, +1):
                    if s.halo_bytes(d):
                        s.halo_pack(d, self.send[r][d].data_ptr())
            for s in self.sims:
                s.sync()  # contexts have their own streams: packs must land before unpacks read
            for r, s in enumerate(self.sims):
                if s.halo_bytes(-1):
                    s.halo_unpack(-1, self.send[r - 1][+1].data_ptr())
                if s.halo_bytes(+1):
                    s.halo_unpack(+1, self.send[r + 1][-1].data_ptr())
            for s in self.sims:
                s.step(1)
            for s in self.sims:
                s.sync()

    def read(self, field):
        import numpy as np
        return np.concatenate([s.read(field) for s in self.sims], axis=0)

    def write(self, field, arr):
        for s in self.sims:
            s.write(field, arr[s.row0:s.row0 + s.rows])

    def close(self):
        for s in self.sims:
            s.close()
