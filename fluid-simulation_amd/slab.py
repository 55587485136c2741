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
    """Moves halo messages between neighbouring ranks (rank-1 below, rank+1 above).

    One exchange = ``start`` (pack the send buffers, post all isend/irecv at once) and, later,
    ``finish`` (wait, unpack).  Several exchanges, told apart by ``tag``, may be in flight at
    the same time, each with its own buffers, so a transfer can hide behind compute.

    ``pack(direction, ptr)`` fills the send buffer at device/host address ``ptr`` with the
    message for the neighbour in ``direction`` (-1 / +1); ``unpack(direction, ptr)`` consumes the
    message received from it.  Buffers are uint8 torch tensors on ``device``.  On CUDA with the
    nccl (= RCCL) backend the transfers are stream-ordered: posted under the stream that ran
    ``pack``, and ``finish`` makes the *current* stream wait for them (no host block).  Any other
    backend reads the buffers from the host side, so the packing stream is synchronised first.
    """

    def __init__(self, rank: int, world: int, device, group=None):
        import torch
        import torch.distributed as dist
        self.rank, self.world, self.group = rank, world, group
        self.device = torch.device(device)
        self.has_lo, self.has_hi = rank > 0, rank < world - 1
        self.host_ordered = self.device.type == "cuda" and dist.get_backend(group) != "nccl"
        self.buf = {}      # tag -> {(kind, dir): tensor}
        self.pending = {}  # tag -> list of requests
        self.bytes_moved = 0

    def _buffers(self, tag, nbytes_lo, nbytes_hi):
        import torch
        b = self.buf.get(tag)
        if b is None:
            mk = lambda n: torch.empty(n, dtype=torch.uint8, device=self.device)
            b = {}
            if self.has_lo and nbytes_lo:
                b[("send", -1)], b[("recv", -1)] = mk(nbytes_lo), mk(nbytes_lo)
            if self.has_hi and nbytes_hi:
                b[("send", +1)], b[("recv", +1)] = mk(nbytes_hi), mk(nbytes_hi)
            self.buf[tag] = b
        return b

    def start(self, tag, nbytes_lo: int, nbytes_hi: int, pack: Callable[[int, int], None]):
        import torch
        import torch.distributed as dist
        assert tag not in self.pending, f"exchange {tag!r} already in flight"
        b = self._buffers(tag, nbytes_lo, nbytes_hi)
        dirs = [d for d in (-1, +1) if ("send", d) in b]
        for d in dirs:
            pack(d, b[("send", d)].data_ptr())
        if self.host_ordered and dirs:
            torch.cuda.current_stream().synchronize()
        ops = []
        for d in dirs:
            peer = self.rank + d
            ops.append(dist.P2POp(dist.isend, b[("send", d)], peer, self.group))
            ops.append(dist.P2POp(dist.irecv, b[("recv", d)], peer, self.group))
            self.bytes_moved += b[("send", d)].numel()
        self.pending[tag] = dist.batch_isend_irecv(ops) if ops else []

    def finish(self, tag, unpack: Optional[Callable[[int, int], None]]):
        """Wait for exchange ``tag``; ``unpack=None`` discards what was received."""
        for req in self.pending.pop(tag):
            req.wait()
        if unpack is not None:
            for (kind, d), t in self.buf[tag].items():
                if kind == "recv":
                    unpack(d, t.data_ptr())

    def in_flight(self, tag) -> bool:
        return tag in self.pending

    def exchange(self, nbytes_lo, nbytes_hi, pack, unpack, tag="all"):
        self.start(tag, nbytes_lo, nbytes_hi, pack)
        self.finish(tag, unpack)


class HostStagedTransport:
    """A caller-provided halo transport for fluid_comm_attach (include/fluid.h, fluid_transport):
    the library's own slab step — exchange posted from C, overlapped with the interior launches —
    with the bytes carried by torch.distributed through host memory.  For rehearsing the C path
    where RCCL cannot run (several ranks on one GPU: tests, FLUID_BENCH_ONE_GPU); it blocks the
    host in every exchange and is not capturable, so the library enqueues its steps directly.

    send: wait for the stream (the rows are final), copy them to the host, post isend;
    recv: post irecv; group_end: wait for all, copy the received rows into the halo rows on the
    stream the library named, and wait for those copies (the staging tensors die with the call).
    """

    def __init__(self, group=None):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import capi
        self.torch, self.dist, self.group = torch, dist, group
        hip = C.CDLL("libamdhip64.so.7")   # by SONAME: the copy this process already runs on
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        hip.hipMemcpyAsync.restype = C.c_int
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        hip.hipStreamSynchronize.restype = C.c_int
        self.hip = hip
        self.ops, self.recvs, self.exchanges = [], [], 0
        self._cbs = (capi.TRANSPORT_GROUP_START(self._group_start), capi.TRANSPORT_SEND(self._send),
                     capi.TRANSPORT_RECV(self._recv), capi.TRANSPORT_GROUP_END(self._group_end))
        self.struct = capi.FluidTransport(None, *self._cbs, 0)

    def _guard(self, fn, *a):
        try:
            return fn(*a) or 0
        except Exception:   # an exception must not unwind through the C caller
            import traceback
            traceback.print_exc()
            return 1

    def _group_start(self, user):
        self.ops, self.recvs = [], []
        return 0

    def _send(self, user, src, nbytes, peer, stream):
        def run():
            t = self.torch.empty(nbytes, dtype=self.torch.uint8)
            if self.hip.hipMemcpyAsync(t.data_ptr(), src, nbytes, 2, stream) or self.hip.hipStreamSynchronize(stream):
                return 1
            self.ops.append(self.dist.P2POp(self.dist.isend, t, peer, self.group))
        return self._guard(run)

    def _recv(self, user, dst, nbytes, peer, stream):
        def run():
            t = self.torch.empty(nbytes, dtype=self.torch.uint8)
            self.ops.append(self.dist.P2POp(self.dist.irecv, t, peer, self.group))
            self.recvs.append((t, dst, nbytes))
        return self._guard(run)

    def _group_end(self, user, stream):
        def run():
            for req in (self.dist.batch_isend_irecv(self.ops) if self.ops else []):
                req.wait()
            for t, dst, nbytes in self.recvs:
                if self.hip.hipMemcpyAsync(dst, t.data_ptr(), nbytes, 1, stream):
                    return 1
            if self.hip.hipStreamSynchronize(stream):
                return 1
            self.ops, self.recvs = [], []
            self.exchanges += 1
        return self._guard(run)


class SlabSim:
    """One rank's slab of a domain decomposed over ``world`` processes (one GPU each).

    Mirrors :class:`FluidSim` (add_source / step / read); ``read`` returns the owned rows,
    ``gather`` assembles the whole field on rank 0.

    ``overlap=True`` (default; env FLUID_SLAB_OVERLAP=0 turns it off) hides the halo traffic
    behind compute (include/fluid.h fluid_step_part).  The step ends with the fused gradient
    subtraction + density advection run in two launches: first the edge rows the neighbours need,
    then the interior; both halo messages for step s+1 are posted between the two on a second
    stream, so

        velocity halos for step s+1 travel while step s's interior rows are finished,
        density halos for step s+1 travel until step s+1's last kernel needs them.

    (With FLUID_FLAG_UNFUSED the two stages are separate kernels: velocity halos then travel during
    the density advection, density halos during the next velocity part.)  Same messages, same
    results; only when they move differs.

    ``transport``: who drives the exchange.
      "torch"  (default; env FLUID_SLAB_TRANSPORT) this class, as described above: the step in parts
               from Python, halos packed into staging tensors and moved by torch.distributed P2P;
      "rccl"   the library itself (include/fluid.h, fluid_comm_*): ncclSend / ncclRecv straight
               out of and into the field rows, posted from C on a second stream between the edge
               and interior launches, whole steps replayed as hipGraphs; this class only makes
               the communicator (unique id broadcast over ``group``) and calls fluid_step;
      "staged" the same C path with :class:`HostStagedTransport` (rehearsal on one GPU).
    """

    def __init__(self, pkg, n: int, *, rank: int, world: int, device_index: int = 0, group=None,
                 overlap: Optional[bool] = None, transport: Optional[str] = None, **params):
        import os
        import torch
        self.pkg, self.rank, self.world = pkg, rank, world
        self.torch = torch
        self.transport = transport or os.environ.get("FLUID_SLAB_TRANSPORT", "torch")
        if self.transport not in ("torch", "rccl", "staged"):
            raise ValueError(f"unknown slab transport {self.transport!r}")
        self.in_library = self.transport != "torch" and world > 1
        torch.cuda.set_device(device_index)
        self.stream = torch.cuda.Stream(device=device_index)       # kernels, unpack
        # pack, transfers.  Normal priority, always: a high-priority second stream was tried in round 1
        # and removed (no gain for these short messages, and its wait on a
        # normal-priority queue could deadlock when four processes share one GPU's hardware queues).
        self.comm_stream = torch.cuda.Stream(device=device_index)
        self.sim = pkg.FluidSim(n, device=device_index, slab_rank=rank, slab_count=world, **params)
        self.sim.set_stream(self.stream.cuda_stream)
        self.n, self.row0, self.rows = self.sim.n, self.sim.row0, self.sim.rows
        self.overlap = (os.environ.get("FLUID_SLAB_OVERLAP", "1") != "0") if overlap is None else overlap
        self.ex = HaloExchanger(rank, world, torch.device("cuda", device_index), group)
        self.V, self.D = pkg.PART_VELOCITY, pkg.PART_DENSITY
        self.P = pkg.PART_PRESSURE_GROUP
        self.groups = self.sim.pressure_groups
        self.split_tail = not (params.get("flags", 0) & pkg.FLAG_UNFUSED)
        # the interior rows of the next step's advection before its velocity halos are waited for
        self.early_advect = world > 1 and os.environ.get("FLUID_SLAB_EARLY_ADVECT", "1") != "0"
        self._staged = None
        if self.in_library:
            import torch.distributed as dist
            if self.transport == "rccl":
                box = [self.sim.lib.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0, group=group)
                self.sim.comm_init_rccl(box[0])          # collective: ncclCommInitRank
            else:
                self._staged = HostStagedTransport(group)
                self.sim.comm_attach(self._staged.struct)

    @property
    def bytes_moved(self) -> int:
        return self.sim.comm_info()["bytes_sent"] if self.in_library else self.ex.bytes_moved

    def _velocity_part(self, tail=True):
        """The velocity part on the compute stream (``tail=False``: without the gradient
        subtraction).  With pressure groups (deep Jacobi loops on short slabs) the sweeps run
        group by group with an exchange of pressure halo rows in between; those exchanges are on
        the critical path."""
        sim = self.sim
        with self.torch.cuda.stream(self.stream):
            if self.groups <= 1 and tail:
                sim.step_part(self.V)
                return
            sim.step_part(self.pkg.PART_VELOCITY_HEAD)
        for _ in range(self.groups - 1):
            self._start(self.P)
            self._finish(self.P)
            with self.torch.cuda.stream(self.stream):
                sim.step_part(self.P)
        if tail:
            with self.torch.cuda.stream(self.stream):
                sim.step_part(self.pkg.PART_VELOCITY_TAIL)

    # -- exchanges -------------------------------------------------------------------------
    def _start(self, part, after_compute=True):
        """Post the halo exchange of one field; it is ordered after everything enqueued on the
        compute stream so far and runs on the comm stream.  ``after_compute=False``: nothing was
        enqueued on the compute stream since the previous ``_start`` (an event record there costs
        about 5 us of bubble between two kernels)."""
        sim, comm = self.sim, self.comm_stream
        if after_compute:
            comm.wait_stream(self.stream)
        with self.torch.cuda.stream(comm):
            self.ex.start(part, sim.halo_bytes_part(-1, part), sim.halo_bytes_part(+1, part),
                          lambda d, ptr: sim.halo_pack_part(d, part, ptr, comm.cuda_stream))

    def _finish(self, part, keep=True):
        sim = self.sim
        with self.torch.cuda.stream(self.stream):
            self.ex.finish(part, (lambda d, ptr: sim.halo_unpack_part(d, part, ptr, self.stream.cuda_stream))
                           if keep else None)

    def _discard_in_flight(self):
        for part in (self.V, self.D, self.P):
            if self.ex.in_flight(part):
                self._finish(part, keep=False)

    # -- the FluidSim interface ----------------------------------------------------------------
    def add_source(self, *a, **k):
        if not self.in_library:
            self._discard_in_flight()  # halos already travelling would be stale after this
        self.sim.add_source(*a, **k)

    def step(self, nsteps: int = 1):
        sim, V, D = self.sim, self.V, self.D
        if self.in_library:          # the library exchanges and overlaps by itself
            sim.step(nsteps)
            return
        if not self.overlap:
            for _ in range(nsteps):
                with self.torch.cuda.stream(self.stream):
                    self.ex.exchange(sim.halo_bytes(-1), sim.halo_bytes(+1), sim.halo_pack, sim.halo_unpack)
                    if self.groups <= 1:
                        sim.step(1)
                        continue
                self._velocity_part()
                with self.torch.cuda.stream(self.stream):
                    sim.step_part(D)
            return
        for _ in range(nsteps):
            if not self.ex.in_flight(V):
                self._start(V)
            if not self.ex.in_flight(D):
                self._start(D)
            if self.split_tail:
                if self.early_advect:
                    with self.torch.cuda.stream(self.stream):
                        sim.step_part(self.pkg.PART_ADVECT_INTERIOR)
                self._finish(V)
                self._velocity_part(tail=False)
                self._finish(D)
                with self.torch.cuda.stream(self.stream):
                    sim.step_part(self.pkg.PART_TAIL_DENSITY_EDGES)
                self._start(V)    # the new edge rows travel while the interior rows are computed
                self._start(D, after_compute=False)
                with self.torch.cuda.stream(self.stream):
                    sim.step_part(self.pkg.PART_TAIL_DENSITY_INTERIOR)
                continue
            self._finish(V)
            self._velocity_part()
            self._start(V)        # next step's velocity halos: travel during the density part
            self._finish(D)
            with self.torch.cuda.stream(self.stream):
                sim.step_part(D)
            self._start(D)        # next step's density halos: travel during the next velocity part

    def sync(self):
        self.sim.sync()

    def read(self, field):
        return self.sim.read(field)

    def diagnostics(self) -> dict:
        """fluid_diagnostics of the whole domain, on every rank: each slab reduces its own rows
        on its GPU (after a refresh of the velocity halo rows the two stencil quantities read),
        then the ranks combine: maxima by all-reduce(MAX), dye sums by all-reduce(SUM)."""
        import torch.distributed as dist
        torch = self.torch
        if self.in_library:
            self.sim.comm_refresh()
        else:
            if not self.ex.in_flight(self.V):   # overlap mode has the next step's halos on the way already
                self._start(self.V)
            self._finish(self.V)
        d = self.sim.diagnostics()
        dev = self.ex.device if dist.get_backend(self.ex.group) == "nccl" else "cpu"
        keys = self.pkg.FluidDiag.MAX_KEYS
        # NaN must survive the combination: all-reduce(MAX) may drop it, so carry it as +inf
        mx = torch.tensor([d[k] for k in keys], dtype=torch.float32, device=dev)
        nan = torch.isnan(mx)
        mx = torch.where(nan, torch.full_like(mx, float("inf")), mx)
        nanf = nan.to(torch.float32)
        sm = torch.tensor(d["dye"], dtype=torch.float64, device=dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.ex.group)
        dist.all_reduce(nanf, op=dist.ReduceOp.MAX, group=self.ex.group)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM, group=self.ex.group)
        mx = torch.where(nanf > 0, torch.full_like(mx, float("nan")), mx)
        out = {k: float(v) for k, v in zip(keys, mx.tolist())}
        out["dye"] = [float(v) for v in sm.tolist()]
        return out

    def gather(self, field):
        """Whole field on rank 0 (None elsewhere); slabs in y order."""
        import numpy as np
        import torch.distributed as dist
        mine = self.read(field)
        parts = [None] * self.world if self.rank == 0 else None
        dist.gather_object(mine, parts, dst=0)
        return np.concatenate(parts, axis=0) if self.rank == 0 else None

    def close(self):
        if not self.in_library:
            self._discard_in_flight()
        self.sim.sync()
        self.sim.close()


class LocalSlabGroup:
    """All slabs of one domain as separate contexts on ONE device, halos moved by handing each
    slab its neighbour's packed send buffer.  Exercises exactly the library's slab path (plan,
    margins, pack/unpack, step parts) without a second GPU; used by the GPU tests to demand
    bit-identical results to the whole-domain context.

    ``split_tail=True`` drives the step the way SlabSim's overlap mode does: the fused gradient
    subtraction + density advection in two launches (edge rows, then interior), with the halo
    blocks for the next step packed between the two."""

    def __init__(self, pkg, n: int, count: int, device_index: int = 0, split_tail: bool = False,
                 early_advect: bool = False, **params):
        import torch
        self.pkg, self.count, self.split_tail = pkg, count, split_tail
        self.early_advect = early_advect   # FLUID_PART_ADVECT_INTERIOR before the velocity halos are unpacked
        self.sims = [pkg.FluidSim(n, device=device_index, slab_rank=r, slab_count=count, **params)
                     for r in range(count)]
        self.groups = self.sims[0].pressure_groups
        dev = torch.device("cuda", device_index)
        self._mk = lambda nbytes: torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        self._bufs = {}
        self.early_valid = False   # halo blocks packed between the edge and interior launches

    def _buf(self, part, r, d):
        key = (part, r, d)
        if key not in self._bufs:
            self._bufs[key] = self._mk(self.sims[r].halo_bytes_part(d, part))
        return self._bufs[key]

    def _sync(self):
        for s in self.sims:
            s.sync()  # contexts have their own streams: packs must land before unpacks read

    def _pack(self, part):
        for r, s in enumerate(self.sims):
            for d in (-1, +1):
                if s.halo_bytes_part(d, part):
                    s.halo_pack_part(d, part, self._buf(part, r, d).data_ptr(), None)

    def _unpack(self, part):
        self._sync()
        for r, s in enumerate(self.sims):
            if s.halo_bytes_part(-1, part):
                s.halo_unpack_part(-1, part, self._buf(part, r - 1, +1).data_ptr(), None)
            if s.halo_bytes_part(+1, part):
                s.halo_unpack_part(+1, part, self._buf(part, r + 1, -1).data_ptr(), None)
        self._sync()

    def add_source(self, *a, **k):
        self.early_valid = False   # blocks packed before the splat are stale
        for s in self.sims:
            s.add_source(*a, **k)

    def _each(self, part):
        for s in self.sims:
            s.step_part(part)

    def step(self, nsteps: int = 1):
        pkg = self.pkg
        V, D, P = pkg.PART_VELOCITY, pkg.PART_DENSITY, pkg.PART_PRESSURE_GROUP
        for _ in range(nsteps):
            if not self.early_valid:
                self._pack(V)
                self._pack(D)
            if self.early_advect:
                self._each(pkg.PART_ADVECT_INTERIOR)
            self._unpack(V)
            self._unpack(D)
            self.early_valid = False
            if self.groups <= 1 and not self.split_tail:
                for s in self.sims:
                    s.step(1)
                self._sync()
                continue
            self._each(pkg.PART_VELOCITY_HEAD)
            for _g in range(self.groups - 1):
                self._pack(P)
                self._unpack(P)
                self._each(P)
            if self.split_tail:
                self._each(pkg.PART_TAIL_DENSITY_EDGES)
                self._pack(V)      # the new edge rows, while the interior rows are still pending
                self._pack(D)
                self.early_valid = True
                self._each(pkg.PART_TAIL_DENSITY_INTERIOR)
            else:
                self._each(pkg.PART_VELOCITY_TAIL)
                self._each(D)
            self._sync()

    def diagnostics(self) -> dict:
        """Whole-domain diagnostics from the slabs' own reductions (velocity halos refreshed first)."""
        import math
        V = self.pkg.PART_VELOCITY
        self._sync()
        self._pack(V)
        self._unpack(V)
        self.early_valid = False
        ds = [s.diagnostics() for s in self.sims]
        out = {}
        for k in self.pkg.FluidDiag.MAX_KEYS:
            vals = [d[k] for d in ds]
            out[k] = float("nan") if any(math.isnan(v) for v in vals) else max(vals)
        out["dye"] = [sum(d["dye"][k] for d in ds) for k in range(4)]
        return out

    def read(self, field):
        import numpy as np
        return np.concatenate([s.read(field) for s in self.sims], axis=0)

    def write(self, field, arr):
        self.early_valid = False
        for s in self.sims:
            s.write(field, arr[s.row0:s.row0 + s.rows])

    def close(self):
        for s in self.sims:
            s.close()
