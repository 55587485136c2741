"""ctypes binding of include/fluid.h.

``FluidLib`` binds any shared library that implements the header; ``FluidSim`` is one
context.  The product path is ``FluidSim(...)`` with the default library
(csrc/libfluid_hip.so).  Tests bind the CPU oracle through the same class by passing its
path explicitly — this package never does.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

FIELD_VELOCITY, FIELD_DENSITY, FIELD_PRESSURE, FIELD_DIVERGENCE, FIELD_VORTICITY = range(5)
FLAG_VORTICITY, FLAG_NO_GRAPH, FLAG_UNFUSED = 1, 2, 4
_FIELD_COMPONENTS = {FIELD_VELOCITY: 2, FIELD_DENSITY: 4, FIELD_PRESSURE: 1, FIELD_DIVERGENCE: 1,
                     FIELD_VORTICITY: 1}

# Every symbol include/fluid.h declares (checked against the header text in tests).
ABI_SYMBOLS = [
    "fluid_default_params", "fluid_init", "fluid_destroy", "fluid_reset", "fluid_add_source",
    "fluid_step", "fluid_prepare_steps", "fluid_step_noalloc", "fluid_debug_sink", "fluid_debug_chain_trace", "fluid_advect_velocity", "fluid_diffuse", "fluid_vorticity", "fluid_project",
    "fluid_advect_density", "fluid_read", "fluid_write", "fluid_sync", "fluid_diagnostics", "fluid_checksum", "fluid_geometry",
    "fluid_strerror", "fluid_last_error", "fluid_time_steps", "fluid_profile_steps",
    "fluid_stream", "fluid_set_stream",
    "fluid_slab_plan", "fluid_halo_bytes", "fluid_halo_pack", "fluid_halo_unpack",
    "fluid_step_part", "fluid_halo_bytes_part", "fluid_halo_pack_part", "fluid_halo_unpack_part",
    "fluid_comm_unique_id", "fluid_comm_init_rccl", "fluid_comm_attach_rccl", "fluid_comm_attach",
    "fluid_comm_detach", "fluid_comm_refresh", "fluid_comm_info", "fluid_comm_selftest_rccl",
]
COMM_ID_BYTES = 128
PART_VELOCITY, PART_DENSITY = 1, 2
PART_VELOCITY_HEAD, PART_PRESSURE_GROUP, PART_VELOCITY_TAIL = 4, 8, 16
PART_TAIL_DENSITY_EDGES, PART_TAIL_DENSITY_INTERIOR = 32, 64
PART_ADVECT_INTERIOR = 128


class FluidParams(C.Structure):
    _fields_ = [
        ("n", C.c_int), ("diffusion_iters", C.c_int), ("pressure_iters", C.c_int),
        ("dt", C.c_float), ("viscosity", C.c_float), ("splat_radius", C.c_float),
        ("device", C.c_int), ("flags", C.c_uint), ("slab_rank", C.c_int), ("slab_count", C.c_int),
        ("halo_rows", C.c_int), ("pressure_groups", C.c_int), ("ny", C.c_int), ("inner_slabs", C.c_int), ("reserved", C.c_int * 2),
        ("n_devices", C.c_int), ("device_ids", C.c_int * 8),
    ]


class SlabInfo(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("n", "row0", "rows", "halo_vel", "halo_dens", "store_row0",
                                       "store_rows", "has_lo", "has_hi", "pressure_groups", "halo_prs")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int), ("total_ms", C.c_float),
                ("alg_bytes", C.c_double), ("moved_bytes", C.c_double)]


class FluidDiag(C.Structure):
    _fields_ = [("max_abs_u", C.c_float), ("max_abs_v", C.c_float), ("max_abs_div", C.c_float),
                ("max_abs_p", C.c_float), ("max_jacobi_update", C.c_float),
                ("max_abs_div_velocity", C.c_float), ("dye", C.c_double * 4)]

    MAX_KEYS = ("max_abs_u", "max_abs_v", "max_abs_div", "max_abs_p", "max_jacobi_update",
                "max_abs_div_velocity")

    def as_dict(self):
        d = {k: float(getattr(self, k)) for k in self.MAX_KEYS}
        d["dye"] = [float(x) for x in self.dye]
        return d


# fluid_transport: the four callbacks of a caller-provided halo transport
TRANSPORT_GROUP_START = C.CFUNCTYPE(C.c_int, C.c_void_p)
TRANSPORT_SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
TRANSPORT_RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
TRANSPORT_GROUP_END = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class FluidTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("group_start", TRANSPORT_GROUP_START), ("send", TRANSPORT_SEND),
                ("recv", TRANSPORT_RECV), ("group_end", TRANSPORT_GROUP_END), ("capturable", C.c_int)]


class FluidCommStats(C.Structure):
    _fields_ = [("attached", C.c_int), ("capturable", C.c_int), ("graph", C.c_int), ("graphs_built", C.c_int),
                ("exchanges", C.c_longlong), ("bytes_sent", C.c_longlong), ("transport", C.c_char * 16),
                ("note", C.c_char * 160), ("ranks", C.c_int), ("reserved", C.c_int)]

    def as_dict(self):
        return {"attached": bool(self.attached), "capturable": bool(self.capturable), "graph": bool(self.graph),
                "graphs_built": self.graphs_built, "exchanges": self.exchanges, "bytes_sent": self.bytes_sent,
                "transport": self.transport.decode(), "note": self.note.decode(), "ranks": self.ranks}


class FluidError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str = ""):
        self.code = code
        super().__init__(f"{what}: error {code}" + (f" ({detail})" if detail else ""))


def hip_lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libfluid_hip.so")


class FluidLib:
    """A loaded implementation of include/fluid.h."""

    def __init__(self, path: Optional[str] = None):
        path = path or hip_lib_path()
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        self.path = path
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME
        # as /opt/rocm's).  Whichever loads first serves both; torch cannot find its GPUs through
        # the system copy, while this library runs on either, so let torch's load first when
        # torch is installed (torch.distributed is this package's multi-GPU transport anyway).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        self.dll = C.CDLL(path, mode=C.RTLD_LOCAL)
        d = self.dll
        P = C.c_void_p

        def sig(name, res, *args, required=True):
            try:
                f = getattr(d, name)
            except AttributeError:
                if required:
                    raise
                return
            f.restype, f.argtypes = res, list(args)

        sig("fluid_default_params", C.c_int, C.POINTER(FluidParams))
        sig("fluid_init", C.c_int, C.POINTER(P), C.POINTER(FluidParams))
        sig("fluid_destroy", None, P)
        sig("fluid_reset", C.c_int, P)
        sig("fluid_add_source", C.c_int, P, C.c_float, C.c_float, C.c_float, C.c_float,
            C.POINTER(C.c_float))
        sig("fluid_step", C.c_int, P, C.c_int)
        sig("fluid_prepare_steps", C.c_int, P, C.c_int, required=False)
        sig("fluid_step_noalloc", C.c_int, P, C.c_int, required=False)
        sig("fluid_debug_sink", C.c_int, P, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_longlong), required=False)
        for s in ("fluid_advect_velocity", "fluid_diffuse", "fluid_vorticity", "fluid_project",
                  "fluid_advect_density", "fluid_sync"):
            sig(s, C.c_int, P)
        sig("fluid_read", C.c_int, P, C.c_int, P, C.c_size_t)
        sig("fluid_write", C.c_int, P, C.c_int, P, C.c_size_t)
        sig("fluid_diagnostics", C.c_int, P, C.POINTER(FluidDiag), required=False)
        sig("fluid_checksum", C.c_int, P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong), required=False)
        sig("fluid_debug_chain_trace", C.c_int, P, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), required=False)
        sig("fluid_geometry", C.c_int, P, C.POINTER(C.c_int), C.POINTER(C.c_int),
            C.POINTER(C.c_int))
        sig("fluid_strerror", C.c_char_p, C.c_int)
        sig("fluid_last_error", C.c_char_p, P)
        # measurement / stream hooks: the oracle does not implement them
        sig("fluid_time_steps", C.c_int, P, C.c_int, C.POINTER(C.c_float), required=False)
        sig("fluid_profile_steps", C.c_int, P, C.c_int, C.POINTER(KernelStat), C.c_int,
            C.POINTER(C.c_int), required=False)
        sig("fluid_stream", P, P, required=False)
        sig("fluid_set_stream", C.c_int, P, P, required=False)
        sig("fluid_slab_plan", C.c_int, C.POINTER(FluidParams), C.POINTER(SlabInfo), required=False)
        sig("fluid_halo_bytes", C.c_size_t, P, C.c_int, required=False)
        sig("fluid_halo_pack", C.c_int, P, C.c_int, P, required=False)
        sig("fluid_halo_unpack", C.c_int, P, C.c_int, P, required=False)
        sig("fluid_step_part", C.c_int, P, C.c_int, required=False)
        sig("fluid_halo_bytes_part", C.c_size_t, P, C.c_int, C.c_int, required=False)
        sig("fluid_halo_pack_part", C.c_int, P, C.c_int, C.c_int, P, P, required=False)
        sig("fluid_halo_unpack_part", C.c_int, P, C.c_int, C.c_int, P, P, required=False)

        sig("fluid_comm_unique_id", C.c_int, P, C.c_size_t, required=False)
        sig("fluid_comm_init_rccl", C.c_int, P, P, C.c_size_t, required=False)
        sig("fluid_comm_attach_rccl", C.c_int, P, P, required=False)
        sig("fluid_comm_attach", C.c_int, P, C.POINTER(FluidTransport), required=False)
        sig("fluid_comm_detach", C.c_int, P, required=False)
        sig("fluid_comm_refresh", C.c_int, P, required=False)
        sig("fluid_comm_info", C.c_int, P, C.POINTER(FluidCommStats), required=False)
        sig("fluid_comm_selftest_rccl", C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, required=False)

    def comm_unique_id(self) -> bytes:
        """fluid_comm_unique_id (ncclGetUniqueId): rank 0 makes it, every rank passes it to comm_init_rccl."""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        rc = self.dll.fluid_comm_unique_id(buf, COMM_ID_BYTES)
        if rc:
            raise FluidError(rc, "fluid_comm_unique_id", self.dll.fluid_strerror(rc).decode())
        return buf.raw

    def comm_selftest_rccl(self, device=0, nbytes=1 << 20, use_graph=True) -> str:
        msg = C.create_string_buffer(256)
        rc = self.dll.fluid_comm_selftest_rccl(device, nbytes, int(use_graph), msg, 256)
        if rc:
            raise FluidError(rc, "fluid_comm_selftest_rccl", msg.value.decode())
        return msg.value.decode()

    def slab_plan(self, n, slab_rank, slab_count, *, diffusion_iters=None, pressure_iters=None,
                  halo_rows=0, pressure_groups=0, ny=0, flags=0) -> dict:
        """fluid_slab_plan: pure host arithmetic, needs no GPU."""
        p = self.default_params()
        p.n, p.slab_rank, p.slab_count, p.halo_rows = n, slab_rank, slab_count, halo_rows
        p.flags = flags   # FLAG_VORTICITY deepens the velocity halo by two rows
        p.pressure_groups = pressure_groups
        p.ny = ny
        if diffusion_iters is not None:
            p.diffusion_iters = diffusion_iters
        if pressure_iters is not None:
            p.pressure_iters = pressure_iters
        info = SlabInfo()
        rc = self.dll.fluid_slab_plan(C.byref(p), C.byref(info))
        if rc:
            raise FluidError(rc, "fluid_slab_plan", self.dll.fluid_strerror(rc).decode())
        return info.as_dict()

    def default_params(self) -> FluidParams:
        p = FluidParams()
        rc = self.dll.fluid_default_params(C.byref(p))
        if rc:
            raise FluidError(rc, "fluid_default_params")
        return p


_default_lib: Optional[FluidLib] = None


def _get_default_lib() -> FluidLib:
    global _default_lib
    if _default_lib is None:
        _default_lib = FluidLib()
    return _default_lib


class FluidSim:
    """One simulation context.  Mirrors the reference's host functions:

    ===================  =============================================
    reference            here
    ===================  =============================================
    initBuffers()        FluidSim(n=...)              kernel.cu:481-516
    resetSimulation()    .reset()                     kernel.cu:518-528
    addForce()           .add_source(sx,sy,fx,fy,rgba) kernel.cu:687-699
    simulationStep()     .step(k)                     kernel.cu:701-708
    D2H readback         .read(field)                 kernel.cu:753
    ===================  =============================================
    """

    def __init__(self, n: Optional[int] = None, *, diffusion_iters: Optional[int] = None,
                 pressure_iters: Optional[int] = None, dt: Optional[float] = None,
                 viscosity: Optional[float] = None, splat_radius: Optional[float] = None,
                 device: int = 0, flags: int = 0, slab_rank: int = 0, slab_count: int = 0,
                 halo_rows: int = 0, pressure_groups: int = 0, ny: int = 0, inner_slabs: int = 0,
                 devices=None, lib: Optional[FluidLib] = None):
        self.lib = lib or _get_default_lib()
        p = self.lib.default_params()
        if n is not None:
            p.n = n
        if diffusion_iters is not None:
            p.diffusion_iters = diffusion_iters
        if pressure_iters is not None:
            p.pressure_iters = pressure_iters
        if dt is not None:
            p.dt = dt
        if viscosity is not None:
            p.viscosity = viscosity
        if splat_radius is not None:
            p.splat_radius = splat_radius
        p.device, p.flags = device, flags
        p.slab_rank, p.slab_count, p.halo_rows = slab_rank, slab_count, halo_rows
        p.pressure_groups = pressure_groups
        p.ny = ny   # rows; 0 = n (square, the reference's case); n is then the number of columns
        p.inner_slabs = inner_slabs   # >= 2: the domain as that many slabs side by side on this GPU (fluid.h)
        if devices:                   # one context over several GPUs: slab g on devices[g] (fluid.h, n_devices)
            p.n_devices = len(devices)
            for i, d in enumerate(devices):
                p.device_ids[i] = d
        self.params = p
        self._h = C.c_void_p()
        rc = self.lib.dll.fluid_init(C.byref(self._h), C.byref(p))
        if rc:
            self._h = C.c_void_p()
            raise FluidError(rc, "fluid_init", self.lib.dll.fluid_strerror(rc).decode())
        n_, r0, rows = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.dll.fluid_geometry(self._h, C.byref(n_), C.byref(r0), C.byref(rows)),
                  "fluid_geometry")
        self.n, self.row0, self.rows = n_.value, r0.value, rows.value
        self.pressure_groups = 1
        if slab_count > 1 and pressure_groups > 1:
            self.pressure_groups = self.lib.slab_plan(
                self.n, slab_rank, slab_count, diffusion_iters=p.diffusion_iters, pressure_iters=p.pressure_iters,
                halo_rows=halo_rows, pressure_groups=pressure_groups, ny=ny, flags=flags)["pressure_groups"]

    # -- plumbing ---------------------------------------------------------------
    def _chk(self, rc: int, what: str):
        if rc:
            detail = self.lib.dll.fluid_strerror(rc).decode()
            last = self.lib.dll.fluid_last_error(self._h)
            if last:
                detail += ": " + last.decode()
            raise FluidError(rc, what, detail)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.dll.fluid_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- the reference's host functions -----------------------------------------
    def reset(self):
        self._chk(self.lib.dll.fluid_reset(self._h), "fluid_reset")

    def add_source(self, sx, sy, fx, fy, rgba=(1.0, 1.0, 1.0, 1.0)):
        col = (C.c_float * 4)(*rgba)
        self._chk(self.lib.dll.fluid_add_source(self._h, sx, sy, fx, fy, col), "fluid_add_source")

    def step(self, nsteps: int = 1):
        self._chk(self.lib.dll.fluid_step(self._h, nsteps), "fluid_step")

    def step_noalloc(self, nsteps: int = 1):
        """fluid_step_noalloc: the same steps by direct launches out of static scratch; allocates nothing."""
        self._chk(self.lib.dll.fluid_step_noalloc(self._h, nsteps), "fluid_step_noalloc")

    def debug_sink(self, which: int, fill: bool, pattern: int = 0x5A5A5A5A) -> int:
        """fluid_debug_sink: fill the context's own sink row (which=0) or the GPU's static scratch
        (which=1) with `pattern`, or (fill=False) count the words that no longer hold it."""
        d = C.c_longlong(-1)
        self._chk(self.lib.dll.fluid_debug_sink(self._h, which, int(fill), pattern, None if fill else C.byref(d)),
                  "fluid_debug_sink")
        return d.value

    def prepare_steps(self, nsteps: int):
        """fluid_prepare_steps: build the graphs step(nsteps) replays, without running anything."""
        if hasattr(self.lib.dll, "fluid_prepare_steps"):
            self._chk(self.lib.dll.fluid_prepare_steps(self._h, nsteps), "fluid_prepare_steps")

    def advect_velocity(self):
        self._chk(self.lib.dll.fluid_advect_velocity(self._h), "fluid_advect_velocity")

    def diffuse(self):
        self._chk(self.lib.dll.fluid_diffuse(self._h), "fluid_diffuse")

    def vorticity(self):
        self._chk(self.lib.dll.fluid_vorticity(self._h), "fluid_vorticity")

    def project(self):
        self._chk(self.lib.dll.fluid_project(self._h), "fluid_project")

    def advect_density(self):
        self._chk(self.lib.dll.fluid_advect_density(self._h), "fluid_advect_density")

    def sync(self):
        self._chk(self.lib.dll.fluid_sync(self._h), "fluid_sync")

    def diagnostics(self) -> dict:
        """fluid_diagnostics: maxima and dye sums over the owned rows (wavefront reductions)."""
        d = FluidDiag()
        self._chk(self.lib.dll.fluid_diagnostics(self._h, C.byref(d)), "fluid_diagnostics")
        return d.as_dict()

    def chain_trace(self, max_records=1 << 20):
        """fluid_debug_chain_trace: (records, 4) uint64 time stamps of the last temporally blocked launch
        (contexts made with FLUID_TB_TRACE=1)."""
        buf = np.zeros(4 * max_records, np.uint64)
        words = C.c_size_t(0)
        self._chk(self.lib.dll.fluid_debug_chain_trace(self._h, buf.ctypes.data, buf.size, C.byref(words)),
                  "fluid_debug_chain_trace")
        return buf[:words.value - words.value % 4].reshape(-1, 4)

    def checksum(self, field: int, row0: int = 0, rows: int = 0) -> int:
        """fluid_checksum: sum of the 32-bit words of global rows [row0, row0 + rows) of the current
        half of `field` modulo 2^64, computed on the device (rows = 0: all owned rows)."""
        out = C.c_ulonglong()
        self._chk(self.lib.dll.fluid_checksum(self._h, field, row0, rows, C.byref(out)), "fluid_checksum")
        return out.value

    def read(self, field: int) -> np.ndarray:
        comps = _FIELD_COMPONENTS[field]
        shape = (self.rows, self.n) if comps == 1 else (self.rows, self.n, comps)
        out = np.empty(shape, dtype=np.float32)
        self._chk(self.lib.dll.fluid_read(self._h, field, out.ctypes.data_as(C.c_void_p),
                                          out.nbytes), "fluid_read")
        return out

    def write(self, field: int, arr: np.ndarray):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        self._chk(self.lib.dll.fluid_write(self._h, field, a.ctypes.data_as(C.c_void_p), a.nbytes),
                  "fluid_write")

    # -- measurement hooks --------------------------------------------------------
    def time_steps(self, nsteps: int) -> float:
        ms = C.c_float()
        self._chk(self.lib.dll.fluid_time_steps(self._h, nsteps, C.byref(ms)), "fluid_time_steps")
        return ms.value

    def profile_steps(self, nsteps: int):
        stats = (KernelStat * 32)()
        cnt = C.c_int()
        self._chk(self.lib.dll.fluid_profile_steps(self._h, nsteps, stats, 32, C.byref(cnt)),
                  "fluid_profile_steps")
        return [dict(name=stats[i].name.decode(), launches=stats[i].launches,
                     total_ms=stats[i].total_ms, alg_bytes=stats[i].alg_bytes,
                     moved_bytes=stats[i].moved_bytes)
                for i in range(cnt.value)]

    # -- slab halos -----------------------------------------------------------------
    def halo_bytes(self, direction: int) -> int:
        return self.lib.dll.fluid_halo_bytes(self._h, direction)

    def halo_pack(self, direction: int, dev_ptr: int):
        self._chk(self.lib.dll.fluid_halo_pack(self._h, direction, C.c_void_p(dev_ptr)), "fluid_halo_pack")

    def halo_unpack(self, direction: int, dev_ptr: int):
        self._chk(self.lib.dll.fluid_halo_unpack(self._h, direction, C.c_void_p(dev_ptr)),
                  "fluid_halo_unpack")

    def step_part(self, part: int):
        self._chk(self.lib.dll.fluid_step_part(self._h, part), "fluid_step_part")

    def halo_bytes_part(self, direction: int, parts: int) -> int:
        return self.lib.dll.fluid_halo_bytes_part(self._h, direction, parts)

    def halo_pack_part(self, direction: int, part: int, dev_ptr: int, stream: int = 0):
        self._chk(self.lib.dll.fluid_halo_pack_part(self._h, direction, part, C.c_void_p(dev_ptr),
                                                    C.c_void_p(stream)), "fluid_halo_pack_part")

    def halo_unpack_part(self, direction: int, part: int, dev_ptr: int, stream: int = 0):
        self._chk(self.lib.dll.fluid_halo_unpack_part(self._h, direction, part, C.c_void_p(dev_ptr),
                                                      C.c_void_p(stream)), "fluid_halo_unpack_part")

    # -- halo transport under the C ABI ---------------------------------------------
    def comm_init_rccl(self, unique_id: bytes):
        """fluid_comm_init_rccl: collective; afterwards step(n) exchanges the halos itself over RCCL."""
        self._chk(self.lib.dll.fluid_comm_init_rccl(self._h, unique_id, len(unique_id)), "fluid_comm_init_rccl")

    def comm_attach(self, transport: "FluidTransport"):
        """fluid_comm_attach: a caller-provided transport (keep `transport` alive while attached)."""
        self._transport = transport
        self._chk(self.lib.dll.fluid_comm_attach(self._h, C.byref(transport)), "fluid_comm_attach")

    def comm_detach(self):
        self._chk(self.lib.dll.fluid_comm_detach(self._h), "fluid_comm_detach")
        self._transport = None

    def comm_refresh(self):
        """fluid_comm_refresh: halo rows of the current state in place (collective)."""
        self._chk(self.lib.dll.fluid_comm_refresh(self._h), "fluid_comm_refresh")

    def comm_info(self) -> dict:
        st = FluidCommStats()
        self._chk(self.lib.dll.fluid_comm_info(self._h, C.byref(st)), "fluid_comm_info")
        return st.as_dict()

    def stream(self) -> int:
        return self.lib.dll.fluid_stream(self._h) or 0

    def set_stream(self, hip_stream: int):
        self._chk(self.lib.dll.fluid_set_stream(self._h, C.c_void_p(hip_stream)),
                  "fluid_set_stream")
