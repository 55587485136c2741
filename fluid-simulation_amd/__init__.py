"""fluid-simulation_amd — MI355X-native stable-fluids grid solver (host-side mirror).

Thin ctypes mirror of the C-ABI in ``include/fluid.h`` (libfluid_hip.so, hand-written
gfx950 HIP kernels).  The directory name carries a hyphen, so load it with
``fluid_amd_loader.load()`` (repo root) or ``importlib`` under the module name
``fluid_simulation_amd``.

There is no CPU fallback: constructing a :class:`FluidSim` without the built HIP
library or without a GPU raises.  Nothing in this package imports ``oracle/``.
"""
from .capi import (FluidLib, FluidSim, FluidError, FluidParams, FluidDiag, KernelStat, SlabInfo, hip_lib_path,
                   FluidTransport, FluidCommStats, COMM_ID_BYTES,
                   FIELD_VELOCITY, FIELD_DENSITY, FIELD_PRESSURE, FIELD_DIVERGENCE,
                   FIELD_VORTICITY, FLAG_VORTICITY, FLAG_NO_GRAPH, FLAG_UNFUSED, ABI_SYMBOLS, PART_VELOCITY, PART_DENSITY,
                   PART_VELOCITY_HEAD, PART_PRESSURE_GROUP, PART_VELOCITY_TAIL,
                   PART_TAIL_DENSITY_EDGES, PART_TAIL_DENSITY_INTERIOR, PART_ADVECT_INTERIOR)
from . import capi, viz, replay as replay_mod
from .replay import replay, parse_events
from .splats import SplatStream, COLOR_PRESETS, alg_bytes_per_cell, run_seeded, DEFAULT_SEED

__all__ = [
    "FluidLib", "FluidSim", "FluidError", "FluidParams", "FluidDiag", "KernelStat", "hip_lib_path",
    "FluidTransport", "FluidCommStats", "COMM_ID_BYTES",
    "FIELD_VELOCITY", "FIELD_DENSITY", "FIELD_PRESSURE", "FIELD_DIVERGENCE", "FIELD_VORTICITY",
    "FLAG_VORTICITY", "FLAG_NO_GRAPH", "FLAG_UNFUSED", "ABI_SYMBOLS", "PART_VELOCITY", "PART_DENSITY",
    "PART_VELOCITY_HEAD", "PART_PRESSURE_GROUP", "PART_VELOCITY_TAIL",
    "PART_TAIL_DENSITY_EDGES", "PART_TAIL_DENSITY_INTERIOR", "PART_ADVECT_INTERIOR",
    "capi", "viz", "replay", "parse_events", "SplatStream", "COLOR_PRESETS", "alg_bytes_per_cell", "run_seeded", "DEFAULT_SEED",
]
