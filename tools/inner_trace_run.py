#!/usr/bin/env python3
"""A short run of an inner-slab context for a rocprofv3 kernel trace: python3 tools/inner_trace_run.py N J G [steps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402
pkg = fluid_amd_loader.load()
n, J, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
sim = pkg.FluidSim(n, pressure_iters=J, inner_slabs=G)
ss = pkg.SplatStream()
sim.add_source(*ss.splat())
sim.step(steps)
sim.sync()
sim.step(steps)
sim.sync()
