#!/usr/bin/env python3
"""Per-kernel average launch time of one step (library's own HIP-event profile): python tools/kernel_times.py N J"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader
pkg = fluid_amd_loader.load()
n, J = int(sys.argv[1]), int(sys.argv[2])
kw = {"lib": pkg.FluidLib(os.environ["FLUID_LIB_PATH"])} if os.environ.get("FLUID_LIB_PATH") else {}
sim = pkg.FluidSim(n, pressure_iters=J, **kw)
ss = pkg.SplatStream()
for _ in range(3):
    sim.add_source(*ss.splat())
sim.step(5)
st = sim.profile_steps(5)
print({s["name"]: round(s["total_ms"] / max(s["launches"], 1), 4) for s in st})
