#!/usr/bin/env python3
"""ms per step (HIP events, min of 3 windows) and per-kernel times for a few whole-domain sizes.
Usage: tools/step_time.py [n:J ...]   (FLUID_TB_* / FLUID_TILE / FLUID_NT apply; FLUID_HIP_LIB=path times an experimental build)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
lib = pkg.FluidLib(os.environ["FLUID_HIP_LIB"]) if os.environ.get("FLUID_HIP_LIB") else None   # an experimental build
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(2048, 40), (4096, 40), (8192, 40), (16384, 40)]
for n, J in cases:
    sim = pkg.FluidSim(n, pressure_iters=J, lib=lib)
    ss = pkg.SplatStream()
    for _ in range(4):
        sim.add_source(*ss.splat())
    steps = 32 if n <= 8192 else 16
    sim.step(steps)
    ms = min(sim.time_steps(steps) for _ in range(3)) / steps
    ks = {s["name"]: round(s["total_ms"] / s["launches"] * 1e3, 1) for s in sim.profile_steps(4)}
    print(f"{n:6d}^2 J={J:3d}: {ms * 1e3:9.1f} us/step  {ks}", flush=True)
    sim.close()
