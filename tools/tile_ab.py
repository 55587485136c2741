#!/usr/bin/env python3
"""A/B of two builds of the library on the small grids: ms per step (HIP events, min of 5 windows)
and the per-kernel times.  Usage: tools/tile_ab.py LIB_A.so LIB_B.so"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
CASES = [(256, 10), (512, 10), (512, 40), (768, 40), (1024, 10), (1024, 40)]
for path in sys.argv[1:]:
    lib = pkg.FluidLib(path)
    print(path)
    for n, J in CASES:
        sim = pkg.FluidSim(n, pressure_iters=J, lib=lib)
        ss = pkg.SplatStream()
        for _ in range(3):
            sim.add_source(*ss.splat())
        sim.step(32)
        ms = min(sim.time_steps(160) for _ in range(5)) / 160
        ks = {s["name"]: round(s["total_ms"] / s["launches"] * 1e3, 2) for s in sim.profile_steps(8)}
        print(f"  {n:5d}^2 J={J:3d}: {ms * 1e3:7.2f} us/step   {ks}")
        sim.close()
