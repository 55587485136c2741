#!/usr/bin/env python3
"""Step time against the skew between the field buffers' base addresses (env FLUID_ARENA_SKEW).
Usage: python tools/skew_sweep.py N J skew,skew,..."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
n, J = int(sys.argv[1]), int(sys.argv[2])
steps = 5 if n > 8192 else 20
for skew in sys.argv[3].split(","):
    os.environ["FLUID_ARENA_SKEW"] = skew
    sim = pkg.FluidSim(n, pressure_iters=J)
    ss = pkg.SplatStream()
    for _ in range(3):
        sim.add_source(*ss.splat())
    sim.step(3)
    wall = sim.time_steps(steps) / steps
    st = sim.profile_steps(steps)
    out = {"skew": int(skew), "ms_per_step": round(wall, 4)}
    for s in st:
        out[s["name"].replace("k_", "")] = round(s["total_ms"] / s["launches"], 3)
    print(json.dumps(out), flush=True)
    sim.close()
