#!/usr/bin/env python3
"""hipGraph replay vs direct launches of the same step: python tools/graph_ab.py N J [steps]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader
pkg = fluid_amd_loader.load()
n, J = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for rep in range(2):
    for name, flags, gs in (("graph", 0, 1), ("graph x4", 0, 4), ("graph x16", 0, 16), ("direct", pkg.FLAG_NO_GRAPH, 1)):
        os.environ["FLUID_GRAPH_STEPS"] = str(gs)
        sim = pkg.FluidSim(n, pressure_iters=J, flags=flags)
        ss = pkg.SplatStream()
        for _ in range(3):
            sim.add_source(*ss.splat())
        sim.step(10)
        print(json.dumps({"n": n, "mode": name, "ms_per_step": round(sim.time_steps(steps) / steps, 4)}), flush=True)
        sim.close()
