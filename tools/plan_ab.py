#!/usr/bin/env python3
"""Careful A/B of launch plans: python tools/plan_ab.py N J "KJ,KD,WAVES[,NT];..." [steps] [repeats]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader
pkg = fluid_amd_loader.load()
n, J = int(sys.argv[1]), int(sys.argv[2])
specs = sys.argv[3].split(";")
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
NAMES = ("FLUID_TB_KJ", "FLUID_TB_KD", "FLUID_TB_WAVES", "FLUID_NT", "FLUID_TILE")
res = {s: [] for s in specs}
for _ in range(reps):
    for spec in specs:
        for k in NAMES:
            os.environ.pop(k, None)
        for k, v in zip(NAMES, spec.split(",")):
            if v.strip():
                os.environ[k] = v.strip()
        kw = {"lib": pkg.FluidLib(os.environ["FLUID_LIB_PATH"])} if os.environ.get("FLUID_LIB_PATH") else {}
        sim = pkg.FluidSim(n, pressure_iters=J, **kw)
        ss = pkg.SplatStream()
        for _i in range(3):
            sim.add_source(*ss.splat())
        sim.step(10)
        res[spec].append(sim.time_steps(steps) / steps)
        sim.close()
for s in specs:
    print(json.dumps({"plan": s, "ms": [round(x, 4) for x in res[s]], "min": round(min(res[s]), 4)}))
