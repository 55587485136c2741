#!/usr/bin/env python3
"""Time whole steps and every kernel for a list of temporal-blocking plans.
Usage: python tools/plan_sweep.py N J "kj,kd,waves[,minchunk[,cj,cd]];kj,kd,waves;..."
A field left empty keeps the library's default for that knob."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
LIB = pkg.FluidLib(os.environ["FLUID_LIB_PATH"]) if os.environ.get("FLUID_LIB_PATH") else None
n, J = int(sys.argv[1]), int(sys.argv[2])
NAMES = ("FLUID_TB_KJ", "FLUID_TB_KD", "FLUID_TB_WAVES", "FLUID_TB_MINCHUNK", "FLUID_TB_CJ", "FLUID_TB_CD")
steps = 3 if n > 8192 else 10

for spec in sys.argv[3].split(";"):
    vals = spec.split(",")
    for name in NAMES:
        os.environ.pop(name, None)
    for name, v in zip(NAMES, vals):
        if v.strip():
            os.environ[name] = v.strip()
    sim = pkg.FluidSim(n, pressure_iters=J, **({"lib": LIB} if LIB else {}))
    ss = pkg.SplatStream()
    for _ in range(3):
        sim.add_source(*ss.splat())
    sim.step(3)
    wall = sim.time_steps(steps) / steps
    st = sim.profile_steps(steps)
    out = {"plan": spec, "ms_per_step": round(wall, 4)}
    for s in st:
        out[s["name"].replace("k_", "")] = f'{s["launches"] // steps}x{s["total_ms"] / s["launches"]:.3f}'
    print(json.dumps(out), flush=True)
    sim.close()
