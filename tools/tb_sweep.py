#!/usr/bin/env python3
"""Sweep the temporal-blocking knobs (env FLUID_TB_*) and print per-kernel times.
Usage: python tools/tb_sweep.py N J [which=jacobi|diffuse]"""
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
J = int(sys.argv[2]) if len(sys.argv) > 2 else 40
which = sys.argv[3] if len(sys.argv) > 3 else "jacobi"
steps = 3 if n > 8192 else 6

if which == "jacobi":
    grid = [(k, c, w) for k in (8, 10, 13, 14, 20) for c in (2, 4) for w in (2048, 4096, 8192) if not (k > 10 and c == 4)]
    names = ("FLUID_TB_KJ", "FLUID_TB_CJ", "FLUID_TB_WAVES")
    kern = "k_jacobi_tb"
    total = J
else:
    grid = [(k, c, w) for k in (5, 10) for c in (2, 4) for w in (2048, 4096, 8192)]
    names = ("FLUID_TB_KD", "FLUID_TB_CD", "FLUID_TB_WAVES")
    kern = "k_diffuse_tb"
    total = 10

for cfg in grid:
    if which == "diffuse" and cfg[0] == 10 and cfg[1] == 4:
        continue  # 267 VGPRs
    for k, v in zip(names, cfg):
        os.environ[k] = str(v)
    sim = pkg.FluidSim(n, pressure_iters=J)
    ss = pkg.SplatStream()
    for _ in range(3):
        sim.add_source(*ss.splat())
    sim.step(2)
    st = {s["name"]: s for s in sim.profile_steps(steps)}
    s = st[kern]
    per_step_ms = s["total_ms"] / steps
    print(json.dumps({"cfg": dict(zip(names, cfg)), "launches_per_step": s["launches"] / steps,
                      "ms_per_launch": round(s["total_ms"] / s["launches"], 4),
                      "ms_per_step_all_sweeps": round(per_step_ms, 4),
                      "us_per_sweep": round(per_step_ms * 1e3 / total, 2)}), flush=True)
    sim.close()
