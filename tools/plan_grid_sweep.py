#!/usr/bin/env python3
"""Grid sweep of the pipeline launch plan (round 3; the rows of kPlan in csrc/fluid_api.hip come from it):
every combination of sweeps per pressure launch, sweeps per diffusion launch and resident rounds
(FLUID_TB_KJ / _KD / _ROUNDS_PCT) against the library's default, ms per step by HIP events (min of 3).
Usage: tools/plan_grid_sweep.py N:J[,N:J..] KJ[,KJ..] KD[,KD..] PCT[,PCT..]"""
import os, sys, itertools, json
sys.path.insert(0, os.getcwd())
import fluid_amd_loader
pkg = fluid_amd_loader.load()
sizes = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1].split(",")]
kjs = [int(v) for v in sys.argv[2].split(",")]
kds = [int(v) for v in sys.argv[3].split(",")]
pcts = [int(v) for v in sys.argv[4].split(",")]
for n, J in sizes:
    res = []
    for kj, kd, pct in itertools.product(kjs, kds, pcts):
        os.environ.update(FLUID_TB_KJ=str(kj), FLUID_TB_KD=str(kd), FLUID_TB_ROUNDS_PCT=str(pct))
        sim = pkg.FluidSim(n, pressure_iters=J)
        ss = pkg.SplatStream()
        for _ in range(3):
            sim.add_source(*ss.splat())
        steps = 32 if n <= 4096 else 16
        sim.step(steps)
        ms = min(sim.time_steps(steps) for _ in range(3)) / steps
        res.append((round(ms * 1e3, 1), kj, kd, pct))
        sim.close()
    for k in ("FLUID_TB_KJ", "FLUID_TB_KD", "FLUID_TB_ROUNDS_PCT"):
        os.environ.pop(k)
    sim = pkg.FluidSim(n, pressure_iters=J)
    ss = pkg.SplatStream()
    for _ in range(3):
        sim.add_source(*ss.splat())
    sim.step(16)
    base = min(sim.time_steps(16) for _ in range(3)) / 16
    sim.close()
    res.sort()
    print(f"n={n} J={J} default={base*1e3:.1f} best:", res[:6], flush=True)
