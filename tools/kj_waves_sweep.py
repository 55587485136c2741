#!/usr/bin/env python3
"""Sweep of the pressure launches' plan (sweeps per launch x target waves) per grid size, with the
diffusion launches' own wave target left at the table's value.  Usage: tools/kj_waves_sweep.py n [J]"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
n = int(sys.argv[1])
J = int(sys.argv[2]) if len(sys.argv) > 2 else 40
res = []
base = None
for kj, waves in itertools.product((0, 8, 10, 13, 14), (2048, 2900, 3072, 3900, 4096, 6100, 8192, 12288)):
    if kj == 0 and waves != 2048:
        continue
    for k in ("FLUID_TB_KJ", "FLUID_TB_WAVES"):
        os.environ.pop(k, None)
    if kj:
        os.environ["FLUID_TB_KJ"], os.environ["FLUID_TB_WAVES"] = str(kj), str(waves)
        os.environ["FLUID_TB_WAVES_D"] = "2900" if n <= 4096 else "0"
    sim = pkg.FluidSim(n, pressure_iters=J)
    ss = pkg.SplatStream()
    for _ in range(4):
        sim.add_source(*ss.splat())
    steps = 32 if n <= 8192 else 8
    sim.step(steps)
    ms = min(sim.time_steps(steps) for _ in range(3)) / steps
    sim.close()
    if kj == 0:
        base = ms
        print(f"{n}^2 J={J} table plan: {ms * 1e3:.1f} us/step", flush=True)
    else:
        res.append((ms, kj, waves))
res.sort()
for ms, kj, waves in res[:8]:
    print(f"   kj={kj:2d} waves={waves:6d}: {ms * 1e3:9.1f} us/step ({(ms / base - 1) * 100:+.1f} %)")
