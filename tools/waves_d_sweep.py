#!/usr/bin/env python3
"""Sweep of the diffusion launches' wave target (FLUID_TB_WAVES_D) per grid size: the diffusion
pipeline holds 160 VGPRs (3 waves per SIMD: 3072 resident), the pressure one 108-140 (4 or 3)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
for n, J in ((2048, 40), (3072, 40), (4096, 40), (8192, 40), (16384, 40)):
    for wd in (0, 1500, 2048, 2900, 3072, 4096, 6000, 9000, 12000):
        os.environ["FLUID_TB_WAVES_D"] = str(wd)
        sim = pkg.FluidSim(n, pressure_iters=J)
        ss = pkg.SplatStream()
        for _ in range(4):
            sim.add_source(*ss.splat())
        steps = 32 if n <= 8192 else 8
        sim.step(steps)
        ms = min(sim.time_steps(steps) for _ in range(3)) / steps
        kd = [s for s in sim.profile_steps(4) if s["name"] == "k_diffuse_tb"][0]
        print(f"{n:6d}^2 waves_d={wd:6d}: {ms * 1e3:9.1f} us/step  k_diffuse_tb {kd['total_ms'] / kd['launches'] * 1e3:8.1f} us", flush=True)
        sim.close()
