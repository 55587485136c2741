#!/usr/bin/env python3
"""Time one slab of an N x N domain cut into G row slabs, on one GPU, without the exchange
(halos go stale, which does not change the work): the per-GPU compute time of the multi-GPU run.
Usage: python tools/slab_time.py N J G [rank]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
n, J, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rank = int(sys.argv[4]) if len(sys.argv) > 4 else G // 2
groups = int(sys.argv[5]) if len(sys.argv) > 5 else 0
LIB = {"lib": pkg.FluidLib(os.environ["FLUID_LIB_PATH"])} if os.environ.get("FLUID_LIB_PATH") else {}
sim = pkg.FluidSim(n, pressure_iters=J, slab_rank=rank, slab_count=G, pressure_groups=groups,
                   flags=int(os.environ.get("FLUID_SLAB_FLAGS", "0")), **LIB)


def velocity_part(tail=True):
    if sim.pressure_groups <= 1 and tail:
        sim.step_part(pkg.PART_VELOCITY)
        return
    sim.step_part(pkg.PART_VELOCITY_HEAD)
    for _ in range(sim.pressure_groups - 1):
        sim.step_part(pkg.PART_PRESSURE_GROUP)
    if tail:
        sim.step_part(pkg.PART_VELOCITY_TAIL)



ss = pkg.SplatStream()
for _ in range(3):
    sim.add_source(*ss.splat())
for _ in range(3):
    velocity_part()
    sim.step_part(pkg.PART_DENSITY)
sim.sync()
for mode in (("whole", "parts", "split") if sim.pressure_groups <= 1 else ("parts", "split")):
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        if mode == "whole":
            sim.step(1)
        elif mode == "split":
            velocity_part(tail=False)
            sim.step_part(pkg.PART_TAIL_DENSITY_EDGES)
            sim.step_part(pkg.PART_TAIL_DENSITY_INTERIOR)
        else:
            velocity_part()
            sim.step_part(pkg.PART_DENSITY)
    try:
        sim.sync()
    except Exception as e:  # stale halos may trip the halo flag; timing is still valid
        print("sync:", e)
    dt = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"n": n, "J": J, "slabs": G, "rank": rank, "rows": sim.rows, "groups": sim.pressure_groups, "mode": mode,
                      "ms_per_step": round(dt, 4), "GCUPS_whole_job_if_all_like_this": round(n * n / dt / 1e6, 2)}), flush=True)
try:
    st = sim.profile_steps(1)
    print({s["name"]: round(s["total_ms"] / max(s["launches"], 1), 4) for s in st})
except Exception as e:
    print("profile:", e)
