#!/usr/bin/env python3
"""Wave-split Jacobi kernel (fluid_kernels_ws.hip, FLUID_WS=K) against the one-wave pipeline: same bits?
per-launch and per-step times.  Usage: tools/ws_ab.py N[:NY] J K[,K..] [BLOCKS,BLOCKS..]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fluid_amd_loader  # noqa: E402
from util import random_state  # noqa: E402

pkg = fluid_amd_loader.load()
shape = [int(v) for v in sys.argv[1].split(":")]
n, ny = shape[0], (shape[1] if len(shape) > 1 else 0)
J = int(sys.argv[2])
Ks = [int(k) for k in sys.argv[3].split(",")]
blocks = [int(b) for b in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]


def run(env):
    for k in ("FLUID_WS", "FLUID_WS_BLOCKS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    sim = pkg.FluidSim(n, ny=ny, pressure_iters=J)
    rows = ny or n
    vel, dens, prs = random_state(max(n, rows), seed=5, vmax=30.0)
    sim.write(pkg.FIELD_VELOCITY, np.ascontiguousarray(vel[:rows, :n]))
    sim.write(pkg.FIELD_DENSITY, np.ascontiguousarray(dens[:rows, :n]))
    sim.write(pkg.FIELD_PRESSURE, np.ascontiguousarray(prs[:rows, :n]))
    sim.step(2)
    out = [sim.read(f).view(np.uint32).copy() for f in range(4)]
    steps = 16
    sim.step(steps)
    ms = min(sim.time_steps(steps) for _ in range(3)) / steps
    ks = {s["name"]: round(s["total_ms"] / s["launches"] * 1e3, 1) for s in sim.profile_steps(4)}
    sim.close()
    return out, ms, ks


base, ms0, ks0 = run({})
print(f"{n}x{ny or n} J={J} base: {ms0 * 1e3:8.1f} us/step {ks0}", flush=True)
for K in Ks:
    for b in blocks:
        env = {"FLUID_WS": str(K)}
        if b:
            env["FLUID_WS_BLOCKS"] = str(b)
        out, ms, ks = run(env)
        same = all(np.array_equal(a, c) for a, c in zip(base, out))
        bad = [int((a != c).sum()) for a, c in zip(base, out)]
        print(f"  ws K={K:2d} blocks={b:5d}: {ms * 1e3:8.1f} us/step  same_bits={same} {bad if not same else ''} {ks}", flush=True)
