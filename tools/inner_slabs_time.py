#!/usr/bin/env python3
"""ms per step of a whole-domain context against the same domain as G inner slabs
(fluid_params.inner_slabs: G slab contexts side by side on one GPU, halo copies included).
Usage: python tools/inner_slabs_time.py N J G[,G..] [steps] [FLUID_x=v ... (for the inner contexts only)]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("FLUID_NO_TORCH"):
    sys.modules["torch"] = None   # the system HIP runtime (7.2) instead of the one torch bundles (7.0)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
n, J = int(sys.argv[1]), int(sys.argv[2])
Gs = [int(g) for g in sys.argv[3].split(",")]
rest = [a for a in sys.argv[4:] if "=" not in a]
overrides = dict(a.split("=", 1) for a in sys.argv[4:] if "=" in a)
steps = int(rest[0]) if rest else 60


def timed(sim):
    ss = pkg.SplatStream()
    for _ in range(3):
        sim.add_source(*ss.splat())
    sim.step(5)
    sim.sync()
    best, enq = None, None
    for _ in range(4):
        t0 = time.perf_counter()
        sim.step(steps)
        t1 = time.perf_counter()
        sim.sync()
        dt = (time.perf_counter() - t0) / steps * 1e3
        if best is None or dt < best:
            best, enq = dt, (t1 - t0) / steps * 1e3
    return round(best, 4), round(enq, 4)


out = {"n": n, "J": J, "plan": overrides}
sim = pkg.FluidSim(n, pressure_iters=J)
out["plain"], out["plain_host_enqueue"] = timed(sim)
del sim
os.environ.update(overrides)
for G in Gs:
    sim = pkg.FluidSim(n, pressure_iters=J, inner_slabs=G)
    out[f"inner{G}"], out[f"inner{G}_host_enqueue"] = timed(sim)
    del sim
print(json.dumps(out), flush=True)
