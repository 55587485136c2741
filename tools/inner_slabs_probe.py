#!/usr/bin/env python3
"""Would cutting ONE GPU's domain into G row slabs that run concurrently (one stream each) fill the
ramp-up and drain of the mid-size launches?  Timing only: the G slab contexts of an N x N domain
are stepped side by side without the exchange (stale halos do not change the work) and compared
with the whole-domain context.
Usage: python tools/inner_slabs_probe.py N J G [steps] [FLUID_TB_x=v ...]   (plan overrides for the slabs only)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
n, J, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rest = [a for a in sys.argv[4:] if "=" not in a]
overrides = dict(a.split("=", 1) for a in sys.argv[4:] if "=" in a)
steps = int(rest[0]) if rest else 40


def timed(sims, label):
    ss = pkg.SplatStream()
    for _ in range(3):
        sp = ss.splat()
        for s in sims:
            s.add_source(*sp)
    for _ in range(3):
        for s in sims:
            s.step(1)
    for s in sims:
        s.sync()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            for s in sims:
                s.step(1)
        t_enq = time.perf_counter() - t0
        for s in sims:
            try:
                s.sync()
            except Exception as e:  # stale halos may trip the halo flag; timing is still valid
                print("sync:", e)
        dt = (time.perf_counter() - t0) / steps * 1e3
        best = dt if best is None else min(best, dt)
    print(json.dumps({"n": n, "J": J, "what": label, "ms_per_step": round(best, 4),
                      "host_enqueue_ms_per_step": round(t_enq / steps * 1e3, 4)}), flush=True)
    return best


whole = pkg.FluidSim(n, pressure_iters=J)
t_whole = timed([whole], "whole domain")
del whole
os.environ.update(overrides)
slabs = [pkg.FluidSim(n, pressure_iters=J, slab_rank=r, slab_count=G, flags=int(os.environ.get("FLUID_PROBE_FLAGS", "0")))
         for r in range(G)]
t_slabs = timed(slabs, f"{G} slabs side by side")
t_one = timed(slabs[G // 2:G // 2 + 1], "one middle slab alone")
print(json.dumps({"n": n, "J": J, "G": G, "plan": overrides, "whole": round(t_whole, 4), "slabs_concurrent": round(t_slabs, 4),
                  "one_slab_times_G": round(t_one * G, 4)}))
