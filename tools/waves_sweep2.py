#!/usr/bin/env python3
"""Target-wave sweep (FLUID_TB_WAVES, pressure and diffusion launches alike unless FLUID_TB_WAVES_D is
set) for whole domains and middle slabs (library-driven step, no-op transport).
Usage: tools/waves_sweep2.py n:J:count[:kj] ...   (count 1 = whole domain)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
cb = (pkg.capi.TRANSPORT_GROUP_START(lambda u: 0), pkg.capi.TRANSPORT_SEND(lambda u, p, b, r, st: 0),
      pkg.capi.TRANSPORT_RECV(lambda u, p, b, r, st: 0), pkg.capi.TRANSPORT_GROUP_END(lambda u, st: 0))
WAVES = [int(v) for v in os.environ.get("SWEEP_WAVES", "0,2048,2600,2900,3072,3900,5800,6100,8700,16384").split(",")]
ROUNDS = os.environ.get("SWEEP_ROUNDS")   # e.g. "0,90,100,200": FLUID_TB_ROUNDS_PCT values instead of wave targets
if ROUNDS:
    WAVES = [int(v) for v in ROUNDS.split(",")]
for arg in sys.argv[1:]:
    parts = [int(v) for v in arg.split(":")]
    n, J, count = parts[:3]
    kj = parts[3] if len(parts) > 3 else 0
    out = []
    for w in WAVES:
        for k in ("FLUID_TB_WAVES", "FLUID_TB_WAVES_D", "FLUID_TB_KJ", "FLUID_TB_ROUNDS_PCT"):
            os.environ.pop(k, None)
        if w and ROUNDS:
            os.environ["FLUID_TB_ROUNDS_PCT"] = str(w)
        elif w:
            os.environ["FLUID_TB_WAVES"] = os.environ["FLUID_TB_WAVES_D"] = str(w)
        if kj:
            os.environ["FLUID_TB_KJ"] = str(kj)
        if count > 1:
            sim = pkg.FluidSim(n, pressure_iters=J, slab_rank=count // 2, slab_count=count)
            sim.comm_attach(pkg.FluidTransport(None, *cb, 1))
        else:
            sim = pkg.FluidSim(n, pressure_iters=J)
        ss = pkg.SplatStream()
        for _ in range(4):
            sim.add_source(*ss.splat())
        steps = 32 if n * n // count <= 1 << 26 else 16
        sim.step(steps)
        sim.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            sim.step(steps)
            sim.sync()
            best = min(best, (time.perf_counter() - t0) / steps)
        out.append((w, best))
        if count > 1:
            sim.comm_detach()
        sim.close()
    base = out[0][1]
    print(f"{n}^2 J={J} " + (f"slab of {count}" if count > 1 else "whole") + (f" kj={kj}" if kj else "") + ": " +
          "  ".join(f"{w or 'table'}:{t * 1e3:.3f}({(t / base - 1) * 100:+.1f}%)" for w, t in out), flush=True)
