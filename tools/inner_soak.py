#!/usr/bin/env python3
"""Long runs of an inner-slab context against the one-stream context, compared bit for bit at
checkpoints: a race between the slabs' streams (a halo row written while it is still read, or read
before it is written) would be rare and timing-dependent, so it gets many steps, call lengths that
vary, and splats at odd intervals.  Usage: python tools/inner_soak.py [n:J:G ...] [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
cases = [a for a in sys.argv[1:] if ":" in a] or ["1280:20:2", "2048:40:2", "2048:40:3", "3072:40:4", "4096:40:2", "1024:10:5"]
steps = next((int(a) for a in sys.argv[1:] if ":" not in a), 600)
F = (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE, pkg.FIELD_DIVERGENCE)
bad = 0
for case in cases:
    n, J, G = (int(v) for v in case.split(":"))
    a = pkg.FluidSim(n, pressure_iters=J)
    b = pkg.FluidSim(n, pressure_iters=J, inner_slabs=G)
    sa, sb = pkg.SplatStream(), pkg.SplatStream()
    rng = np.random.default_rng(n + J + G)
    done, checks = 0, 0
    while done < steps:
        if rng.random() < 0.35:
            a.add_source(*sa.splat())
            b.add_source(*sb.splat())
        k = int(rng.integers(1, 24))
        a.step(k)
        b.step(k)
        done += k
        if rng.random() < 0.15 or done >= steps:
            checks += 1
            for f in F:
                x, y = a.read(f), b.read(f)
                if not np.array_equal(x.view(np.uint32), y.view(np.uint32)):
                    bad += 1
                    print(f"MISMATCH {case} field {f} after {done} steps", flush=True)
                    done = steps
                    break
    print(f"{case}: {done} steps, {checks} checkpoints, mismatches so far {bad}", flush=True)
    a.close(); b.close()
sys.exit(1 if bad else 0)
