#!/usr/bin/env python3
"""Would the density advection hide behind the next step's velocity chain?  Two independent
contexts on two streams: A runs velocity parts, B runs density parts, alone and together."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
J = 40
a = pkg.FluidSim(n, pressure_iters=J)
b = pkg.FluidSim(n, pressure_iters=J)
ss = pkg.SplatStream()
for _ in range(3):
    sp = ss.splat()
    a.add_source(*sp)
    b.add_source(*sp)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a.set_stream(s1.cuda_stream)
b.set_stream(s2.cuda_stream)
V, D = pkg.PART_VELOCITY, pkg.PART_DENSITY


def run(fa, fb, reps=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if fa:
            a.step_part(V)
        if fb:
            b.step_part(D)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(2):
    run(True, True, 2)
print("velocity part alone   %.3f ms" % run(True, False))
print("density part alone    %.3f ms" % run(False, True))
print("both, two streams     %.3f ms" % run(True, True))
