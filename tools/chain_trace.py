"""Where the waves of a temporally blocked launch spend their time: per-wave stamps (device-wide 100 MHz counter:
start, pipeline filled, tail begins, end) of the last such launch of a step, by strip and by chunk-row
(FLUID_TB_TRACE=1; FLUID_TB_CHAIN=0 traces the plain kernels, the default here is chained; TRACE_NSTR = strips per
chunk-row rounded up to 4, 44 at 4096^2).  usage: chain_trace.py N [J]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FLUID_TB_CHAIN", "1")
os.environ["FLUID_TB_TRACE"] = "1"
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()

n = int(sys.argv[1])
J = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sim = pkg.FluidSim(n, pressure_iters=J, flags=pkg.FLAG_NO_GRAPH)
ss = pkg.SplatStream()
for _ in range(4):
    sim.add_source(*ss.splat())
sim.step(3)
sim.sync()
t = sim.chain_trace().astype(np.int64)
idx = np.flatnonzero(t[:, 0] > 0)
idx = idx[t[idx, 0] > t[idx, 0].max() - int(os.environ.get("TRACE_WINDOW", "5000"))]   # the last launch (records of earlier ones linger)
u = t[idx] - t[idx, 0].min()
nstr = int(os.environ.get("TRACE_NSTR", "44"))
by, strip = idx // nstr, idx % nstr
dur = u[:, 3] - u[:, 0]
print(f"last launch: {len(u)} waves, chunk-rows {by.min()}..{by.max()}, strips {strip.min()}..{strip.max()}; spans {u[:, 3].max()} clocks")


def row(name, v):
    print(f"  {name:14s} min/med/p90/max {int(v.min()):7d} {int(np.median(v)):7d} {int(np.percentile(v, 90)):7d} {int(v.max()):7d}")


row("start", u[:, 0])
row("fill", u[:, 1] - u[:, 0])
row("after fill", u[:, 3] - u[:, 1])
row("duration", dur)
row("end", u[:, 3])
print("by strip (median duration, median start):")
for s_ in range(strip.min(), strip.max() + 1):
    m = strip == s_
    if m.any():
        print(f"  strip {s_:3d}: dur {int(np.median(dur[m])):7d}  max {int(dur[m].max()):7d}  start {int(np.median(u[m, 0])):6d}  end max {int(u[m, 3].max()):7d}")
print("by chunk-row (every 4th):")
for b in range(by.min(), by.max() + 1, 4):
    m = by == b
    if m.any():
        print(f"  row {b:3d}: dur med {int(np.median(dur[m])):7d} max {int(dur[m].max()):7d}  start med {int(np.median(u[m, 0])):6d}  end max {int(u[m, 3].max()):7d}")
print("histogram of start times (1 us bins; stamps are 10 ns ticks):")
h, edges = np.histogram(u[:, 0], bins=np.arange(0, u[:, 0].max() + 100, 100))
for c_, e_ in zip(h, edges):
    if c_:
        print(f"  {int(e_):7d}: {c_}")
print("duration against start (median by start bin):")
for e_ in edges[:-1]:
    m = (u[:, 0] >= e_) & (u[:, 0] < e_ + 100)
    if m.sum() >= 8:
        print(f"  {int(e_):7d}: dur {int(np.median(dur[m])):7d}  fill {int(np.median((u[:, 1] - u[:, 0])[m])):7d}")
