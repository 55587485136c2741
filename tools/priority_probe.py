#!/usr/bin/env python3
"""When does work on a second stream run while a step occupies the GPU?  A small and a large
device copy on a normal / high-priority side stream, submitted right after the step.
python tools/priority_probe.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fluid_amd_loader
pkg = fluid_amd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
sim = pkg.FluidSim(n, pressure_iters=40)
comp = torch.cuda.Stream()
sim.set_stream(comp.cuda_stream)
ss = pkg.SplatStream()
sim.add_source(*ss.splat())
sim.step(2); sim.sync()
for mb in (8, 1024):
    src = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); dst.copy_(src, non_blocking=True); b.record(); torch.cuda.synchronize()
    alone = a.elapsed_time(b)
    for name, prio in (("normal", 0), ("high", -1)):
        side = torch.cuda.Stream(priority=prio)
        for rep in range(2):
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            with torch.cuda.stream(comp):
                e0.record()
                sim.step(1)
                e2.record()
            with torch.cuda.stream(side):
                dst.copy_(src, non_blocking=True)
                e1.record()
            torch.cuda.synchronize()
            print(f"{mb} MB copy ({alone:.3f} ms alone), {name} priority: done {e0.elapsed_time(e1):.3f} ms after the step "
                  f"started; step took {e0.elapsed_time(e2):.3f} ms", flush=True)
sim.close()
