#!/usr/bin/env python3
"""How fast does a halo-sized transfer kernel run on a high-priority side stream while the split
tail's interior launch (HBM-bound) or the pressure sweeps occupy the GPU?  One slab of N^2 / G.
Stand-ins for the message: the runtime's blit copy (few, large workgroups) and an elementwise
kernel (many 256-thread workgroups).  (An RCCL send/recv to self on a 1-rank group hangs: not used.)
python tools/message_probe.py [N] [G]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fluid_amd_loader
pkg = fluid_amd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sim = pkg.FluidSim(n, pressure_iters=40, slab_rank=G // 2, slab_count=G)
comp = torch.cuda.Stream()
side = torch.cuda.Stream(priority=-1 if os.environ.get("PROBE_HIGH_PRIORITY", "0") == "1" else 0)
sim.set_stream(comp.cuda_stream)
ss = pkg.SplatStream()
sim.add_source(*ss.splat())
nbytes = sim.halo_bytes(-1)
print("halo message", nbytes / 1e6, "MB; rows", sim.rows, flush=True)
src = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
dst = torch.empty_like(src)
ev = lambda: torch.cuda.Event(enable_timing=True)

def message(kind):
    if kind == "pack":   # the library's own pack of the velocity block for the lower neighbour
        sim.halo_pack_part(-1, pkg.PART_VELOCITY, dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
    elif kind == "blit":
        dst.copy_(src, non_blocking=True)
    else:
        torch.add(src, 1.0, out=dst)

def one(kind, where):
    torch.cuda.synchronize()
    t0, t1, s0, s1, mid = ev(), ev(), ev(), ev(), ev()
    with torch.cuda.stream(comp):
        t0.record()
        sim.step_part(pkg.PART_VELOCITY_HEAD)
        if where == "head":
            mid.record()
    if where == "head":
        with torch.cuda.stream(side):
            s0.record(); message(kind); s1.record()
    with torch.cuda.stream(comp):
        sim.step_part(pkg.PART_TAIL_DENSITY_EDGES)
        if where == "interior":
            mid.record()
    if where == "interior":
        with torch.cuda.stream(side):
            side.wait_event(mid)
            s0.record(); message(kind); s1.record()
    with torch.cuda.stream(comp):
        sim.step_part(pkg.PART_TAIL_DENSITY_INTERIOR)
        t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1), s0.elapsed_time(s1), t0.elapsed_time(s0), t0.elapsed_time(s1)

for kind in ("pack", "blit", "elementwise"):
    torch.cuda.synchronize()
    a, b = ev(), ev()
    message(kind); message(kind); torch.cuda.synchronize(); a.record(); message(kind); b.record(); torch.cuda.synchronize()
    print(f"{kind}: alone {a.elapsed_time(b)*1e3:.0f} us", flush=True)
    for where in ("interior", "head"):
        one(kind, where)
        for rep in range(3):
            step, msg, beg, end = one(kind, where)
            print(f"  {kind} during {where}: step {step:.3f} ms, message {msg*1e3:.0f} us (from {beg:.3f} to {end:.3f} ms)", flush=True)
try:
    sim.sync()
except Exception as e:
    print("sync:", e)
st = sim.profile_steps(1)
print({s["name"]: round(s["total_ms"] / max(s["launches"], 1), 4) for s in st})
