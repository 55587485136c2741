#!/usr/bin/env python3
"""Host-side cost of driving one slab the way bench.py --gpus N does (SlabSim, overlap mode), with
the transport replaced by no-ops (halos go stale; the kernels, packs, unpacks, events and stream
switches are the real ones): how long the host needs to enqueue a step against how long the GPU
needs to run it.  python tools/slab_host_overhead.py [N] [J] [G]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import fluid_amd_loader
pkg = fluid_amd_loader.load()
from fluid_simulation_amd.slab import SlabSim

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
J = int(sys.argv[2]) if len(sys.argv) > 2 else 40
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8


class Done:
    def wait(self):
        return True


dist.get_backend = lambda group=None: "nccl"
dist.P2POp = lambda op, tensor, peer, group=None: (op, tensor, peer)
dist.batch_isend_irecv = lambda ops: [Done()]
sim = SlabSim(pkg, n, rank=G // 2, world=G, pressure_iters=J)
ss = pkg.SplatStream()
sim.add_source(*ss.splat())
sim.step(10)
try:
    sim.sync()
except Exception as e:
    print("sync:", e)
for steps in (50, 200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step(steps)
    t1 = time.perf_counter()
    try:
        sim.sync()
    except Exception as e:
        print("sync:", e)
    t2 = time.perf_counter()
    print(f"{n}^2 J={J} slab of {G}: host enqueue {1e3 * (t1 - t0) / steps:.3f} ms/step, "
          f"until the GPU is done {1e3 * (t2 - t0) / steps:.3f} ms/step", flush=True)
