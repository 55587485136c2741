#!/usr/bin/env python3
"""Per-step time of ONE slab driven by the library itself (fluid_comm_attach) with a
transport whose calls do nothing: what a rank's GPU and host spend per step apart from the
messages — replayed as hipGraphs (capturable transport) and enqueued directly.
Usage: tools/slab_lib_time.py [n J count ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

pkg = fluid_amd_loader.load()
args = [int(v) for v in sys.argv[1:]]
cases = list(zip(args[0::3], args[1::3], args[2::3])) or [(16384, 40, 2), (16384, 40, 4), (16384, 40, 8), (8192, 200, 8)]
cb = (pkg.capi.TRANSPORT_GROUP_START(lambda u: 0), pkg.capi.TRANSPORT_SEND(lambda u, p, b, r, st: 0),
      pkg.capi.TRANSPORT_RECV(lambda u, p, b, r, st: 0), pkg.capi.TRANSPORT_GROUP_END(lambda u, st: 0))
for n, J, count in cases:
    whole = pkg.FluidSim(n, pressure_iters=J)
    ss = pkg.SplatStream()
    for _ in range(4):
        whole.add_source(*ss.splat())
    whole.step(8)
    t1 = min(whole.time_steps(16) for _ in range(3)) / 16
    whole.close()
    for graph in (1, 0):
        os.environ["FLUID_COMM_GRAPH"] = str(graph)
        sim = pkg.FluidSim(n, pressure_iters=J, slab_rank=count // 2, slab_count=count)
        sim.comm_attach(pkg.FluidTransport(None, *cb, 1))
        ss = pkg.SplatStream()
        for _ in range(4):
            sim.add_source(*ss.splat())
        sim.step(32)
        sim.sync()
        best, host = 1e9, 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            sim.step(64)
            th = time.perf_counter() - t0        # host time to enqueue
            sim.sync()
            best = min(best, (time.perf_counter() - t0) / 64)
            host = min(host, th / 64)
        info = sim.comm_info()
        print(f"{n}^2 J={J} slab {count // 2} of {count}: {best * 1e3:7.3f} ms/step ({'graphs' if info['graph'] else 'direct'}), "
              f"host {host * 1e6:6.1f} us/step to enqueue; whole domain {t1:7.3f} ms -> x{t1 / (best * 1e3):.2f} before any message",
              flush=True)
        sim.comm_detach()
        sim.close()
