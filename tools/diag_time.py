import sys, time
sys.path.insert(0, "/root/repo")
import fluid_amd_loader
pkg = fluid_amd_loader.load()
for n in (16384, 4096):
    sim = pkg.FluidSim(n, pressure_iters=40)
    ss = pkg.SplatStream()
    for _ in range(3):
        sim.add_source(*ss.splat())
    sim.step(2); sim.sync()
    sim.diagnostics()
    t0 = time.perf_counter()
    for _ in range(10):
        d = sim.diagnostics()
    print(n, "diagnostics ms", (time.perf_counter() - t0) / 10 * 1e3, d["max_abs_u"])
    sim.close()
