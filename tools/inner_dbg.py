import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "notorch" in sys.argv:
    sys.modules["torch"] = None   # capi.py then loads the system HIP runtime, not torch's bundled one
import fluid_amd_loader
pkg = fluid_amd_loader.load()
n, G = int(sys.argv[1]), int(sys.argv[2])
sim = pkg.FluidSim(n, inner_slabs=G, pressure_iters=10)
sim.add_source(0.5, 0.5, 10.0, -5.0)
for k in (1, 2, 3):
    print("step", k, flush=True)
    sim.step(k)
    sim.sync()
print("ok", flush=True)
