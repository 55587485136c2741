import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fluid_amd_loader
from util import random_state, load_state, bits
pkg = fluid_amd_loader.load()
oracle = pkg.FluidLib(os.path.join(ROOT, "oracle", "liboracle_fluid.so"))
n = int(os.environ.get("N", 600))
for (kj, cj, waves, J) in [(10, 2, 64, 20), (10,2,64,30), (5, 2, 64, 10), (2, 2, 64, 4), (1, 2, 64, 2), (1,2,64,3), (8, 4, 64, 16), (10, 4, 64, 20), (4, 4, 64, 8), (4,2,64,8), (8,2,64,16), (8,2,64,24), (5,2,64,15), (5,4,64,15)]:
    os.environ.update(FLUID_TB_KJ=str(kj), FLUID_TB_CJ=str(cj), FLUID_TB_WAVES=str(waves))
    gpu = pkg.FluidSim(n, pressure_iters=J)
    ref = pkg.FluidSim(n, pressure_iters=J, lib=oracle)
    st = random_state(n, seed=21, vmax=30.0)
    load_state(gpu, pkg, *st); load_state(ref, pkg, *st)
    gpu.project(); ref.project()
    a, b = gpu.read(pkg.FIELD_PRESSURE), ref.read(pkg.FIELD_PRESSURE)
    bad = np.argwhere(bits(a) != bits(b))
    msg = "OK" if len(bad) == 0 else f"{len(bad)} bad; rows {bad[:,0].min()}..{bad[:,0].max()} cols {bad[:,1].min()}..{bad[:,1].max()} maxabs {np.abs(a-b).max():.3e} first {tuple(bad[0])} cols sample {sorted(set(bad[:,1]))[:12]}"
    print((kj, cj, waves, J), msg, flush=True)
