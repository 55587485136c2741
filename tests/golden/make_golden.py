"""Generate the committed golden fixtures from the CPU oracle.

    python tests/golden/make_golden.py

The reference has no fixtures of its own and cannot be run, so these vectors are
outputs of oracle/fluid_ref.c — "parity unpinned" with respect to the reference; they exist to
detect oracle drift between sessions and to give the GPU path fixed targets that do not depend
on the oracle being rebuilt.  Each .npz holds the inputs' definition (params, seed/splat) and
the expected fields.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402

CASES = [
    # name, n, D, J, steps, mode
    ("seeded_n64_s1", 64, 10, 10, 1, "seeded"),
    ("seeded_n64_s10", 64, 10, 10, 10, "seeded"),
    ("seeded_n128_s25", 128, 10, 10, 25, "seeded"),
    ("seeded_n96_j40_s12", 96, 10, 40, 12, "seeded"),
    ("centre_n64_s5", 64, 10, 10, 5, "centre"),   # the reference-faithful case: (0.5,0.5), (10,-5), white
    ("seeded_n50_d3_j7_s11", 50, 3, 7, 11, "seeded"),  # odd iteration counts, n not a multiple of 32
    ("seeded_96x40_s12", (96, 40), 10, 10, 12, "seeded"),  # non-square: 96 columns x 40 rows
]


def run_case(pkg, lib, n, D, J, steps, mode, **kw):
    sim = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, lib=lib, **kw)
    if mode == "seeded":
        pkg.run_seeded(sim, steps)
    else:
        sim.add_source(0.5, 0.5, 10.0, -5.0, (1.0, 1.0, 1.0, 1.0))
        sim.step(steps)
    out = dict(velocity=sim.read(pkg.FIELD_VELOCITY), density=sim.read(pkg.FIELD_DENSITY),
               pressure=sim.read(pkg.FIELD_PRESSURE), divergence=sim.read(pkg.FIELD_DIVERGENCE))
    sim.close()
    return out


def main():
    pkg = fluid_amd_loader.load()
    lib = pkg.FluidLib(os.path.join(ROOT, "oracle", "liboracle_fluid.so"))
    for name, n, D, J, steps, mode in CASES:
        path = os.path.join(HERE, name + ".npz")
        if os.path.exists(path) and "--all" not in sys.argv:
            continue   # committed fixtures stay as they are; --all regenerates every one
        n, ny = n if isinstance(n, tuple) else (n, 0)
        out = run_case(pkg, lib, n, D, J, steps, mode, ny=ny)
        np.savez_compressed(path, n=n, ny=ny, D=D, J=J, steps=steps, mode=mode, **out)
        print(name, {k: float(np.abs(v).max()) for k, v in out.items()})


if __name__ == "__main__":
    main()
