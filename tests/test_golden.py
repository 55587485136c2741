"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle).

CPU: the oracle must still reproduce them bit for bit (oracle drift detection).
GPU: the HIP path must match them within the north_star tolerance 1e-5 (field-normalised);
the only arithmetic that may differ from the oracle is expf inside add_source.
"""
import glob
import os

import numpy as np
import pytest

from util import GOLDEN, FIELDS, assert_bit_equal, rel_err, read_all

CASES = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))
TOL = 1e-5  # north_star: max relative error <= 1e-5 per field


def _run(pkg, lib, g, **kw):
    ny = int(g["ny"]) if "ny" in g.files else 0   # rows; 0 = square
    sim = pkg.FluidSim(int(g["n"]), diffusion_iters=int(g["D"]), pressure_iters=int(g["J"]),
                       lib=lib, ny=ny, **kw)
    if str(g["mode"]) == "seeded":
        pkg.run_seeded(sim, int(g["steps"]))
    else:
        sim.add_source(0.5, 0.5, 10.0, -5.0, (1.0, 1.0, 1.0, 1.0))
        sim.step(int(g["steps"]))
    return read_all(sim, pkg)


def test_fixtures_exist():
    assert len(CASES) >= 7


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_reproduces_golden(pkg, oracle, path):
    g = np.load(path)
    out = _run(pkg, oracle, g)
    for f in FIELDS:
        assert_bit_equal(out[f], g[f], f)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 4, 2], ids=["fused-graph", "unfused-graph", "fused-nograph"])
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_hip_matches_golden(pkg, hip, path, flags):
    g = np.load(path)
    out = _run(pkg, hip, g, flags=flags)
    for f in FIELDS:
        e = rel_err(out[f], g[f])
        assert e <= TOL, f"{f}: rel err {e:.3e} > {TOL}"
