"""the reference's view mappings and input conversion, headless."""
import json

import numpy as np
import pytest

from util import assert_bit_equal

F32 = np.float32


def test_view_mappings_follow_the_reference_kernels(pkg, tmp_path):
    rng = np.random.default_rng(0)
    vel = rng.uniform(-3, 3, (8, 8, 2)).astype(F32)
    prs = rng.uniform(-2, 2, (8, 8)).astype(F32)
    dens = rng.uniform(0, 1, (8, 8, 4)).astype(F32)
    v = pkg.viz.velocity_image(vel)                                   # kernel.cu:411-414
    assert np.array_equal(v[..., 0], (F32(1) + vel[..., 0]) / F32(2)) and np.all(v[..., 2:] == 0)
    p = pkg.viz.pressure_image(prs)                                   # kernel.cu:427-429
    assert all(np.array_equal(p[..., c], (F32(1) + prs) / F32(2)) for c in range(4))
    assert_bit_equal(pkg.viz.density_image(dens), dens)               # kernel.cu:397-398
    pkg.viz.write_ppm(str(tmp_path / "a.ppm"), dens)
    pkg.viz.write_pfm(str(tmp_path / "a.pfm"), dens)
    head = open(tmp_path / "a.ppm", "rb").read(15)
    assert head.startswith(b"P6\n8 8\n255\n")
    raw = np.frombuffer(open(tmp_path / "a.pfm", "rb").read()[len(b"PF\n8 8\n-1.0\n"):], "<f4").reshape(8, 8, 3)
    assert np.array_equal(raw, dens[..., :3])


def test_replay_converts_events_like_mousemove_and_addforce(pkg, oracle):
    events = pkg.parse_events(json.dumps([
        {"frame": 0, "type": "down", "x": 100, "y": 200},
        {"frame": 0, "type": "move", "x": 110, "y": 195},      # force (10, +5): y flipped
        {"frame": 2, "type": "key", "key": "3"},               # green
        {"frame": 2, "type": "move", "x": 104, "y": 205},      # delta from the LAST event: (-6, -10)
        {"frame": 4, "type": "key", "key": "r"},
        {"frame": 5, "type": "move", "x": 105, "y": 205},
    ]))
    n = 64
    a = pkg.FluidSim(n, lib=oracle)
    pkg.replay(a, events, frames=7, width=512, height=512)
    b = pkg.FluidSim(n, lib=oracle)                              # the same by hand
    b.add_source(0.5, 0.5, 10.0, 5.0, (1, 1, 1, 1)); b.step(2)   # splat always at the window centre
    b.add_source(0.5, 0.5, -6.0, -10.0, (0, 1, 0, 0)); b.step(2)
    b.reset(); b.step(1)
    b.add_source(0.5, 0.5, 1.0, -0.0, (0, 1, 0, 0)); b.step(2)
    for f in (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE):
        assert_bit_equal(a.read(f), b.read(f), f"field {f}")
    # non-square window: integer halves, float division (kernel.cu:875, :689-690)
    c = pkg.FluidSim(n, lib=oracle)
    pkg.replay(c, [{"frame": 0, "type": "move", "x": 3, "y": 0}], frames=1, width=500, height=300)
    d = pkg.FluidSim(n, lib=oracle)
    d.add_source(float(F32(150) / F32(500)), float(F32(250) / F32(300)), 3.0, -0.0, (1, 1, 1, 1)); d.step(1)
    assert_bit_equal(c.read(pkg.FIELD_VELOCITY), d.read(pkg.FIELD_VELOCITY), "non-square window")


@pytest.mark.gpu
def test_view_of_a_gpu_run(pkg, hip, tmp_path):
    sim = pkg.FluidSim(128, lib=hip)
    pkg.replay(sim, [{"frame": 0, "type": "down", "x": 0, "y": 0}, {"frame": 1, "type": "move", "x": 30, "y": -20}], frames=5)
    for m in (0, 1, 2):
        img = pkg.viz.image_of(sim, pkg, m)
        assert img.shape == (128, 128, 4) and np.isfinite(img).all()
    assert pkg.viz.image_of(sim, pkg, 0).max() > 0
    pkg.viz.write_ppm(str(tmp_path / "d.ppm"), pkg.viz.image_of(sim, pkg, 0))
