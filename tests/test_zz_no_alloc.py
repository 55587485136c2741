"""fluid_step_noalloc (include/fluid.h): the entry point for callers that cannot allocate.

Same kernels as fluid_step, launched directly, with the dump row of the temporally blocked kernels
in the library's file-scope device scratch (its device address taken with hipGetSymbolAddress at
fluid_init).  Checked against fluid_step itself, bit for bit (the bar for every stage of the
step), through the C-ABI, on two calls in a row — the second on a smaller grid, so leftovers of
the first call in the shared scratch would show — and under the caller's own stream capture, where
an allocation would fail.  Collected last (file name): `pytest -x` stops at the first failure."""
import ctypes as C

import numpy as np
import pytest

from util import assert_bit_equal

pytestmark = pytest.mark.gpu

FIELD_NAMES = ("velocity", "density", "pressure", "divergence")


def twin_run(pkg, hip, n, J, steps, flags=0):
    """The same splats and steps through fluid_step and through fluid_step_noalloc."""
    a = pkg.FluidSim(n, lib=hip, pressure_iters=J, flags=flags)
    b = pkg.FluidSim(n, lib=hip, pressure_iters=J, flags=flags)
    ss_a, ss_b = pkg.SplatStream(), pkg.SplatStream()
    for s in range(steps):
        a.add_source(*ss_a.splat())
        b.add_source(*ss_b.splat())
        a.step(1)
        b.step_noalloc(1)
    a.step(3)
    b.step_noalloc(3)
    for name, fid in zip(FIELD_NAMES, (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE,
                                       pkg.FIELD_DIVERGENCE)):
        x, y = a.read(fid), b.read(fid)
        assert float(np.abs(x).max()) > 0, name
        assert_bit_equal(y, x, f"{name} n={n}: fluid_step_noalloc vs fluid_step")
    a.close()
    b.close()


def test_noalloc_step_equals_step_twice_second_smaller(pkg, hip, monkeypatch):
    # 2048^2 and 1280^2 run the register pipeline (the kernels that use the dump row) by default;
    # 600^2 with FLUID_TILE=0 forces it on a grid whose strips leave the domain on the right
    twin_run(pkg, hip, 2048, 40, 2)
    twin_run(pkg, hip, 1280, 14, 2)
    monkeypatch.setenv("FLUID_TILE", "0")
    twin_run(pkg, hip, 600, 10, 2)
    monkeypatch.delenv("FLUID_TILE")
    twin_run(pkg, hip, 256, 10, 2)                      # LDS tiles: no scratch, still the same bits
    twin_run(pkg, hip, 1280, 10, 1, pkg.FLAG_UNFUSED)   # straight kernels


def test_noalloc_step_under_the_callers_stream_capture(pkg, hip):
    """What the entry point is for: the caller captures its own hipGraph around the step (any
    hipMalloc or graph instantiation inside the call would break the capture), replays it, and
    gets the bits of fluid_step."""
    import torch
    n, J, replays = 1536, 20, 3
    ref = pkg.FluidSim(n, lib=hip, pressure_iters=J)
    sim = pkg.FluidSim(n, lib=hip, pressure_iters=J)
    for s in (ref, sim):
        ss = pkg.SplatStream()
        for _ in range(3):
            s.add_source(*ss.splat())
    ref.step(2 * replays)
    stream = torch.cuda.Stream()
    sim.set_stream(stream.cuda_stream)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        graph.capture_begin()   # capture mode "global": an allocation anywhere in the process fails the capture
        sim.step_noalloc(2)     # two steps: every ping-pong index is back where the capture started
        graph.capture_end()
    for _ in range(replays):
        graph.replay()
    torch.cuda.synchronize()
    for name, fid in zip(FIELD_NAMES, (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE,
                                       pkg.FIELD_DIVERGENCE)):
        assert_bit_equal(sim.read(fid), ref.read(fid), f"{name}: replayed capture of fluid_step_noalloc")
    sim.set_stream(0)
    sim.close()
    ref.close()


def test_noalloc_step_rejects_what_it_cannot_serve(pkg, hip):
    slab = pkg.FluidSim(512, lib=hip, slab_rank=1, slab_count=4)
    with pytest.raises(pkg.FluidError) as ei:
        slab.step_noalloc(1)
    assert ei.value.code == -7
    slab.close()
    assert hip.dll.fluid_step_noalloc(None, 1) == -1
    big = pkg.FluidSim(16384 + 64, lib=hip, pressure_iters=1, diffusion_iters=1)   # beyond FLUID_NOALLOC_MAX_N
    with pytest.raises(pkg.FluidError) as ei:
        big.step_noalloc(1)
    assert ei.value.code == -4 and "static scratch" in str(ei.value)
    big.step(1)            # the allocating entry point serves it
    big.sync()
    big.close()
