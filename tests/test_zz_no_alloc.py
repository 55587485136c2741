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
    # the plan whose divergence + pressure launch still writes the scratch row (end of round 3: the default plan's
    # launches drop what they used to sink, see test_noalloc_step_hands_the_kernels_the_static_scratch)
    monkeypatch.setenv("FLUID_TB_DIV", "0")
    twin_run(pkg, hip, 2048, 40, 2)
    twin_run(pkg, hip, 1280, 14, 2)
    monkeypatch.delenv("FLUID_TB_DIV")
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


def test_noalloc_step_hands_the_kernels_the_static_scratch(pkg, hip, monkeypatch):
    """The sink row is write-only, so equal results cannot tell WHICH row the kernels were handed.
    fluid_debug_sink can: canary both rows, step, and see which one was written.  fluid_step_noalloc
    must leave the context's own hipMalloc'ed row untouched and write the GPU's static scratch;
    fluid_step the other way round.  600^2 with the pipeline forced: strips leave the domain on the
    right, so sink columns are certainly written.

    End of round 3: the pipelines store their output rows through a buffer resource and drop the lanes and rows
    they used to sink (fluid_kernels_tb.hip, TB_BUFST); the sink is still what the divergence + pressure launch
    writes the divergence rows it must not store to, i.e. on plans where the last diffusion launch does not write
    the divergence itself (FLUID_TB_DIV=0 here; vorticity confinement, odd diffusion counts).  On the default plan
    neither entry point writes either row any more, which is checked too."""
    monkeypatch.setenv("FLUID_TILE", "0")
    for tb_div in ("0", None):
        if tb_div is None:
            monkeypatch.delenv("FLUID_TB_DIV", raising=False)
        else:
            monkeypatch.setenv("FLUID_TB_DIV", tb_div)
        for n, J in ((600, 10), (2048, 16)):
            sim = pkg.FluidSim(n, lib=hip, pressure_iters=J)
            sim.add_source(0.5, 0.5, 10.0, -5.0)
            for which in (0, 1):
                sim.debug_sink(which, fill=True)
                assert sim.debug_sink(which, fill=False) == 0
            sim.step_noalloc(2)
            assert sim.debug_sink(0, fill=False) == 0, "fluid_step_noalloc wrote the context's own sink row"
            if tb_div == "0":
                assert sim.debug_sink(1, fill=False) > 0, "fluid_step_noalloc did not write the static scratch"
            sim.debug_sink(1, fill=True)
            sim.step(2)
            assert sim.debug_sink(1, fill=False) == 0, "fluid_step wrote the static scratch"
            if tb_div == "0":
                assert sim.debug_sink(0, fill=False) > 0, "fluid_step did not write the context's own sink row"
            else:
                assert sim.debug_sink(0, fill=False) == 0, "the default plan's launches still sink rows"
            sim.close()


FRESH = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import fluid_amd_loader
pkg = fluid_amd_loader.load()
n, J = %(n)d, %(J)d
# the captured context is the FIRST thing this process launches: first use of the code objects (and
# the > 64 KB LDS attribute of the tiles) happens inside the caller's global-mode capture
sim = pkg.FluidSim(n, pressure_iters=J)
stream = torch.cuda.Stream()
sim.set_stream(stream.cuda_stream)
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(stream):
    graph.capture_begin()
    sim.step_noalloc(2)
    graph.capture_end()
for _ in range(2):
    graph.replay()
torch.cuda.synchronize()
got = [sim.read(f) for f in range(4)]
ref = pkg.FluidSim(n, pressure_iters=J)
ref.step(4)
for f in range(4):
    assert np.array_equal(got[f].view(np.uint32), ref.read(f).view(np.uint32)), f
# a state that is not all zeros, through the same replayed graph
for s in (sim, ref):
    s.add_source(0.4, 0.6, 20.0, -10.0)
graph.replay()
torch.cuda.synchronize()
ref.step(2)
for f in range(4):
    a, b = sim.read(f), ref.read(f)
    assert float(np.abs(b).max()) > 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32)), f
print("fresh capture ok")
"""


@pytest.mark.parametrize("n,J,tile", [(1536, 20, "0"), (512, 10, "1")])
def test_noalloc_capture_as_the_first_launch_of_a_fresh_process(n, J, tile):
    """in the test above `ref` has run every kernel before the capture starts.  Here a new
    process captures fluid_step_noalloc before anything else has launched, on the pipeline path and
    on the LDS-tile path."""
    import os
    import subprocess
    import sys
    from util import ROOT
    env = dict(os.environ, FLUID_TILE=tile)
    r = subprocess.run([sys.executable, "-c", FRESH % {"root": ROOT, "n": n, "J": J}], env=env,
                       capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and "fresh capture ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


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
