"""Slab decomposition (multi-GPU path).  CPU: plan arithmetic and the torch.distributed halo
protocol over gloo with world_size 2 and 3.  GPU: all slabs of a domain as separate contexts on
one device must reproduce the whole-domain context bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from util import ROOT, FIELDS, assert_bit_equal, random_state, read_all


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_slab_plan_arithmetic(pkg, hip):
    # halo_vel = J + 1 + D + A, halo_dens = A (include/fluid.h); slabs tile the domain exactly
    n, world, D, J, A = 1024, 8, 10, 40, 8
    plans = [hip.slab_plan(n, r, world, diffusion_iters=D, pressure_iters=J, halo_rows=A) for r in range(world)]
    assert plans[0]["row0"] == 0 and plans[-1]["row0"] + plans[-1]["rows"] == n
    for a, b in zip(plans, plans[1:]):
        assert a["row0"] + a["rows"] == b["row0"]
    for r, p in enumerate(plans):
        assert p["halo_vel"] == J + 1 + D + A and p["halo_dens"] == A
        assert p["has_lo"] == (r > 0) and p["has_hi"] == (r < world - 1)
        assert p["store_row0"] == p["row0"] - (p["halo_vel"] if r > 0 else 0)
        assert p["store_rows"] == p["rows"] + p["halo_vel"] * (int(r > 0) + int(r < world - 1))
    # uneven split
    rows = [hip.slab_plan(1000, r, 3)["rows"] for r in range(3)]
    assert sum(rows) == 1000 and max(rows) - min(rows) <= 1
    # default advection reach
    assert hip.slab_plan(4096, 1, 4)["halo_dens"] == 8
    # whole domain
    p = hip.slab_plan(512, 0, 1)
    assert (p["row0"], p["rows"], p["halo_vel"], p["store_rows"]) == (0, 512, 0, 512)
    # vorticity confinement between diffusion and divergence reads two rows further (include/fluid.h)
    pv = hip.slab_plan(n, 3, world, diffusion_iters=D, pressure_iters=J, halo_rows=A, flags=pkg.FLAG_VORTICITY)
    assert pv["halo_vel"] == J + 1 + D + A + 2 and pv["halo_dens"] == A
    assert pv["store_rows"] == pv["rows"] + 2 * pv["halo_vel"]
    # a slab shorter than its halo cannot be served by the adjacent slab alone
    with pytest.raises(pkg.FluidError):
        hip.slab_plan(256, 0, 8, pressure_iters=40)
    with pytest.raises(pkg.FluidError):
        hip.slab_plan(256, 3, 2)


@pytest.mark.parametrize("world,n,D,J,A", [(2, 96, 4, 6, 1), (3, 120, 3, 5, 2), (2, 64, 10, 10, 1)])
def test_halo_exchange_protocol_over_gloo(world, n, D, J, A):
    """world_size > 1 on CPU: see tests/slab_gloo_worker.py."""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "slab_gloo_worker.py"), str(n), str(D), str(J), str(A), "5"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert f"rank {r} ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 4, "pipeline"], ids=["tiles", "unfused", "pipeline"])
@pytest.mark.parametrize("count,n,D,J,A", [(2, 256, 10, 10, 1), (4, 512, 10, 40, 8), (3, 300, 3, 7, 2),
                                           (8, 1024, 10, 40, 8)])
def test_slabs_equal_whole_domain(pkg, hip, count, n, D, J, A, flags, monkeypatch):
    """K slab contexts + halo pack/unpack == one whole-domain context, bit for bit, over several
    steps with seeded splats (A=1 makes the halo exactly as deep as the dependency cone, so any
    off-by-one in a stage's row range shows).  Small slabs take the LDS-tile sweeps by default;
    "pipeline" forces the register pipeline the large grids use."""
    import fluid_simulation_amd.slab as slab
    if flags == "pipeline":
        monkeypatch.setenv("FLUID_TILE", "0")
        flags = 0
    whole = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, flags=flags)
    group = slab.LocalSlabGroup(pkg, n, count, diffusion_iters=D, pressure_iters=J, halo_rows=A, flags=flags)
    ss_a, ss_b = pkg.SplatStream(), pkg.SplatStream()
    for s in range(6):
        if s % 2 == 0:
            whole.add_source(*ss_a.splat())
            group.add_source(*ss_b.splat())
        whole.step(1)
        group.step(1)
    for f, fid in (("velocity", pkg.FIELD_VELOCITY), ("density", pkg.FIELD_DENSITY),
                   ("pressure", pkg.FIELD_PRESSURE), ("divergence", pkg.FIELD_DIVERGENCE)):
        assert_bit_equal(group.read(fid), whole.read(fid), f"{f} count={count}")
    group.close()


@pytest.mark.gpu
@pytest.mark.parametrize("count,n,D,J,A,G", [(2, 256, 10, 10, 1, 0), (4, 512, 10, 40, 8, 0), (3, 300, 3, 7, 2, 2),
                                             (8, 1024, 10, 40, 8, 0), (2, 64, 2, 3, 1, 0), (4, 128, 10, 10, 8, 0)])
def test_slabs_with_split_tail_equal_whole_domain(pkg, hip, count, n, D, J, A, G):
    """The step as the overlapped multi-GPU driver runs it — fused gradient subtraction + density
    advection on the edge rows, halo blocks packed, then the interior rows — gives the bits of the
    whole-domain run, also when the two edges cover the whole slab (interior empty)."""
    import fluid_simulation_amd.slab as slab
    whole = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J)
    group = slab.LocalSlabGroup(pkg, n, count, split_tail=True, diffusion_iters=D, pressure_iters=J,
                                halo_rows=A, pressure_groups=G)
    ss_a, ss_b = pkg.SplatStream(), pkg.SplatStream()
    for s in range(7):
        if s % 3 == 0:
            whole.add_source(*ss_a.splat())
            group.add_source(*ss_b.splat())
        whole.step(1)
        group.step(1)
    for f, fid in (("velocity", pkg.FIELD_VELOCITY), ("density", pkg.FIELD_DENSITY),
                   ("pressure", pkg.FIELD_PRESSURE), ("divergence", pkg.FIELD_DIVERGENCE)):
        assert_bit_equal(group.read(fid), whole.read(fid), f"{f} count={count}")
    # fluid_diagnostics per slab (stencils across the seams read refreshed halo rows), combined
    dg, dw = group.diagnostics(), whole.diagnostics()
    for k in pkg.FluidDiag.MAX_KEYS:
        assert np.float32(dg[k]).view(np.uint32) == np.float32(dw[k]).view(np.uint32), (k, dg[k], dw[k])
    for a, b in zip(dg["dye"], dw["dye"]):
        assert abs(a - b) <= 1e-6 * max(abs(b), 1.0)
    whole.step(1)
    group.step(1)   # the halo refresh of the diagnostics must not disturb the next step
    assert_bit_equal(group.read(pkg.FIELD_DENSITY), whole.read(pkg.FIELD_DENSITY), "density after diagnostics")
    with pytest.raises(pkg.FluidError):
        group.sims[0].step_part(pkg.PART_TAIL_DENSITY_INTERIOR)  # no edges launch pending
    group.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["tiles", "pipeline", "unfused"])
@pytest.mark.parametrize("count,n,D,J,A,G,split,early", [(2, 256, 10, 10, 1, 0, False, False), (3, 300, 3, 7, 2, 2, True, False),
                                                         (4, 512, 10, 40, 8, 0, True, True), (2, 128, 1, 1, 1, 0, False, False),
                                                         (3, 384, 10, 30, 1, 3, False, False)])
def test_slabs_with_vorticity_confinement_equal_whole_domain(pkg, hip, count, n, D, J, A, G, split, early, path, monkeypatch):
    """FLUID_FLAG_VORTICITY on slab contexts (kernel.cu:207-263, :557-580, the call at :705): the curl and the
    confinement force are recomputed in the halo like every other stage, two velocity rows deeper.  A = 1 makes the
    halo exactly as deep as the dependency cone.  Bit for bit against the whole-domain context with the same flag
    (itself held to the oracle by test_vorticity_stage_bit_exact), the vorticity field included."""
    import fluid_simulation_amd.slab as slab
    flags = pkg.FLAG_VORTICITY | (pkg.FLAG_UNFUSED if path == "unfused" else 0)
    if path != "unfused":
        monkeypatch.setenv("FLUID_TILE", "1" if path == "tiles" else "0")
    if path == "unfused":
        split = early = False
    whole = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, flags=flags)
    plain = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, flags=flags & ~pkg.FLAG_VORTICITY)
    group = slab.LocalSlabGroup(pkg, n, count, split_tail=split, early_advect=early, diffusion_iters=D, pressure_iters=J,
                                halo_rows=A, pressure_groups=G, flags=flags)
    assert group.sims[1].halo_bytes(-1) == n * ((-(-J // max(G, 1)) + 1 + D + A + 2) * 8 + A * 16)
    ss = [pkg.SplatStream() for _ in range(3)]
    for s in range(6):
        if s % 2 == 0:
            for sim, st in zip((whole, plain, group), ss):
                sim.add_source(*st.splat())
        for sim in (whole, plain, group):
            sim.step(1)
    for f, fid in (("velocity", pkg.FIELD_VELOCITY), ("density", pkg.FIELD_DENSITY), ("pressure", pkg.FIELD_PRESSURE),
                   ("divergence", pkg.FIELD_DIVERGENCE), ("vorticity", pkg.FIELD_VORTICITY)):
        assert_bit_equal(group.read(fid), whole.read(fid), f"{f} count={count} {path}")
    # the stage did something: the run with the flag differs from the run without it
    assert not np.array_equal(whole.read(pkg.FIELD_VELOCITY), plain.read(pkg.FIELD_VELOCITY))
    group.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pieces", [1, 3, 8])
def test_interior_of_the_split_tail_in_several_launches(pkg, hip, pieces, monkeypatch):
    """FLUID_PART_TAIL_DENSITY_INTERIOR goes out as several launches so that halo traffic queued on
    another stream gets in between them.  Any number of pieces, also more pieces
    than rows would fill: the bits of the whole-domain run."""
    import fluid_simulation_amd.slab as slab
    monkeypatch.setenv("FLUID_INTERIOR_SPLIT", str(pieces))
    monkeypatch.setenv("FLUID_INTERIOR_PIECE_MIN", "1")
    n, count = 384, 2
    whole = pkg.FluidSim(n, pressure_iters=12)
    group = slab.LocalSlabGroup(pkg, n, count, split_tail=True, pressure_iters=12, halo_rows=3)
    ss_a, ss_b = pkg.SplatStream(), pkg.SplatStream()
    for s in range(4):
        whole.add_source(*ss_a.splat())
        group.add_source(*ss_b.splat())
        whole.step(1)
        group.step(1)
    for f, fid in (("velocity", pkg.FIELD_VELOCITY), ("density", pkg.FIELD_DENSITY), ("pressure", pkg.FIELD_PRESSURE)):
        assert_bit_equal(group.read(fid), whole.read(fid), f"{f} pieces={pieces}")
    group.close()
    whole.close()


@pytest.mark.gpu
@pytest.mark.parametrize("split_tail", [True, False], ids=["split", "whole-step"])
@pytest.mark.parametrize("count,n,D,J,A,G", [(2, 256, 10, 10, 1, 0), (4, 512, 10, 40, 8, 0), (3, 300, 3, 7, 2, 2),
                                             (8, 1024, 10, 40, 8, 0), (4, 128, 10, 10, 8, 0), (2, 1136, 4, 27, 5, 2)])
def test_interior_advection_ahead_of_the_halos(pkg, hip, count, n, D, J, A, G, split_tail):
    """FLUID_PART_ADVECT_INTERIOR: the interior rows of a step's advection run before the step's
    velocity halos are unpacked (the halo rows still hold the previous step's leftovers), the
    rest after.  Same bits as the whole domain, also when the band is empty (slab shorter than
    two halos), with pressure groups, and when a splat or a write cancels it."""
    import fluid_simulation_amd.slab as slab
    whole = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J)
    group = slab.LocalSlabGroup(pkg, n, count, split_tail=split_tail, early_advect=True, diffusion_iters=D,
                                pressure_iters=J, halo_rows=A, pressure_groups=G)
    ss_a, ss_b = pkg.SplatStream(), pkg.SplatStream()
    for s in range(6):
        if s % 3 == 0:
            whole.add_source(*ss_a.splat())
            group.add_source(*ss_b.splat())
        if s == 4:   # ahead of a splat: cancelled, the step advects every row itself
            for sim in group.sims:
                sim.step_part(pkg.PART_ADVECT_INTERIOR)
            with pytest.raises(pkg.FluidError):
                group.sims[0].step_part(pkg.PART_ADVECT_INTERIOR)   # once per step
            whole.add_source(*ss_a.splat())
            group.add_source(*ss_b.splat())
        whole.step(1)
        group.step(1)
    for f, fid in (("velocity", pkg.FIELD_VELOCITY), ("density", pkg.FIELD_DENSITY),
                   ("pressure", pkg.FIELD_PRESSURE), ("divergence", pkg.FIELD_DIVERGENCE)):
        assert_bit_equal(group.read(fid), whole.read(fid), f"{f} count={count}")
    with pytest.raises(pkg.FluidError):
        whole.step_part(pkg.PART_ADVECT_INTERIOR)   # slab contexts only
    group.close()
    whole.close()


@pytest.mark.gpu
@pytest.mark.parametrize("early", [False, True], ids=["plain", "early-advect"])
def test_slabs_long_backtraces_and_halo_violation(pkg, hip, early):
    """Backtraces of several rows are fine while they stay within halo_rows; one that leaves it
    must be reported (FLUID_E_HALO), never silently wrong — also when the interior rows are
    advected ahead of the halos (FLUID_PART_ADVECT_INTERIOR)."""
    import fluid_simulation_amd.slab as slab
    n, count = 512, 4
    vel, dens, _ = random_state(n, seed=5, vmax=60.0)  # dt*v up to 6 rows
    whole = pkg.FluidSim(n)
    whole.write(pkg.FIELD_VELOCITY, vel)
    whole.write(pkg.FIELD_DENSITY, dens)
    group = slab.LocalSlabGroup(pkg, n, count, halo_rows=8, early_advect=early)
    group.write(pkg.FIELD_VELOCITY, vel)
    group.write(pkg.FIELD_DENSITY, dens)
    whole.step(1)
    group.step(1)
    assert_bit_equal(group.read(pkg.FIELD_VELOCITY), whole.read(pkg.FIELD_VELOCITY), "velocity")
    assert_bit_equal(group.read(pkg.FIELD_DENSITY), whole.read(pkg.FIELD_DENSITY), "density")
    group.close()
    # halo_rows = 2 is too shallow for 6-row backtraces
    group = slab.LocalSlabGroup(pkg, n, count, halo_rows=2, early_advect=early)
    group.write(pkg.FIELD_VELOCITY, vel)
    group.write(pkg.FIELD_DENSITY, dens)
    with pytest.raises(pkg.FluidError) as ei:
        group.step(1)
        group.read(pkg.FIELD_VELOCITY)
    assert ei.value.code == -6


@pytest.mark.gpu
def test_slab_contexts_reject_what_they_cannot_do(pkg, hip):
    s = pkg.FluidSim(512, slab_rank=1, slab_count=4)
    assert (s.row0, s.rows) == (128, 128)
    with pytest.raises(pkg.FluidError):
        s.step(2)          # halos must be refreshed before every step
    with pytest.raises(pkg.FluidError):
        s.diffuse()        # single stages are whole-domain only
    assert s.halo_bytes(-1) == 512 * ((10 + 1 + 10 + 8) * 8 + 8 * 16) == s.halo_bytes(+1)
    assert pkg.FluidSim(512, slab_rank=0, slab_count=4).halo_bytes(-1) == 0


def test_slab_plan_with_pressure_groups(pkg, hip):
    """J sweeps in G groups: halo_vel = Jg + 1 + D + A with Jg = ceil(J / G); halo_prs = Jg + 1
    pressure rows travel between groups; a group count that would leave the last group empty is
    reduced; one group (or a whole-domain context) exchanges no pressure."""
    p = hip.slab_plan(8192, 3, 8, diffusion_iters=10, pressure_iters=200, pressure_groups=4)
    assert (p["pressure_groups"], p["halo_prs"], p["halo_vel"]) == (4, 51, 50 + 1 + 10 + 8)
    p1 = hip.slab_plan(8192, 3, 8, diffusion_iters=10, pressure_iters=200)
    assert (p1["pressure_groups"], p1["halo_prs"], p1["halo_vel"]) == (1, 0, 219)
    p = hip.slab_plan(1024, 0, 2, diffusion_iters=3, pressure_iters=10, pressure_groups=4)  # Jg = 3: 3+3+3+1
    assert (p["pressure_groups"], p["halo_prs"]) == (4, 4)
    p = hip.slab_plan(1024, 0, 2, diffusion_iters=3, pressure_iters=9, pressure_groups=6)   # Jg = 2: 5 groups
    assert (p["pressure_groups"], p["halo_prs"]) == (5, 3)
    p = hip.slab_plan(1024, 0, 2, diffusion_iters=3, pressure_iters=5, pressure_groups=50)  # at most J groups
    assert (p["pressure_groups"], p["halo_prs"]) == (5, 2)
    assert hip.slab_plan(512, 0, 1, pressure_groups=4)["pressure_groups"] == 1
    # 8192^2, J = 200 on 8 GPUs only fits... both ways, but 16 slabs need the groups
    with pytest.raises(pkg.FluidError):
        hip.slab_plan(2048, 3, 16, pressure_iters=200)
    assert hip.slab_plan(2048, 3, 16, pressure_iters=200, pressure_groups=4)["halo_vel"] == 69


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 4], ids=["fused", "unfused"])
@pytest.mark.parametrize("count,n,D,J,A,G", [(2, 256, 10, 10, 1, 2), (4, 512, 10, 40, 8, 4), (3, 300, 3, 7, 2, 3),
                                             (4, 1024, 10, 200, 8, 5), (2, 128, 2, 9, 1, 9),
                                             # found by tools/fuzz_parity.py: the divergence tile of the
                                             # second slab starts at its local row 0 (read row -1)
                                             (2, 1136, 4, 27, 5, 2)])
def test_slabs_with_pressure_groups_equal_whole_domain(pkg, hip, count, n, D, J, A, G, flags):
    """The grouped pressure loop (head, pressure-halo exchange + group, ..., tail) gives the bits of
    the whole-domain run: uneven last groups, one sweep per group, temporal-blocking passes that
    do not divide the group length."""
    import fluid_simulation_amd.slab as slab
    whole = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, flags=flags)
    group = slab.LocalSlabGroup(pkg, n, count, diffusion_iters=D, pressure_iters=J, halo_rows=A, flags=flags,
                                pressure_groups=G)
    assert group.groups > 1
    ss_a, ss_b = pkg.SplatStream(), pkg.SplatStream()
    for s in range(4):
        if s % 2 == 0:
            whole.add_source(*ss_a.splat())
            group.add_source(*ss_b.splat())
        whole.step(1)
        group.step(1)
    for f, fid in (("velocity", pkg.FIELD_VELOCITY), ("density", pkg.FIELD_DENSITY),
                   ("pressure", pkg.FIELD_PRESSURE), ("divergence", pkg.FIELD_DIVERGENCE)):
        assert_bit_equal(group.read(fid), whole.read(fid), f"{f} count={count} groups={G}")
    s0 = group.sims[0]
    with pytest.raises(pkg.FluidError):
        s0.step(1)                         # the exchanges between groups are the caller's
    with pytest.raises(pkg.FluidError):
        s0.step_part(pkg.PART_PRESSURE_GROUP)  # no head yet in this step
    group.close()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [0, 1], ids=["plain", "overlap"])
def test_slabsim_multiprocess_with_pressure_groups(overlap):
    """SlabSim driving grouped pressure sweeps over a real process group (gloo, ranks share GPU 0)."""
    world, port = 3, free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "slab_gpu_worker.py"), "gloo", "384", "30", "4", str(overlap), "3"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [0, 1], ids=["plain", "overlap"])
@pytest.mark.parametrize("world", [2, 4])
def test_slabsim_multiprocess_on_one_gpu(world, overlap):
    """The bench's multi-rank driver (SlabSim: stream hand-over, pack -> P2P -> unpack -> step) with
    one process per slab.  Only one GPU is available to tests and RCCL refuses two ranks on one
    device, so the ranks share device 0 and the transport is gloo; the RCCL leg differs only in
    the backend string."""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "slab_gpu_worker.py"), "gloo", "512", "10", "6", str(overlap)],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,J,groups,flags", [(2, 512, 10, 0, 0), (4, 512, 10, 0, 0), (3, 384, 30, 3, 0),
                                                    (3, 384, 10, 0, 1)], ids=["2", "4", "3-groups", "3-vorticity"])
def test_library_driven_exchange_multiprocess_on_one_gpu(world, n, J, groups, flags):
    """fluid_step on slab contexts with a transport attached (fluid_comm_attach): the library posts
    the exchange itself, straight out of and into the field rows, between the edge and interior
    launches of the last kernel, and runs the interior advection ahead of the halos.  One process
    per slab sharing GPU 0; the bytes travel through HostStagedTransport (gloo) because RCCL
    refuses two ranks per device.  Gathered result == whole-domain context, bit for bit, through
    splats between steps (halos invalidated), several steps per call and diagnostics."""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "slab_gpu_worker.py"), "gloo", str(n), str(J), "6", "1",
             str(groups), "staged", str(flags)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"


@pytest.mark.gpu
@pytest.mark.parametrize("n,J,groups,flags", [(512, 10, 0, 0), (2048, 12, 0, 0), (384, 30, 3, 0), (512, 10, 0, 4)])
def test_library_step_as_graphs_equals_direct_enqueue(pkg, hip, n, J, groups, flags, monkeypatch):
    """The capture logic of the library-driven slab step — compute stream and comm stream forked and
    joined by events inside one hipGraph, batches of 16 + 4 + 1 steps, the state machine of the
    halos across add_source — with a capturable transport whose calls do nothing (a middle slab
    whose neighbours never answer: its halo rows keep what the cone recompute left, the same in
    both runs).  Graph replay must give the bits of direct enqueueing, report graph mode, and
    have built its graphs; a non-capturable transport must fall back to direct enqueueing."""
    def run(graph, capturable=1):
        monkeypatch.setenv("FLUID_COMM_GRAPH", str(graph))
        sim = pkg.FluidSim(n, lib=hip, slab_rank=1, slab_count=4, pressure_iters=J, halo_rows=2, pressure_groups=groups,
                           flags=flags)
        cbs = (pkg.capi.TRANSPORT_GROUP_START(lambda u: 0), pkg.capi.TRANSPORT_SEND(lambda u, p, b, r, st: 0),
               pkg.capi.TRANSPORT_RECV(lambda u, p, b, r, st: 0), pkg.capi.TRANSPORT_GROUP_END(lambda u, st: 0))
        tr = pkg.FluidTransport(None, *cbs, capturable)
        sim.comm_attach(tr)
        ss = pkg.SplatStream()
        sim.add_source(*ss.splat())
        sim.step(1)
        sim.step(5)
        sim.add_source(*ss.splat())     # halos invalid: the next call exchanges eagerly first
        sim.step(21)                    # 16 + 4 + 1
        sim.sync()
        out = [sim.read(f) for f in (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE)]
        info = sim.comm_info()
        sim.comm_detach()
        sim.close()
        return out, info, cbs
    a, ia, _ = run(1)
    b, ib, _ = run(0)
    c, ic, _ = run(1, capturable=0)
    assert ia["graph"] and ia["graphs_built"] >= 3 and ia["note"] == "", ia
    assert not ib["graph"] and ib["graphs_built"] == 0, ib
    assert not ic["graph"] and ic["graphs_built"] == 0, ic
    assert ia["exchanges"] == ib["exchanges"] == ic["exchanges"] and ia["bytes_sent"] == ib["bytes_sent"] > 0
    for x, y, z, f in zip(a, b, c, ("velocity", "density", "pressure")):
        assert float(np.abs(x).max()) > 0
        assert_bit_equal(x, y, f"{f}: graphs vs direct")
        assert_bit_equal(z, y, f"{f}: non-capturable vs direct")


@pytest.mark.gpu
def test_rccl_binding_on_one_gpu(pkg, hip):
    """The RCCL leg of the transport on the one GPU a test box has: a one-rank communicator sends
    to itself through the calls the slab step makes (ncclGroupStart / ncclSend / ncclRecv /
    ncclGroupEnd bound by dlopen), eagerly and as a captured hipGraph replayed twice."""
    msg = hip.comm_selftest_rccl(0, 3 << 20, True)
    print(msg)
    assert "2 graph replays" in msg
    assert len(hip.comm_unique_id()) == pkg.COMM_ID_BYTES
    whole = pkg.FluidSim(256, lib=hip)
    with pytest.raises(pkg.FluidError):
        whole.comm_init_rccl(hip.comm_unique_id())   # a whole-domain context has no halos
    whole.close()


@pytest.mark.gpu
def test_step_parts_equal_whole_step(pkg, hip):
    """fluid_step(1) == fluid_step_part(VELOCITY) + fluid_step_part(DENSITY), whole-domain and slab
    contexts alike (the slab variant also moves the two fields' halos separately)."""
    import fluid_simulation_amd.slab as slab
    n = 384
    a, b = pkg.FluidSim(n, pressure_iters=12), pkg.FluidSim(n, pressure_iters=12)
    for s in (a, b):
        pkg.run_seeded(s, 3)
    a.step(2)
    for _ in range(2):
        b.step_part(pkg.PART_VELOCITY)
        b.step_part(pkg.PART_DENSITY)
    for f in (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE, pkg.FIELD_DIVERGENCE):
        assert_bit_equal(a.read(f), b.read(f), f"field {f}")
    s = pkg.FluidSim(512, slab_rank=1, slab_count=4)
    assert s.halo_bytes_part(-1, pkg.PART_VELOCITY) + s.halo_bytes_part(-1, pkg.PART_DENSITY) == s.halo_bytes(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,J,count", [(16384, 40, 8), (8192, 200, 8), (16384, 40, 4)])
def test_baseline_multi_gpu_configs_at_full_size(pkg, hip, n, J, count):
    """BASELINE configs 4 and 5 at their real sizes and launch plans (the plan table of fluid_init
    picks different sweep counts and wave counts for a 2048-row slab than for the whole domain):
    the slabs, stepped the way the overlapped driver does, give the bits of the whole-domain
    context.  One GPU holds everything (whole domain + all slabs: 33 GB at 16384^2); fields are
    compared one at a time to bound host memory."""
    import fluid_simulation_amd.slab as slab
    whole = pkg.FluidSim(n, pressure_iters=J)
    group = slab.LocalSlabGroup(pkg, n, count, split_tail=True, pressure_iters=J)
    ss_a, ss_b = pkg.SplatStream(), pkg.SplatStream()
    for _ in range(3):
        sx, sy, fx, fy, rgba = ss_a.splat()
        whole.add_source(sx, sy, 30 * fx, 30 * fy, rgba)   # backtraces of several cells
        sx, sy, fx, fy, rgba = ss_b.splat()
        group.add_source(sx, sy, 30 * fx, 30 * fy, rgba)
    whole.step(2)
    group.step(2)
    for f, fid in (("velocity", pkg.FIELD_VELOCITY), ("pressure", pkg.FIELD_PRESSURE),
                   ("density", pkg.FIELD_DENSITY)):
        a = group.read(fid)
        b = whole.read(fid)
        assert a.shape == b.shape
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{f} differs at {n}^2 / {count} slabs"
        del a, b
    dg, dw = group.diagnostics(), whole.diagnostics()
    for k in pkg.FluidDiag.MAX_KEYS:
        assert np.float32(dg[k]).view(np.uint32) == np.float32(dw[k]).view(np.uint32), (k, dg[k], dw[k])
    assert dw["max_abs_u"] > 0 and sum(dw["dye"]) > 0
    group.close()
    whole.close()


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent (which never touches the GPU) starts
    the two ranks itself, relays rank 0's JSON line and exits 0.  Rehearsal transport: gloo, both
    ranks on GPU 0 (RCCL refuses two ranks per device); the measured configuration differs only in
    the backend and the device index."""
    import json
    env = dict(os.environ, FLUID_BENCH_BACKEND="gloo", FLUID_BENCH_ONE_GPU="1", GLOO_SOCKET_IFNAME="lo",
               FLUID_BENCH_TIMEOUT="300")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "1024_J40",
                          "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]   # stdout is the JSON line and nothing else (gloo's chatter goes to stderr)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["overlap_matches_plain_exchange"] is True
    assert rec["value"] > 0 and rec["unit"] == "GCUPS"
    # the library-driven exchange was verified against the whole-domain run and timed
    assert rec["exchange_mode"] == "staged" and rec["slab_matches_whole_domain"] is True and rec["attempt"] == 0
    assert rec["comm"]["attached"] and rec["halo_bytes_per_step_per_rank"] > 0
    # what makes a first multi-GPU line auditable
    assert rec["rccl_ranks"] is None                      # the rehearsal transport is not RCCL
    assert len(rec["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in rec["per_rank_ms_per_step"])
    assert 0 < rec["compute_only_ms_per_step"] and len(rec["compute_only_per_rank_ms_per_step"]) == 2
    assert abs(rec["exposed_comm_ms_per_step"] - (rec["ms_per_step"] - rec["compute_only_ms_per_step"])) < 1e-3
    assert rec["roofline"]["bound"] == "hbm" and rec["roofline"]["frac"] > 0 and rec["roofline"]["kernel"] in rec["slab_kernels_rank0"]
    assert rec["verified_after_steps"] == [3, 19]


@pytest.mark.gpu
def test_bench_moves_on_when_an_attempt_dies():
    """An exchange mode that kills a rank (or hangs: same path, through the watchdog) costs one
    attempt: every rank's supervisor starts the next attempt one mode further down the list, on
    the next rendezvous port, and the line still comes out.  Under torch.distributed.run-style
    environment variables (no launcher of bench.py's own)."""
    import json
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FLUID_BENCH_BACKEND="gloo", FLUID_BENCH_ONE_GPU="1", GLOO_SOCKET_IFNAME="lo",
                   FLUID_BENCH_TEST_DIE="0", FLUID_BENCH_ATTEMPT_TIMEOUT="40",
                   FLUID_BENCH_RENDEZVOUS_MARGIN="20")
        env.pop("FLUID_BENCH_ATTEMPT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload",
                                       "512_J10", "--steps", "2", "--warmup", "1"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][0][-2000:]
    rec = json.loads(lines[0])
    assert rec["attempt"] == 1 and rec["exchange_mode"] == "torch" and rec["n_gpus"] == 2
    assert "attempt 0" in outs[0][1]


def test_bench_launcher_reports_a_failed_rank(tmp_path):
    """The launcher half alone, on CPU: a rank that dies makes the parent end the others and exit
    non-zero; a job that outlives FLUID_BENCH_TIMEOUT is ended with 124."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "mode = sys.argv[1]\n"
                      "if mode == 'ok': print('{\"rank\": %d}' % r) if r == 0 else None\n"
                      "elif mode == 'die' and r == 1: sys.exit(3)\n"
                      "else: time.sleep(60)\n")
    real = bench.__file__
    try:
        bench.__file__ = str(script)          # launch_ranks starts children of "this file"
        assert bench.launch_ranks(3, ["ok"]) == 0
        assert bench.launch_ranks(3, ["die"]) == 1
        os.environ["FLUID_BENCH_TIMEOUT"] = "1"
        assert bench.launch_ranks(3, ["hang"]) == 124
    finally:
        bench.__file__ = real
        os.environ.pop("FLUID_BENCH_TIMEOUT", None)
