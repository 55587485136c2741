"""One rank of the CPU (gloo) slab test; started by tests/test_slab.py with RANK/WORLD_SIZE set.

Exercises the product's host-side multi-rank logic without a GPU:
  * fluid_slab_plan (libfluid_hip.so, pure host arithmetic): owned rows and halo depths;
  * HaloExchanger (fluid-simulation_amd/slab.py): the torch.distributed P2P protocol.
The compute stands in through the CPU oracle (checker only): every rank holds a whole-domain
oracle state, poisons with NaN every row the plan says it does not have, lets the exchanger fill
the halo rows, and steps.  If the plan's halo were one row short, or the exchange delivered the
wrong rows, NaNs or stale values would reach the owned rows, which must stay bit-identical to the
undecomposed oracle run.
"""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n, D, J, A, steps = (int(x) for x in sys.argv[1:6])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = fluid_amd_loader.load()
    from fluid_simulation_amd.slab import HaloExchanger
    hip = pkg.FluidLib()  # host-side planning only; no GPU call is made
    oracle = pkg.FluidLib(os.path.join(ROOT, "oracle", "liboracle_fluid.so"))
    plan = hip.slab_plan(n, rank, world, diffusion_iters=D, pressure_iters=J, halo_rows=A)
    row0, rows, hv, hd = plan["row0"], plan["rows"], plan["halo_vel"], plan["halo_dens"]
    row1 = row0 + rows
    assert plan["store_row0"] == row0 - (hv if rank > 0 else 0)
    nbytes = n * (hv * 8 + hd * 16)
    ex = HaloExchanger(rank, world, torch.device("cpu"))

    base = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, lib=oracle)   # undecomposed run
    mine = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, lib=oracle)   # my slab's view
    ss_a, ss_b = pkg.SplatStream(), pkg.SplatStream()
    for s in range(steps):
        if s % 2 == 0:
            base.add_source(*ss_a.splat())
            mine.add_source(*ss_b.splat())
        vel, dens = mine.read(pkg.FIELD_VELOCITY), mine.read(pkg.FIELD_DENSITY)
        # rows this rank does not own are unknown until the exchange delivers them
        own = np.zeros(n, bool)
        own[row0:row1] = True
        vel[~own] = np.nan
        dens[~own] = np.nan

        def pack(d, ptr):
            v = vel[row0:row0 + hv] if d < 0 else vel[row1 - hv:row1]
            q = dens[row0:row0 + hd] if d < 0 else dens[row1 - hd:row1]
            buf = np.concatenate([np.ascontiguousarray(v).view(np.uint8).ravel(),
                                  np.ascontiguousarray(q).view(np.uint8).ravel()])
            assert buf.size == nbytes
            ctypes.memmove(ptr, buf.ctypes.data, nbytes)

        def unpack(d, ptr):
            buf = np.empty(nbytes, np.uint8)
            ctypes.memmove(buf.ctypes.data, ptr, nbytes)
            v = buf[:n * hv * 8].view(np.float32).reshape(hv, n, 2)
            q = buf[n * hv * 8:].view(np.float32).reshape(hd, n, 4)
            if d < 0:
                vel[row0 - hv:row0] = v
                dens[row0 - hd:row0] = q
            else:
                vel[row1:row1 + hv] = v
                dens[row1:row1 + hd] = q

        ex.exchange(nbytes, nbytes, pack, unpack)
        mine.write(pkg.FIELD_VELOCITY, vel)
        mine.write(pkg.FIELD_DENSITY, dens)
        mine.step(1)
        base.step(1)
        for f in (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE):
            a, b = mine.read(f)[row0:row1], base.read(f)[row0:row1]
            assert np.isfinite(a).all(), f"rank {rank} step {s} field {f}: NaN reached the owned rows"
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
                f"rank {rank} step {s} field {f}: owned rows differ from the undecomposed run"
    # the halo really was needed: without it the owned edge rows go bad
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok rows [{row0},{row1}) halo_vel={hv} halo_dens={hd}")


if __name__ == "__main__":
    main()
