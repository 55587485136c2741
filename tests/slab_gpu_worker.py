"""One rank of the multi-process GPU slab test (all ranks share device 0; backend from argv).
Runs the product's SlabSim exactly as bench.py --gpus N does and checks the gathered result
against a whole-domain context bit for bit on rank 0."""
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
    backend, n, J, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    overlap = bool(int(sys.argv[5]))
    groups = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    transport = sys.argv[7] if len(sys.argv) > 7 else "torch"   # "staged": the exchange posted by the library itself
    flags = int(sys.argv[8]) if len(sys.argv) > 8 else 0        # fluid_params.flags (1: vorticity confinement)
    dist.init_process_group(backend, rank=rank, world_size=world)
    pkg = fluid_amd_loader.load()
    from fluid_simulation_amd.slab import SlabSim
    sim = SlabSim(pkg, n, rank=rank, world=world, device_index=0, pressure_iters=J, halo_rows=2,
                  overlap=overlap, pressure_groups=groups, transport=transport, flags=flags)
    ss = pkg.SplatStream()
    for s in range(steps):
        if s % 2 == 0:
            sim.add_source(*ss.splat())   # also exercises discarding halos already in flight
        sim.step(1)
    sim.step(3)                            # several steps in one call: exchanges chained back to back
    sim.sync()
    diag = sim.diagnostics()               # slab reductions + all-reduce; every rank gets the whole domain's
    sim.step(1)                            # and the halos it consumed are sent again
    fields = {f: sim.gather(f) for f in (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE)}
    if rank == 0:
        whole = pkg.FluidSim(n, pressure_iters=J, flags=flags)
        ss = pkg.SplatStream()
        for s in range(steps):
            if s % 2 == 0:
                whole.add_source(*ss.splat())
            whole.step(1)
        whole.step(3)
        want = whole.diagnostics()
        for k in pkg.FluidDiag.MAX_KEYS:
            assert np.float32(diag[k]).view(np.uint32) == np.float32(want[k]).view(np.uint32), (k, diag[k], want[k])
        for a, b in zip(diag["dye"], want["dye"]):
            assert abs(a - b) <= 1e-6 * max(abs(b), 1.0), (diag["dye"], want["dye"])
        whole.step(1)
        for f, got in fields.items():
            want = whole.read(f)
            assert got.shape == want.shape
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"field {f} differs"
    if transport != "torch":
        info = sim.sim.comm_info()
        assert info["attached"] and info["exchanges"] >= steps + 4 and info["bytes_sent"] > 0, info
        assert not info["graph"] and info["transport"] == "caller", info   # host-staged: not capturable
    sim.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
