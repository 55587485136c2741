"""Multi-GPU leg of bench.py: one rank per GPU (torch.distributed.run), row slabs, RCCL halos."""
import json
import os
import time


def run_multi(pkg, args, n, D, J, flags):
    import torch
    import torch.distributed as dist
    from fluid_simulation_amd.slab import SlabSim
    from bench import HBM_PEAK_GBPS, seeded_warm_state

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # FLUID_BENCH_BACKEND=gloo + FLUID_BENCH_ONE_GPU=1: rehearsal of this code path with several
    # ranks on a single GPU (RCCL refuses two ranks per device); never the measured configuration.
    backend = os.environ.get("FLUID_BENCH_BACKEND", "nccl")
    if os.environ.get("FLUID_BENCH_ONE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    # A rank that waits for a peer that never comes (or a transfer that never ends) must not hang
    # the job: after FLUID_BENCH_RANK_TIMEOUT seconds every rank dumps its Python stacks and exits
    # non-zero (the launcher in bench.py / torch.distributed.run then ends the others).
    import datetime
    import faulthandler
    limit = float(os.environ.get("FLUID_BENCH_RANK_TIMEOUT", "1200"))
    faulthandler.dump_traceback_later(limit, exit=True)
    tmo = datetime.timedelta(seconds=min(limit, 600))
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local),
                                timeout=tmo)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    # Deep pressure loops on short slabs: the J sweeps can be cut into groups with a pressure-halo
    # exchange in between (include/fluid.h, pressure_groups).  Measured on one MI355X (tools/
    # slab_time.py, 8192^2, J = 200, one of 8 slabs, no exchange): 1.03 ms with one group (219 halo
    # rows), 0.95-1.01 ms with 2-5 groups — less than the extra exposed exchanges would cost, so
    # groups are used only when a slab is too short for the one-exchange halo.
    rows_min = n // world
    groups = 1
    while groups < J and (J + groups - 1) // groups + 1 + D + 8 > rows_min:
        groups += 1
    groups = int(os.environ.get("FLUID_BENCH_GROUPS", groups))
    gkw = {"pressure_groups": groups} if groups > 1 else {}

    # Self-check on the real transport before timing: the overlapped exchange must give the bits
    # of the plain one (3 steps from the same splats); otherwise time the plain path.
    import numpy as np
    overlap_ok = True
    if os.environ.get("FLUID_SLAB_OVERLAP", "1") != "0":
        sums = []
        for ov in (True, False):
            s_ = SlabSim(pkg, n, rank=rank, world=world, device_index=local, diffusion_iters=D,
                         pressure_iters=J, flags=flags, overlap=ov, **gkw)
            seeded_warm_state(pkg, s_)
            s_.step(3)
            s_.sync()
            sums.append([int(s_.read(f).view(np.uint32).sum(dtype=np.uint64))
                         for f in (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE)])
            s_.close()
        ok = torch.tensor([int(sums[0] == sums[1])], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        overlap_ok = bool(ok.item())
    sim = SlabSim(pkg, n, rank=rank, world=world, device_index=local, diffusion_iters=D,
                  pressure_iters=J, flags=flags, overlap=None if overlap_ok else False, **gkw)
    seeded_warm_state(pkg, sim)          # every rank applies the same splats to its own rows
    # The early interior advection (FLUID_PART_ADVECT_INTERIOR) buys the velocity message 0.1 ms more
    # cover for 12 us of launches per step; which way that nets out depends on the links of the node
    # at hand, so both ways are tried on the real transport, before the warm-up and the timed region.
    # Every rank takes the same decision (times are max-reduced over the ranks).
    tune = None
    if sim.overlap and sim.split_tail and sim.early_advect and os.environ.get("FLUID_SLAB_EARLY_ADVECT") is None:
        tune, trial = {}, (20 if backend == "nccl" else 2)
        for early in (True, False):
            sim.early_advect = early
            sim.step(3)
            sim.sync()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            sim.step(trial)
            sim.sync()
            torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tune["early" if early else "late"] = round(float(tt.item()) / trial * 1e3, 5)
        sim.early_advect = tune["early"] <= tune["late"]
    sim.step(args.warmup)
    sim.sync()
    torch.cuda.synchronize()
    dist.barrier()
    moved0 = sim.ex.bytes_moved
    t0 = time.perf_counter()
    sim.step(args.steps)
    sim.sync()
    torch.cuda.synchronize()
    dist.barrier()
    wall = time.perf_counter() - t0
    moved = sim.ex.bytes_moved - moved0   # sent in the timed region only (the tuning trials above moved more)
    t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())

    diag = sim.diagnostics()   # slab reductions + all-reduce, after the timed region
    if rank == 0:
        t_step = wall / args.steps
        bpc = pkg.alg_bytes_per_cell(D, J)
        plan = pkg.FluidLib().slab_plan(n, 0, world, diffusion_iters=D, pressure_iters=J, pressure_groups=groups)
        out = {
            "metric": "cell-updates/sec per full timestep (GCUPS)",
            "value": round(n * n / t_step / 1e9, 4), "unit": "GCUPS", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t_step * 1e3, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (seeded splats, SURVEY.md 8d)",
            "config": {"workload": f"{n}x{n} grid, D={D} diffusion + J={J} Jacobi sweeps per step",
                       "name": args.workload, "path": "unfused" if args.unfused else "fused",
                       "decomposition": f"{world} row slabs of {plan['rows']} rows, one halo exchange per "
                                        f"step ({plan['halo_vel']} velocity + {plan['halo_dens']} density rows "
                                        f"per neighbour"
                                        + (f", pressure sweeps in {plan['pressure_groups']} groups with {plan['halo_prs']} "
                                           f"pressure rows exchanged between groups" if plan['pressure_groups'] > 1 else "")
                                        + f") over {'RCCL send/recv' if backend == 'nccl' else backend}"},
            "step_alg_bytes_per_cell": bpc,
            "step_effective_GBps": round(bpc * n * n / t_step / 1e9, 1),
            "step_frac_of_hbm_peak": round(bpc * n * n / t_step / 1e9 / (HBM_PEAK_GBPS * world), 4),
            "transport": "RCCL (torch.distributed nccl)" if backend == "nccl" else backend,
            "halo_bytes_per_step_per_rank": moved // max(args.steps, 1),
            "overlap": sim.overlap, "overlap_matches_plain_exchange": overlap_ok,
            "early_interior_advection": bool(sim.overlap and sim.split_tail and sim.early_advect),
            "early_interior_advection_trial_ms_per_step": tune,
            "diagnostics_after_run": diag,
        }
        print(json.dumps(out), flush=True)
    sim.close()
    dist.barrier()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()
