"""Multi-GPU leg of bench.py: one rank per GPU, row slabs, halo rows over RCCL.

Process layout (bench.py): launcher or torch.distributed.run -> one RANK process per GPU, which is
only a supervisor (supervise_rank: touches no GPU) -> one CHILD per attempt, which runs run_multi.
Nothing multi-rank can be rehearsed on the one-GPU boxes this was built on, so the child verifies
before it times, and a hang costs one attempt, not the job:

  * every rank first runs the WHOLE domain on its own GPU for three steps (it fits: 16 GB at
    16384^2) and keeps checksums of its own rows: the truth;
  * the exchange modes are tried in order — "rccl-graph" (halos posted by the library itself from
    C, ncclSend/ncclRecv out of and into the field rows, whole steps replayed as hipGraphs),
    "rccl" (the same, enqueued directly), "torch" (round 1's Python-driven exchange through
    torch.distributed P2P) — and the first whose slabs reproduce the truth bit for bit on every rank
    is the one that is timed;
  * a mode that hangs or kills a rank ends the attempt (watchdog / supervisor), and the next
    attempt starts one mode further down the list on a fresh rendezvous port.
"""
import json
import os
import sys
import time


def default_modes():
    rehearsal = os.environ.get("FLUID_BENCH_BACKEND", "nccl") != "nccl"
    # last resort on a node whose RCCL does not come up at all: the library-driven exchange with the
    # bytes staged through the host (slow, and labelled as such in the line)
    return "staged,torch" if rehearsal else "rccl-graph,rccl,torch,staged"


def rendezvous_wait(attempt, limit):
    """Seconds the ranks of attempt `attempt` wait for each other: more than the previous attempt can
    last on a rank that hung in it (its own rendezvous + the watchdog + process start-up, where the
    first `import torch` on a fresh box takes a minute or two)."""
    margin = float(os.environ.get("FLUID_BENCH_RENDEZVOUS_MARGIN", "150"))
    return (attempt + 1) * (limit + margin)


def supervise_rank(argv):
    """One rank, no GPU: run the benchmark in a child per attempt; on a failed or hung attempt
    start the next with the first mode dropped.  Every rank does the same, so the ranks of an
    attempt find each other on MASTER_PORT + 1 + attempt."""
    import signal
    import subprocess
    import threading
    rank = os.environ.get("RANK", "0")
    modes = os.environ.get("FLUID_BENCH_MODES", default_modes()).split(",")
    base_port = int(os.environ.get("MASTER_PORT", "29500"))
    limit = float(os.environ.get("FLUID_BENCH_ATTEMPT_TIMEOUT", "240"))   # watchdog of one attempt, once its ranks have met
    child = [None]

    def on_term(signum, frame):   # the launcher ends this rank: take the child along
        if child[0] is not None and child[0].poll() is None:
            child[0].kill()
        sys.exit(128 + signum)
    signal.signal(signal.SIGTERM, on_term)
    signal.signal(signal.SIGINT, on_term)
    for attempt in range(len(modes)):
        env = dict(os.environ, FLUID_BENCH_ATTEMPT=str(attempt), FLUID_BENCH_MODES=",".join(modes[attempt:]),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(base_port + 1 + attempt),
                   FLUID_BENCH_RANK_TIMEOUT=str(limit))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)   # rank 0's child hosts the store of its attempt
        p = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench.py")]
                             + list(argv), env=env, stdout=subprocess.PIPE,
                             text=True)   # same session: whoever ends this rank's session ends the child too
        child[0] = p
        # The job's stdout is ONE JSON line.  Libraries write there too (gloo announces its
        # connections on stdout), so the child's stdout is read here: the JSON line is held until
        # the attempt has ended well, everything else goes to stderr.
        result = []

        def relay(pipe=p.stdout, result=result):
            for line in pipe:
                if line.lstrip().startswith("{"):
                    result.append(line)
                else:
                    sys.stderr.write(line)
                    sys.stderr.flush()
        reader = threading.Thread(target=relay, daemon=True)
        reader.start()
        try:
            rc = p.wait(timeout=rendezvous_wait(attempt, limit) + limit + 60)
        except subprocess.TimeoutExpired:
            p.kill()   # exactly the child started above
            p.wait()
            rc = 124
        reader.join(timeout=10)
        if rc == 0:
            for line in result:
                sys.stdout.write(line)
            sys.stdout.flush()
            return 0
        print(f"bench.py rank {rank}: attempt {attempt} (modes {','.join(modes[attempt:])}) ended with {rc}",
              file=sys.stderr, flush=True)
    return 1


def _checksums(pkg, sim, row0=0, rows=0):
    """fluid_checksum of the three state fields: computed on the device, 8 bytes each cross the bus
    (round 2 read the rows through the host: 7.4 GB per rank at 16384^2)."""
    return [sim.checksum(f, row0, rows) for f in (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE)]


VERIFY_STEPS = 3          # 2 + 1: small batches, and the eager first exchange
VERIFY_GRAPH_STEPS = 16   # then one full batch: in "rccl-graph" mode a replayed 16-step hipGraph


def _null_transport(pkg, capturable):
    """A transport whose calls do nothing: what a rank spends per step apart from the messages."""
    cb = (pkg.capi.TRANSPORT_GROUP_START(lambda u: 0), pkg.capi.TRANSPORT_SEND(lambda u, p, b, r, st: 0),
          pkg.capi.TRANSPORT_RECV(lambda u, p, b, r, st: 0), pkg.capi.TRANSPORT_GROUP_END(lambda u, st: 0))
    return pkg.FluidTransport(None, *cb, int(capturable)), cb


def run_multi(pkg, args, n, D, J, flags):
    import datetime
    import faulthandler
    import numpy as np
    import torch
    import torch.distributed as dist
    from fluid_simulation_amd.slab import SlabSim
    from bench import HBM_PEAK_GBPS, seeded_warm_state

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or with no launcher at all")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node by contract; the host name may not resolve
    # FLUID_BENCH_BACKEND=gloo + FLUID_BENCH_ONE_GPU=1: rehearsal of this code path with several
    # ranks on a single GPU (RCCL refuses two ranks per device); never the measured configuration.
    backend = os.environ.get("FLUID_BENCH_BACKEND", "nccl")
    if os.environ.get("FLUID_BENCH_ONE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    # A rank that waits for a peer that never comes (or a transfer that never ends) must not hang
    # the job: after FLUID_BENCH_RANK_TIMEOUT seconds every rank dumps its Python stacks and exits
    # non-zero (its supervisor then starts the next attempt).
    limit = float(os.environ.get("FLUID_BENCH_RANK_TIMEOUT", "240"))
    attempt = int(os.environ.get("FLUID_BENCH_ATTEMPT", "0"))
    # control plane (barriers, the reductions of flags and times, the RCCL id): gloo, CPU tensors.
    # A rank whose previous attempt died early waits here for the ranks whose attempt ran into its
    # watchdog: the rendezvous of attempt a waits rendezvous_wait(a), longer than attempt a - 1 can last.
    dist.init_process_group("gloo", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=rendezvous_wait(attempt, limit)))
    faulthandler.dump_traceback_later(limit, exit=True)   # armed once the ranks have met
    if os.environ.get("FLUID_BENCH_TEST_DIE") == str(attempt) and rank == world - 1:
        os._exit(3)   # tests: the last rank of this attempt dies; the others must notice and move on
    p2p_group = None   # the "torch" mode's data plane, made when that mode is reached

    def all_min(flag):
        t = torch.tensor([int(flag)])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def all_max(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

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

    # The truth: the whole domain on this rank's own GPU, VERIFY_STEPS steps from the seeded splats.
    whole = pkg.FluidSim(n, device=local, diffusion_iters=D, pressure_iters=J, flags=flags)
    seeded_warm_state(pkg, whole)
    plan = pkg.FluidLib().slab_plan(n, rank, world, diffusion_iters=D, pressure_iters=J, pressure_groups=groups)
    whole.step(VERIFY_STEPS)
    truth = _checksums(pkg, whole, plan["row0"], plan["rows"])
    whole.step(VERIFY_GRAPH_STEPS)
    truth_batch = _checksums(pkg, whole, plan["row0"], plan["rows"])
    whole.close()

    modes = os.environ.get("FLUID_BENCH_MODES", default_modes()).split(",") if world > 1 else ["torch"]
    tried, sim, mode = [], None, None
    for m in modes:
        note, ok, s_ = "", False, None
        try:
            if m in ("rccl-graph", "rccl", "staged"):
                os.environ["FLUID_COMM_GRAPH"] = "1" if m == "rccl-graph" else "0"   # read when the transport is attached
                s_ = SlabSim(pkg, n, rank=rank, world=world, device_index=local, diffusion_iters=D, pressure_iters=J,
                             flags=flags, transport="staged" if m == "staged" else "rccl", **gkw)
            elif m in ("torch", "torch-plain"):
                if p2p_group is None and backend == "nccl":
                    p2p_group = dist.new_group(backend="nccl", device_id=torch.device("cuda", local))
                s_ = SlabSim(pkg, n, rank=rank, world=world, device_index=local, diffusion_iters=D, pressure_iters=J,
                             flags=flags, transport="torch", group=p2p_group, overlap=(m == "torch"), **gkw)
            else:
                raise ValueError(f"unknown mode {m!r}")
            seeded_warm_state(pkg, s_)
            s_.step(VERIFY_STEPS)
            s_.sync()
            ok = _checksums(pkg, s_.sim) == truth
            if not ok:
                note = "slab rows differ from the whole-domain run"
            if ok:   # ... and after one more full batch of steps (a replayed hipGraph in "rccl-graph" mode)
                s_.step(VERIFY_GRAPH_STEPS)
                s_.sync()
                ok = _checksums(pkg, s_.sim) == truth_batch
                if not ok:
                    note = f"slab rows differ from the whole-domain run after a batch of {VERIFY_GRAPH_STEPS} more steps"
            if ok and m == "rccl-graph" and not s_.sim.comm_info()["graph"]:
                ok, note = False, "graph capture refused: " + s_.sim.comm_info()["note"]
            if ok and m in ("rccl-graph", "rccl") and s_.sim.comm_info()["ranks"] != world:
                ok, note = False, f"RCCL reports {s_.sim.comm_info()['ranks']} ranks, not {world}"
        except Exception as e:   # noqa: BLE001 - any failure means: next mode
            note = f"{type(e).__name__}: {e}"
        ok = all_min(ok)
        tried.append({"mode": m, "ok": ok, "note": note})
        if ok:
            sim, mode = s_, m
            break
        if s_ is not None:
            try:
                s_.close()
            except Exception:   # noqa: BLE001
                pass
    if sim is None:
        if rank == 0:
            print(f"bench.py: no exchange mode reproduced the whole-domain run: {tried}", file=sys.stderr, flush=True)
        sys.exit(2)

    # The early interior advection (FLUID_PART_ADVECT_INTERIOR) buys the velocity message 0.1 ms more
    # cover for 12 us of launches per step; in the Python-driven mode both ways are tried on the real
    # transport before the warm-up (times max-reduced: every rank takes the same decision).
    tune = None
    if mode == "torch" and sim.split_tail and sim.early_advect and os.environ.get("FLUID_SLAB_EARLY_ADVECT") is None:
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
            tune["early" if early else "late"] = round(all_max(time.perf_counter() - t0) / trial * 1e3, 5)
        sim.early_advect = tune["early"] <= tune["late"]
    sim.step(args.warmup)
    sim.sync()
    torch.cuda.synchronize()
    dist.barrier()
    moved0 = sim.bytes_moved
    t0 = time.perf_counter()
    sim.step(args.steps)
    sim.sync()
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0   # this rank's own GPU is done (the barrier below adds the others' lag)
    dist.barrier()
    wall = all_max(time.perf_counter() - t0)
    moved = sim.bytes_moved - moved0   # sent in the timed region only
    per_rank = [None] * world
    dist.all_gather_object(per_rank, round(mine / args.steps * 1e3, 5))

    diag = sim.diagnostics()   # slab reductions + all-reduce, after the timed region
    comm = sim.sim.comm_info() if sim.in_library else None

    # Where the time went, on the same slab in the same run: the steps again with a transport whose
    # calls do nothing (same launches, same graphs or direct enqueueing, no message), so
    # exposed communication = timed - compute-only; and one step kernel by kernel for the roofline of
    # the slab's dominant launch.  The fields are garbage afterwards (stale halos): nothing reads them.
    compute_only, per_rank_compute, roofline, kernels = None, None, None, None
    if sim.in_library:
        try:
            sim.sim.comm_detach()
            tr, keep = _null_transport(pkg, capturable=(mode == "rccl-graph"))
            sim.sim.comm_attach(tr)
            k = max(2, min(args.steps, 20))
            sim.sim.step(max(args.warmup, 2))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sim.sim.step(k)
            torch.cuda.synchronize()
            mine_c = (time.perf_counter() - t1) / k * 1e3
            sim.sim.comm_detach()
            stats = [st for st in sim.sim.profile_steps(1) if st["launches"]]
            kernels = {st["name"]: {"launches": st["launches"], "ms_per_launch": round(st["total_ms"] / st["launches"], 5)}
                       for st in stats}
            dom = max(stats, key=lambda st: st["total_ms"])
            avg = dom["total_ms"] / dom["launches"]
            roofline = {"bound": "hbm", "kernel": dom["name"], "rank": rank, "launches_per_step": dom["launches"],
                        "avg_launch_ms": round(avg, 5),
                        "achieved": round(dom["alg_bytes"] / (avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(dom["alg_bytes"] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                        "moved_bytes_per_launch": dom["moved_bytes"],
                        "frac_moved": round(dom["moved_bytes"] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                        "traffic": None, "timing": "HIP events on the slab context's stream, one step, direct launches"}
        except pkg.FluidError as e:   # e.g. FLUID_E_HALO from the stale halos: the line still comes out
            mine_c = None
            roofline = {"error": str(e)}
        per_rank_compute = [None] * world
        dist.all_gather_object(per_rank_compute, None if mine_c is None else round(mine_c, 5))
        if all(v is not None for v in per_rank_compute):
            compute_only = max(per_rank_compute)
    if rank == 0:
        t_step = wall / args.steps
        bpc = pkg.alg_bytes_per_cell(D, J)
        carrier = {"rccl-graph": "RCCL ncclSend/ncclRecv posted by the library, whole steps replayed as hipGraphs",
                   "rccl": "RCCL ncclSend/ncclRecv posted by the library, steps enqueued directly",
                   "staged": "host-staged transport (gloo through host memory; rehearsal / last resort), exchange posted by the library",
                   "torch": "torch.distributed P2P (" + ("RCCL" if backend == "nccl" else backend) + "), step driven in parts from Python",
                   "torch-plain": "torch.distributed P2P, exchange before the step, no overlap"}[mode]
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
                                        + f"); {carrier}"},
            "step_alg_bytes_per_cell": bpc,
            "step_effective_GBps": round(bpc * n * n / t_step / 1e9, 1),
            "step_frac_of_hbm_peak": round(bpc * n * n / t_step / 1e9 / (HBM_PEAK_GBPS * world), 4),
            "transport": carrier, "exchange_mode": mode, "modes_tried": tried,
            "attempt": attempt,
            "halo_bytes_per_step_per_rank": moved // max(args.steps, 1),
            # what RCCL itself says (ncclCommCount through fluid_comm_info); None: not an RCCL mode
            "rccl_ranks": comm["ranks"] if comm and mode in ("rccl-graph", "rccl") else None,
            "per_rank_ms_per_step": per_rank,
            "compute_only_ms_per_step": None if compute_only is None else round(compute_only, 5),
            "compute_only_per_rank_ms_per_step": per_rank_compute,
            "exposed_comm_ms_per_step": None if compute_only is None else round(t_step * 1e3 - compute_only, 5),
            "roofline": roofline, "slab_kernels_rank0": kernels,
            "verified_after_steps": [VERIFY_STEPS, VERIFY_STEPS + VERIFY_GRAPH_STEPS],
            # the timed configuration reproduced the whole-domain run (every rank's own GPU, 3 steps)
            # bit for bit on every rank's rows before it was timed
            "slab_matches_whole_domain": True, "overlap_matches_plain_exchange": True,
            "overlap": mode != "torch-plain", "comm": comm,
            "early_interior_advection": bool(sim.early_advect) if mode == "torch" else (mode != "torch-plain"),
            "early_interior_advection_trial_ms_per_step": tune,
            "diagnostics_after_run": diag,
        }
        print(json.dumps(out), flush=True)
    try:
        sim.close()
        dist.barrier()
        dist.destroy_process_group()
    except Exception:   # noqa: BLE001 - the line is out; a teardown hiccup must not fail the attempt
        pass
    faulthandler.cancel_dump_traceback_later()
    sys.stdout.flush()
    os._exit(0)
