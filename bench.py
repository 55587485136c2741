#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X stable-fluids solver.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one full simulationStep() (kernel.cu:701-708) over the whole grid.  Metric:
cell-updates/s per full timestep (GCUPS = n^2 / t_step / 1e9) and the HBM-roofline fraction
(BASELINE.json).  Prints ONE JSON line on rank 0.

For N > 1 there is one rank per GPU; the grid is cut into N row slabs (strong scaling: the total
grid is fixed) and halo rows travel over RCCL.  Two ways in:
  - under a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*): this
    process is one rank;
  - plain `python bench.py --gpus N`: this process touches no GPU, starts N fresh children of
    itself with those variables set (launch_ranks), relays rank 0's JSON line, and exits non-zero
    if a rank fails or the wall-clock limit (FLUID_BENCH_TIMEOUT, seconds) passes.
Either way a rank process only supervises (bench_multi.supervise_rank): the benchmark itself runs
in a child per attempt, so that an exchange mode that hangs on the node at hand costs one attempt
and the next, more conservative mode still produces the line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (n, diffusion_iters, pressure_iters)      BASELINE.json configs
    "512_J10": (512, 10, 10),        # configs[0] reference default
    "1024_J40": (1024, 10, 40),      # configs[1]
    "4096_J40": (4096, 10, 40),      # configs[2] HBM-roofline characterisation
    "16384_J40": (16384, 10, 40),    # configs[3] 1/2/4/8 GPU
    "8192_J200": (8192, 10, 200),    # configs[4]
}
DEFAULT_WORKLOAD = "16384_J40"


def seeded_warm_state(pkg, sim, nsplats=4):
    """Splats are applied before the timed window; the timed region holds no
    add_source / read."""
    ss = pkg.SplatStream()
    for _ in range(nsplats):
        sim.add_source(*ss.splat())


def measure_single(pkg, n, D, J, steps, warmup, flags=0, inner_slabs=0):
    import torch
    sim = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, flags=flags, inner_slabs=inner_slabs)
    seeded_warm_state(pkg, sim)
    sim.step(warmup)
    sim.prepare_steps(steps)   # builds the hipGraphs the timed call replays; runs nothing
    sim.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step(steps)
    sim.sync()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return sim, wall


def cpu_threads():
    """Threads for the multi-core baseline: OMP_NUM_THREADS when set (16 on the GPU box: its CPU
    share; os.cpu_count() shows the whole machine there), else the visible CPUs, at most 16."""
    try:
        return max(1, int(os.environ.get("OMP_NUM_THREADS", "")))
    except ValueError:
        return max(1, min(16, os.cpu_count() or 1))


def cpu_baseline(pkg, D, J, budget_s=12.0, threads=1, n=1024):
    """The host-C oracle (oracle/fluid_ref.c) on a bounded sample of the workload: an n x n grid
    (1024^2 unless told otherwise) with the workload's D and J, as many steps as fit the budget.
    threads=1: the serial build (the contract's `cpu_baseline`); threads>1: the same source built
    with -fopenmp (bit-identical, tests/test_oracle_kat.py), reported as `cpu_baseline_parallel`."""
    path = os.path.join(ROOT, "oracle", "liboracle_fluid.so" if threads == 1 else "liboracle_fluid_omp.so")
    if threads > 1:
        os.environ["OMP_NUM_THREADS"] = str(threads)  # read by libgomp when the library is loaded
    lib = pkg.FluidLib(path)
    sim = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, lib=lib)
    seeded_warm_state(pkg, sim)
    if budget_s > 0:
        sim.step(1)  # warm caches / leave the all-zero state (the one-step full-size sample runs cold)
    steps, t0 = 0, time.perf_counter()
    while True:
        sim.step(1)
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 400:
            break
    sim.close()   # 16 GB of host memory at 16384^2
    return {
        "value": round(n * n * steps / el / 1e9, 6), "unit": "GCUPS", "cores": threads, "kind": "port",
        "sample": f"{n}x{n} grid, D={D}, J={J}, {steps} steps, seeded splats, "
                  + ("serial C oracle" if threads == 1 else f"C oracle with OpenMP row loops, {threads} threads")
                  + f" (gcc -O2 -ffp-contract=off), {el:.1f} s",
    }


def kernel_sources_sha():
    """sha256 over the kernel sources (csrc/fluid_kernels*.hip, fluid_internal.h) in name order: what
    profiles/pmc_traffic.json's counters were collected on is compared with what runs now."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "fluid-simulation_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "fluid_kernels*.hip")) + [os.path.join(csrc, "fluid_internal.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc(workload):
    """profiles/pmc_traffic.json: HBM bytes per launch from rocprofv3 PMC passes (2*FETCH_SIZE +
    WRITE_SIZE, KiB -> bytes: MI355X_MICROARCH.md's gfx950 correction), written by
    tools/summarize_profile.py from a run of THIS file on the GPU box.  Counters cannot be
    collected inside an ordinary bench run, so the table is replayed here, stamped with the
    commit, date and workload it was collected on; None when it is for another workload."""
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        pmc = json.load(open(pmc_path))
    except Exception:
        return None
    return pmc if pmc.get("_workload") == workload else None


def roofline_from_profile(sim, steps, workload, t_step):
    stats = sim.profile_steps(steps)
    pmc = load_pmc(workload)
    dom = max(stats, key=lambda s: s["total_ms"])
    avg_ms = dom["total_ms"] / dom["launches"]
    achieved = dom["alg_bytes"] / (avg_ms * 1e-3) / 1e9
    moved_rate = dom["moved_bytes"] / (avg_ms * 1e-3) / 1e9
    per_kernel, step_moved, step_traffic, step_kernel_ms = {}, 0.0, 0.0, 0.0
    for s in stats:
        a_ms = s["total_ms"] / s["launches"]
        tr = pmc.get(s["name"], {}).get("hbm_bytes_per_launch") if pmc else None
        per_kernel[s["name"]] = {
            "launches_per_step": s["launches"] / steps, "avg_ms": round(a_ms, 5),
            "alg_GBps": round(s["alg_bytes"] / (a_ms * 1e-3) / 1e9, 1),
            "moved_GBps": round(s["moved_bytes"] / (a_ms * 1e-3) / 1e9, 1),
            "traffic_GBps": round(tr / (a_ms * 1e-3) / 1e9, 1) if tr else None}
        step_moved += s["moved_bytes"] * s["launches"] / steps
        step_kernel_ms += s["total_ms"] / steps
        if tr is not None and step_traffic is not None:
            step_traffic += tr * s["launches"] / steps
        else:
            step_traffic = None
    traffic = pmc.get(dom["name"], {}).get("hbm_bytes_per_launch") if pmc else None
    # `achieved`/`frac` price every reference launch's algorithmic bytes (the contract, SURVEY.md
    # 8d), so a launch that fuses reference launches is credited bytes that never cross HBM:
    # `frac_moved` prices only what the launch itself must move, `traffic_frac` what the PMC
    # counters saw crossing HBM, both over the same launch duration.
    moved = traffic / (avg_ms * 1e-3) / 1e9 if traffic else None
    roof = {
        "bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "traffic": traffic, "avg_launch_ms": round(avg_ms, 5),
        "alg_bytes_per_launch": dom["alg_bytes"],
        "moved_bytes_per_launch": dom["moved_bytes"],
        "frac_moved": round(moved_rate / HBM_PEAK_GBPS, 4),
        "traffic_GBps": round(moved, 1) if moved else None,
        "traffic_frac": round(moved / HBM_PEAK_GBPS, 4) if moved else None,
        # the replayed counters belong to other kernel sources than the ones this run used
        "traffic_stale": (pmc.get("_kernels_sha") != kernel_sources_sha()) if pmc else None,
        "traffic_collected": ({"commit": pmc.get("_commit"), "date": pmc.get("_date"), "workload": pmc.get("_workload"),
                               "kernels_sha": pmc.get("_kernels_sha"),
                               "source": pmc.get("_source"), "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                               "bench.py on an MI355X; replayed, not measured in this run"} if pmc else None),
    }
    # Step level, honest version: the launches actually issued move `step_moved_bytes` (compulsory)
    # and `step_traffic_bytes` (counters) per step; over the wall-clock step time.
    step = {
        "step_moved_bytes": step_moved,
        "step_moved_bytes_per_cell": round(step_moved / (sim.n * sim.n), 2),
        "step_moved_frac": round(step_moved / t_step / 1e9 / HBM_PEAK_GBPS, 4),
        "step_traffic_bytes": step_traffic,
        "step_traffic_frac": round(step_traffic / t_step / 1e9 / HBM_PEAK_GBPS, 4) if step_traffic else None,
        "step_kernel_ms_sum": round(step_kernel_ms, 5),
    }
    return roof, per_kernel, step


def cpu_baseline_config0(pkg):
    """BASELINE.json configs[0] verbatim (SURVEY.md 8d "Config -> run (1)"): the serial host-C
    transcription, 512^2, D = J = 10, 100 steps on the seeded splat script, one core."""
    lib = pkg.FluidLib(os.path.join(ROOT, "oracle", "liboracle_fluid.so"))
    n, steps = 512, 100
    sim = pkg.FluidSim(n, diffusion_iters=10, pressure_iters=10, lib=lib)
    t0 = time.perf_counter()
    pkg.run_seeded(sim, steps)
    el = time.perf_counter() - t0
    return {"value": round(n * n * steps / el / 1e9, 6), "unit": "GCUPS", "cores": 1, "kind": "port",
            "ms_per_step": round(el / steps * 1e3, 3),
            "sample": f"{n}x{n} grid, D=10, J=10, {steps} steps, seeded splats every 10th step, serial C oracle "
                      f"(gcc -O2 -ffp-contract=off), {el:.1f} s"}


def launch_ranks(nranks, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this file.

    The parent never imports torch or touches HIP (a process that has initialised the GPU must not
    be re-executed, and it would hold a context the ranks do not need).  Every rank gets RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT (a port found free just now) and
    its own session, so that the whole group can be ended by process-group id.  Rank 0's stdout is
    this process's stdout; the other ranks' output goes to stderr.  Returns the exit code."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    limit = float(os.environ.get("FLUID_BENCH_TIMEOUT", "2400"))
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FLUID_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr, start_new_session=True))

    def end_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)   # the rank's own session: exactly what was started here
                except ProcessLookupError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass

    t0, rc = time.monotonic(), 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                print(f"bench.py: rank {bad[0][0]} exited with {bad[0][1]}; ending the other ranks", file=sys.stderr)
                rc = 1
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() - t0 > limit:
                print(f"bench.py: {nranks} ranks not done after {limit:.0f} s (FLUID_BENCH_TIMEOUT); ending them",
                      file=sys.stderr)
                rc = 124
                break
            time.sleep(0.05)
    finally:
        end_all()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--unfused", action="store_true", help="straight one-launch-per-reference-launch path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra single-GPU workloads")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))   # before anything imports torch / touches HIP
    if (args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1) and "FLUID_BENCH_ATTEMPT" not in os.environ:
        # one rank, under a launcher or launch_ranks: supervise attempts, touch no GPU (bench_multi.py)
        from bench_multi import supervise_rank
        sys.exit(supervise_rank(sys.argv[1:]))

    import fluid_amd_loader
    pkg = fluid_amd_loader.load()
    n, D, J = WORKLOADS[args.workload]
    flags = pkg.FLAG_UNFUSED if args.unfused else 0
    world = int(os.environ.get("WORLD_SIZE", "1"))

    if args.gpus > 1 or world > 1 or os.environ.get("FLUID_BENCH_FORCE_MULTI"):
        from bench_multi import run_multi  # slab-decomposed path
        return run_multi(pkg, args, n, D, J, flags)

    sim, wall = measure_single(pkg, n, D, J, args.steps, args.warmup, flags)
    t_step = wall / args.steps
    bpc = pkg.alg_bytes_per_cell(D, J)
    gcups = n * n / t_step / 1e9
    ev_ms = sim.time_steps(args.steps) / args.steps
    roof, per_kernel, step_acct = roofline_from_profile(sim, max(2, min(args.steps, 5)), args.workload, t_step)
    diag = sim.diagnostics()   # state after the run (wavefront reductions; outside the timed region)
    sim.close()

    out = {
        "metric": "cell-updates/sec per full timestep (GCUPS)", "value": round(gcups, 4),
        "unit": "GCUPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(t_step * 1e3, 5), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded splats, SURVEY.md 8d)",
        "config": {"workload": f"{n}x{n} grid, D={D} diffusion + J={J} Jacobi sweeps per step",
                   "name": args.workload, "path": "unfused" if args.unfused else "fused",
                   "graph": True},
        "step_alg_bytes_per_cell": bpc,
        "step_effective_GBps": round(bpc * n * n / t_step / 1e9, 1),
        "step_frac_of_hbm_peak": round(bpc * n * n / t_step / 1e9 / HBM_PEAK_GBPS, 4),
        "hip_event_ms_per_step": round(ev_ms, 5),
        **step_acct,
        "roofline": roof,
        "kernels": per_kernel,
        "diagnostics_after_run": diag,
    }
    if not args.no_extra:
        extra = {}
        for name in ("4096_J40", "1024_J40", "512_J10"):
            if name == args.workload:
                continue
            n2, D2, J2 = WORKLOADS[name]
            k2 = 50 if n2 > 1024 else 160   # multiples of the 16 steps a graph holds on the launch-bound grids
            s2, w2 = measure_single(pkg, n2, D2, J2, k2, 16, flags)
            b2 = pkg.alg_bytes_per_cell(D2, J2)
            t2 = w2 / k2
            extra[name] = {"ms_per_step": round(t2 * 1e3, 5), "GCUPS": round(n2 * n2 / t2 / 1e9, 4),
                           "step_effective_GBps": round(b2 * n2 * n2 / t2 / 1e9, 1),
                           "step_frac_of_hbm_peak": round(b2 * n2 * n2 / t2 / 1e9 / HBM_PEAK_GBPS, 4)}
            s2.close()
            if name == "4096_J40" and not flags:
                # the same grid as two slabs side by side on the GPU (fluid_params.inner_slabs, opt-in:
                # the slab contexts' advection reach applies); same bits, tests/test_inner_slabs.py
                s3, w3 = measure_single(pkg, n2, D2, J2, k2, 16, flags, inner_slabs=2)
                extra[name]["ms_per_step_inner_slabs_2"] = round(w3 / k2 * 1e3, 5)
                s3.close()
        out["other_workloads"] = extra
    if not args.no_cpu_baseline:
        # `cpu_baseline`: the serial port on the workload's own grid when ONE step of it fits the
        # contract's 10-30 s of CPU work (16384^2, J = 40: about 20 s on one core, 16 GB of host memory),
        # judged from a 3 s probe at 1024^2 (kept as cpu_baseline_probe); else the 1024^2 sample.
        probe = cpu_baseline(pkg, D, J, budget_s=3.0)
        est = n * n / (probe["value"] * 1e9)
        if n > 1024 and est <= float(os.environ.get("FLUID_BENCH_CPU_STEP_LIMIT", "40")):
            full = cpu_baseline(pkg, D, J, budget_s=0.0, n=n)   # budget 0: exactly one timed step
            out["cpu_baseline"], out["cpu_baseline_probe"] = full, probe
        else:
            out["cpu_baseline"] = cpu_baseline(pkg, D, J)
        out["cpu_baseline_config0"] = cpu_baseline_config0(pkg)
        nthr = cpu_threads()
        if nthr > 1 and os.path.exists(os.path.join(ROOT, "oracle", "liboracle_fluid_omp.so")):
            out["cpu_baseline_parallel"] = cpu_baseline(pkg, D, J, budget_s=6.0, threads=nthr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
