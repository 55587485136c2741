#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X stable-fluids solver.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one full simulationStep() (kernel.cu:701-708) over the whole grid.  Metric:
cell-updates/s per full timestep (GCUPS = n^2 / t_step / 1e9) and the HBM-roofline fraction
(BASELINE.json).  Prints ONE JSON line on rank 0.

For N > 1 the driver launches this file under torch.distributed.run, one rank per GPU; the grid
is cut into N row slabs (strong scaling: the total grid is fixed) and halo rows travel over
RCCL (torch.distributed backend "nccl").
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


def measure_single(pkg, n, D, J, steps, warmup, flags=0):
    import torch
    sim = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, flags=flags)
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


def cpu_baseline(pkg, D, J, budget_s=12.0, threads=1):
    """The host-C oracle (oracle/fluid_ref.c) on a bounded sample of the workload: a 1024^2 grid
    with the workload's D and J, as many steps as fit the budget.  threads=1: the serial build (the
    contract's `cpu_baseline`); threads>1: the same source built with -fopenmp (bit-identical,
    tests/test_oracle_kat.py), reported separately as `cpu_baseline_parallel`."""
    path = os.path.join(ROOT, "oracle", "liboracle_fluid.so" if threads == 1 else "liboracle_fluid_omp.so")
    if threads > 1:
        os.environ["OMP_NUM_THREADS"] = str(threads)  # read by libgomp when the library is loaded
    lib = pkg.FluidLib(path)
    n = 1024
    sim = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, lib=lib)
    seeded_warm_state(pkg, sim)
    sim.step(1)  # warm caches / leave the all-zero state
    steps, t0 = 0, time.perf_counter()
    while True:
        sim.step(1)
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 400:
            break
    return {
        "value": round(n * n * steps / el / 1e9, 6), "unit": "GCUPS", "cores": threads, "kind": "port",
        "sample": f"{n}x{n} grid, D={D}, J={J}, {steps} steps, seeded splats, "
                  + ("serial C oracle" if threads == 1 else f"C oracle with OpenMP row loops, {threads} threads")
                  + f" (gcc -O2 -ffp-contract=off), {el:.1f} s",
    }


def roofline_from_profile(sim, steps, workload):
    stats = sim.profile_steps(steps)
    dom = max(stats, key=lambda s: s["total_ms"])
    avg_ms = dom["total_ms"] / dom["launches"]
    achieved = dom["alg_bytes"] / (avg_ms * 1e-3) / 1e9
    per_kernel = {s["name"]: {"launches_per_step": s["launches"] / steps,
                              "avg_ms": round(s["total_ms"] / s["launches"], 5),
                              "alg_GBps": round(s["alg_bytes"] / (s["total_ms"] / s["launches"] * 1e-3) / 1e9, 1)}
                  for s in stats}
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            if pmc.get("_workload") == workload:  # counters are per launch of THAT workload's kernels
                traffic = pmc.get(dom["name"], {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # `achieved` prices every sweep's algorithmic bytes (contract), so a temporally blocked
    # launch of K sweeps can exceed the HBM peak; `traffic_frac` is the bytes the PMC counters
    # saw actually crossing HBM over the same launch duration.
    moved = traffic / (avg_ms * 1e-3) / 1e9 if traffic else None
    return {
        "bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "traffic": traffic, "avg_launch_ms": round(avg_ms, 5),
        "alg_bytes_per_launch": dom["alg_bytes"],
        "traffic_GBps": round(moved, 1) if moved else None,
        "traffic_frac": round(moved / HBM_PEAK_GBPS, 4) if moved else None,
    }, per_kernel


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
    roof, per_kernel = roofline_from_profile(sim, max(2, min(args.steps, 5)), args.workload)
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
        "roofline": roof,
        "kernels": per_kernel,
        "diagnostics_after_run": diag,
    }
    if not args.no_extra:
        extra = {}
        for name in ("4096_J40", "1024_J40"):
            if name == args.workload:
                continue
            n2, D2, J2 = WORKLOADS[name]
            s2, w2 = measure_single(pkg, n2, D2, J2, 50, 10, flags)
            b2 = pkg.alg_bytes_per_cell(D2, J2)
            t2 = w2 / 50
            extra[name] = {"ms_per_step": round(t2 * 1e3, 5), "GCUPS": round(n2 * n2 / t2 / 1e9, 4),
                           "step_effective_GBps": round(b2 * n2 * n2 / t2 / 1e9, 1),
                           "step_frac_of_hbm_peak": round(b2 * n2 * n2 / t2 / 1e9 / HBM_PEAK_GBPS, 4)}
            s2.close()
        out["other_workloads"] = extra
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, D, J)
        nthr = cpu_threads()
        if nthr > 1 and os.path.exists(os.path.join(ROOT, "oracle", "liboracle_fluid_omp.so")):
            out["cpu_baseline_parallel"] = cpu_baseline(pkg, D, J, budget_s=6.0, threads=nthr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
