#!/usr/bin/env python3
"""Reduce a gpurun_out/prof_TAG directory (rocprofv3 csv output) to one small JSON + text table.
Usage: python tools/summarize_profile.py gpurun_out/prof_TAG profiles/TAG"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


out = {"kernels": {}}
# kernel trace: per-dispatch durations
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    agg = defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in agg.items():
        v.sort()
        out["kernels"][k] = {"calls": len(v), "avg_us": round(sum(v) / len(v), 2),
                             "median_us": round(v[len(v) // 2], 2), "total_us": round(sum(v), 1)}
# counters: per dispatch rows (Kernel_Name, Counter_Name, Counter_Value)
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, ctrs in pmc.items():
    d = out["kernels"].setdefault(k, {})
    d["pmc_avg_per_dispatch"] = {c: round(sum(v) / len(v), 1) for c, v in sorted(ctrs.items())}
# HBM traffic per launch, corrected as MI355X_MICROARCH.md (HBM) prescribes for gfx950:
# FETCH_SIZE counts 64 B per 128-B read request => double it; WRITE_SIZE is exact; unit KiB.
traffic = {}
for k, v in out["kernels"].items():
    p = v.get("pmc_avg_per_dispatch", {})
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        v["hbm_bytes_per_launch"] = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
        base = k.split("<")[0]
        traffic.setdefault(base, []).append((v.get("calls", 1), v["hbm_bytes_per_launch"]))
out["traffic_by_kernel_family"] = {
    b: {"hbm_bytes_per_launch": sum(c * x for c, x in lst) / max(sum(c for c, _ in lst), 1)}
    for b, lst in traffic.items()}
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
if len(sys.argv) > 3:  # also write the table bench.py reads for roofline.traffic
    table = dict(out["traffic_by_kernel_family"])
    table["_workload"] = sys.argv[4] if len(sys.argv) > 4 else "16384_J40"  # bench.py applies it to this workload only
    table["_source"] = dst + ".json"
    import datetime
    import subprocess
    try:   # the commit the profiled tree was taken from (this script runs in the repo, after the GPU call)
        table["_commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                          check=True).stdout.strip()
        if subprocess.run(["git", "status", "--porcelain", "--untracked-files=no"], capture_output=True,
                          text=True).stdout.strip():
            table["_commit"] += "+uncommitted"
    except Exception:
        table["_commit"] = None
    table["_date"] = datetime.date.today().isoformat()
    # what the counters were collected ON: bench.py compares this with the kernel sources it runs with
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_sources_sha
    table["_kernels_sha"] = kernel_sources_sha()
    json.dump(table, open(sys.argv[3], "w"), indent=1, sort_keys=True)
json.dump(out, open(dst + ".json", "w"), indent=1, sort_keys=True)
with open(dst + ".txt", "w") as fh:
    tot = sum(v.get("total_us", 0) for v in out["kernels"].values())
    fh.write(f"{'kernel':44s} {'calls':>6s} {'avg_us':>10s} {'total_us':>10s} {'%':>6s}\n")
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("total_us", 0)):
        if "calls" in v:
            fh.write(f"{k:44s} {v['calls']:6d} {v['avg_us']:10.2f} {v['total_us']:10.1f} {100 * v['total_us'] / max(tot, 1e-9):6.1f}\n")
    for k, v in sorted(out["kernels"].items()):
        if "pmc_avg_per_dispatch" in v:
            fh.write(f"\n[{k}] counters, mean per dispatch\n")
            for c, x in v["pmc_avg_per_dispatch"].items():
                fh.write(f"  {c:28s} {x:16.1f}\n")
print(open(dst + ".txt").read())
