#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel.  Usage: tools/pmc_by_kernel.py DIR [DIR ...]"""
import collections
import csv
import glob
import re
import sys

for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1)
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k in sorted(agg):
            ds = list(dur[k].values())
            print(f"{k}  launches={len(ds)}  avg_us={sum(ds) / len(ds) / 1e3:.1f}")
            for c, v in sorted(agg[k].items()):
                print(f"    {c:28s} {sum(v) / len(v):14.0f}")
