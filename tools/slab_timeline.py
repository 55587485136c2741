#!/usr/bin/env python3
"""One steady-state step of the slab driver from a rocprofv3 kernel trace of
tools/slab_host_overhead.py: python tools/slab_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def name(r):
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
    return m.group(1) if m else r["Kernel_Name"][:40]


marks = [i for i, r in enumerate(rows) if "k_diffuse_tb" in r["Kernel_Name"]]
i0, i1 = marks[-20], marks[-19]          # one period: from one diffusion launch to the next
t0 = int(rows[i0]["Start_Timestamp"])
print(f"period {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us")
print(f"{'start us':>9} {'dur us':>8} {'queue':>5}  kernel  (grid in threads)")
busy = {}
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {r['Queue_Id']:>5}  {name(r)}  "
          f"({r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']})")
    busy[r["Queue_Id"]] = busy.get(r["Queue_Id"], 0) + (e - s)
for q, b in sorted(busy.items()):
    print(f"queue {q}: busy {b / 1e3:.1f} us")
