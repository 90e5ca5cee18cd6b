#!/usr/bin/env python3
"""Per-launch durations of the Gauss-Seidel kernels in a rocprofv3 kernel trace, in launch order (one V-cycle's worth from the end).
    python tools/gs_trace.py <kernel_trace.csv> [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gs = [(r["Kernel_Name"].split("(")[0][-40:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows
      if "k_gs_" in r["Kernel_Name"] and "expand" not in r["Kernel_Name"] and "indeg" not in r["Kernel_Name"]]
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print(len(gs), "launches")
for k, d in gs[-cnt:]:
    print(f"{k:42s} {d:9.1f} us")
