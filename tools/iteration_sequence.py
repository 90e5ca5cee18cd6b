#!/usr/bin/env python3
"""Dispatch sequence of the LAST PCG iteration-like period of a rocprofv3 kernel trace: every kernel between the last two launches of
k_cg_update, with start offset, duration and gap to its predecessor (ns), plus totals of the dispatches shorter than 30 us.
python tools/iteration_sequence.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(sys.argv[1]))))
upd = [i for i, r in enumerate(rows) if "k_cg_update" in r[2]]
if len(upd) < 3:
    sys.exit("no PCG iterations in this trace")
a, b = upd[-3], upd[-2]
t0, prev = rows[a][0], rows[a][0]
small = gaps = 0
nsmall = 0
for s, e, k in rows[a:b]:
    d = e - s
    print(f"{s - t0:9d} {d:8d} gap {s - prev:7d}  {k}")
    if d < 30000:
        small += d
        nsmall += 1
        gaps += max(s - prev, 0)
    prev = e
print(f"iteration: {rows[b][0] - t0} ns, {b - a} dispatches; {nsmall} shorter than 30 us: {small} ns in kernels + {gaps} ns of gaps before them")
