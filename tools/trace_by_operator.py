#!/usr/bin/env python3
"""Split rocprofv3's per-kernel statistics by operator size.

The multigrid cycle launches one kernel name on every level, so `--stats` averages launches that
differ by orders of magnitude.  This reads the per-dispatch kernel trace of the same run and
groups each kernel's launches into duration clusters (a gap of more than 1.5x starts a new
cluster; levels of an AMG hierarchy differ by 2x or more), largest first:

    python tools/trace_by_operator.py gpurun_out/r01c/trace/run_kernel_trace.csv profiles/r01c_kernel_by_operator.csv"""
import csv
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1:3]
    d = defaultdict(list)
    for r in csv.DictReader(open(src)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for k, v in d.items():
        v.sort(reverse=True)
        cl = [[v[0]]]
        for x in v[1:]:
            if cl[-1][-1] > 1.5 * x:
                cl.append([x])
            else:
                cl[-1].append(x)
        for i, c in enumerate(cl):
            rows.append((sum(c), k, i, len(c), sum(c) / len(c), min(c), max(c)))
    rows.sort(reverse=True)
    with open(dst, "w") as o:
        o.write("kernel,cluster,calls,total_ns,average_ns,min_ns,max_ns\n")
        for tot, k, i, n, avg, mn, mx in rows:
            if tot < 200000:
                continue
            o.write(f'"{k.split("(")[0]}",{i},{n},{tot},{avg:.1f},{mn},{mx}\n')


if __name__ == "__main__":
    main()
