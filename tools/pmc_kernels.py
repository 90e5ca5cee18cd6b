#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 counter passes (one small --pmc group per pass, counters only):

    python tools/pmc_kernels.py <dir with p1/, p2/, ... pass directories> [kernel-name substring ...]

For every kernel whose name contains one of the substrings (default: all) and every counter: calls, max, and the median of the
launches within 30 % of the maximum (one kernel name serves every level of a hierarchy: the top cluster is the biggest operator)."""
import collections
import csv
import glob
import statistics
import sys


def summarise(root, keep):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if keep and not any(s in k for s in keep):
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for k in sorted(agg):
        for c, v in sorted(agg[k].items()):
            m = max(v)
            top = [x for x in v if x >= 0.7 * m] if m > 0 else v
            rows.append((k, c, len(v), m, statistics.median(top)))
    return rows


if __name__ == "__main__":
    print("kernel,counter,calls,max,median_of_top_cluster")
    for k, c, n, m, t in summarise(sys.argv[1], sys.argv[2:]):
        print(f'"{k}",{c},{n},{m:.6g},{t:.6g}')
