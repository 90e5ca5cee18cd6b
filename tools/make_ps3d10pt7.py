#!/usr/bin/env python3
"""Regenerate the `ps3d10pt7` data set the reference's examples/ex1.yml / ex2.yml read
(Zenodo record 17471036 is unreachable offline): 7-pt Laplace on 10x10x10, b = ones, in
hypre's ASCII IJ format (`<prefix>.<rank %05d>`; docs/usrman-src/driver_examples.rst:171-173,
examples/refOutput/ex1.txt:17: 1000 rows / 6400 nonzeros).  Writes np1 and np4 partitions."""
import os
import sys


def rows(n=10):
    for z in range(n):
        for y in range(n):
            for x in range(n):
                i = (z * n + y) * n + x
                ent = []
                if z > 0: ent.append((i - n * n, -1.0))
                if y > 0: ent.append((i - n, -1.0))
                if x > 0: ent.append((i - 1, -1.0))
                ent.append((i, 6.0))
                if x < n - 1: ent.append((i + 1, -1.0))
                if y < n - 1: ent.append((i + n, -1.0))
                if z < n - 1: ent.append((i + n * n, -1.0))
                yield i, ent


def write(root, nparts, n=10):
    N = n ** 3
    d = os.path.join(root, "data", "ps3d10pt7", f"np{nparts}")
    os.makedirs(d, exist_ok=True)
    allrows = list(rows(n))
    for r in range(nparts):
        lo, hi = r * N // nparts, (r + 1) * N // nparts - 1
        with open(os.path.join(d, f"IJ.out.A.{r:05d}"), "w") as f:
            f.write(f"{lo} {hi} {lo} {hi}\n")
            for i, ent in allrows[lo:hi + 1]:
                for j, v in ent:
                    f.write(f"{i} {j} {v:.14e}\n")
        with open(os.path.join(d, f"IJ.out.b.{r:05d}"), "w") as f:
            f.write(f"{lo} {hi}\n")
            for i in range(lo, hi + 1):
                f.write(f"{i} {1.0:.14e}\n")


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    write(root, 1)
    write(root, 4)
    print("wrote", os.path.join(root, "data", "ps3d10pt7"))
