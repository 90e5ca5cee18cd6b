#!/usr/bin/env python3
"""Extract the numeric pins the reference keeps in its checked-in outputs
(examples/refOutput/*.txt) into tests/golden/ref_pins.json.

Run in the build container (needs /root/reference); the JSON is committed because the
reference does not travel to the GPU box.  Only numbers are extracted -- no reference
source text is stored.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_pins.json")


def stats_rows(txt):
    rows = []
    for m in re.finditer(r"^\|\s+(\d+) \|\s+([\d.]*) \|\s+([\d.]+) \|\s+([\d.]+) \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|",
                         txt, re.M):
        rows.append(dict(entry=int(m.group(1)), r0=float(m.group(5)), rel=float(m.group(6)),
                         iters=int(m.group(7))))
    return rows


def main():
    pins = {}
    ro = os.path.join(REF, "examples", "refOutput")
    ex1 = open(os.path.join(ro, "ex1.txt")).read()
    m = re.search(r"with (\d+) rows and (\d+) nonzeros", ex1)
    pins["ex1"] = dict(source="examples/refOutput/ex1.txt:17,27", rows=int(m.group(1)),
                       nnz=int(m.group(2)), stats=stats_rows(ex1),
                       config="10^3 7-pt, b=ones, PCG + BoomerAMG CPU defaults (HMIS, ext+i, hl1GS 13/14)")
    ex2 = open(os.path.join(ro, "ex2.txt")).read()
    ops = [dict(lev=int(a), rows=int(b), nnz=int(c)) for a, b, c in
           re.findall(r"^\s+(\d+)\s+(\d+)\s+(\d+)\s+[\d.]+\s+\d+\s+\d+\s+[\d.]+\s+\S+\s+\S+\s*$", ex2, re.M)]
    interp = [dict(lev=int(a), rows=int(b), cols=int(c), min=int(d), max=int(e)) for a, b, c, d, e in
              re.findall(r"^\s+(\d+)\s+(\d+) x (\d+)\s+(\d+)\s+(\d+)\s+[\d.]+", ex2, re.M)]
    hist = [(int(a), float(b), float(c)) for a, b, c in
            re.findall(r"^\s+(\d+)\s+([\d.]+e[+-]\d+)\s+[\d.]+\s+([\d.]+e[+-]\d+)\s*$", ex2, re.M)]
    cx = re.search(r"grid = ([\d.]+)\s+operator = ([\d.]+)", ex2)
    bb = re.search(r"<b,b>: (\S+)", ex2)
    pins["ex2"] = dict(source="examples/refOutput/ex2.txt:124-139,160-169", operators=ops,
                       interp=interp, grid_complexity=float(cx.group(1)),
                       operator_complexity=float(cx.group(2)), bdotb=float(bb.group(1)),
                       history=[dict(it=a, rnorm=b, rel=c) for a, b, c in hist],
                       stats=stats_rows(ex2),
                       config="same matrix on 4 ranks, PMIS + ext+i(4) + hl1GS 13/14 + FSAI level-0 smoother")
    lap = open(os.path.join(ro, "laplacian.txt")).read()
    pins["laplacian"] = dict(source="examples/refOutput/laplacian.txt:10-16,34-38",
                             stats=stats_rows(lap),
                             config="10^3 generator defaults, b=1 on y=0 plane, presets pcg+poisson, CPU defaults")
    # second example driver: time-dependent convection-diffusion, GMRES + BoomerAMG (examples/src/C_convdif)
    cd = open(os.path.join(ro, "convdif.txt")).read()
    steps = [dict(step=int(a), lin=int(b), cmax=float(c), mass=float(d)) for a, b, c, d in
             re.findall(r"^Time step:\s+(\d+) \|.*\| Lin:\s+(\d+) \| min\(c\)=\s*\S+ max\(c\)=\s*(\S+) mass=(\S+)", cd, re.M)]
    paths = [dict(path=a, r0=float(b), rel=float(c), iters=int(d)) for a, b, c, d in
             re.findall(r"^\|\s+(\d+\.\d+) \|\s+[\d.]* \|\s+[\d.]+ \|\s+[\d.]+ \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", cd, re.M)]
    pins["convdif"] = dict(source="examples/refOutput/convdif.txt:26-35,55-64", steps=steps, paths=paths,
                           config="64x16x16 cells, 10 growing time steps, presets gmres + poisson (CPU defaults)")
    # third example driver: linear elasticity, PCG + systems AMG (examples/src/C_elasticity)
    el = open(os.path.join(ro, "elasticity.txt")).read()
    pins["elasticity"] = dict(source="examples/refOutput/elasticity.txt:9-16,37-41", stats=stats_rows(el),
                              config="30x10x10 nodes Q1 hexahedra, 3 dofs/node, presets pcg + elasticity_3D "
                                     "(num_functions 3, strong_th 0.8), CPU defaults")
    # AMG variants on the 10^3 system (examples/ex8.yml as it stood when the output was made: the echoed input tree in the
    # output itself, ex8.txt:26-78, names every variant; the initial residual 1.58e+01 = sqrt(250) against ex2's sqrt(1000) on
    # the same np4 files says only the first of the four right-hand-side parts was read on one rank)
    ex8 = open(os.path.join(ro, "ex8.txt")).read()
    pins["ex8"] = dict(source="examples/refOutput/ex8.txt:26-78,92-96", stats=stats_rows(ex8),
                       config="10^3 7-pt (np4 files on 1 rank), b = 1 on rows 0..249, PCG rtol 1e-9; variants: "
                              "0 HMIS 0.25 + mm-ext+i + Chebyshev(2); 1 HMIS 0.5 + mm-ext+i + Chebyshev(4, fraction 0.1); "
                              "2 HMIS 0.8 + mm-ext+i + l1sym-hgs; 3 HMIS 0.9 + mm-ext+i + Chebyshev(2) + ILU(0) level-0 smoother; "
                              "4 PMIS 0.5 + standard interpolation + l1-hsgs x2 (not restated)")
    # analytic unit-test anchors (tests/test_linsys.c:4126-4155, tests/test_setmatrix_from_csr.c:397-417)
    pins["unit"] = dict(norms_of_1_m2_3=dict(L1=6.0, L2=14.0 ** 0.5, Linf=3.0),
                        one_by_one=dict(a=3.0, b=6.0, x_norm=2.0, tol=1e-6))
    json.dump(pins, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
