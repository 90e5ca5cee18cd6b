#!/usr/bin/env python3
"""Bytes-vs-time accounting of the AMG setup ("prec" timer; reference src/internal/solver.c:288-302, HYPRE_BoomerAMGSetup): the largest
setup kernels of ONE warm setup at 256^3 with their time, the HBM bytes the counters saw (FETCH_SIZE x the gfx950 factor + WRITE_SIZE,
as tools/pmc_traffic.py calibrates them), the rate that makes, and a lower bound on the bytes the step needs (its inputs read once, its
outputs written once).

    python tools/setup_accounting.py <kernel_trace.csv> <pmc_fetch dir> <pmc_write dir> <setup_only.json> > profiles/<tag>_setup_accounting.md

Dispatches are those between the two hda::k_marker launches (the second setup of tools/setup_only.py); launches of one kernel name are
split into the hierarchy's levels by order of appearance where the level loop launches it once per level, else reported summed."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

FETCH_FACTOR = 2.0  # gfx950: FETCH_SIZE reports half the bytes of a streamed read (MI355X_MICROARCH.md, HBM section; pmc_traffic.py re-derives 1.9-2.0)


def short(name):
    k = name.split("(")[0]
    for pre in ("void ", "hda::"):
        if k.startswith(pre):
            k = k[len(pre):]
    return k.replace("hda::", "")


def between_markers(rows, name_key):
    idx = [i for i, r in enumerate(rows) if "k_marker" in r[name_key]]
    return rows[idx[0] + 1:idx[1]] if len(idx) >= 2 else rows


def load_trace(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    rows = between_markers(rows, "Kernel_Name")
    return [(short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]


def load_pmc(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"])))
    rows.sort()
    rows = [{"k": k, "v": v} for _, k, v in rows]
    idx = [i for i, r in enumerate(rows) if "k_marker" in r["k"]]
    rows = rows[idx[0] + 1:idx[1]] if len(idx) >= 2 else rows
    return [(r["k"], r["v"]) for r in rows]


def main():
    trace, fdir, wdir, dims_json = sys.argv[1:5]
    T = load_trace(trace)
    F, W = load_pmc(fdir, "FETCH_SIZE"), load_pmc(wdir, "WRITE_SIZE")
    info = json.loads([ln for ln in open(dims_json) if ln.startswith("{")][-1])
    lv = info["levels"]
    # per kernel name: lists in launch order (the three runs launch the same sequence)
    t_by, f_by, w_by = defaultdict(list), defaultdict(list), defaultdict(list)
    for k, d in T: t_by[k].append(d)
    for k, v in F: f_by[k].append(v * 1024.0 * FETCH_FACTOR)
    for k, v in W: w_by[k].append(v * 1024.0)
    total_ns = sum(d for _, d in T)
    out = []
    for k, ds in t_by.items():
        fs, ws = f_by.get(k, []), w_by.get(k, [])
        same = len(fs) == len(ds) == len(ws)
        # largest launch on its own (the level it belongs to is named by the caller's table below), the rest summed
        order = sorted(range(len(ds)), key=lambda i: -ds[i])
        big = order[0]
        out.append((ds[big], k, 1, (fs[big] + ws[big]) if same else None, "largest launch"))
        if len(ds) > 1:
            rest = [i for i in order[1:]]
            out.append((sum(ds[i] for i in rest), k, len(rest), sum(fs[i] + ws[i] for i in rest) if same else None, "all other launches"))
    out.sort(reverse=True)
    n0, z0 = lv[0]["rows"], lv[0]["nnz"]
    n1, z1 = (lv[1]["rows"], lv[1]["nnz"]) if len(lv) > 1 else (0, 0)
    p0, p1 = lv[0]["P_nnz"], (lv[1]["P_nnz"] if len(lv) > 1 else 0)
    # lower bounds (inputs once + outputs once), by kernel name and the level its largest launch works on
    bounds = {
        "k_interp_wave<64>": ("extended+i interpolation of level 1 (rows up to ~100 entries): A_1 once (12 B/entry) + strength mask + C/F marker, P_1 written", 13.0 * z1 + 4.0 * n1 + 12.0 * p1),
        "k_interp_wave<8>": ("extended+i interpolation of level 0 (7-point rows): A_0 once + mask + marker, P_0 written", 13.0 * z0 + 4.0 * n0 + 12.0 * p0),
        "k_spgemm_esc<8, 256>": ("Galerkin product, largest chunked launch (expand-sort-compress): both factors once, the product written", None),
        "k_spgemm_esc<8, 512>": ("Galerkin product (4096-product chunks)", None),
        "k_spgemm_esc<8, 1024>": ("Galerkin product (8192-product chunks)", None),
        "k_pmis_setF": ("PMIS: undecided rows look for a strong C neighbour: A_0 pattern + mask + marker", 5.0 * z0 + 8.0 * n0),
        "k_pmis_mark": ("PMIS: one edge visit decides both endpoints: A_0 pattern + mask + measures + marker", 5.0 * z0 + 16.0 * n0),
        "k_pmis_init": ("PMIS measures: in-degrees, hash, marker", 20.0 * n0),
        "k_sort_rows": ("column sort of the level-1 operator's rows", 24.0 * z1),
        "k_l1_grp<32>": ("l1 row norms of level 1", 12.0 * z1 + 12.0 * n1),
        "k_strength_grp<32>": ("strength of connection on level 1: A_1 once, one mask byte per entry", 13.0 * z1 + 4.0 * n1),
        "k_vhist": ("value histogram of the coding pass", None),
        "k_win_count": ("windowed-CSR plan: distinct columns per chunk", None),
    }
    print("| kernel (launches) | what it does | time, ms | share of the setup's kernel time | counter bytes, MB | rate, GB/s | lower bound on bytes, MB | bound / (time x 8 TB/s) |")
    print("|---|---|---|---|---|---|---|---|")
    for d, k, c, by, which in out[:14]:
        desc, lb = bounds.get(k, ("", None)) if which == "largest launch" else (which, None)
        ms = d / 1e6
        cb = f"{by / 1e6:.0f}" if by else "n/a"
        rate = f"{by / d:.0f}" if by else "n/a"
        lbs = f"{lb / 1e6:.0f}" if lb else "—"
        fr = f"{lb / (d * 8000.0):.3f}" if lb else "—"
        print(f"| `{k}` ({c}) | {desc} | {ms:.2f} | {100.0 * d / total_ns:.1f} % | {cb} | {rate} | {lbs} | {fr} |")
    print()
    print(f"Kernel time of the setup between the markers: {total_ns / 1e6:.1f} ms in {len(T)} dispatches; wall {info['setup_ms'][1]:.1f} ms (second setup; first {info['setup_ms'][0]:.1f} ms).")
    print("Levels: " + "; ".join(f"{x['level']}: {x['rows']} rows / {x['nnz']} entries" for x in lv))


if __name__ == "__main__":
    main()
