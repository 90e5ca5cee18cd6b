#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes over the same bench.py command into HBM traffic per launch.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o run -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o run -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01c

FETCH_SIZE / WRITE_SIZE come in KB.  On gfx950 FETCH_SIZE reports half of the bytes of a
streamed read (MI355X_MICROARCH.md, HBM section); the factor is re-derived here from this run's
own k_cg_dir launches, whose byte count is known exactly (reads z, p; writes p; 8 B per lane),
and applied to every kernel.  Writes: <tag>_pmc_summary.csv (per kernel: calls, max / median of
both counters) and profiles/traffic.json (the launches bench.py quotes)."""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def compute(fdir, wdir, tag, summary_csv=None):
    """The traffic dictionary of one pair of counter passes (bench.py calls this on passes it has just run itself); tag names the
    round / run; summary_csv: where to write the per-kernel table, if anywhere."""
    F, W = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    byk = defaultdict(lambda: {"F": [], "W": []})
    for _, k, v in F:
        byk[k]["F"].append(v)
    for _, k, v in W:
        byk[k]["W"].append(v)
    with open(summary_csv or os.devnull, "w") as o:
        o.write("kernel,calls,FETCH_SIZE_max_KB,FETCH_SIZE_median_KB,WRITE_SIZE_max_KB,WRITE_SIZE_median_KB\n")
        for k, d in sorted(byk.items(), key=lambda kv: -sum(kv[1]["F"])):
            f, w = d["F"] or [0.0], d["W"] or [0.0]
            o.write(f'"{k}",{len(d["F"])},{max(f):.1f},{statistics.median(f):.1f},{max(w):.1f},{statistics.median(w):.1f}\n')

    def med(name_part, counter, top_cluster=False):
        vals = [v for k, d in byk.items() if name_part in k for v in d[counter]]
        if not vals:
            return None
        if top_cluster:  # the launches on the largest operator: within 5 % of the maximum
            m = max(vals)
            vals = [v for v in vals if v >= 0.95 * m]
        return statistics.median(vals)

    n = 256 ** 3
    cal_f, cal_w = med("k_cg_dir", "F", True), med("k_cg_dir", "W", True)
    fetch_factor = (2 * 8.0 * n / 1024) / cal_f if cal_f else 2.0
    write_factor = (8.0 * n / 1024) / cal_w if cal_w else 1.0
    out = {"formula": "traffic = fetch_factor * FETCH_SIZE*1024 + write_factor * WRITE_SIZE*1024",
           "calibration": {"kernel": "k_cg_dir on 256^3 (reads z,p = 268435456 B, writes p = 134217728 B, 8 B/lane streaming)",
                           "FETCH_SIZE_KB": cal_f, "WRITE_SIZE_KB": cal_w, "fetch_factor": fetch_factor, "write_factor": write_factor},
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 "
                     f"--no-cpu-baseline --no-kernel-table; summary in {os.path.basename(tag)}_pmc_summary.csv"}

    def traffic(name_part, top_cluster):
        f, w = med(name_part, "F", top_cluster), med(name_part, "W", top_cluster)
        if f is None:
            return None, None, None
        return fetch_factor * f * 1024 + write_factor * (w or 0.0) * 1024, f, w

    names = "".join(byk.keys())
    level0_coded = any(q in names for q in ("k_spmv_rowclass<0, true", "k_spmv_coded_row<0, true"))
    for key, part, top in (("k_spmv_level0", "k_spmv_rowclass<0, true", False), ("k_spmv_level0", "k_spmv_coded_row<0, true", False),
                           ("k_spmv_level0", "k_spmv_stream<0, true", True)):
        t, f, w = traffic(part, top)
        if t is not None and key + "_bytes_per_launch" not in out:
            out[key + "_bytes_per_launch"] = t
            out[key] = {"kernel": part, "FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w}
    # Jacobi sweep on the largest plain-CSR operator (bench.py's dominant kernel): level 1 when level 0 is coded
    t, f, w = traffic("k_spmv_win<2, false", True)
    kname = "k_spmv_win<2, false, ...> (windowed CSR)"
    if t is None:
        t, f, w = traffic("k_spmv_stream<2, false", True)
        kname = "k_spmv_stream<2, false, ...>"
    if t is not None:
        lvl = 1 if level0_coded else 0
        out[f"k_spmv_stream_jacobi_level{lvl}_bytes_per_launch"] = t
        out[f"k_spmv_stream_jacobi_level{lvl}"] = {"kernel": kname + ", launches within 5 % of the largest FETCH_SIZE",
                                                   "FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w}
    # level-0 transfer operators: the value-coded stream kernel in PLAIN mode without dot; P (x += P e) writes 8 B per fine row,
    # R (f_c = R t) 8 B per coarse row -- told apart by WRITE_SIZE
    # (since round 2 one of the two runs on the windowed kernel, and the restriction also writes the next level's first sweep:
    # 2 x 8 B per coarse row)
    pr = [k for k in byk if "k_spmv_stream<0, false, true" in k or "k_spmv_win<0, false, true" in k]
    if pr and len(F) == len(W):
        wmap = {d: v for d, _, v in W}
        pairs = [(v, wmap.get(d)) for d, k, v in F if k in pr and wmap.get(d) is not None]
        if pairs:
            wmax = max(w for _, w in pairs)
            Pp = [(f, w) for f, w in pairs if w >= 0.9 * wmax]
            Rr = [(f, w) for f, w in pairs if 0.2 * wmax <= w <= 0.75 * wmax and f >= 0.5 * max(q for q, _ in pairs)]
            for key, sel in (("level0_prolongation", Pp), ("level0_restriction", Rr)):
                if sel:
                    f, w = statistics.median([q for q, _ in sel]), statistics.median([q for _, q in sel])
                    out[key + "_bytes_per_launch"] = fetch_factor * f * 1024 + write_factor * w * 1024
                    out[key] = {"kernel": "k_spmv_stream / k_spmv_win <0, false, true, false> (value-coded level-0 transfer operator)", "launches": len(sel),
                                "FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w}
    # the whole solve phase: every dispatch between the two hda::k_marker launches bench.py puts around its timed solves, both counters
    # summed with the calibrated factors (setup kernels, uploads and the probes' own launches stay outside the markers)
    sp = solve_phase(F, W, fetch_factor, write_factor)
    if sp:
        out.update(sp)
    out["round"] = os.path.basename(tag)
    return out


def between_markers(rows):
    """(dispatch id, kernel, value) rows of the dispatches strictly between the first and the last hda::k_marker launch; None without two markers"""
    marks = [d for d, k, _ in rows if "k_marker" in k]
    if len(marks) < 2:
        return None
    lo, hi = marks[0], marks[-1]
    return [(d, k, v) for d, k, v in rows if lo < d < hi]


def solve_phase(F, W, fetch_factor, write_factor):
    f, w = between_markers(F), between_markers(W)
    if f is None or w is None:
        return None
    fk, wk = sum(v for _, _, v in f), sum(v for _, _, v in w)
    per = defaultdict(lambda: [0, 0.0, 0.0])
    for _, k, v in f:
        per[k][0] += 1
        per[k][1] += v
    for _, k, v in w:
        per[k][2] += v
    top = sorted(per.items(), key=lambda kv: -(fetch_factor * kv[1][1] + write_factor * kv[1][2]))[:8]
    return {"solve_phase_traffic_bytes": fetch_factor * fk * 1024 + write_factor * wk * 1024,
            "solve_phase": {"what": "sum over every dispatch between the two hda::k_marker launches around the timed solve(s): "
                                    "fetch_factor * FETCH_SIZE + write_factor * WRITE_SIZE",
                            "dispatches": len(f), "FETCH_SIZE_KB_sum": fk, "WRITE_SIZE_KB_sum": wk,
                            "top_kernels": [{"kernel": k.split("(")[0][-60:], "calls": c, "bytes": fetch_factor * a * 1024 + write_factor * b * 1024}
                                            for k, (c, a, b) in top]}}


def main():
    fdir, wdir, tag = sys.argv[1:4]
    out = compute(fdir, wdir, tag, tag + "_pmc_summary.csv")
    json.dump(out, open(os.path.join(os.path.dirname(tag) or ".", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
