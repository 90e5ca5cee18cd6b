#!/usr/bin/env python3
"""Series B of SURVEY.md 8(d): the reference's CPU-build defaults (HMIS coarsening, hybrid l1 Gauss-Seidel 13 / 14;
/root/reference/src/internal/amg.c:141-146, 182-189) on the benchmark's Laplacian through the HYPREDRV_* API, on V row blocks
(HDA_BLOCKS; unset = the setup's own choice).  Prints one JSON line per run.

    python tools/series_b.py --grid 128 [--blocks 32] [--steps 5] [--oracle]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

YAML_CPU_DEFAULTS = ("solver: pcg\npreconditioner:\n  amg:\n    coarsening:\n      type: hmis\n    relaxation:\n"
                     "      down_type: forward-hl1gs\n      up_type: backward-hl1gs\n      coarse_type: ge\n")


def run(n, steps=5, warmup=1, blocks=None, oracle=False, yaml=YAML_CPU_DEFAULTS):
    if blocks is not None:
        os.environ["HDA_BLOCKS"] = str(blocks)
    import hypredrive_amd as hh
    from hypredrive_amd import hypredrv as hd
    h = hd.Hypredrv(yaml)
    h.set_laplacian7((n, n, n))
    ts = []
    for rep in range(2):
        hh.sync()
        t0 = time.perf_counter()
        h.create_and_setup()
        hh.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
        if rep == 0:
            h.destroy_solver()
    A, amg = hh._lib.borrow(h)
    g, o = amg.complexities
    V = amg.blocks
    for _ in range(warmup):
        h.apply()
    hh.sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = h.apply()
    hh.sync()
    ms = (time.perf_counter() - t0) * 1e3 / max(steps, 1)
    res = {"what": "same system, same API path, the reference's CPU-build defaults (HMIS, hybrid l1 Gauss-Seidel 13 / 14) on V row blocks "
                   "= the reference at np = V", "grid": n, "V": V, "rows_per_block": n ** 3 // max(V, 1),
           "ms_per_step": ms, "value": n ** 3 / (ms * 1e-3), "iters": last["iters"], "converged": last["converged"], "final_rel": last["final_rel"],
           "setup_ms": ts[1], "setup_cold_ms": ts[0], "operator_complexity": o, "grid_complexity": g, "num_levels": amg.num_levels}
    del A, amg
    h.destroy_solver()
    h.close()
    if oracle:
        from oracle import oracle_ffi as orc
        Ao, b = orc.lap7(n, n, n)
        t0 = time.perf_counter()
        ho = orc.Amg(Ao, orc.amg_params(False, blocks=V))
        t1 = time.perf_counter()
        ro = orc.pcg(Ao, b, ho)
        res["oracle"] = {"iters": ro["iters"], "final_rel": ro["final_rel"], "setup_s": t1 - t0, "solve_s": time.perf_counter() - t1,
                         "num_levels": ho.num_levels, "operator_complexity": ho.operator_complexity}
        res["iters_match"] = ro["iters"] == res["iters"]
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=None)
    ap.add_argument("--oracle", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.grid, a.steps, a.warmup, a.blocks, a.oracle)), flush=True)
