#!/usr/bin/env python3
"""Series B of SURVEY.md 8(d): the reference's CPU-build defaults (HMIS coarsening, hybrid l1 Gauss-Seidel 13 / 14;
/root/reference/src/internal/amg.c:141-146, 182-189) on the benchmark's Laplacian through the HYPREDRV_* API, on V row blocks
(HDA_BLOCKS; unset = the setup's own choice).  Prints one JSON line per run.

    python tools/series_b.py --grid 128 [--blocks 32] [--steps 5] [--oracle]
    python tools/series_b.py --grid 256 --rank-grid 8      # the system numbered as the reference's generator numbers it at np = 8^3
                                                           # (-P 8 8 8): one row block per rank = a 32^3 sub-cube
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


def lap7_rank_blocks(n, p):
    """The n^3 7-point Laplacian in the numbering the reference's generator gives it at np = p^3 (`-P p p p`,
    /root/reference/examples/src/C_laplacian/laplacian.c:504-520 grid2idx): rank blocks in Cartesian rank order (z fastest), x fastest
    inside a block -- so rows [q (n/p)^3, (q + 1)(n/p)^3) ARE rank q's sub-cube.  rhs as the generator's (1 on the y = 0 face,
    laplacian.c:898-905).  Returns indptr, cols, vals, b (int64 / float64)."""
    import numpy as np
    nl = n // p
    assert nl * p == n, "the grid must divide evenly into the rank grid"
    r = np.arange(n ** 3, dtype=np.int64)
    q, rem = r // nl ** 3, r % nl ** 3
    g = [(q // (p * p)) * nl + rem % nl, ((q // p) % p) * nl + (rem // nl) % nl, (q % p) * nl + rem // (nl * nl)]  # gx, gy, gz
    del q, rem

    def rid(gx, gy, gz):
        return (((gx // nl) * p + gy // nl) * p + gz // nl) * nl ** 3 + ((gz % nl) * nl + gy % nl) * nl + gx % nl

    cols = np.empty((n ** 3, 7), dtype=np.int64)
    ok = np.ones((n ** 3, 7), dtype=bool)
    cols[:, 0] = r
    k = 1
    for d in range(3):
        for s in (-1, 1):
            gg = list(g)
            gg[d] = g[d] + s
            ok[:, k] = (gg[d] >= 0) & (gg[d] < n)
            gg[d] = np.clip(gg[d], 0, n - 1)
            cols[:, k] = rid(*gg)
            k += 1
    vals = np.where(np.arange(7) == 0, 6.0, -1.0) * np.ones((n ** 3, 1))
    indptr = np.concatenate(([0], np.cumsum(ok.sum(axis=1)))).astype(np.int64)
    b = (g[1] == 0).astype(np.float64)
    return indptr, cols[ok], vals[ok], b


def run(n, steps=5, warmup=1, blocks=None, oracle=False, yaml=YAML_CPU_DEFAULTS, rank_grid=None):
    if rank_grid:  # the reference at np = p^3 with `-P p p p`: its numbering, one row block per rank
        blocks = rank_grid ** 3
    if blocks is not None:
        os.environ["HDA_BLOCKS"] = str(blocks)
    import hypredrive_amd as hh
    from hypredrive_amd import hypredrv as hd
    h = hd.Hypredrv(yaml)
    system = None
    if rank_grid:
        system = lap7_rank_blocks(n, rank_grid)
        h.set_matrix_csr(0, n ** 3 - 1, system[0], system[1], system[2])
        h.set_rhs_array(0, n ** 3 - 1, system[3])
        h.finish_system()
    else:
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
    if rank_grid:
        res["numbering"] = f"rank blocks of -P {rank_grid} {rank_grid} {rank_grid} (laplacian.c grid2idx): block q = rank q's {n // rank_grid}^3 sub-cube"
    del A, amg
    h.destroy_solver()
    h.close()
    if oracle:
        from oracle import oracle_ffi as orc
        if system is not None:
            Ao, b = orc.Csr.from_arrays(n ** 3, n ** 3, system[0], system[1], system[2]), system[3]
        else:
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
    ap.add_argument("--rank-grid", type=int, default=None, help="p: the system in the reference's numbering at np = p^3 (-P p p p), one row block per rank")
    a = ap.parse_args()
    print(json.dumps(run(a.grid, a.steps, a.warmup, a.blocks, a.oracle, rank_grid=a.rank_grid)), flush=True)
