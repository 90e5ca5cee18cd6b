#!/usr/bin/env python3
"""Exact ILU(0) substitutions (tri_solve 1, the reference's CPU-build default, /root/reference/src/internal/ilu.c:15-28) on the
benchmark's Laplacian through the HYPREDRV_* API: one block (level-scheduled launches) against V row blocks (bj-iluk at np = V on
the block Gauss-Seidel kernels).  Prints one JSON line per run.

    python tools/ilu_blocks.py --grid 128 --blocks 1
    python tools/ilu_blocks.py --grid 128            # the setup's own choice of V
    python tools/ilu_blocks.py --grid 128 --smoother # BoomerAMG (PMIS, l1-Jacobi) with the ILU as level-0 complex smoother
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

YAML_ILU = "solver: pcg\npreconditioner:\n  ilu:\n    type: bj-iluk\n    tri_solve: 1\n"
YAML_SMOOTHER = ("solver: pcg\npreconditioner:\n  amg:\n    smoother:\n      type: ilu\n      num_levels: 1\n      ilu:\n        type: bj-iluk\n"
                 "        tri_solve: 1\n")


def run(n, steps, warmup, blocks, smoother, tri_solve):
    if blocks is not None:
        os.environ["HDA_BLOCKS"] = str(blocks)
    import hypredrive_amd as hh
    from hypredrive_amd import hypredrv as hd
    yaml = (YAML_SMOOTHER if smoother else YAML_ILU).replace("tri_solve: 1", f"tri_solve: {tri_solve}")
    h = hd.Hypredrv(yaml)
    h.set_laplacian7((n, n, n))
    hh.sync()
    t0 = time.perf_counter()
    h.create_and_setup()
    hh.sync()
    setup_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(warmup):
        h.apply()
    hh.sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = h.apply()
    hh.sync()
    ms = (time.perf_counter() - t0) * 1e3 / max(steps, 1)
    res = {"what": "PCG + " + ("BoomerAMG with ILU(0) level-0 smoother" if smoother else "ILU(0)") + f", tri_solve {tri_solve}", "grid": n,
           "HDA_BLOCKS": os.environ.get("HDA_BLOCKS", "unset (auto)"), "ms_per_step": ms, "iters": last["iters"], "converged": last["converged"],
           "final_rel": last["final_rel"], "ms_per_iteration": ms / max(last["iters"], 1), "setup_ms": setup_ms}
    h.destroy_solver()
    h.close()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=None)
    ap.add_argument("--smoother", action="store_true")
    ap.add_argument("--tri-solve", type=int, default=1)
    a = ap.parse_args()
    print(json.dumps(run(a.grid, a.steps, a.warmup, a.blocks, a.smoother, a.tri_solve)), flush=True)
