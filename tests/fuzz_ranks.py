#!/usr/bin/env python3
"""Random row partitions of random irregular matrices on 2-8 thread ranks against the oracle (test seam hypredrive_amd/_lib.py
run_thread_ranks; checker oracle/): uneven blocks, blocks of a handful of rows, EMPTY blocks, every level partitioned or a replicated
tail at a random depth, host-staged or asynchronous device transport, HDA_DIST_CHECK on a third of the cases (the partitioned setup
compared level by level with the replicated one inside the library).  usage: tests/fuzz_ranks.py <cases> [first seed]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from dist_worker import random_mmatrix  # noqa: E402
from hypredrive_amd import _lib  # noqa: E402
from hypredrive_amd import hypredrv as hd  # noqa: E402
from oracle import oracle_ffi as orc  # noqa: E402


def solve_ranks(cuts, S, b, yaml):
    def body(rank, world):
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        blk = S[lo:hi]
        h = hd.Hypredrv(yaml)
        try:
            h.set_matrix_csr(lo, hi - 1, blk.indptr, blk.indices, blk.data)
            h.set_rhs_array(lo, hi - 1, b[lo:hi])
            h.finish_system()
            r = h.solve()
            x = np.array(h.solution(), copy=True) if hi > lo else np.zeros(0)
            return r, x
        finally:
            h.close()

    outs = _lib.run_thread_ranks(len(cuts) - 1, body)
    assert len({o[0]["iters"] for o in outs}) == 1, [o[0]["iters"] for o in outs]
    return outs[0][0], np.concatenate([o[1] for o in outs])


# (yaml, oracle Krylov, oracle AMG parameters or None when the multi-rank preconditioner is not the one-rank one, iteration slack)
VARIANTS = [
    ("solver: pcg\npreconditioner: amg\n", "pcg", {}, 1),
    ("solver: pcg\npreconditioner: amg\n", "pcg", {}, 1),
    ("solver: gmres\npreconditioner: amg\n", "gmres", {}, 1),
    ("solver: pcg\npreconditioner:\n  amg:\n    relaxation:\n      down_type: 16\n      up_type: 16\n", "pcg", dict(relax_down=16, relax_up=16), 1),
    ("solver: pcg\npreconditioner:\n  amg:\n    aggressive:\n      num_levels: 1\n", "pcg", dict(agg_num_levels=1), 1),
    ("solver: pcg\npreconditioner:\n  amg:\n    relaxation:\n      down_type: 13\n      up_type: 14\n", "pcg", None, 4),   # hybrid GS: block Jacobi by rank
    ("solver: gmres\npreconditioner:\n  amg:\n    smoother:\n      type: ilu\n      num_levels: 1\n      ilu:\n        type: bj-iluk\n        tri_solve: 0\n", "gmres", None, 4),
    ("solver: pcg\npreconditioner:\n  amg:\n    interpolation:\n      max_nnz_row: 2\n    coarsening:\n      strong_th: 0.5\n", "pcg", dict(pmax=2, strong_th=0.5), 1),
]


def main():
    cases = int(sys.argv[1])
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for c in range(cases):
        rng = np.random.default_rng(1000 + seed0 + c)
        n = int(rng.integers(400, 7000))
        ranks = int(rng.integers(2, 9))
        S = random_mmatrix(int(rng.integers(0, 1 << 30)), n)
        inner = np.sort(rng.integers(0, n + 1, ranks - 1))
        if rng.uniform() < 0.3 and ranks > 2:  # an empty block
            inner[int(rng.integers(1, ranks - 1))] = inner[0]
            inner = np.sort(inner)
        cuts = np.concatenate([[0], inner, [n]])
        rep = int(rng.choice([0, 0, 300, 2000, 100000]))
        transport = str(rng.choice(["host", "device"]))
        check = "1" if rng.uniform() < 0.33 else "0"
        os.environ.update(HDA_REPLICATE_ROWS=str(rep), HDA_THREAD_TRANSPORT=transport, HDA_DIST_CHECK=check)
        b = rng.standard_normal(n)
        vi = int(rng.integers(0, len(VARIANTS)))
        yaml, kry, okw, slack = VARIANTS[vi]
        tag = dict(case=c, n=n, ranks=ranks, cuts=[int(v) for v in cuts], rep=rep, transport=transport, check=check, variant=vi)
        try:
            res, x = solve_ranks(cuts, S, b, yaml)
            Ao = orc.Csr.from_scipy(S)
            if okw is None:  # the one-rank default hierarchy as a yardstick for the iteration count
                ao = orc.Amg(Ao, orc.amg_params(True))
            else:
                ao = orc.Amg(Ao, orc.amg_params(True, **okw))
            ref = (orc.gmres if kry == "gmres" else orc.pcg)(Ao, b, ao, orc.krylov_params(kry == "gmres"))
            err = float(np.linalg.norm(x - ref["x"]) / np.linalg.norm(ref["x"]))
            true_res = float(np.linalg.norm(b - S @ x) / np.linalg.norm(b))
            ok = bool(res["converged"]) and true_res < 3e-6
            if okw is not None:
                ok = ok and abs(res["iters"] - ref["iters"]) <= slack and err < 1e-5
            else:
                ok = ok and res["iters"] <= ref["iters"] + slack
            tag.update(iters=res["iters"], ref_iters=ref["iters"], err=err, true_res=true_res, ok=ok)
        except Exception as e:  # noqa: BLE001
            tag.update(ok=False, error=repr(e)[:500])
            hd.lib().HYPREDRV_ErrorCodeClear()
        bad += 0 if tag["ok"] else 1
        print(json.dumps(tag), flush=True)
    print(json.dumps(dict(cases=cases, failed=bad)), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
