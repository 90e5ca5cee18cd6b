"""Which knob moves the oracle's 6 / 5 / 6 / 6 iterations on examples/refOutput/ex8.txt:92-95 toward the reference's 7 / 6 / 6 / 7?
(three of four variants one iteration low, all on the same side.)  Not a test: run by hand,
`python tests/ex8_knobs.py`; the table it prints is recorded in DESIGN.md section 3.  Uses the CPU oracle only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_ffi as orc  # noqa: E402
from test_oracle_pins import EX8_VARIANTS, ex8_system  # noqa: E402

REF = [(7, 4.11e-10), (6, 1.67e-11), (6, 9.03e-10), (7, 3.00e-11)]  # iterations, true relative residual (ex8.txt:92-95)


def run(A, b, S, k, **kw):
    amg = orc.Amg(A, orc.amg_params(False, **kw))
    if k == 3:
        amg.set_ilu_smoother(1, 1)
    r = orc.pcg(A, b, amg, orc.krylov_params(False, rtol=1e-9, max_iter=500))
    tr = np.linalg.norm(b - S @ r["x"]) / np.linalg.norm(b)
    return r["iters"], tr, [amg.level_A(l).nrows for l in range(amg.num_levels)]


def main():
    A, b = ex8_system(orc)
    S = A.to_scipy()
    print("per-iteration rate = (final true relative residual)^(1 / iterations)")
    for k, v in enumerate(EX8_VARIANTS):
        it, rr = REF[k]
        print(f"variant {k}: reference {it} it, {rr:.2e}, rate {rr ** (1 / it):.4f}")
        knobs = [("restated (HMIS = Ruge first pass, CG eigen-estimate)", {})]
        knobs.append(("PMIS grids instead of HMIS", dict(coarsen_type=8)))
        if v.get("relax_down") == 16:
            knobs.append(("Gershgorin eigen-estimate (eig_est 0)", dict(cheby_eig_est=0)))
            knobs.append(("20 CG iterations for the estimate", dict(cheby_eig_est=20)))
        for name, over in knobs:
            it2, rr2, rows = run(A, b, S, k, **{**v, **over})
            print(f"   {name:58s} {it2} it, {rr2:.2e}, rate {rr2 ** (1 / it2):.4f}, rows per level {rows}")
    # the numbering of the np4 part files read on one rank (block numbering of a rank grid) with the first 250 rows as right-hand side
    for P in [(1, 1, 4), (4, 1, 1), (2, 2, 1)]:
        Ab, _ = orc.lap7(10, 10, 10, b_mode=1, P=P)
        Sb = Ab.to_scipy()
        bb = np.r_[np.ones(250), np.zeros(750)]
        print(f"block numbering of a {P} rank grid:", [run(Ab, bb, Sb, k, **v)[0] for k, v in enumerate(EX8_VARIANTS)])


if __name__ == "__main__":
    main()
