"""Random operators of every row length through the AMG setup, device against oracle, bit for bit (run under HDA_GUARD=1 HDA_POISON=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import hypredrive_amd as hd
from oracle import oracle_ffi as orc

bad = 0
seeds = [int(a) for a in sys.argv[2:]] if len(sys.argv) > 2 else range(int(sys.argv[1]) if len(sys.argv) > 1 else 60)
for seed in seeds:
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(40, 3000))
    avg = float(rng.choice([2, 4, 7, 9, 13, 17, 25, 40, 70]))
    M = sp.random(n, n, density=min(1.0, avg / n), random_state=rng, format="csr")
    M = (M + M.T) * 0.5
    sign = -1.0 if seed % 3 else np.where(rng.random(M.nnz) < 0.8, -1.0, 0.4)
    M.data = sign * np.abs(M.data)
    M = ((M + M.T) * 0.5).tolil()
    M.setdiag(np.asarray(abs(M.tocsr()).sum(axis=1)).ravel() * rng.choice([0.55, 1.0, 1.3]) + 0.1)
    if seed % 5 == 0:
        M[int(rng.integers(0, n)), :] = 0  # an empty row
    M = M.tocsr(); M.eliminate_zeros(); M.sort_indices()
    Ao, Ah = orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)
    kw = dict(strong_th=float(rng.choice([0.25, 0.5, 0.7])), pmax=int(rng.choice([0, 2, 4, 6])), interp_type=int(rng.choice([6, 6, 3])))
    r4 = os.environ.get("FUZZ_R4", "1") != "0" and seed % 2 == 1
    if r4:  # round 4: standard and mm-ext+i interpolation, HMIS and hybrid Gauss-Seidel on row blocks (the reference at np = V)
        kw["interp_type"] = int(rng.choice([6, 8, 17, 3]))
        kw["coarsen_type"] = int(rng.choice([8, 10, 10]))
        gs = [(18, 18), (13, 14), (3, 4), (8, 8), (6, 6)][int(rng.integers(0, 5))]
        kw["relax_down"], kw["relax_up"] = gs
        kw["blocks"] = int(rng.choice([1, 1, 3, 7])) if (kw["coarsen_type"] == 10 or gs[0] != 18) else 1
    if os.environ.get("FUZZ_AGG", "1") != "0" and not r4:  # round 3: aggressive levels (second PMIS pass, multipass interpolation, its truncation)
        kw.update(agg_num_levels=int(rng.choice([0, 0, 1, 2])), agg_num_paths=int(rng.choice([1, 1, 2])), agg_pmax=int(rng.choice([0, 0, 3])),
                  agg_trunc_factor=float(rng.choice([0.0, 0.0, 0.2])))
    try:
        ho, hh = orc.Amg(Ao, orc.amg_params(True, **kw)), hd.Amg(Ah, hd.AmgParams.default(**kw))
        if r4 and kw["blocks"] > 1:
            for l in range(min(ho.num_levels, hh.num_levels)):
                if not np.array_equal(hh.level_blocks(l), ho.level_block_part(l)): raise AssertionError("block starts differ on level %d" % l)
        ok = hh.num_levels == ho.num_levels
        why = [] if ok else ["levels %d vs %d" % (hh.num_levels, ho.num_levels)]
        for l in range(min(ho.num_levels, hh.num_levels)):
            rp, cj, v = hh.level_matrix(l, 0).download(); Al = ho.level_A(l)
            if not (np.array_equal(rp, Al.rowptr) and np.array_equal(cj, Al.col)): why.append("A%d pattern" % l)
            elif not np.array_equal(v, Al.val): why.append("A%d values %.3g" % (l, np.abs(v - Al.val).max()))
            if l < min(ho.num_levels, hh.num_levels) - 1:
                if not np.array_equal(hh.level_cf(l), ho.level_cf(l)): why.append("cf%d" % l)
                rp, cj, v = hh.level_matrix(l, 1).download(); Pl = ho.level_P(l)
                if not (np.array_equal(rp, Pl.rowptr) and np.array_equal(cj, Pl.col)): why.append("P%d pattern" % l)
                elif not np.array_equal(v, Pl.val): why.append("P%d values %.3g" % (l, np.abs(v - Pl.val).max()))
        r = np.random.default_rng(seed).standard_normal(n)
        zo, zh = ho.vcycle(r), hh.vcycle(r)
        # the cycle is reported, not judged: an empty row makes it NaN on both sides, and a near-singular coarsest operator (condition
        # 1e12 happens with positive off-diagonals) amplifies the rounding of the two dense coarse solves
        if np.all(np.isfinite(zo)) and not (np.linalg.norm(zh - zo) <= 1e-11 * max(np.linalg.norm(zo), 1e-300)):
            print("seed", seed, "note: V-cycle differs by %.3g of %.3g" % (np.linalg.norm(zh - zo), np.linalg.norm(zo)), flush=True)
        ok = not why
        if why: print("seed", seed, "why:", why[:6], flush=True)
    except Exception as e:  # noqa: BLE001
        ok = False
        print("seed", seed, "raised", repr(e)[:200], flush=True)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "n", n, "avg", avg, kw, flush=True)
print("fuzz done, mismatches:", bad, flush=True)
sys.exit(1 if bad else 0)
