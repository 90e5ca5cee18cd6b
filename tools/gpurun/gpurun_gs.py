"""hybrid l1 Gauss-Seidel (the reference's CPU defaults: HMIS + relax 13/14) at n^3: solve time and iterations."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hypredrive_amd as h
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
coarsen = int(sys.argv[2]) if len(sys.argv) > 2 else 8
A = h.lap7(n, n, n, want_rhs=False)
p = h.AmgParams.default(coarsen_type=coarsen, relax_down=13, relax_up=14)
t0 = time.perf_counter(); amg = h.Amg(A, p); h.sync(); t1 = time.perf_counter()
kp = h.KrylovParams.default(False)
h.solve_device(A, amg, kp, nsolves=1, profile_k1=False)
r = h.solve_device(A, amg, kp, nsolves=3, profile_k1=False)
print(f"hl1GS {n}^3 coarsen {coarsen}: setup {t1 - t0:.2f} s, iters {r['iters']}, solve ms {[round(float(x), 2) for x in r['solve_ms']]}, levels {amg.num_levels}", flush=True)
