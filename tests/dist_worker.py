"""Worker for multi-rank tests (launched with torch.distributed.run).

mode "transport": CPU only -- exercises the gloo-backed staged transport callbacks with the
                  message pattern of a row-partitioned SpMV (oracle as the local kernel).
mode "solve":     GPU -- AMG-PCG through the HYPREDRV_* API on a row-partitioned Laplacian,
                  several ranks sharing the visible GPU(s) through the staged transport.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mode_transport(out):
    import ctypes as C
    import torch
    import torch.distributed as dist
    from hypredrive_amd import dist as hdist
    from oracle import oracle_ffi as o
    rank, world, _ = hdist.env_rank()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ar, a2a = hdist._make_callbacks(dist, torch, rank, world)
    # all-reduce: f64 sum and i64 max
    v = np.array([rank + 1.0, 2.0 * rank], dtype=np.float64)
    assert ar(v.ctypes.data_as(C.c_void_p), 2, 0, 0) == 0
    assert np.allclose(v, [world * (world + 1) / 2, world * (world - 1)])
    w = np.array([rank, 7 - rank], dtype=np.int64)
    assert ar(w.ctypes.data_as(C.c_void_p), 2, 1, 1) == 0
    assert list(w) == [world - 1, 7]
    # row-partitioned SpMV on the generator's block numbering.  The PLAN comes from the library itself: its communicator is
    # joined with these callbacks (no device needed) and hda_halo_plan_host runs the partition code of the GPU path -- owner
    # lookup, count exchange, request lists -- collectively over gloo; only the local product is scipy's.
    from hypredrive_amd import hypredrv as hd
    import hypredrive_amd as hh
    hdist._keep["cbs"] = (ar, a2a)
    hd.check(hd.lib().HYPREDRV_AMD_CommInitCallbacks(rank, world, -1, ar, a2a))
    n, P = 6, (world, 1, 1)
    A, _ = o.lap7(n, n, n, P=P)
    S = A.to_scipy().tocsr()
    N = n ** 3
    part = np.array([o.lap7_partition(n, n, n, P, r)[0] for r in range(world)] + [N], dtype=np.int64)
    lo, hi = int(part[rank]), int(part[rank + 1])
    loc = S[lo:hi]
    cols = np.unique(loc.indices)
    ghosts = np.ascontiguousarray(cols[(cols < lo) | (cols >= hi)], dtype=np.int64)
    sc, rc = np.zeros(world, dtype=np.int32), np.zeros(world, dtype=np.int32)
    idx = np.zeros(max(hi - lo, 1), dtype=np.int32)
    tot = C.c_int()
    L = hh.load()
    ll, ip = C.POINTER(C.c_longlong), C.POINTER(C.c_int)
    rcode = L.hda_halo_plan_host(hi - lo, part.ctypes.data_as(ll), ghosts.ctypes.data_as(ll), len(ghosts), sc.ctypes.data_as(ip),
                                 rc.ctypes.data_as(ip), idx.ctypes.data_as(ip), len(idx), C.byref(tot))
    assert rcode == 0, L.hda_last_error()
    # the plan against an independent construction: ghosts grouped by owner, requests inside the owned range
    owner = np.searchsorted(part, ghosts, side="right") - 1
    assert list(rc) == [int((owner == p).sum()) for p in range(world)] and rc[rank] == 0
    assert tot.value == int(sc.sum()) and np.all((idx[:tot.value] >= 0) & (idx[:tot.value] < hi - lo))
    # the halo exchange with the library's plan: pack by send_idx, one alltoallv, ghosts land in ascending global id
    x = np.sin(np.arange(N, dtype=np.float64))
    sendv = np.ascontiguousarray(x[lo + idx[:tot.value]]) if tot.value else np.zeros(1)
    recvv = np.zeros(max(len(ghosts), 1))
    sb = (C.c_long * world)(*[8 * int(c) for c in sc])
    rb = (C.c_long * world)(*[8 * int(c) for c in rc])
    assert a2a(sendv.ctypes.data_as(C.c_void_p), sb, recvv.ctypes.data_as(C.c_void_p), rb) == 0
    xg = np.zeros(N)
    xg[lo:hi] = x[lo:hi]
    xg[ghosts] = recvv[:len(ghosts)]
    y = loc @ xg
    assert np.allclose(y, (S @ x)[lo:hi], rtol=1e-14)
    hd.lib().HYPREDRV_AMD_CommFinalize()
    dist.barrier()
    if rank == 0:
        json.dump({"ok": True, "world": world}, open(out, "w"))
    dist.destroy_process_group()


def mode_solve(out, n, solver):
    from hypredrive_amd import dist as hdist
    from hypredrive_amd import hypredrv as hd
    rank, world = hdist.init("staged")
    import hypredrive_amd as h
    assert h.load().hda_comm_selftest() == 0, h.load().hda_last_error()
    P = tuple(int(v) for v in os.environ["HDA_TEST_P"].split(",")) if os.environ.get("HDA_TEST_P") else hdist.factor3(world)
    yaml = f"solver: {solver}\npreconditioner:\n  preset: poisson\n"
    h = hd.Hypredrv(yaml)
    h.set_laplacian7((n, n, n), P)
    # rank-to-rank operations of the solve alone (Setup's are not counted): Create + Setup, reset the counters, Apply
    hd.check(hd.lib().HYPREDRV_LinearSystemResetInitialGuess(h.h))
    h.create_and_setup()
    from hypredrive_amd import _lib
    _lib.comm_stats(reset=True)
    r = h.apply(reset=False)
    cs = _lib.comm_stats()
    vcycles = _lib.load().hda_last_precond_calls()
    h.destroy_solver()
    x = h.solution()
    nrm = h.solution_norm("L2")
    l1 = h.solution_norm("L1")
    linf = h.solution_norm("Linf")
    np.save(f"{out}.x{rank}.npy", x)
    if rank == 0:
        json.dump({"iters": r["iters"], "converged": r["converged"], "final_rel": r["final_rel"], "norm": nrm,
                   "l1": l1, "linf": linf, "world": world, "P": P, "comm": cs, "vcycles": vcycles}, open(out, "w"))
    h.close()
    hdist.finalize()


def random_mmatrix(seed, n):
    """Irregular symmetric M-matrix, identical on every rank (seeded)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    m = 6 * n
    i, j = rng.integers(0, n, m), rng.integers(0, n, m)
    # mostly local couplings plus a few long-range ones: ghost layers of very different sizes
    j = np.where(rng.uniform(size=m) < 0.9, np.clip(i + rng.integers(-40, 41, m), 0, n - 1), j)
    keep = i != j
    i, j = i[keep], j[keep]
    w = rng.uniform(0.05, 1.0, i.size)
    W = sp.coo_matrix((w, (i, j)), shape=(n, n)).tocsr()
    W = W + W.T
    d = np.asarray(W.sum(axis=1)).ravel() + rng.uniform(0.01, 0.2, n)
    A = (sp.diags(d) - W).tocsr()
    A.sort_indices()
    return A


def mode_csr(out, n, seed):
    """Row blocks of an irregular matrix through HYPREDRV_LinearSystemSetMatrixFromCSR on every rank
    (reference tests/test_setmatrix_from_csr_mpi.c), AMG-PCG on the row-partitioned system."""
    from hypredrive_amd import dist as hdist
    from hypredrive_amd import hypredrv as hd
    rank, world = hdist.init("staged")
    A = random_mmatrix(seed, n)
    lo, hi = rank * n // world, (rank + 1) * n // world
    blk = A[lo:hi]
    h = hd.Hypredrv(os.environ.get("HDA_TEST_YAML", "solver: pcg\npreconditioner: amg\n"))
    h.set_matrix_csr(lo, hi - 1, blk.indptr, blk.indices, blk.data)
    h.set_rhs_array(lo, hi - 1, np.ones(hi - lo))
    h.finish_system()
    if os.environ.get("HDA_TEST_DOFMAP_MOD"):  # label of global row i = i mod k
        import ctypes as C
        lab = np.ascontiguousarray(np.arange(lo, hi) % int(os.environ["HDA_TEST_DOFMAP_MOD"]), dtype=np.int32)
        hd.check(hd.lib().HYPREDRV_LinearSystemSetDofmap(h.h, hi - lo, lab.ctypes.data_as(C.POINTER(C.c_int))))
    r = h.solve()
    nrm = h.solution_norm("L2")
    if rank == 0:
        json.dump({"iters": r["iters"], "converged": r["converged"], "final_rel": r["final_rel"], "norm": nrm, "world": world},
                  open(out, "w"))
    h.close()
    hdist.finalize()


def mode_mgr(out, n):
    """Row blocks of the 3-field model system (rows cut anywhere, also inside a cell) with their slice of the dofmap;
    GMRES + MGR from the YAML in HDA_TEST_YAML."""
    from hypredrive_amd import dist as hdist
    from hypredrive_amd import hypredrv as hd
    rank, world = hdist.init("staged")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_pins import three_field_system
    S, labels = three_field_system(n, seed=4)
    N = S.shape[0]
    lo, hi = rank * N // world, (rank + 1) * N // world
    blk = S[lo:hi]
    h = hd.Hypredrv(os.environ["HDA_TEST_YAML"])
    h.set_matrix_csr(lo, hi - 1, blk.indptr, blk.indices, blk.data)
    h.set_rhs_array(lo, hi - 1, np.ones(hi - lo))
    h.finish_system()
    lab = np.ascontiguousarray(labels[lo:hi], dtype=np.int32)
    import ctypes as C
    hd.check(hd.lib().HYPREDRV_LinearSystemSetDofmap(h.h, hi - lo, lab.ctypes.data_as(C.POINTER(C.c_int))))
    r = h.solve()
    nrm = h.solution_norm("L2")
    if rank == 0:
        json.dump({"iters": r["iters"], "converged": r["converged"], "final_rel": r["final_rel"], "norm": nrm, "world": world}, open(out, "w"))
    h.close()
    hdist.finalize()


def mode_threads(out, n, P, solver, want_x=True):
    """ONE process, prod(P) thread ranks on the visible GPU (hypredrive_amd/csrc/hda_thread_ranks.hip): the public HYPREDRV_* sequence
    on every rank, the in-process staged transport between them."""
    from hypredrive_amd import _lib
    P = tuple(int(v) for v in P.split(","))
    yaml = f"solver: {solver}\npreconditioner:\n  preset: poisson\n"
    r = _lib.thread_ranks_lap7(P[0] * P[1] * P[2], (n, n, n), P, yaml, want_x=want_x)
    if want_x:
        np.save(out + ".x.npy", r.pop("x"))
    json.dump(r, open(out, "w"))


if __name__ == "__main__":
    if sys.argv[1] == "threads":
        mode_threads(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5], want_x=(len(sys.argv) < 7 or sys.argv[6] == "1"))
    elif sys.argv[1] == "transport":
        mode_transport(sys.argv[2])
    elif sys.argv[1] == "mgr":
        mode_mgr(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "csr":
        mode_csr(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        mode_solve(sys.argv[2], int(sys.argv[3]), sys.argv[4])
