#!/usr/bin/env python3
"""Eight (or N) thread ranks sharing ONE MI355X over the device-direct thread transport (hda_comm.hip DeviceThreadComm: exchanges and
all-reduces are only enqueued, like RCCL): what partitioning costs when the ranks' kernels time-share one card and no host staging is in
the way.  Prints one JSON line per layout: ms per solve (max over ranks), iterations, collectives per iteration; and the one-rank run of
the same global grid.  usage: gpurun_thread_timing.py <global n> <P0,P1,P2> [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from hypredrive_amd import _lib  # noqa: E402
from hypredrive_amd import hypredrv as hd  # noqa: E402

YAML = "solver: pcg\npreconditioner:\n  preset: poisson\n"


def solve_loop(n, P, steps):
    def body(rank, world):
        h = hd.Hypredrv(YAML)
        try:
            h.set_laplacian7((n, n, n), P)
            h.create_and_setup()
            h.apply()
            _lib.sync()
            _lib.comm_stats(reset=True)
            t0 = time.perf_counter()
            for _ in range(steps):
                last = h.apply()
            _lib.sync()
            ms = (time.perf_counter() - t0) * 1e3 / steps
            cs = _lib.comm_stats()
            part = _lib.load().hda_amd_partitioned_levels(h.h)
            h.destroy_solver()
            return dict(ms=ms, iters=last["iters"], allreduce=cs["allreduce"] / steps / max(last["iters"], 1),
                        exchange=cs["exchange"] / steps / max(last["iters"], 1), overlapped=cs["overlapped"] / steps / max(last["iters"], 1), part=part)
        finally:
            h.close()

    nr = P[0] * P[1] * P[2]
    if nr == 1:
        return [body(0, 1)]
    return _lib.run_thread_ranks(nr, body)


def main():
    n = int(sys.argv[1])
    P = tuple(int(v) for v in sys.argv[2].split(","))
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    outs = solve_loop(n, P, steps)
    print(json.dumps(dict(n=n, P=P, transport=os.environ.get("HDA_THREAD_TRANSPORT", "host"), overlap=os.environ.get("HDA_OVERLAP", "default"),
                          ms_per_solve=max(o["ms"] for o in outs), iters=outs[0]["iters"], allreduces_per_iter=outs[0]["allreduce"],
                          exchanges_per_iter=outs[0]["exchange"], overlapped_per_iter=outs[0]["overlapped"], partitioned_levels=outs[0]["part"])), flush=True)


if __name__ == "__main__":
    main()
