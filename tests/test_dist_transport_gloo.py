"""N > 1 on CPU (gloo, world_size 2 and 3): the staged transport the multi-rank GPU tests and single-GPU rehearsals run on.
The library's communicator is joined with the gloo callbacks and the LIBRARY's own halo-plan code (hda_halo_plan_host: the
host half of make_halo_plan, hda_dist.hip) builds the exchange plan collectively; the halo values then travel through the same
callbacks and the row-partitioned SpMV is checked against the oracle's global one."""
import json
import os
import subprocess
import sys

import pytest

from conftest import free_port  # a port nobody listens on: two suites on one host do not collide

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_staged_transport_partitioned_spmv(tmp_path, world):
    out = str(tmp_path / "res.json")
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "transport", out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert json.load(open(out)) == {"ok": True, "world": world}


def test_factor3():
    from hypredrive_amd.dist import factor3
    assert factor3(1) == (1, 1, 1) and factor3(2) == (1, 1, 2) and factor3(4) == (1, 2, 2) and factor3(8) == (2, 2, 2)
    assert factor3(6) == (1, 2, 3)
