"""Transport fallback rehearsal: two ranks on ONE GPU ask for RCCL (which refuses duplicate GPUs);
every rank must agree to fall back to the host-staged transport and the solve must succeed."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
from hypredrive_amd import dist as hdist
from hypredrive_amd import hypredrv as hd
rank, world = hdist.init("rccl")
h = hd.Hypredrv("solver: pcg\npreconditioner: amg\n")
h.set_laplacian7((16, 16, 16), hdist.factor3(world))
r = h.solve()
if rank == 0:
    print("transport:", hdist.transport(), "iters:", r["iters"], "converged:", r["converged"], flush=True)
h.close()
hdist.finalize()
