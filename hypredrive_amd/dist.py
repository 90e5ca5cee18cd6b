"""One process per GPU: join the library's communicator from a torchrun-style environment.

Transport choice:
  * "rccl"   -- every rank has its own GPU (LOCAL_RANK < device count): the C library talks
                RCCL over xGMI directly (halo ncclSend/ncclRecv, fused dot ncclAllReduce);
                torch.distributed (gloo) is only used to hand rank 0's ncclUniqueId around.
  * "staged" -- several ranks share one GPU (tests) or no RCCL: the same messages are
                staged through host callbacks implemented on torch.distributed/gloo.
"""
import ctypes as C
import os
import sys

import numpy as np

_keep = {}


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def _make_callbacks(dist, torch, rank, world):
    from . import hypredrv as hd

    def guarded(fn):
        # ctypes prints and swallows an exception raised inside a callback; the C side would then carry on with
        # unfilled buffers.  Record it and hand the library a non-zero code instead: it raises hda::Error there.
        def wrapper(*a):
            try:
                fn(*a)
                return 0
            except BaseException as e:  # noqa: BLE001 - anything here must reach the C caller as a failure
                _keep["callback_error"] = e
                print(f"[hypredrive_amd] rank {rank}: staged-transport callback failed: {e!r}", file=sys.stderr, flush=True)
                return 1
        return wrapper

    @guarded
    def allreduce(buf, count, dtype, op):
        ct = C.c_double if dtype == 0 else C.c_int64
        arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(ct)), shape=(count,))
        t = torch.from_numpy(arr)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)

    @guarded
    def alltoallv(send, sbytes, recv, rbytes):
        sb = [sbytes[p] for p in range(world)]
        rb = [rbytes[p] for p in range(world)]
        st, rt = sum(sb), sum(rb)
        s = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(max(st, 1),))
        r = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(max(rt, 1),))
        so = np.concatenate([[0], np.cumsum(sb)]).astype(np.int64)
        ro = np.concatenate([[0], np.cumsum(rb)]).astype(np.int64)
        reqs, bufs = [], []
        for p in range(world):
            if p == rank:
                n = min(sb[p], rb[p])
                if n:
                    r[ro[p]:ro[p] + n] = s[so[p]:so[p] + n]
                continue
            if rb[p]:
                t = torch.empty(rb[p], dtype=torch.uint8)
                bufs.append((p, t))
                reqs.append(dist.irecv(t, src=p))
        for p in range(world):
            if p != rank and sb[p]:
                t = torch.from_numpy(s[so[p]:so[p] + sb[p]].copy())
                reqs.append(dist.isend(t, dst=p))
        for q in reqs:
            q.wait()
        for p, t in bufs:
            r[ro[p]:ro[p] + rb[p]] = t.numpy()

    return hd.ALLREDUCE_CB(allreduce), hd.ALLTOALLV_CB(alltoallv)


class TransportError(RuntimeError):
    """RCCL could not be joined and falling back was not allowed.  stage: uid | init | halo_comm | selftest | devices | peer
    (peer: this rank was fine, another one failed).  nccl_debug: what RCCL itself logged on this rank (NCCL_DEBUG=WARN)."""

    def __init__(self, stage, rank, msg, nccl_debug=None):
        super().__init__(msg)
        self.stage, self.rank, self.nccl_debug = stage, rank, nccl_debug


def _rccl_log_path():
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"hda_rccl_{os.getpid()}.log")


def rccl_debug_tail(limit=1500):
    """the tail of this process's RCCL log (NCCL_DEBUG_FILE, set by init() unless the caller chose one); '' when there is none"""
    path = _keep.get("nccl_debug_file")
    try:
        with open(path, errors="replace") as f:
            return f.read()[-limit:]
    except (OSError, TypeError):
        return ""


def init(transport="auto", strict=False):
    """Join the world described by RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*; returns (rank, world).  strict: an RCCL failure raises
    TransportError on every rank instead of falling back to the host-staged transport."""
    from . import hypredrv as hd
    rank, world, local = env_rank()
    if world == 1:
        return rank, world
    # RCCL / device-memory sharing across processes needs dmabuf IPC on this driver, whoever launched the ranks; it is read when the
    # HIP runtime starts, so it has to be in the environment before the first GPU call of the process (nothing here has made one)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # a first contact that fails should say why: RCCL's own warnings go to a per-process file whose tail joins the error
    if "NCCL_DEBUG" not in os.environ:
        os.environ["NCCL_DEBUG"] = "WARN"
        os.environ.setdefault("NCCL_DEBUG_FILE", _rccl_log_path())
    _keep["nccl_debug_file"] = os.environ.get("NCCL_DEBUG_FILE")
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        # a one-node world that meets on the loopback address needs no other interface: gloo otherwise looks its interface up through the
        # host NAME, which containers do not always resolve
        if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ndev = torch.cuda.device_count()
    forced = os.environ.get("HDA_TRANSPORT")  # "rccl" | "staged": overrides the choice below
    if forced:
        transport = forced
    if transport == "auto":
        transport = "rccl" if ndev >= 1 else "staged"  # (decided for good below: one PHYSICAL GPU per rank, or the staged transport)
    L = hd.lib()
    if transport == "rccl":
        # RCCL refuses two ranks on one device, and the refusing rank leaves its peers blocked inside
        # ncclCommInitRank: find that out BEFORE touching RCCL.  The identity of a device is its PCI bus id, not its index: a launcher
        # that narrows every rank's view to its own GPU (HIP_VISIBLE_DEVICES per rank) makes every index 0.
        mine = local if local < max(ndev, 1) else 0
        ident = mine
        buf = C.create_string_buffer(64)
        if L.hda_device_pci_bus_id(mine, buf, 64) == 0:
            ident = buf.value.decode(errors="replace")
        devs = [None] * world
        dist.all_gather_object(devs, (os.uname().nodename, ident))
        if len(set(devs)) < world:
            if forced == "rccl":
                raise TransportError("devices", rank, f"RCCL needs one GPU per rank; ranks share devices: {devs}")
            if rank == 0:
                print("[hypredrive_amd] several ranks share a GPU: using the host-staged transport", file=sys.stderr, flush=True)
            transport = "staged"
    if transport == "rccl":
        # join RCCL and prove the communicator works (all-reduce + neighbour exchange self-test);
        # if ANY rank fails, every rank falls back to the host-staged transport: slower
        # messages, same kernels and results.  Set HDA_TRANSPORT=rccl (or strict=True: bench.py) to make this fatal.
        # `stage` names where it went wrong -- uid (rank 0's ncclGetUniqueId), init (ncclCommInitRank of the main communicator),
        # halo_comm (the second communicator of the neighbour exchanges), selftest (first collectives on both) -- for the one-line
        # diagnosis a first multi-GPU run needs
        err, stage = None, None
        _keep["stage"] = "uid"
        # rank 0's id (or its failure) reaches every rank in ONE collective, whatever happened on rank 0:
        # a rank that skipped the broadcast would leave the others waiting in it
        uid, box = (C.c_ubyte * 128)(), [None]
        if rank == 0:
            try:
                hd.check(L.HYPREDRV_AMD_CommGetUniqueId(uid))
                box = [bytes(uid)]
            except Exception as e:  # noqa: BLE001
                err, stage, box = e, "uid", [None]
        dist.broadcast_object_list(box, src=0)
        try:
            if box[0] is None:
                stage = "uid"
                raise err or RuntimeError("rank 0 could not create an RCCL unique id")
            uid = (C.c_ubyte * 128).from_buffer_copy(box[0])
            stage = _keep["stage"] = "init"  # (_keep["stage"]: what a watchdog reports when a call never returns)
            try:
                hd.check(L.HYPREDRV_AMD_CommInit(rank, world, mine, uid))  # (mine: LOCAL_RANK, or 0 when the launcher shows this rank one device)
            except Exception as e:  # noqa: BLE001 - the library names the communicator that failed
                if "halo_comm" in str(e):
                    stage = "halo_comm"
                raise
            from . import load
            stage = _keep["stage"] = "selftest"
            if load().hda_comm_selftest() != 0:
                raise RuntimeError("RCCL self-test failed: " + load().hda_last_error().decode())
            stage = None
            _keep["stage"] = "agree"
        except Exception as e:  # noqa: BLE001 - agreement below decides what happens
            err = e
        ok = torch.tensor([0 if err else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0:
            if forced == "rccl" or strict:
                raise TransportError(stage or "peer", rank, f"RCCL transport unavailable on some rank (this rank: {err})", rccl_debug_tail())
            if rank == 0:
                print(f"[hypredrive_amd] RCCL transport unavailable ({err}); using the host-staged transport", file=sys.stderr, flush=True)
            L.HYPREDRV_AMD_CommFinalize()
            transport = "staged"
    if transport == "staged":
        ar, a2a = _make_callbacks(dist, torch, rank, world)
        _keep["cbs"] = (ar, a2a)  # keep the ctypes trampolines alive
        hd.check(L.HYPREDRV_AMD_CommInitCallbacks(rank, world, local if local < max(ndev, 1) else 0, ar, a2a))
    _keep["transport"] = transport
    _keep["stage"] = "joined"
    return rank, world


def transport():
    return _keep.get("transport", "self")


def finalize():
    from . import hypredrv as hd
    hd.lib().HYPREDRV_AMD_CommFinalize()
    _keep.clear()
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
    except Exception:
        pass


def factor3(p):
    """P0 x P1 x P2 = p, as cubic as possible, z (the slowest block index) largest."""
    best = (1, 1, p)
    for a in range(1, p + 1):
        if p % a:
            continue
        for b in range(a, p // a + 1):
            if (p // a) % b:
                continue
            c = p // a // b
            if c >= b and (c - a) < (best[2] - best[0]):
                best = (a, b, c)
    return best
