"""Row-block sharding glue (one process per GPU).

The sharded algorithm itself lives in the C++ drivers (include/RandLAPACK_amd/*.hh): every reduction over the
row index is followed by `Queue::allreduce_sum`, which is a no-op for one rank.  This module only performs the
rendezvous: rank 0 creates the RCCL unique id, torch.distributed broadcasts the 128 bytes, every rank joins
the communicator bound to its rlhip context.  After that the data-path collectives are issued by librlhip.so
itself on the context's HIP stream (RCCL over xGMI) -- torch.distributed is not in the data path.

(A numpy statement of the same exchange pattern lives in tests/_sharded_model.py: test scaffolding, not product code.)
"""
from __future__ import annotations

import ctypes as C

from . import _lib


_keepalive = {}


def init_comm(ctx, dist, force_hook: bool = False) -> str:
    """Join this rank's rlhip context to an RCCL communicator spanning dist's world.  Returns the transport name:
    "rccl" (librlhip.so issues ncclAllReduce itself, on the context's stream) or "torch.distributed" (the library's
    all-reduce hook hands the buffer to torch's RCCL communicator -- used only if EVERY rank failed to bind RCCL
    directly, e.g. a host process with a static RCCL; still a device-side RCCL all-reduce over xGMI)."""
    import sys

    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    dev = f"cuda:{ctx.device}"
    host_pg = dist.get_backend() == "gloo"      # tests: several ranks share one GPU, the exchange runs on the host
    if host_pg:
        force_hook = True
    pg_dev = "cpu" if host_pg else dev
    idbuf = (C.c_ubyte * 128)()
    # ncclCommInitRank is collective: a rank that cannot bind RCCL would leave the others blocked inside it, so the ranks first agree
    # (MIN) that EVERY one of them can; only then does anybody join
    can = torch.tensor([0 if force_hook else int(ctx.lib.rlhip_comm_can_load())], dtype=torch.int32, device=pg_dev)
    dist.all_reduce(can, op=dist.ReduceOp.MIN)
    if int(can.item()) == 0:
        force_hook = True
    ok = 1
    if rank == 0 and not force_hook:
        ok = int(ctx.lib.rlhip_comm_unique_id(idbuf) == 0)
    t = torch.tensor([ok] + list(idbuf), dtype=torch.uint8, device=pg_dev)
    dist.broadcast(t, src=0)
    vals = t.cpu().tolist()
    native = 0
    if vals[0] and not force_hook:
        idbuf = (C.c_ubyte * 128)(*vals[1:])
        native = int(ctx.lib.rlhip_comm_init(ctx.h, world, rank, idbuf) == 0)
    flag = torch.tensor([native], dtype=torch.int32, device=pg_dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return "rccl"
    if native:
        ctx.lib.rlhip_comm_destroy(ctx.h)
    if rank == 0:
        print("[randlapack_amd] direct RCCL binding failed on some rank; all-reduce goes through torch.distributed", file=sys.stderr)
    staging = {}

    def hook(_user, dev_ptr, count, is_f64):
        try:
            dt = torch.float64 if is_f64 else torch.float32
            buf = staging.get(dt)
            if buf is None or buf.numel() < count:
                buf = torch.empty(max(int(count), 1 << 20), dtype=dt, device=pg_dev)
                staging[dt] = buf
            nbytes = int(count) * (8 if is_f64 else 4)
            if host_pg:
                _lib.check(ctx.lib.rlhip_memcpy_d2h(ctx.h, buf.data_ptr(), dev_ptr, nbytes), "memcpy_d2h")
                dist.all_reduce(buf[:count])
                _lib.check(ctx.lib.rlhip_memcpy_h2d(ctx.h, dev_ptr, buf.data_ptr(), nbytes), "memcpy_h2d")
                return 0
            _lib.check(ctx.lib.rlhip_memcpy_d2d(ctx.h, buf.data_ptr(), dev_ptr, nbytes), "memcpy_d2d")
            ctx.sync()
            dist.all_reduce(buf[:count])
            torch.cuda.synchronize()
            _lib.check(ctx.lib.rlhip_memcpy_d2d(ctx.h, dev_ptr, buf.data_ptr(), nbytes), "memcpy_d2d")
            return 0
        except Exception as e:  # noqa: BLE001 - reported through the C return code
            print(f"[randlapack_amd] all-reduce hook failed: {e}", file=sys.stderr)
            return -1

    cb = _lib.HOOK(hook)
    _keepalive[id(ctx)] = (cb, staging)
    _lib.check(ctx.lib.rlhip_comm_set_hook(ctx.h, cb, None, world, rank), "rlhip_comm_set_hook")
    return "torch.distributed"


def rsvd_rowsharded(ctx, dist, A_local, m_local, n, k, key=(0, 0), b_sz=None, tol=1e-12, p=0, q=1):
    """RSVD of the row-sharded matrix whose local block is A_local (m_local x n).  Returns this rank's block of
    U (k, m_local) plus the replicated S and V."""
    from . import device as dev

    if ctx.lib.rlhip_comm_size(ctx.h) != dist.get_world_size():
        ctx.comm_transport = init_comm(ctx, dist)
    return dev.drv_rsvd(ctx, A_local, m_local, n, k, b_sz or k, tol, p, q, key=key)
