"""Row-block sharding glue (one process per GPU).

The sharded algorithm itself lives in the C++ drivers (include/RandLAPACK_amd/*.hh): every reduction over the
row index is followed by `Queue::allreduce_sum`, which is a no-op for one rank.  This module only performs the
rendezvous: rank 0 creates the RCCL unique id, torch.distributed broadcasts the 128 bytes, every rank joins
the communicator bound to its rlhip context.  After that the data-path collectives are issued by librlhip.so
itself on the context's HIP stream (RCCL over xGMI) -- torch.distributed is not in the data path.

`rowsharded_rsvd_model` is a numpy statement of the same exchange pattern; tests run it under gloo with
world_size 2 on CPU to pin down WHICH quantities are all-reduced (it is test scaffolding for the host
logic, it is not a fallback and is never called by the product path).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def init_comm(ctx, dist) -> None:
    """Join this rank's rlhip context to an RCCL communicator spanning dist's world."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    idbuf = (C.c_ubyte * 128)()
    if rank == 0:
        _lib.check(ctx.lib.rlhip_comm_unique_id(idbuf), "rlhip_comm_unique_id")
    t = torch.tensor(list(idbuf), dtype=torch.uint8, device=f"cuda:{ctx.device}")
    dist.broadcast(t, src=0)
    idbuf = (C.c_ubyte * 128)(*t.cpu().tolist())
    _lib.check(ctx.lib.rlhip_comm_init(ctx.h, world, rank, idbuf), "rlhip_comm_init")


def rsvd_rowsharded(ctx, dist, A_local, m_local, n, k, key=(0, 0), b_sz=None, tol=1e-12, p=0, q=1):
    """RSVD of the row-sharded matrix whose local block is A_local (m_local x n).  Returns this rank's block of
    U (k, m_local) plus the replicated S and V."""
    from . import device as dev

    if ctx.lib.rlhip_comm_size(ctx.h) != dist.get_world_size():
        init_comm(ctx, dist)
    return dev.drv_rsvd(ctx, A_local, m_local, n, k, b_sz or k, tol, p, q, key=key)


# ------------------------------------------------------------------------------------------------------
def rowsharded_rsvd_model(A_local, k, Omega, allreduce):
    """numpy model of the exchange pattern for p = 0, one QB block:
         Y_g = A_g Omega | G = allreduce(Y_g^T Y_g) | R = chol(G) | Q_g = Y_g R^-1 |
         B^T = allreduce(A_g^T Q_g) | SVD(B^T) replicated | U_g = Q_g Uhat
    `allreduce(x)` must return the element-wise sum over ranks."""
    Y = A_local @ Omega
    G = allreduce(Y.T @ Y)
    R = np.linalg.cholesky(G).T
    Q = np.linalg.solve(R.T, Y.T).T
    BT = allreduce(A_local.T @ Q)
    V, S, UT = np.linalg.svd(BT, full_matrices=False)
    U = Q @ UT.T
    return U, S, V
