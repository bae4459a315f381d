"""N > 1 host logic on CPU: two gloo ranks run the row-sharded exchange pattern (numpy model of what the C++
drivers do between collectives) and must reproduce the single-process oracle RSVD on the stacked matrix."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m, n, k, out):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from _sharded_model import rowsharded_rsvd_model

    rng = np.random.default_rng(123)
    A = rng.standard_normal((m, n)) @ np.diag(np.linspace(1, 0.01, n))
    Omega, _ = oracle.fill_dense(n, k)            # every rank regenerates the same sketch from (ctr, key)
    rows = np.array_split(np.arange(m), world)[rank]

    def allreduce(x):
        t = torch.from_numpy(np.ascontiguousarray(x))
        dist.all_reduce(t)
        return t.numpy()

    U, S, V = rowsharded_rsvd_model(A[rows], k, Omega, allreduce)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rows, U))
    if rank == 0:
        Ufull = np.zeros((m, k))
        for r, u in gathered:
            Ufull[r] = u
        ref = oracle.rsvd(A, k, k, 1e-12, 0, 1)
        out["S_err"] = float(np.max(np.abs(S - ref["S"]) / ref["S"]))
        out["recon"] = float(np.linalg.norm(A - (Ufull * S) @ V.T) / np.linalg.norm(A))
        out["recon_ref"] = float(np.linalg.norm(A - (ref["U"] * ref["S"]) @ ref["V"].T) / np.linalg.norm(A))
        out["orth"] = float(np.linalg.norm(Ufull.T @ Ufull - np.eye(k)))
    dist.destroy_process_group()


def test_rowsharded_rsvd_world2_matches_oracle():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 301, 64, 12, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out["S_err"] < 1e-11
    assert abs(out["recon"] - out["recon_ref"]) < 1e-10
    assert out["orth"] < 1e-9
