"""N ranks of a row-sharded driver in ONE process on ONE device (test scaffolding, also used by scripts/ranks_on_one_device.py).

Every rank is a thread with its own rlhip context (own scratch arena, mailbox, communicator record: world = N, rank = r); all contexts are
bound to the SAME HIP stream, so the device runs the ranks' kernels strictly one after the other.  The library's all-reduce hook
(include/rlhip.h: rlhip_comm_set_hook) is served in place: the ranks meet at a barrier, rank 0 sums the N device buffers IN RANK ORDER with
torch ops on the shared stream and copies the sum back into every rank's buffer.  This is the real sharded code path -- Queue::allreduce_sum
at every reduction point, shard_extent, block-cyclic rows -- with a deterministic transport."""
import threading

import numpy as np
import torch

from randlapack_amd import _lib, device as d


class World:
    def __init__(self, n):
        self.n = n
        self.ctx = [d.Context(0) for _ in range(n)]
        self.bar = threading.Barrier(n, timeout=600)
        self.slots = [None] * n
        self.bytes_reduced = 0
        self.collectives = 0
        self.err = []
        self._cbs = []
        for r in range(n):
            cb = _lib.HOOK(self._make_hook(r))
            self._cbs.append(cb)
            _lib.check(self.ctx[r].lib.rlhip_comm_set_hook(self.ctx[r].h, cb, None, n, r), "rlhip_comm_set_hook")

    def close(self):
        for c in self.ctx:
            c.lib.rlhip_comm_destroy(c.h)
            c.close()
        self.ctx = []

    @staticmethod
    def _view(ptr, count, is_f64):
        class H:
            pass
        h = H()
        h.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8" if is_f64 else "<f4", "data": (int(ptr), False), "version": 3}
        return torch.as_tensor(h, device="cuda:0")

    def _make_hook(self, r):
        def hook(_user, dev_ptr, count, is_f64):
            try:
                self.slots[r] = (int(dev_ptr), int(count), int(is_f64))
                self.bar.wait()
                if r == 0:
                    c0, f0 = self.slots[0][1], self.slots[0][2]
                    assert all(s[1] == c0 and s[2] == f0 for s in self.slots), f"ranks disagree on a collective: {self.slots}"
                    bufs = [self._view(*s) for s in self.slots]
                    acc = bufs[0]
                    for b in bufs[1:]:
                        acc += b                       # fixed rank order: deterministic
                    for b in bufs[1:]:
                        b.copy_(acc)
                    self.bytes_reduced += c0 * (8 if f0 else 4)
                    self.collectives += 1
                self.bar.wait()
                return 0
            except Exception as e:  # noqa: BLE001
                self.err.append(f"rank {r}: {e}")
                try:
                    self.bar.abort()
                except Exception:
                    pass
                return -1
        return hook

    def run(self, fn):
        """fn(rank, ctx) on every rank, concurrently; returns the list of results"""
        out = [None] * self.n
        exc = []

        def work(r):
            try:
                out[r] = fn(r, self.ctx[r])
            except Exception as e:  # noqa: BLE001
                exc.append((r, e))
                try:
                    self.bar.abort()
                except Exception:
                    pass
        ts = [threading.Thread(target=work, args=(r,)) for r in range(self.n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if exc or self.err:
            self.bar.reset()
            err, self.err = self.err, []
            raise RuntimeError(f"{exc} {err}")
        return out


def block_cyclic_rows(rank, world, m, b):
    """global row indices of `rank` when blocks of b rows are dealt round-robin (block g on rank g % world)"""
    idx = [np.arange(g * b, min((g + 1) * b, m)) for g in range(rank, (m + b - 1) // b, world)]
    return np.concatenate(idx).astype(np.int64) if idx else np.zeros(0, dtype=np.int64)


def contiguous_rows(rank, world, m):
    return np.array_split(np.arange(m, dtype=np.int64), world)[rank]
