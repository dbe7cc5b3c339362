"""Multi-GPU sharding of the independent units (rollouts / knots / bundle knots): one process per GPU, no collective
on the data path -- every trajectory's backward (Riccati) pass is local to the rank that rolled it out.  The only
exchange the path can have is an all-gather of the per-knot linearisation for an outer loop that runs elsewhere
(SURVEY.md 8(e)); it ships the COMPACT form, x+ (2nq) and dq3/d(q1, q2, u) (nq x (2nq+nu)) per knot -- the dense A / B
of the reference's callbacks are those numbers plus constant rows (src/dynamics.jl:105-111,125) and 2.4x the bytes --
as one all_gather_into_tensor per array into a preallocated buffer, read through a view (no concatenation copy).
The reference has no distributed code at all (single Julia process)."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


class Communicator:
    """od_comm_* (include/od_mi355x.h): the all-gather of the compact linearisation over RCCL behind the C ABI, on the handle's stream --
    what a Julia process per GPU calls; this class is the same calls from Python.  `unique_id()` on rank 0, the 128
    bytes to every rank by whatever the host has (a file, a socket, torch.distributed), then `Communicator(im, id, rank, world)` on all
    ranks (a collective call).  `gather_compact(out)` -> X_all (world, 2nq, T+1, B), G_all (world, nq, 2nq+nu, T, B): block r is rank
    r's arrays."""

    @staticmethod
    def unique_id(lib=None):
        lib = lib or _lib.default_library()
        buf = (C.c_char * _lib.COMM_ID_BYTES)()
        lib.check(lib.cdll.od_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, im, uid, rank, world):
        assert len(uid) == _lib.COMM_ID_BYTES
        self.im, self.lib = im, im.lib
        h = C.c_void_p()
        self.lib.check(self.lib.cdll.od_comm_create(im._h, C.c_char_p(uid), int(rank), int(world), C.byref(h)))
        self._c = h
        w, r, d = C.c_int(), C.c_int(), C.c_int()
        self.lib.check(self.lib.cdll.od_comm_info(self._c, C.byref(w), C.byref(r), C.byref(d)))
        self.world, self.rank, self.device_index = w.value, r.value, d.value      # as the communicator reports them (ncclCommCount / ncclCommUserRank)

    def gather_compact(self, out, bufs=None):
        """`out`: the buffer dict ImplicitDynamics.rollout_compact returns last (X: 2nq x (T+1) x B; G: the raw nq (2nq+nu) x T x B
        array, column-major per knot) -> X_all (world, 2nq, T+1, B), G_all (world, nq, 2nq+nu, T, B) (a view), bufs (pass back in to
        reuse); asynchronous on the handle's stream (the current torch stream: ImplicitDynamics sets it per call)"""
        X, G = out["X"], out["G"]
        nq = self.im.model.nq
        T, B = G.shape[-2], G.shape[-1]
        if bufs is None:
            bufs = (torch.empty((self.world,) + tuple(X.shape), dtype=X.dtype, device=X.device), torch.empty((self.world,) + tuple(G.shape), dtype=G.dtype, device=G.device))
        self.im._use_current_stream()
        self.lib.check(self.lib.cdll.od_allgather_compact(self.im._h, self._c, B, T, X.data_ptr(), G.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr()))
        return bufs[0], bufs[1].view(self.world, G.shape[0] // nq, nq, T, B).transpose(1, 2), bufs

    def gather(self, t, out=None):
        """any contiguous device tensor -> (world,) + t.shape"""
        t = t.contiguous()
        if out is None:
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self.im._use_current_stream()
        self.lib.check(self.lib.cdll.od_comm_allgather(self.im._h, self._c, t.data_ptr(), out.data_ptr(), t.numel() * t.element_size()))
        return out

    def close(self):
        if getattr(self, "_c", None):
            self.lib.cdll.od_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_range(n, world, rank):
    """contiguous, balanced [lo, hi) slice of n units for `rank` of `world`"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_batch(tensors, group=None, bufs=None):
    """All-gather per-rank tensors (..., B_local) along the batch (last) axis; every rank must hold the same B_local
    (pad the last shard otherwise).  Returns views (world, ..., B_local) of the gather buffers: slab r is rank r's
    shard -- `.movedim(0, -2).reshape(..., world*B_local)` orders them like the unsharded batch when a flat batch axis
    is needed.  `bufs` (returned as second value) can be passed back in to reuse the buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [t.unsqueeze(0) for t in tensors], bufs
    world = dist.get_world_size(group)
    if bufs is None:
        bufs = [torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device) for t in tensors]
    for buf, t in zip(bufs, tensors):
        dist.all_gather_into_tensor(buf.view(-1), t.contiguous().view(-1), group=group)
    return bufs, bufs


def gather_linearization(X, G, group=None):
    """compact linearisation of sharded rollouts, concatenated along the batch axis: X (2nq, T+1, B), G (nq, 2nq+nu, T, B)"""
    (Xg, Gg), _ = gather_batch([X, G], group)
    return tuple(torch.cat(list(t.unbind(0)), dim=-1) for t in (Xg, Gg))


def dense_linearization(X, G):
    """fx / fu of the reference from the compact form: A (2nq, 2nq, T, B) = [[0 I]; [dq3/dq1 dq3/dq2]], Bm (2nq, nu, T, B)
    = [0; dq3/du]  (src/dynamics.jl:105-111,125)"""
    nq = G.shape[0]
    n = 2 * nq
    nu = G.shape[1] - n
    T, B = G.shape[2], G.shape[3]
    A = torch.zeros(n, n, T, B, dtype=G.dtype, device=G.device)
    A[:nq, nq:] = torch.eye(nq, dtype=G.dtype, device=G.device)[:, :, None, None]
    A[nq:, :] = G[:, :n]
    Bm = torch.zeros(n, nu, T, B, dtype=G.dtype, device=G.device)
    Bm[nq:] = G[:, n:]
    return A, Bm
