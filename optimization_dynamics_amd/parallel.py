"""Multi-GPU sharding of the independent units (rollouts / knots / bundle knots): one process per
GPU (torchrun), no collective on the data path.  The only exchange the path has is the all-gather
of the per-knot linearisation (x_{t+1}, A_t, B_t) that an outer iLQR backward pass consumes once per
iteration; over RCCL this is one fused all_gather_into_tensor per array.
The reference has no distributed code at all (single Julia process)."""
import torch
import torch.distributed as dist


def shard_range(n, world, rank):
    """contiguous, balanced [lo, hi) slice of n units for `rank` of `world`"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_linearization(X, A, Bm, group=None):
    """All-gather per-rank rollout outputs along the batch (last) axis.
    X: (2nq, T+1, B_local), A: (2nq, 2nq, T, B_local), Bm: (2nq, nu, T, B_local); every rank must hold
    the same B_local (pad the last shard otherwise).  Returns the concatenated tensors."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return X, A, Bm
    world = dist.get_world_size(group)
    outs = []
    for t in (X, A, Bm):
        t = t.contiguous()
        buf = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf.view(-1), t.view(-1), group=group)
        # (world, ..., B_local) -> (..., world*B_local)
        outs.append(torch.cat(list(buf.unbind(0)), dim=-1))
    return tuple(outs)
