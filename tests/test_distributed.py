"""N>1 path on CPU: two gloo ranks shard a batch of rollouts (no data-path collective), then
all-gather the linearisation (x+, A, B) the way an outer iLQR backward pass would consume it.
Each rank drives the host-emulation build of the product library (CPU test tier)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, emu_path, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import parity_checks as P
    import workloads as W
    from optimization_dynamics_amd import _lib, parallel
    lib = _lib.Library(emu_path)
    B, T = 16, 8
    x1, U = W.hopper_rollout_inputs(B, T, seed=3, u_sigma=0.3)
    lo, hi = parallel.shard_range(B, world, rank)
    im = P.make_im("hopper", lib, "cpu")
    X, A, Bm, st, it, _ = im.rollout(torch.tensor(x1[:, lo:hi]), torch.tensor(U[:, :, lo:hi]))
    Xg, Ag, Bg = parallel.gather_linearization(X.contiguous(), A.contiguous(), Bm.contiguous())
    if rank == 0:
        np.savez(os.path.join(outdir, "gathered.npz"), X=Xg.numpy(), A=Ag.numpy(), B=Bg.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from optimization_dynamics_amd.parallel import shard_range
    for n in (1, 7, 8, 4096, 8192):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_sharded_rollout_and_allgather(emu_lib, tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, emu_lib.path, str(tmp_path)), nprocs=world, join=True)
    g = np.load(tmp_path / "gathered.npz")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_checks as P
    import workloads as W
    x1, U = W.hopper_rollout_inputs(16, 8, seed=3, u_sigma=0.3)
    im = P.make_im("hopper", emu_lib, "cpu")
    X, A, Bm, st, it, _ = im.rollout(torch.tensor(x1), torch.tensor(U))
    # sharded + gathered == unsharded, bit for bit (units are independent)
    assert np.array_equal(g["X"], X.numpy())
    assert np.array_equal(g["A"], A.numpy())
    assert np.array_equal(g["B"], Bm.numpy())
