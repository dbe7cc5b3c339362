"""N>1 path on CPU: two gloo ranks shard a batch of rollouts (no data-path collective), then
all-gather the compact linearisation (x+, dq3) the way an outer loop elsewhere would consume it; and
bench.py's own multi-rank launch (`python bench.py --gpus 2`).
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
    X, G, st, it, _ = im.rollout_compact(torch.tensor(x1[:, lo:hi]), torch.tensor(U[:, :, lo:hi]))
    Xg, Gg = parallel.gather_linearization(X.contiguous(), G.contiguous())
    Ag, Bg = parallel.dense_linearization(Xg, Gg)
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


def test_bench_gpus_flag_spawns_ranks(emu_lib):
    """`python bench.py --gpus 2` without a launcher: two ranks appear and the line says n_gpus = 2 (host-emulation
    library over gloo here; on a GPU box the same path runs one rank per GPU over RCCL)"""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
           "--horizon", "6", "--gather", "--test-emu-lib", emu_lib.path]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["units_per_step_per_gpu"] == 16 * 6
    assert rec["value"] > 0 and "all-gather" in rec["config"]["parallelism"]
