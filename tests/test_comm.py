"""od_comm_* (include/od_mi355x.h): the all-gather of the compact linearisation behind the C ABI.
CPU tier: the product's od_comm.inc in the host build, RCCL replaced by the harness stand-in over /dev/shm (tests/host_emu/emu_rccl.cpp)
-- one rank, and two ranks in two processes, each against the unsharded rollout bit for bit.  GPU tier: one rank over RCCL on the MI355X."""
import ctypes as C
import glob
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _cleanup_shm():
    for f in glob.glob("/dev/shm/odemu_%d_*" % os.getpid()):
        try:
            os.remove(f)
        except OSError:
            pass


def _skip_without_rccl(lib):
    """od_comm_* answer OD_ERR_UNSUPPORTED where librccl cannot be loaded: an environment without it skips, it does not fail"""
    buf = (C.c_ubyte * 128)()
    if lib.cdll.od_comm_unique_id(buf) == -2:
        pytest.skip("librccl not loadable on this box: " + lib.cdll.od_last_error().decode())


def _one_rank(lib, device):
    import bench
    import parity_checks as P
    from optimization_dynamics_amd import parallel as par
    B, T = 24, 12
    x1, U = bench.workload_slice(0, B, B, T)
    im = P.make_im("hopper", lib, device)
    X, G, st, it, out = im.rollout_compact(torch.tensor(x1, device=device), torch.tensor(U, device=device))
    comm = par.Communicator(im, par.Communicator.unique_id(lib), 0, 1)
    assert (comm.world, comm.rank) == (1, 0)
    Xa, Ga, bufs = comm.gather_compact(out)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    assert Xa.shape == (1,) + tuple(X.shape) and Ga.shape == (1,) + tuple(G.shape)
    assert torch.equal(Xa[0], X) and torch.equal(Ga[0], G)
    # any buffer; the gains of a backward pass, say
    t = torch.arange(37, dtype=torch.float64, device=device)
    assert torch.equal(comm.gather(t)[0], t)
    # argument errors are error codes
    cd = lib.cdll
    assert cd.od_comm_create(im._h, None, 0, 1, C.byref(C.c_void_p())) == -1
    assert cd.od_comm_create(im._h, C.c_char_p(b"x" * 128), 2, 2, C.byref(C.c_void_p())) == -1
    assert cd.od_allgather_compact(im._h, None, B, T, out["X"].data_ptr(), None, bufs[0].data_ptr(), None) == -1
    assert cd.od_allgather_compact(im._h, comm._c, B, T, out["X"].data_ptr(), None, None, None) == -1
    assert cd.od_allgather_compact(im._h, comm._c, B, T, out["X"].data_ptr(), None, bufs[0].data_ptr(), None) == 0      # X alone
    assert cd.od_comm_destroy(None) == 0
    comm.close()
    return X, G


def test_one_rank_allgather_is_the_rollout_cpu(emu_lib):
    try:
        _one_rank(emu_lib, "cpu")
    finally:
        _cleanup_shm()


def _worker(rank, world, uid, path, q):
    try:
        torch.set_num_threads(1)
        import bench
        import parity_checks as P
        from optimization_dynamics_amd import _lib, parallel as par
        lib = _lib.Library(path)
        B, T = 12, 9                       # per rank
        x1, U = bench.workload_slice(rank * B, (rank + 1) * B, world * B, T)
        im = P.make_im("hopper", lib, "cpu")
        X, G, st, it, out = im.rollout_compact(torch.tensor(x1), torch.tensor(U))
        comm = par.Communicator(im, uid, rank, world)
        bufs = None
        for rep in range(4):               # (several gathers: the stand-in recycles its blocks two gathers back)
            Xa, Ga, bufs = comm.gather_compact(out, bufs)
        q.put((rank, comm.world, comm.rank, Xa.numpy().copy(), Ga.numpy().copy()))
        comm.close()
    except Exception as e:                  # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def test_two_ranks_allgather_equals_the_unsharded_rollout_cpu(emu_lib):
    """two processes, each rolls out its shard through the host build and calls od_allgather_compact; every rank ends with both shards,
    in rank order, equal bit for bit to the single-process rollout of all trajectories"""
    import bench
    import parity_checks as P
    from optimization_dynamics_amd import parallel as par
    world, B, T = 2, 12, 9
    uid = par.Communicator.unique_id(emu_lib)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, uid, emu_lib.path, q)) for r in range(world)]
    try:
        for p in ps:
            p.start()
        res = [q.get(timeout=240) for _ in ps]
        for p in ps:
            p.join(60)
    finally:
        for p in ps:
            if p.is_alive():
                p.kill()
        for f in glob.glob("/dev/shm/" + uid.split(b"\0")[0].decode() + "_*"):
            os.remove(f)
    assert all(r[1] != "error" for r in res), [r[2] for r in res if r[1] == "error"]
    x1, U = bench.workload_slice(0, world * B, world * B, T)
    X, G, st, it, _ = P.make_im("hopper", emu_lib, "cpu").rollout_compact(torch.tensor(x1), torch.tensor(U))
    X, G = X.numpy(), G.numpy()
    for rank, w, r, Xa, Ga in res:
        assert (w, r) == (world, rank)
        for k in range(world):
            assert np.array_equal(Xa[k], X[:, :, k * B:(k + 1) * B]) and np.array_equal(Ga[k], G[:, :, :, k * B:(k + 1) * B]), (rank, k)


@pytest.mark.gpu
def test_one_rank_allgather_over_rccl_gpu(gpu_lib):
    """ncclCommInitRank + two ncclAllGather through od_comm_* on the MI355X (librccl resolved at run time), on the handle's stream right
    after the rollout that produced the arrays: bit-identical to them; the record goes to gpurun_out/od_comm_rccl.json"""
    import json
    _skip_without_rccl(gpu_lib)
    X, G = _one_rank(gpu_lib, "cuda:0")
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(dict(ranks=1, backend="RCCL through od_comm_* (C ABI)", X_bytes=int(X.numel() * 8), G_bytes=int(G.numel() * 8), bit_identical=True),
              open(os.path.join(d, "od_comm_rccl.json"), "w"), indent=1)
