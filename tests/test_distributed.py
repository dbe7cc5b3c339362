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


def _ilqr_worker(rank, world, port, emu_path, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ilqr_checks as C
    from optimization_dynamics_amd import _lib, ilqr as IL, parallel
    lib = _lib.Library(emu_path)
    B, T = 6, 25
    im, obj, x1, U0 = C.constrained_problem(lib, "cpu", "cartpole", B, T)
    lo, hi = parallel.shard_range(B, world, rank)
    X, U, J, hist = IL.ILQR(im, obj, T).solve(torch.tensor(x1[:, lo:hi]), torch.tensor(U0[:, :, lo:hi]), max_iter=10, max_al_iter=4,
                                              obj_tol=1e-7, con_tol=1e-4)
    (Xg, Ug, Jg), _ = parallel.gather_batch([X.contiguous(), U.contiguous(), J.contiguous()])
    if rank == 0:
        cat = lambda t: torch.cat(list(t.unbind(0)), dim=-1).numpy()
        np.savez(os.path.join(outdir, "ilqr.npz"), X=cat(Xg), U=cat(Ug), J=cat(Jg))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_ilqr_solves(emu_lib, tmp_path):
    """the iLQR iteration shards like the path itself: the problems of a solver are independent solves (per-trajectory solver
    state), so two ranks each solving half of the batch with od_ilqr_solve -- no collective inside the solve -- and gathering the
    results give the unsharded solve's trajectories, controls and costs bit for bit"""
    world = 2
    mp.spawn(_ilqr_worker, args=(world, _free_port(), emu_lib.path, str(tmp_path)), nprocs=world, join=True)
    g = np.load(tmp_path / "ilqr.npz")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ilqr_checks as C
    from optimization_dynamics_amd import ilqr as IL
    im, obj, x1, U0 = C.constrained_problem(emu_lib, "cpu", "cartpole", 6, 25)
    X, U, J, hist = IL.ILQR(im, obj, 25).solve(torch.tensor(x1), torch.tensor(U0), max_iter=10, max_al_iter=4, obj_tol=1e-7, con_tol=1e-4)
    assert np.array_equal(g["X"], X.numpy()) and np.array_equal(g["U"], U.numpy()) and np.array_equal(g["J"], J.numpy())


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


def test_scaling_command_config5_leg_glue(emu_lib, monkeypatch):
    """the config-5 leg of an N > 1 run needs the MI355X (device-resident iLQR iteration); its glue -- per-rank measurement, ONE agreeing
    all-reduce, the block in the line, composition with the gather leg under one watchdog -- runs here with the measurement stubbed"""
    monkeypatch.setenv("OD_BENCH_TEST_CONFIG5_STUB", "1")
    rec = _run_bench(emu_lib, ["--gpus", "2"])
    c5 = rec["config5_sharded"]
    assert c5["problems_per_gpu"] == 2048 and c5["ms_per_iteration_slowest_rank"] == 2.0 and c5["ms_per_iteration_fastest_rank"] == 1.0
    assert abs(c5["value"] - 4096 * 60 * 12 / 2e-3) < 1e-3 and rec["with_gather"]["ranks_seen"] == 2


def test_scaling_command_also_times_the_gather(emu_lib):
    """the driver's scaling command carries no --gather: a run on N > 1 ranks times, after the headline region, the same steps once more
    with od_allgather_compact after every step and reports them beside `value` (`with_gather`), so one invocation per N yields the
    collective's cost too; a single-rank run has no such block"""
    rec = _run_bench(emu_lib, ["--gpus", "2"])
    wg = rec["with_gather"]
    assert wg and wg["ranks_seen"] == 2 and wg["value"] > 0 and "od_allgather_compact" in wg["collective"], wg
    assert wg["gathered_bytes_per_rank_per_step"] == 8 * (8 * 7 + 40 * 6) * 16 * 2
    assert rec["collective"] is None and rec["strong_scaling"]["value"] > 0          # the headline itself: no collective
    c4 = rec["config4_sharded"]                                                       # BASELINE config 4 as worded: 8192 rollouts over the ranks
    assert c4["rollouts_per_gpu"] == 4096 and c4["value"] > 0


def test_a_gather_leg_that_hangs_does_not_take_the_line_down(emu_lib, monkeypatch):
    """the watchdog of the extra gather leg: if the leg does not finish (an RCCL rendezvous that never returns, simulated) every rank
    leaves after the limit, rank 0 having printed the headline line with the cut-off noted -- exit code 0, one JSON line"""
    monkeypatch.setenv("OD_BENCH_TEST_HANG_GATHER_LEG", "1")
    monkeypatch.setenv("OD_BENCH_GATHER_LEG_TIMEOUT", "3")
    rec = _run_bench(emu_lib, ["--gpus", "2"])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["strong_scaling"]["value"] > 0
    assert "did not finish" in rec["with_gather"]["error"]


def _run_bench(emu_lib, extra, timeout=600):
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "16", "--horizon", "6",
           "--no-cpu-baseline", "--test-emu-lib", emu_lib.path] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 only
    return json.loads(lines[0])


def test_bench_strong_scaling_shards_the_same_workload(emu_lib, tmp_path):
    """`bench.py --gpus 2 --scaling strong`: the FIXED batch of the single-GPU run (block 0 of the seeded workload) is
    sharded over the ranks; gathered result == the unsharded rollout, bit for bit; value counts the batch once"""
    dump = str(tmp_path / "strong.npz")
    rec = _run_bench(emu_lib, ["--gpus", "2", "--scaling", "strong", "--gather", "--test-dump", dump])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["scaling"] == "strong" and rec["backend"] == "gloo"
    assert rec["config"]["total_batch"] == 16 and rec["config"]["units_per_step_per_gpu"] == 8 * 6
    assert abs(rec["value"] - 16 * 6 * rec["steps"] / (rec["ms_per_step"] * 1e-3 * rec["steps"])) < 1e-6 * rec["value"]
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    import parity_checks as P
    x1, U = bench.workload_slice(0, 16, 16, 6)
    x1b, Ub = bench.make_inputs(16, 6, seed=0)
    assert np.array_equal(x1, x1b) and np.array_equal(U, Ub)          # block 0 IS the single-GPU workload
    im = P.make_im("hopper", emu_lib, "cpu")
    X, G, st, it, _ = im.rollout_compact(torch.tensor(x1), torch.tensor(U))
    g = np.load(dump)
    assert np.array_equal(g["X"], X.numpy()) and np.array_equal(g["G"], G.numpy())


def test_bench_weak_run_reports_strong_mode_too(emu_lib):
    """a weak run on N > 1 ranks (what the driver launches) times the sharded-fixed-batch mode as well"""
    rec = _run_bench(emu_lib, ["--gpus", "2"])
    assert rec["scaling"] == "weak" and rec["config"]["total_batch"] == 32
    s = rec["strong_scaling"]
    assert s["scaling"] == "strong" and s["total_batch"] == 16 and s["rollouts_per_gpu"] == 8 and s["value"] > 0
    rec1 = _run_bench(emu_lib, [])
    assert rec1["n_gpus"] == 1 and "strong_scaling" not in rec1 and rec1["config"]["total_batch"] == 16


def test_workload_slices_are_consistent():
    sys.path.insert(0, ROOT)
    import bench
    a, ua = bench.workload_slice(0, 48, 16, 4)
    for lo, hi in ((0, 16), (16, 32), (5, 37), (40, 48)):
        b, ub = bench.workload_slice(lo, hi, 16, 4)
        assert np.array_equal(b, a[:, lo:hi]) and np.array_equal(ub, ua[:, :, lo:hi])
    x1, U = bench.make_inputs(16, 4, seed=2)
    assert np.array_equal(a[:, 32:48], x1)


def test_bench_force_dist_runs_the_multi_rank_path_with_one_rank(emu_lib, tmp_path):
    """`bench.py --force-dist --gather`: ONE rank under torch.distributed.run, process group initialised, barriers, max-over-ranks
    all-reduce and od_allgather_compact (the product's collective behind the C ABI, csrc/od_comm.inc; here over the harness stand-in
    for librccl, in the -m gpu twin below over RCCL on the MI355X) all executed -- and the gathered result is the plain rollout"""
    dump = str(tmp_path / "forced.npz")
    rec = _run_bench(emu_lib, ["--force-dist", "--gather", "--test-dump", dump])
    assert rec["n_gpus"] == 1 and rec["ranks_seen"] == 1 and rec["backend"] == "gloo" and "od_allgather_compact" in rec["collective"]
    assert "all-gather" in rec["config"]["parallelism"] and rec["value"] > 0
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    import parity_checks as P
    x1, U = bench.workload_slice(0, 16, 16, 6)
    X, G, st, it, _ = P.make_im("hopper", emu_lib, "cpu").rollout_compact(torch.tensor(x1), torch.tensor(U))
    g = np.load(dump)
    assert np.array_equal(g["X"], X.numpy()) and np.array_equal(g["G"], G.numpy())


@pytest.mark.gpu
def test_bench_force_dist_over_rccl_gpu(tmp_path):
    """the multi-rank path of bench.py on the hardware: torch.distributed.run --nproc-per-node 1, backend nccl (= RCCL), --gather --
    init_process_group, barrier, all_reduce(MAX) through torch.distributed, and the data path's collective through the C ABI:
    od_comm_create (ncclCommInitRank) + od_allgather_compact (two ncclAllGather on the handle's stream) on device arrays; the record
    says so (ranks_seen = ncclCommCount), and the gathered linearisation equals the single-process rollout bit for bit"""
    import json
    import subprocess
    dump = str(tmp_path / "rccl.npz")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--gather", "--steps", "3", "--warmup", "1", "--batch", "256",
           "--horizon", "10", "--no-cpu-baseline", "--test-dump", dump]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL rendezvous / bench subprocess did not finish in 120 s on this box (environment, not the product's data path)")
    if out.returncode != 0 and any(t in out.stderr for t in ("init_process_group", "ncclCommInitRank", "NCCL error", "ncclSystemError",
                                                             "ncclUnhandledCudaError", "DistNetworkError", "DistBackendError")):
        pytest.skip("torch.distributed / RCCL could not be brought up on this box: " + out.stderr.strip().splitlines()[-1][:300])
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["ranks_seen"] == 1 and rec["backend"] == "nccl" and "od_allgather_compact" in rec["collective"], rec
    assert "all-gather" in rec["config"]["parallelism"] and rec["value"] > 0
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(rec, open(os.path.join(d, "force_dist_rccl.json"), "w"), indent=1)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    import parity_checks as P
    from optimization_dynamics_amd import _lib
    x1, U = bench.workload_slice(0, 256, 256, 10)
    X, G, st, it, _ = P.make_im("hopper", _lib.default_library(), "cuda:0").rollout_compact(torch.tensor(x1), torch.tensor(U))
    g = np.load(dump)
    assert np.array_equal(g["X"], X.cpu().numpy()) and np.array_equal(g["G"], G.cpu().numpy())
