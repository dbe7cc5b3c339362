#!/usr/bin/env python
"""Headline benchmark: contact-implicit steps + implicit gradients per second,
hopper, T=100, batch=4096 rollouts per GPU (BASELINE.json metric).

One "step" = one od_rollout_compact call (two launches: the time recursion, then all gradients):
4096 trajectories x 100 knots = 409 600 units, each unit = (q1,q2,u) -> (q3, dq3/dq1, dq3/dq2, dq3/du)
honouring kappa_eval for the state and kappa_grad for the gradient (the reference's f + fx + fu,
src/dynamics.jl:81-128; fx / fu are these blocks plus constants).  Inputs are resident in HBM
before the timed region.  Multi-GPU: one process per GPU, trajectories sharded, no data-path
collective -- every trajectory's Riccati pass is rank-local.  The workload is ONE seeded set of trajectories in blocks
of `--batch` (block k drawn with seed k; block 0 is the single-GPU workload):
  --scaling weak   (default) rank r rolls out block r: `--batch` rollouts per GPU, N x the work on N GPUs;
  --scaling strong the metric as BASELINE.json words it -- a FIXED total batch (`--batch`, block 0) sharded over the
                   ranks (shard_range), value = total units / max-over-ranks time.
A weak run on N > 1 ranks times the strong mode as well (same K steps, after the headline region) and reports it under
"strong_scaling", so one driver invocation per N yields both curves.
`--gather` adds the one exchange the path can have, the all-gather of the linearisation for an outer
loop that runs elsewhere, in compact form (x+ and dq3/d(q1,q2,u): 2.4x fewer bytes than x+, A, B), through the product's own
entry points (od_comm_* / od_allgather_compact: RCCL behind the C ABI).  Without the flag a run on N > 1 ranks times the steps
once more WITH the gather after the headline region and reports them under "with_gather" (never in `value`).

Launch: under `python -m torch.distributed.run ... bench.py --gpus N` (RANK / WORLD_SIZE in the
environment) this process is one rank; `python bench.py --gpus N` on its own re-executes itself under
torch.distributed.run with N ranks on 127.0.0.1.  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector == fp64 matrix peak
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md
HOPPER = dict(nq=4, nu=2, nz=20, nth=13)


def algorithmic_flops_per_unit(iters_eval, stats):
    """F = I_e (F_rz + 2/3 nz^3 + 4 nz^2 + 2 F_r + c_cone) + (F_rz + F_rth + 2/3 nz^3 + 2 nz^2 nth)
    = the reference's dense-LU algorithm, with the measured mean iteration count I_e."""
    nz, nth = HOPPER["nz"], HOPPER["nth"]
    F_r, F_rz, F_rth = stats["ops_r"], stats["ops_rz"], stats["ops_rth"]
    c_cone = 100.0
    per_iter = F_rz + (2.0 / 3.0) * nz ** 3 + 4.0 * nz ** 2 + 2.0 * F_r + c_cone
    grad = F_rz + F_rth + (2.0 / 3.0) * nz ** 3 + 2.0 * nz ** 2 * nth
    return iters_eval * per_iter + grad, per_iter, grad


HEADLINE_SOURCES = ("od_math.h", "od_solver.h", "od_units.h", "od_coop.h", "od_coop3.h", "od_vtable.h", "od_model_tu.inc", "od_capi.hip",
                    "od_model_hopper.hip", "gen/hopper.h", "gen/coop_hopper.h", "gen/coop3_hopper.h")


def kernel_source_hash():
    """hash of the device sources the headline kernels are compiled from: profiles/*_traffic.json records it, a PMC measurement of
    other code is stale"""
    d = os.path.join(ROOT, "optimization_dynamics_amd", "csrc")
    hsh = hashlib.sha1()
    for fn in HEADLINE_SOURCES:
        p = os.path.join(d, fn)
        if os.path.exists(p):
            hsh.update(fn.encode())
            hsh.update(open(p, "rb").read())
    return hsh.hexdigest()[:16]


def measured_traffic(batch, horizon, gather):
    """HBM bytes per od_rollout and the executed-work counters from the newest profiles/*_traffic.json
    (tools/summarize_profile.py after tools/profile_round.sh: separate rocprofv3 --pmc passes, 2*FETCH_SIZE + WRITE_SIZE over both
    kernels; wavefront-level fp64 instruction counts).  -> (bytes | None, note, record | None)"""
    if gather or batch != 4096 or horizon != 100:
        return None, "PMC traffic is recorded for the default workload only", None
    pd = os.path.join(ROOT, "profiles")
    cands = sorted(f for f in os.listdir(pd) if f.endswith("_traffic.json"))
    if not cands:
        return None, "no profiles/*_traffic.json", None
    rec = json.load(open(os.path.join(pd, cands[-1])))
    if rec.get("source_hash") != kernel_source_hash():
        return None, "profiles/%s is stale: kernels changed since that PMC run (re-run tools/profile_round.sh)" % cands[-1], None
    return float(rec["hbm_bytes_per_step"]), "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (2*FETCH + WRITE, both kernels)" % cands[-1], rec


def algorithmic_bytes_per_unit():
    nq, nu = HOPPER["nq"], HOPPER["nu"]
    return 8 * ((2 * nq + nu) + nq + nq * (2 * nq + nu))     # 432 B


def workload_slice(lo, hi, block, horizon):
    """trajectories [lo, hi) of the global workload: blocks of `block` trajectories, block k = make_inputs(block, T, seed=k)"""
    xs, us = [], []
    for k in range(lo // block, (hi - 1) // block + 1):
        x1, U = make_inputs(block, horizon, seed=k)
        a, b = max(lo, k * block) - k * block, min(hi, (k + 1) * block) - k * block
        xs.append(x1[:, a:b]); us.append(U[:, :, a:b])
    return np.ascontiguousarray(np.concatenate(xs, 1)), np.ascontiguousarray(np.concatenate(us, 2))


def make_inputs(batch, horizon, seed, h=0.05):
    """q = [0, 0.5 + r_foot, 0, 0.5] + N(0, 0.02^2), x1 = [q; q];
    u_t = [0; g m_body h/2] + N(0,1) (examples/hopper.jl:178,270)."""
    rng = np.random.default_rng(seed)
    q = np.array([0.0, 0.5 + 0.05, 0.0, 0.5])[:, None] + rng.normal(0.0, 0.02, (4, batch))
    x1 = np.vstack([q, q])
    U = np.array([0.0, 9.81 * 3.0 * 0.5 * h])[:, None, None] + rng.normal(0.0, 1.0, (2, horizon, batch))
    return x1, U


def effective_cores():
    """host cores this process may actually use: CPU count, affinity mask and the cgroup CPU quota (a container on a
    256-thread box can be limited to 16 CPUs' worth of time; 256 OpenMP threads would only be throttled)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(batch, horizon, seed, budget_s=12.0):
    """Oracle (CPU restatement: three dense-LU solves per knot like the reference) on a bounded
    sample of the same workload, OpenMP over trajectories on all host cores."""
    from oracle import oracle as O
    cores = effective_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    try:                                     # the runtime may already be initialised: set the team size explicitly too
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except Exception:
        pass
    sim = O.make_sim("hopper", 0.05, kappa_tol=1e-4, kappa_grad_tol=1e-3)
    x1, U = make_inputs(batch, horizon, seed)
    nb = min(batch, max(cores, 16))
    t0 = time.time()
    O.rollout(sim, x1[:, :nb], U[:, :, :nb])
    dt = time.time() - t0
    rate = nb * horizon / dt
    nb2 = int(min(batch, max(nb, rate * budget_s / horizon)))
    nb2 = max(cores, (nb2 // cores) * cores)
    bufs = {}
    xs, Us = np.asfortranarray(x1[:, :nb2]), np.asfortranarray(U[:, :, :nb2])
    O.rollout(sim, xs[:, :cores], Us[:, :, :cores])                     # threads up
    X_, A_, B_, bad = O.rollout(sim, xs, Us, bufs=bufs)                 # output arrays allocated and touched
    t0 = time.time()
    _, _, _, bad = O.rollout(sim, xs, Us, bufs=bufs)                    # timed: the solves only, like the GPU side
    dt = time.time() - t0
    # the same code on one thread (single-thread and all-cores figures)
    single = None
    try:
        import ctypes
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(1)
        ns = min(batch, 8)
        t1 = time.time()
        O.rollout(sim, x1[:, :ns], U[:, :, :ns])
        single = ns * horizon / (time.time() - t1)
        omp.omp_set_num_threads(cores)
    except Exception:
        pass
    return dict(value=nb2 * horizon / dt, unit="steps+grads/s", cores=cores, kind="port", single_thread_value=single,
                sample="%d of %d trajectories x T=%d (same seed), %.1f s, OpenMP over trajectories; "
                       "CPU restatement of the reference algorithm (f, fx, fu = 3 dense-LU IP solves per knot), "
                       "not the Julia reference" % (nb2, batch, horizon, dt),
                nonconverged_solves=int(bad))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="rollouts per GPU (weak) / in total (strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch rollouts per GPU; strong: --batch rollouts in total, sharded over the GPUs (BASELINE.json's wording)")
    ap.add_argument("--test-dump", default=None, help="TEST HARNESS ONLY: rank 0 writes the gathered (X, G) of the last step to this .npz")
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--gather", action="store_true", help="all-gather the compact linearisation (x+, dq3) after every step (RCCL)")
    ap.add_argument("--dense", action="store_true", help="write the dense fx / fu matrices (od_rollout) instead of the compact dq3 (od_rollout_compact)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux-configs", action="store_true", help="skip the aux_config_2 / 3 / 5 blocks (bench_configs.py)")
    ap.add_argument("--ppw", type=int, default=0, help="problems per wavefront (0 = library default)")
    ap.add_argument("--wpb", type=int, default=0, help="wavefronts per workgroup of the solve pass: 1 or 4 (0 = library default)")
    ap.add_argument("--coop", type=int, default=0, help="cooperative solve pass: 0 automatic, 1 never, 2 always")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the multi-rank code path even with ONE rank: torch.distributed.run --nproc-per-node 1, init_process_group "
                         "(nccl = RCCL on the GPU), barriers, the max-over-ranks all-reduce and (with --gather) od_allgather_compact -- so "
                         "that every line of the N > 1 path has executed on whatever single MI355X is at hand (tests/test_distributed.py)")
    ap.add_argument("--test-emu-lib", default=None,
                    help="TEST HARNESS ONLY (tests/test_distributed.py): run the ranks on CPU over gloo against the host-emulation build")
    args = ap.parse_args()

    if "RANK" not in os.environ and (args.gpus > 1 or args.force_dist):
        # stand-alone launch: become N ranks (one per GPU) under torch.distributed.run on this node
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)
    emu = args.test_emu_lib is not None
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if emu else "nccl")
    if emu:
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU path in the product library)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = lambda: torch.cuda.synchronize(dev)

    from optimization_dynamics_amd import ImplicitDynamics, hopper
    lib = None
    if emu:
        from optimization_dynamics_amd import _lib
        lib = _lib.Library(args.test_emu_lib)
    im = ImplicitDynamics(hopper, 0.05, r_tol=1e-8, kappa_eval_tol=1e-4, kappa_grad_tol=1e-3, device=dev, lib=lib)  # examples/hopper.jl:42
    if args.ppw or args.wpb:
        im.set_launch_config(args.ppw, args.wpb)
    if args.coop:
        im.set_cooperative(args.coop)
    T = args.horizon
    from optimization_dynamics_amd.parallel import shard_range
    stream = None if emu else torch.cuda.current_stream(dev)

    comm = None
    if args.gather and dist is not None:
        from optimization_dynamics_amd.parallel import Communicator
        box = [Communicator.unique_id(im.lib) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)            # the 128-byte id, out of band (a Julia host: a file or a socket)
        comm = Communicator(im, box[0], rank, world)
        assert comm.world == world and comm.rank == rank

    def inputs_for(mode):
        """(lo, hi) of this rank in the global workload"""
        if mode == "weak":
            return rank * args.batch, (rank + 1) * args.batch
        return shard_range(args.batch, world, rank)

    def timed_region(mode, dump=None):
        """W untimed + K timed steps of this rank's share under `mode`; -> (max-over-ranks seconds, mean event ms, status, iterations, B_local)"""
        lo, hi = inputs_for(mode)
        x1, U = workload_slice(lo, hi, args.batch, T)
        x1d, Ud = torch.tensor(x1, device=dev), torch.tensor(U, device=dev)
        out = None
        gather_bufs = None

        def step():
            nonlocal out, gather_bufs
            # the unit's outputs exactly: x+ = [q2; q3] and dq3/d(q1, q2, u) (nq x (2nq+nu)) per knot -- od_rollout's dense
            # A / B are the same numbers padded with the constant [0 I] and zero rows of fx / fu (--dense times those)
            if args.dense:
                X, A, Bm, st, it, out = im.rollout(x1d, Ud, out=out)
                return st, it
            X, G, st, it, out = im.rollout_compact(x1d, Ud, out=out)
            if comm is not None:
                # the product's collective: od_allgather_compact (two ncclAllGather on the handle's stream, behind the C ABI a Julia
                # process per GPU would call) -- not torch.distributed
                _, _, gather_bufs = comm.gather_compact(out, gather_bufs)
            return st, it

        for _ in range(args.warmup):
            step()
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        ev = None if emu else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t0 = time.perf_counter()
        for k in range(args.steps):
            if ev:
                ev[k][0].record(stream)      # the rollout kernels are launched on this (torch current) stream
            st, it = step()
            if ev:
                ev[k][1].record(stream)
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else elapsed / args.steps * 1e3
        if dump and rank == 0 and gather_bufs is not None:
            Bl = hi - lo
            Xg = gather_bufs[0].reshape(world, 8, T + 1, Bl).movedim(0, -2).reshape(8, T + 1, world * Bl)
            Gg = gather_bufs[1].reshape(world, 10, 4, T, Bl).movedim(0, -2).reshape(10, 4, T, world * Bl).transpose(0, 1)
            np.savez(dump, X=Xg.cpu().numpy(), G=Gg.cpu().numpy())
        return elapsed, kernel_ms, st, it, hi - lo

    if args.gather and args.scaling == "strong" and args.batch % world:
        raise SystemExit("bench.py: --gather with --scaling strong needs --batch divisible by the number of ranks (equal shards)")
    elapsed, kernel_ms, st, it, B = timed_region(args.scaling, dump=args.test_dump)
    strong = None
    if world > 1 and args.scaling == "weak" and not (args.gather and args.batch % world):
        e2, k2, st2, it2, B2 = timed_region("strong")
        strong = dict(scaling="strong", total_batch=args.batch, rollouts_per_gpu=B2, value=args.batch * T * args.steps / e2,
                      unit="steps+grads/s", ms_per_step=e2 / args.steps * 1e3, kernel_ms_rank0=k2)
    # BASELINE config 4 as worded -- "hopper gait, batch = 8192 rollouts sharded across 8 x MI355X" -- on N > 1 ranks: the FIXED batch of 8192
    # (block 0 of the workload at that block size) sharded over the ranks, same K steps, beside the headline's figures
    config4 = None
    if world > 1 and args.scaling == "weak" and not args.gather and 8192 % world == 0 and args.batch != 8192:
        keep = args.batch
        args.batch = 8192
        try:
            e4, k4, _, _, B4 = timed_region("strong")
            config4 = dict(workload="BASELINE config 4: hopper, 8192 rollouts x T = %d sharded over %d GPUs (strong scaling)" % (T, world), rollouts_per_gpu=B4,
                           value=8192 * T * args.steps / e4, unit="steps+grads/s", ms_per_step=e4 / args.steps * 1e3, kernel_ms_rank0=k4)
        finally:
            args.batch = keep

    # BASELINE config 5 as worded -- "rocket thrust-cone SOCP step inside full iLQR outer loop with implicit grads, fp32, 8 x MI355X" -- on
    # N > 1 ranks: 4096 independent landing problems sharded over the ranks (every problem an independent solve: no collective in the
    # iteration), the device-resident iteration timed per rank, the slowest rank's time reported.  No collective inside the leg; what each
    # rank measured (or that it failed) is agreed on by ONE all-reduce afterwards, so no rank is left waiting.
    stub5 = bool(os.environ.get("OD_BENCH_TEST_CONFIG5_STUB"))         # TEST HARNESS ONLY: the leg's glue on the host build, the measurement replaced by a constant
    want_config5_leg = world > 1 and args.scaling == "weak" and not args.gather and (not emu or stub5) and not args.no_aux_configs and 4096 % world == 0

    def config5_leg():
        ms5, r5 = -1.0, None
        try:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
            import bench_configs
            if stub5:
                r5 = dict(ms_per_iteration=1.0 + rank)
            else:
                r5, _, _ = bench_configs._config5_one(dev, "examples/rocket.jl inputs", torch.float32, False, B=4096 // world)
            ms5 = float(r5["ms_per_iteration"])
        except Exception as e:               # noqa: BLE001
            r5 = dict(error=repr(e)[:300])
        t5 = torch.tensor([ms5, -ms5], dtype=torch.float64, device=dev)
        dist.all_reduce(t5, op=dist.ReduceOp.MAX)
        slow, fast = float(t5[0].item()), -float(t5[1].item())
        if fast > 0:                         # every rank measured
            n_units = (4096 // world) * world * 60 * 12         # 11 candidates' state steps + the linearisation's step per knot
            return dict(workload="BASELINE config 5: rocket landing with the thrust-cone projection inside the iLQR iteration on the device, fp32 (mixed precision: "
                                    "projection and refinement in double), 4096 problems sharded over %d GPUs, 11 step sizes, T = 61, inputs of examples/rocket.jl" % world,
                           problems_per_gpu=4096 // world, ms_per_iteration_slowest_rank=slow, ms_per_iteration_fastest_rank=fast,
                           value=n_units / (slow * 1e-3), unit="projected rocket steps/s (all ranks)")
        return dict(error="at least one rank could not run the leg", rank0=r5 if isinstance(r5, dict) and "error" in r5 else None)

    # N > 1 without --gather (the driver's scaling command): after the headline region, the same K steps once more WITH the path's one
    # exchange -- od_allgather_compact over RCCL behind the C ABI after every step -- so that one invocation per N also yields what the
    # collective costs over xGMI.  Reported beside `value`, never in it; a failure here (librccl absent, rendezvous) is recorded, not raised,
    # and a leg that does not finish (a rendezvous that hangs) is cut off by a watchdog on every rank: rank 0 prints the line it has.
    def gather_leg():
        nonlocal comm
        if os.environ.get("OD_BENCH_TEST_HANG_GATHER_LEG"):      # TEST HARNESS ONLY (tests/test_distributed.py): a rendezvous that never returns
            time.sleep(3600)
        try:
            from optimization_dynamics_amd.parallel import Communicator
            box = [None]
            if rank == 0:
                try:
                    box = [Communicator.unique_id(im.lib)]
                except Exception as e:       # noqa: BLE001  (every rank must leave the broadcast below)
                    box = ["error: " + repr(e)[:200]]
            dist.broadcast_object_list(box, src=0)
            if not isinstance(box[0], bytes):
                raise RuntimeError(str(box[0]))
            try:
                comm = Communicator(im, box[0], rank, world)
            except Exception:                # noqa: BLE001
                comm = None
            agreed = torch.tensor([1.0 if comm is not None else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)        # all ranks time the gather, or none does (no rank left waiting in a collective)
            if agreed.item() < 1.0:
                comm = None
                raise RuntimeError("od_comm_create failed on at least one rank")
            e3, k3, _, _, B3 = timed_region("weak")
            return dict(collective="od_allgather_compact: ncclAllGather x 2 behind the C ABI (od_comm_*)", ranks_seen=comm.world,
                        value=world * args.batch * T * args.steps / e3, unit="steps+grads/s", ms_per_step=e3 / args.steps * 1e3,
                        gathered_bytes_per_rank_per_step=8 * (8 * (T + 1) + 40 * T) * B3 * world)
        except Exception as e:           # noqa: BLE001
            return dict(error=repr(e)[:300])
        finally:
            comm = None

    want_gather_leg = world > 1 and comm is None and args.scaling == "weak"
    line = None

    def run_extra_legs():
        """every rank, just before rank 0 prints: -> (with_gather, config5_sharded) blocks (None where a leg does not apply).  The legs run
        under a watchdog: whatever has not finished after the limit is cut off, rank 0 prints the line it has, every rank leaves"""
        if not (want_gather_leg or want_config5_leg):
            return None, None
        import threading
        done = threading.Event()
        limit = float(os.environ.get("OD_BENCH_GATHER_LEG_TIMEOUT", "180"))
        got = {}

        def watchdog():
            if not done.wait(limit):
                if rank == 0 and line is not None:
                    cut = dict(error="did not finish within %.0f s of the extra legs' start (RCCL rendezvous / collective / a rank that fell over); cut off -- the headline figures above are unaffected" % limit)
                    line["with_gather"] = got.get("wg", cut if want_gather_leg else None)
                    line["config5_sharded"] = got.get("c5", cut if want_config5_leg else None)
                    print(json.dumps(line), flush=True)
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        if want_gather_leg:
            got["wg"] = gather_leg()
        if want_config5_leg:
            try:
                got["c5"] = config5_leg()
            except Exception as e:           # noqa: BLE001  (e.g. the agreeing all-reduce on a context an earlier failure poisoned)
                got["c5"] = dict(error=repr(e)[:300])
        done.set()
        return got.get("wg"), got.get("c5")

    units_per_rank = B * T
    total_units = (world * args.batch if args.scaling == "weak" else args.batch) * T
    value = total_units * args.steps / elapsed
    stc = torch.bincount(st.flatten(), minlength=8).tolist()
    it_eval = float(it[0].double().mean().item())
    it_max = int(it.max().item())

    latency_floor_ms = dense_ms = None
    if rank == 0 and world == 1 and not emu and not args.no_cpu_baseline:      # (--no-cpu-baseline = the bare timed region: profiling passes)
        # the latency of ONE trajectory's T sequential knots (64 rollouts: one per wavefront, the chip nearly empty): no batch
        # of any size finishes a step faster than this; and the same step writing the dense fx / fu (od_rollout)
        xs, Us = workload_slice(0, 64, args.batch, T)
        xsd, Usd = torch.tensor(xs, device=dev), torch.tensor(Us, device=dev)
        o64 = None
        for _ in range(3):
            o64 = im.rollout_compact(xsd, Usd, out=o64)[-1]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record(stream); o64 = im.rollout_compact(xsd, Usd, out=o64)[-1]; e1.record(stream); sync()
            ts.append(e0.elapsed_time(e1))
        latency_floor_ms = float(np.median(ts))
        if not args.dense:
            x1f, Uf = workload_slice(0, args.batch, args.batch, T)
            x1fd, Ufd = torch.tensor(x1f, device=dev), torch.tensor(Uf, device=dev)
            od_ = None
            for _ in range(2):
                od_ = im.rollout(x1fd, Ufd, out=od_)[-1]
            ts = []
            for _ in range(5):
                e0.record(stream); od_ = im.rollout(x1fd, Ufd, out=od_)[-1]; e1.record(stream); sync()
                ts.append(e0.elapsed_time(e1))
            dense_ms = float(np.median(ts))
            del od_, x1fd, Ufd

    aux = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not emu:
        # auxiliary, NOT the headline: the same unit as independent knots (no time recursion), which is the
        # regime where the chip is full (reported so the roofline fraction can be read in both regimes)
        Bk = 262144
        rng = np.random.default_rng(123)
        q = np.array([0.0, 0.55, 0.0, 0.5])[:, None]
        q1 = q + rng.normal(0, 0.02, (4, Bk)); q2 = q1 + 0.5 * rng.normal(0, 0.02, (4, Bk))
        Xk = torch.tensor(np.vstack([q1, q2]), device=dev)
        Uk = torch.tensor(np.array([0.0, 9.81 * 3.0 * 0.5 * 0.05])[:, None] + rng.normal(size=(2, Bk)), device=dev)
        im.step_grad(Xk, Uk); torch.cuda.synchronize(dev)
        tk = time.perf_counter()
        for _ in range(5):
            _, _, _, stk, itk = im.step_grad(Xk, Uk)
        torch.cuda.synchronize(dev)
        tk = (time.perf_counter() - tk) / 5
        aux = dict(workload="od_step_grad, hopper, %d independent knots" % Bk, units_per_s=Bk / tk, ms=tk * 1e3,
                   mean_iterations=float(itk[0].double().mean().item()))

    aux_roll = aux_c4 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not emu:
        # auxiliary, NOT the headline: the same rollout step at BASELINE.json's configs[3] (8192 rollouts: 8 lanes per problem) and
        # at a batch that fills the chip (65 536 trajectories x T knots, 64 per wavefront) -- where the roofline fraction of this
        # path stands when the batch is not the limit
        def large(Br):
            xr, Ur = workload_slice(0, Br, Br, T)
            xrd, Urd = torch.tensor(xr, device=dev), torch.tensor(Ur, device=dev)
            orr = im.rollout_compact(xrd, Urd, out=None)[-1]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(3):
                e0.record(stream); r_ = im.rollout_compact(xrd, Urd, out=orr); orr = r_[-1]; e1.record(stream); sync()
                ts.append(e0.elapsed_time(e1))
            return dict(workload="od_rollout_compact, hopper, %d rollouts x T=%d" % (Br, T), ms=float(np.median(ts)), units_per_s=Br * T / (np.median(ts) * 1e-3),
                        mean_iterations=float(r_[3][0].double().mean().item()))
        aux_c4 = large(8192)
        aux_roll = large(65536)

    if rank == 0:
        stats = json.load(open(os.path.join(ROOT, "optimization_dynamics_amd", "csrc", "gen", "stats.json")))["hopper"]
        F, per_iter, grad = algorithmic_flops_per_unit(it_eval, stats)
        ach_tflops = F * units_per_rank / (kernel_ms * 1e-3) / 1e12
        ach_gbs = algorithmic_bytes_per_unit() * units_per_rank / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_note, pmc = measured_traffic(B, T, args.gather or args.dense)
        coop_on = bool(im.lib.cdll.od_uses_cooperative(im._h, B)) if hasattr(im.lib.cdll, "od_uses_cooperative") else False
        line = {
            "metric": "contact-implicit steps+grads/sec, hopper T=%d batch=%d%s" % (T, args.batch, "" if args.scaling == "weak" else " (fixed total, sharded)"),
            "value": value, "unit": "steps+grads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "ranks_seen": (comm.world if comm is not None else dist.get_world_size() if dist is not None else 1), "backend": (dist.get_backend() if dist is not None else "none (single process)"),
            "collective": ("od_allgather_compact: ncclAllGather x 2 behind the C ABI (od_comm_*), ranks_seen = ncclCommCount" if comm is not None else None),
            "with_gather": None,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "hopper (RoboDojo path-following contact), T=%d, batch=%d rollouts %s, "
                                   "h=0.05, kappa_eval=1e-4, kappa_grad=1e-3, r_tol=1e-8; od_rollout_compact = f+fx+fu per knot (q3, dq3/d(q1,q2,u))"
                                   % (T, args.batch, "per GPU" if args.scaling == "weak" else "in total, sharded over the GPUs"),
                       "total_batch": world * args.batch if args.scaling == "weak" else args.batch,
                       "units_per_step_per_gpu": units_per_rank, "parallelism": "rollouts sharded x%d, no collective%s" % (world, " + all-gather(x+, dq3) compact" if args.gather else "")},
            # the governing roof is the fp64 VECTOR-ALU rate (78.6 TFLOP/s, equal to the fp64 matrix peak on MI355X): the
            # path is arithmetic on 4..20-wide systems with no GEMM-shaped work; the one GEMM-shaped kernel of the repository is the
            # Riccati pass of the rocket's iLQR iteration (aux_config_5), which does run on MFMA
            "roofline": {"bound": "fp64-valu", "bound_detail": "fp64 vector ALU roof; algorithmic flops of the reference's dense-LU algorithm (SURVEY.md 8d) over the kernel time -- the device executes a sparse elimination, so this is a useful-work ratio",
                         "achieved": ach_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tflops / FP64_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_note": "%s; algorithmic = %d" % (traffic_note, algorithmic_bytes_per_unit() * units_per_rank),
                         # (od_model_tu.inc: 16 lanes per problem up to 4096 rollouts per launch, 8 lanes up to 16 384, one lane beyond)
                         "kernel": (("k_rollout_state_coop<Coop_hopper> (one problem per 16 lanes)" if B <= 4096 else "k_rollout_state_coop3<Coop3_hopper> (one problem per 8 lanes)")
                                    if coop_on else "k_rollout_state<Model_hopper,double>")
                                   + " (98 %) + k_grad_knots<Model_hopper,double>", "kernel_ms": kernel_ms,
                         "algorithmic_flops_per_unit": F, "mean_iterations_to_kappa_eval": it_eval, "max_iterations": it_max,
                         "hbm_algorithmic_GBps": ach_gbs, "hbm_frac": ach_gbs / HBM_PEAK_GBS,
                         # executed, not algorithmic: wavefront-level fp64 instruction counts x 64 lane slots (PMC record); the
                         # device runs a sparse block elimination, a fraction of the dense-LU count `frac` is priced with
                         "executed_flops_per_unit": (pmc["executed_lane_slot_flops_per_step"] / units_per_rank) if pmc and "executed_lane_slot_flops_per_step" in pmc else None,
                         "executed_frac": (pmc["executed_lane_slot_flops_per_step"] / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if pmc and "executed_lane_slot_flops_per_step" in pmc else None,
                         "valu_issue_util": pmc.get("valu_issue_util") if pmc else None,
                         "latency_floor_ms": latency_floor_ms,
                         "why_not_0.40": "one trajectory is T=100 sequential knots x ~8.2 interior-point iterations x ~870 dependent instructions: %s ms "
                                         "for ANY batch (latency_floor_ms); at batch 4096 every SIMD holds one wavefront of four trajectories, so the step "
                                         "cannot be shorter than that and frac <= %.2f; the fp64 roof is approached by batches that fill the lanes: "
                                         "65 536 rollouts (aux_large_batch_rollouts) and independent knots (aux_independent_knots)" % ("%.2f" % latency_floor_ms if latency_floor_ms else "~2.0",
                                                                     (ach_tflops / FP64_PEAK_TFLOPS) * kernel_ms / latency_floor_ms if latency_floor_ms else 0.20)},
            "dense_fx_fu": {"ms_per_step": dense_ms, "note": "the same step through od_rollout (dense 2nq x 2nq fx and 2nq x nu fu with their constant rows, what the reference's callbacks fill) instead of the compact dq3"},
            "solver_status_counts": {"converged(7)": stc[7], "other": int(sum(stc) - stc[7])},
        }
        if strong is not None:
            line["strong_scaling"] = strong
        if config4 is not None:
            line["config4_sharded"] = config4
        for key_, blk in (("aux_config_4", aux_c4), ("aux_large_batch_rollouts", aux_roll)):
            if blk is not None:
                Fr_, _, _ = algorithmic_flops_per_unit(blk["mean_iterations"], stats)
                blk["algorithmic_tflops"] = Fr_ * blk["units_per_s"] / 1e12
                blk["algorithmic_frac"] = blk["algorithmic_tflops"] / FP64_PEAK_TFLOPS
                line[key_] = blk
        if aux is not None:
            Fk, _, _ = algorithmic_flops_per_unit(aux["mean_iterations"], stats)
            aux["algorithmic_tflops"] = Fk * aux["units_per_s"] / 1e12
            aux["algorithmic_frac"] = aux["algorithmic_tflops"] / FP64_PEAK_TFLOPS     # dense-LU flop count over time: useful work, not hardware utilisation
            line["aux_independent_knots"] = aux
        if not args.no_cpu_baseline and world == 1 and not emu and not args.no_aux_configs:
            # the other BASELINE.json configs with their own roofline and CPU baseline (bench_configs.py)
            import bench_configs
            line.update(bench_configs.all_configs(dev, cpu=True))
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(B, T, seed=0)
            except Exception as e:   # the oracle is a reported baseline, never a dependency of the timed path
                line["cpu_baseline"] = {"value": None, "unit": "steps+grads/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
    wg, c5 = run_extra_legs()
    if rank == 0:
        line["with_gather"] = wg
        if c5 is not None:
            line["config5_sharded"] = c5
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
