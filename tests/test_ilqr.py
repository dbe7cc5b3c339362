"""iLQR around the path: backward pass against the numpy restatement, forward pass
consistency, and a decreasing cost on cartpole with joint friction -- CPU tier on the host emulation,
GPU tier through the shipped library."""
import pytest
import torch

import ilqr_checks as C


def test_backward_forward_cpu(oracle, emu_lib):
    C.check_backward_and_forward(oracle, emu_lib, "cpu")


def test_solver_decreases_cost_cpu(emu_lib):
    C.check_solver_decreases_cost(emu_lib, "cpu", B=4, T=20)


def test_one_bad_trajectory_does_not_hurt_the_batch_cpu(emu_lib):
    C.check_one_bad_trajectory_does_not_hurt_the_batch(emu_lib, "cpu")


@pytest.mark.gpu
def test_one_bad_trajectory_does_not_hurt_the_batch_gpu(gpu_lib):
    C.check_one_bad_trajectory_does_not_hurt_the_batch(gpu_lib, "cuda:0", B=16, T=30)


@pytest.mark.gpu
def test_backward_every_size_instantiation_gpu(gpu_lib):
    C.check_backward_sizes(gpu_lib, "cuda:0")
    C.check_backward_sizes(gpu_lib, "cuda:0", sizes=((12, 3),), batches=(1030, 2047, 2049), T=7)      # the matrix-core kernel at 8 and 16 trajectories per workgroup, ragged
    for T in (1, 2, 3):                                                                               # ... and its prologue / prefetch at the shortest horizons
        C.check_backward_sizes(gpu_lib, "cuda:0", sizes=((12, 3), (8, 2), (4, 1)), batches=(3, 70), T=T)


def test_backward_sizes_cpu(emu_lib):
    C.check_backward_sizes(emu_lib, "cpu", sizes=((12, 3), (6, 2)), batches=(5,), T=4)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_backward_retry_gpu(gpu_lib, dtype):
    C.check_backward_retry(gpu_lib, "cuda:0", dtype=dtype)


def test_backward_retry_cpu(emu_lib):
    C.check_backward_retry(emu_lib, "cpu", B=6, T=8)


@pytest.mark.gpu
def test_backward_trajectories_per_workgroup_gpu(gpu_lib):
    C.check_backward_trajectories_per_workgroup(gpu_lib, "cuda:0")


@pytest.mark.gpu
def test_backward_forward_gpu(oracle, gpu_lib):
    C.check_backward_and_forward(oracle, gpu_lib, "cuda:0")


@pytest.mark.gpu
def test_solver_decreases_cost_gpu(gpu_lib):
    J0, Jf = C.check_solver_decreases_cost(gpu_lib, "cuda:0", B=256, T=50)


def test_rocket_projection_ilqr_cpu(oracle, emu_lib):
    """BASELINE config 5 (reduced): rocket thrust-cone SOCP step inside the iLQR loop with implicit gradients"""
    C.check_rocket_ilqr(oracle, emu_lib, "cpu", B=3, T=15)


@pytest.mark.gpu
def test_rocket_projection_ilqr_gpu_f64(oracle, gpu_lib):
    C.check_rocket_ilqr(oracle, gpu_lib, "cuda:0", B=64, T=40)


@pytest.mark.gpu
def test_rocket_projection_ilqr_gpu_f32(oracle, gpu_lib):
    import torch
    C.check_rocket_ilqr(oracle, gpu_lib, "cuda:0", B=64, T=40, dtype=torch.float32)


def test_device_iteration_cpu(emu_lib):
    """od_ilqr_solve against the loop composed on the host, cartpole with joint friction (terminal equality constraints:
    two augmented-Lagrangian rounds) and the rocket with the thrust-cone projection in both precisions"""
    C.check_device_iteration(emu_lib, "cpu", "cartpole")
    C.check_device_iteration(emu_lib, "cpu", "rocket", B=3, T=10, max_iter=5, max_al_iter=1)
    import torch
    C.check_device_iteration(emu_lib, "cpu", "rocket", B=3, T=10, max_iter=5, max_al_iter=1, dtype=torch.float32)


def test_device_iteration_constrained_cpu(emu_lib):
    """stage and terminal affine constraints, inequalities by active set (od_ilqr_set_constraints)"""
    C.check_device_iteration_constrained(emu_lib, "cpu", "cartpole")
    C.check_device_iteration_constrained(emu_lib, "cpu", "rocket", B=3, T=10, max_iter=4, max_al_iter=3, expect_feasible=False)   # (0.5 s: cannot reach the goal)


def test_batch_independence_cpu(emu_lib):
    C.check_batch_independence(emu_lib, "cpu", "cartpole")
    C.check_batch_independence(emu_lib, "cpu", "rocket", B=3, T=10, max_iter=4, max_al_iter=3, pick=(1,))


@pytest.mark.gpu
def test_batch_independence_gpu(gpu_lib):
    import torch
    C.check_batch_independence(gpu_lib, "cuda:0", "cartpole", B=40, pick=(0, 17, 39))
    C.check_batch_independence(gpu_lib, "cuda:0", "rocket", B=70, T=30, max_iter=5, max_al_iter=3, pick=(0, 69), dtype=torch.float32)


@pytest.mark.gpu
def test_device_iteration_constrained_gpu(gpu_lib):
    import torch
    # (eight multiplier rounds: the penalty reaches 1e8 and with it the sensitivity of the iteration to the last bits of the costs,
    # which the cost kernels and torch round differently on the device -- 1e-6 here, 1e-9 / bit for bit on the host build)
    C.check_device_iteration_constrained(gpu_lib, "cuda:0", "cartpole", B=64, T=25, max_iter=15, max_al_iter=8, tol=1e-6)
    C.check_device_iteration_constrained(gpu_lib, "cuda:0", "rocket", B=128, T=30, max_iter=5, max_al_iter=3, expect_feasible=False)
    C.check_device_iteration_constrained(gpu_lib, "cuda:0", "rocket", B=128, T=30, max_iter=5, max_al_iter=3, dtype=torch.float32, expect_feasible=False)


@pytest.mark.gpu
def test_device_iteration_gpu(gpu_lib):
    import torch
    C.check_device_iteration(gpu_lib, "cuda:0", "cartpole", B=64, T=40)
    C.check_device_iteration(gpu_lib, "cuda:0", "rocket", B=200, T=30, max_iter=6, max_al_iter=1)
    C.check_device_iteration(gpu_lib, "cuda:0", "rocket", B=200, T=30, max_iter=6, max_al_iter=1, dtype=torch.float32)
    # the matrix-core Riccati kernel at 8 and 16 trajectories per workgroup with the objective's constant Hessians (the solver) against
    # the same kernel with per-knot Hessians (the host-composed loop), which switches its workgroup size at other batch sizes
    for B in (1100, 2100):
        C.check_device_iteration(gpu_lib, "cuda:0", "rocket", B=B, T=12, max_iter=5, max_al_iter=1)
    C.check_device_iteration(gpu_lib, "cuda:0", "rocket", B=2100, T=12, max_iter=5, max_al_iter=1, dtype=torch.float32)


def test_reused_forward_states_cpu(emu_lib):
    C.check_reused_forward_states(emu_lib, "cpu", B=4, T=15)


@pytest.mark.gpu
def test_reused_forward_states_gpu(gpu_lib):
    C.check_reused_forward_states(gpu_lib, "cuda:0")


def test_quad_cost_cpu(emu_lib):
    C.check_quad_cost(emu_lib, "cpu")


@pytest.mark.gpu
def test_quad_cost_gpu(gpu_lib):
    C.check_quad_cost(gpu_lib, "cuda:0")


def test_config5_inputs_cpu(oracle, emu_lib):
    """the inputs of examples/rocket.jl (BASELINE config 5) through the whole chain on the host build, a handful of problems"""
    C.check_config5(oracle, emu_lib, "cpu", B=3, iters=3)


@pytest.mark.gpu
def test_config5_rocket_projection_ilqr_as_stated(oracle, gpu_lib):
    """BASELINE config 5 at its size: T = 61, u_max = 12.5, the example's x1 / objective / initial controls, 1024 problems x 11 step
    sizes, double and single precision"""
    C.check_config5(oracle, gpu_lib, "cuda:0", B=1024, iters=12)


def test_config5_rocket_projection_ilqr_host_build(oracle, emu_lib):
    """the CPU twin of test_config5_rocket_projection_ilqr_as_stated: the same checks (chained rollouts against the oracle's own chain, the
    Riccati pass against numpy, device iteration against the host-composed loop, single against double precision) at 64 problems"""
    C.check_config5(oracle, emu_lib, "cpu", B=64, iters=6)


def test_rocket_example_with_its_constraints_cpu(emu_lib):
    """examples/rocket.jl `:projection` in full (stage inequality, terminal box and equalities) through od_ilqr_solve, one problem"""
    C.check_rocket_example(emu_lib, "cpu", B=1)


@pytest.mark.gpu
def test_rocket_example_with_its_constraints_gpu(gpu_lib):
    import torch
    C.check_rocket_example(gpu_lib, "cuda:0", B=64)
    C.check_rocket_example(gpu_lib, "cuda:0", B=64, dtype=torch.float32)


def test_rocket_example_as_shipped_nominal_cpu(emu_lib):
    """examples/rocket.jl with the mode the file ends up in (`:nominal`, :11-12): thrust limits as stage constraints, no projection"""
    C.check_rocket_example_nominal(emu_lib, "cpu", B=4)


@pytest.mark.gpu
def test_rocket_example_as_shipped_nominal_gpu(gpu_lib):
    import json, os
    r = C.check_rocket_example_nominal(gpu_lib, "cuda:0", B=64, need=0.4)      # (measured: 62 % of 64, 60 % of 1024 starts)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/rocket_example_nominal.json", "w") as f:
        json.dump(r, f, indent=1)


def test_reference_examples_on_the_device_cpu(emu_lib):
    """examples/cartpole.jl (frictionless, the file's default) and examples/planar_push.jl `:translate` through od_ilqr_solve, one problem each (host build)"""
    C.check_reference_example(emu_lib, "cpu", "cartpole:frictionless")
    C.check_reference_example(emu_lib, "cpu", "planar_push:translate")


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["cartpole:frictionless", "planar_push:rotate", "planar_push:translate"])
def test_reference_examples_on_the_device_gpu(gpu_lib, which):
    C.check_reference_example(gpu_lib, "cuda:0", which, B=32, need=0.9)


def test_cartpole_friction_example_on_the_device_cpu(emu_lib):
    """examples/cartpole.jl `:friction` (:11, joint friction 0.35, kappa_eval 1e-4 / kappa_grad 1e-3): the swing-up through the friction
    cones is sensitive to its start (the example's own comment at :77 retunes the first control per friction coefficient); of starts
    perturbed by 1e-2 more than half reach the goal to con_tol, each an independent solve"""
    C.check_reference_example(emu_lib, "cpu", "cartpole:friction", B=8, need=0.5)


@pytest.mark.gpu
def test_cartpole_friction_example_on_the_device_gpu(gpu_lib):
    # (measured: 42 % of 64 and 44 % of 1024 starts, the example's own start among them -- profiles/r5_examples_device.json)
    C.check_reference_example(gpu_lib, "cuda:0", "cartpole:friction", B=64, need=0.3)


def test_hopper_example_on_the_device_cpu(emu_lib):
    """examples/hopper.jl's gait problem with the initial configuration fixed, through od_ilqr_solve, one problem (host build)"""
    C.check_hopper_example(emu_lib, "cpu", B=1)


@pytest.mark.gpu
def test_hopper_example_on_the_device_gpu(gpu_lib):
    C.check_hopper_example(gpu_lib, "cuda:0", B=64, need=0.9)


# ---- od_ilqr_solve against the independent numpy AL-iLQR driven by the oracle's dynamics (oracle/ilqr_np.py::solve): every decision ------------
@pytest.mark.parametrize("case", ["cartpole", "cartpole_constrained", "rocket", "rocket_projected"])
def test_solver_decisions_against_the_numpy_oracle_cpu(oracle, emu_lib, case):
    C.check_against_numpy_oracle(oracle, emu_lib, "cpu", case, B=4)


def test_acrobot_solver_decisions_against_the_numpy_oracle_cpu(oracle, emu_lib):
    C.check_against_numpy_oracle(oracle, emu_lib, "cpu", "acrobot", B=1)


def test_acrobot_as_shipped_solver_decisions_against_the_numpy_oracle_cpu(oracle, emu_lib):
    """examples/acrobot.jl in the mode the file ends up in (`:nominal`, :11-12: no joint limits): every decision of all ~300 iterations"""
    st = C.check_against_numpy_oracle(oracle, emu_lib, "cpu", "acrobot_nominal", B=1)
    assert st["agreeing_iterations"] == st["iterations_oracle"] and st["iterations_oracle"][0] > 100


@pytest.mark.gpu
def test_solver_decisions_against_the_numpy_oracle_gpu(oracle, gpu_lib):
    """cartpole with two augmented-Lagrangian rounds, the constrained cartpole (stage and terminal rows), the acrobot swing-up of
    examples/acrobot.jl (with joint limits, and `:nominal` as the file ships: all 303 iterations) and the rocket in double precision
    (with and without the thrust-cone projection), 8 problems each (2 for `:nominal` in this test): accepted
    step index, regularisation, penalty and cost of every iteration, final trajectory and flags; gpurun_out/ilqr_oracle_parity.json"""
    import json
    import os
    out = [C.check_against_numpy_oracle(oracle, gpu_lib, "cuda:0", case, B=(2 if case == "acrobot_nominal" else 8))       # (303 iterations of numpy per problem; 8 problems: profiles/r5_ilqr_oracle_parity_acrobot_nominal_8.json)
           for case in ("cartpole", "cartpole_constrained", "acrobot", "acrobot_nominal", "rocket", "rocket_projected")]
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "ilqr_oracle_parity.json"), "w"), indent=1)


# ---- examples/hopper.jl as shipped: parameter stage (initial configurations optimised), generated nonlinear constraint ------------------------
def test_hopper_example_full_on_the_device_cpu(oracle, emu_lib):
    """host build of the product sources: one problem, every decision against the numpy oracle on the reference's own formulation"""
    st = C.check_hopper_example_full(oracle, emu_lib, "cpu", B=1, n_oracle=1)
    assert st["agreeing_iterations"][0] == st["iterations_oracle"][0] == st["iterations"]


@pytest.mark.gpu
def test_hopper_example_full_on_the_device_gpu(oracle, gpu_lib):
    """examples/hopper.jl as shipped -- the first stage of its own dimensions (8 -> 16 states, 10 controls), nonlinear foot-position
    constraints, the terminal constraint coupled with the optimised initial configurations -- through od_ilqr_solve with no host
    synchronisation per iteration: 64 problems, the first four against the numpy oracle decision by decision"""
    import json
    import os
    st = C.check_hopper_example_full(oracle, gpu_lib, "cuda:0", B=64, n_oracle=4)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(st, open(os.path.join(d, "hopper_example_full.json"), "w"), indent=1)
    assert 25 <= st["iterations"] <= 60


@pytest.mark.parametrize("gait", [2, 3])
def test_hopper_example_other_gaits_on_the_device_cpu(oracle, emu_lib, gait):
    """examples/hopper.jl:190-203: GAIT 2 / GAIT 3 (other cost weights; animations/hopper_gait_2.gif, _3.gif of the reference)"""
    st = C.check_hopper_example_full(oracle, emu_lib, "cpu", B=1, n_oracle=1, gait=gait)
    assert st["agreeing_iterations"][0] == st["iterations_oracle"][0] == st["iterations"]


@pytest.mark.gpu
@pytest.mark.parametrize("gait", [2, 3])
def test_hopper_example_other_gaits_on_the_device_gpu(oracle, gpu_lib, gait):
    import json
    import os
    st = C.check_hopper_example_full(oracle, gpu_lib, "cuda:0", B=64, n_oracle=2, gait=gait)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(st, open(os.path.join(d, "hopper_example_full_gait%d.json" % gait), "w"), indent=1)


# ---- the forward pass leaves a candidate at its first failed knot ----------------------------------------------------------------------------------
def test_forward_pass_early_exit_cpu(emu_lib):
    assert C.check_forward_pass_early_exit(emu_lib, "cpu") > 0


@pytest.mark.gpu
def test_forward_pass_early_exit_gpu(gpu_lib):
    assert C.check_forward_pass_early_exit(gpu_lib, "cuda:0", B=64) > 0


# ---- examples/planar_push.jl with GB = true: the gradient bundle as the solver's linearisation ---------------------------------------------------
def test_gradient_bundle_linearisation_on_the_device_cpu(oracle, emu_lib):
    C.check_bundle_linearisation(oracle, emu_lib, "cpu", mode="rotate", B=2, n_oracle=1)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rotate", "translate"])
def test_gradient_bundle_linearisation_on_the_device_gpu(oracle, gpu_lib, mode):
    import json
    import os
    st = C.check_bundle_linearisation(oracle, gpu_lib, "cuda:0", mode=mode, B=32, n_oracle=2, need=0.9)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(st, open(os.path.join(d, "planar_push_gb_%s.json" % mode), "w"), indent=1)


def test_constraint_generator_builds_a_new_constraint(tmp_path):
    """python -m optimization_dynamics_amd.codegen --add-constraint: a user's sympy constraint becomes device code (into a scratch root)"""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = tmp_path / "circle.py"
    spec.write_text("from optimization_dynamics_amd.codegen.constraints import ConstraintSpec\n"
                    "def constraint():\n"
                    "    return ConstraintSpec('unit_circle', 8, 1, lambda x, p: [x[0]**2 + x[1]**2 - p[0]**2, x[4] - x[0]], 'test')\n")
    scratch = tmp_path / "root"
    (scratch / "optimization_dynamics_amd" / "csrc" / "gen").mkdir(parents=True)
    (scratch / "optimization_dynamics_amd" / "codegen" / "user_models").mkdir(parents=True)
    out = subprocess.run([sys.executable, "-m", "optimization_dynamics_amd.codegen", "--add-constraint", str(spec), "--root", str(scratch)],
                         cwd=root, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    hdr = (scratch / "optimization_dynamics_amd" / "csrc" / "gen" / "con_unit_circle.h").read_text()
    assert "struct Con_unit_circle" in hdr and "NC = 2, NX = 8, NP = 1" in hdr
    lst = (scratch / "optimization_dynamics_amd" / "csrc" / "gen" / "con_list.h").read_text()
    assert "X(hopper_foot, 0)" in lst and "X(unit_circle, 1)" in lst
    # the generated code compiles and evaluates: c and dc/dx against numpy (the header includes "../od_math.h" like every generated header)
    import shutil
    shutil.copy(os.path.join(root, "optimization_dynamics_amd", "csrc", "od_math.h"), scratch / "optimization_dynamics_amd" / "csrc" / "od_math.h")
    src = tmp_path / "t.cpp"
    src.write_text('#include <cstdio>\n#include <hip/hip_runtime.h>\n#include "gen/con_unit_circle.h"\n'
                   'int main() { double x[8] = {0.6, 0.9, 0, 0, 0.1, 0, 0, 0}, p[1] = {1.5}, c[2], cx[16];\n'
                   '  od::Con_unit_circle::eval<double>(x, p, c, cx); printf("%.17g %.17g %.17g %.17g %.17g %.17g\\n", c[0], c[1], cx[0], cx[2], cx[1], cx[9]); }\n')
    exe = tmp_path / "t"
    cc = subprocess.run(["g++", "-std=c++17", "-I", os.path.join(root, "tests", "host_emu"), "-I", os.path.join(root, "optimization_dynamics_amd", "csrc"),
                         "-I", str(scratch / "optimization_dynamics_amd" / "csrc"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    v = [float(t) for t in subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()]
    assert abs(v[0] - (0.36 + 0.81 - 2.25)) < 1e-15 and abs(v[1] - (0.1 - 0.6)) < 1e-15
    assert abs(v[2] - 1.2) < 1e-15 and abs(v[3] - 1.8) < 1e-15 and v[4] == -1.0 and v[5] == 1.0
