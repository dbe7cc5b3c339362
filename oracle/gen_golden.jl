# Produces TRUE reference vectors with the real reference (thowell/optimization_dynamics + RoboDojo.jl +
# IterativeLQR.jl as pinned by its Project.toml).  Cannot run in the build environment (no Julia);
# run it where Julia >= 1.6 and the packages exist:
#
#     python tests/golden/export_inputs.py                       # writes tests/golden/inputs/*.bin
#     julia --project=/path/to/optimization_dynamics oracle/gen_golden.jl tests/golden/inputs tests/golden/reference
#     python -m pytest tests/test_reference_golden.py            # oracle vs reference (CPU), -m gpu: HIP path vs reference
#
# Every array is a little-endian Float64 file, column-major, in the layout of tests/golden/oracle_v1.npz.
#
#   mechanical models (f, fx, fu: src/dynamics.jl:81-128)
#     in : <model>_X.bin (2nq x B), <model>_U.bin (nu x B)
#     out: <model>_D.bin (2nq x B), <model>_DX.bin (2nq x 2nq x B), <model>_DU.bin (2nq x nu x B),
#          <model>_IT.bin (3 x B: interior-point iterations of the f, fx and fu solves; -1 if the field is not there),
#          <model>_ST.bin (3 x B: 1 = the solver reported success, 0 = failure, -1 = not observable)
#          <model>_ZG.bin (nz x B: the iterate z the fx solve differentiated at -- grad_sim.ip.z --, NaN if not observable;
#                          lets the comparison arbitrate ill-conditioned gradients in binary128, tests/test_reference_golden.py)
#   rocket (src/models/rocket/dynamics.jl:101-268), u_max = 12.5, h = 0.05 (examples/rocket.jl:16,19)
#     in : rocket_X.bin (12 x B), rocket_U.bin (3 x B)
#     out: rocket_Y.bin, rocket_DX.bin (12 x 12 x B), rocket_DU.bin (12 x 3 x B)            f/fx/fu_rocket
#          rocket_Yp.bin, rocket_DXp.bin, rocket_DUp.bin                                    f/fx/fu_rocket_proj
#          rocket_UP.bin (3 x B), rocket_DP.bin (3 x 3 x B)                                 soc_projection(_gradient)
#   gradient bundle (src/gradient_bundle.jl:87-104) with the exported perturbations eta
#     in : bundle_<model>_X.bin (2nq x B), bundle_<model>_U.bin (nu x B), bundle_<model>_eta.bin ((2nq+nu) x N)
#     out: bundle_<model>_DZ.bin (nq x (2nq+nu) x B)
using OptimizationDynamics
using LinearAlgebra
using Random
const RoboDojo = OptimizationDynamics.RoboDojo

indir, outdir = ARGS[1], ARGS[2]
mkpath(outdir)

readmat(path, n) = (v = reinterpret(Float64, read(path)); reshape(collect(v), n, :))
writearr(path, a) = write(path, reinterpret(UInt8, vec(Float64.(a))))

# iteration count / status of the last solve of a simulator, where the installed RoboDojo exposes them
ip_of(sim) = hasproperty(sim, :ip) ? sim.ip : nothing
iters_of(sim) = (ip = ip_of(sim); ip !== nothing && hasproperty(ip, :iterations) ? Float64(ip.iterations) : -1.0)
function status_of(sim)
    ip = ip_of(sim)
    ip === nothing && return -1.0
    o = ip.opts
    (hasproperty(ip, :r) && hasproperty(ip, :idx)) || return -1.0
    try
        rv = RoboDojo.residual_violation(ip, ip.r)
        kv = RoboDojo.bilinear_violation(ip, ip.r)
        return (rv < o.r_tol && kv < o.κ_tol) ? 1.0 : 0.0
    catch
        return -1.0
    end
end

function run_mech(name, im_dyn, nq, nu)
    X = readmat(joinpath(indir, name * "_X.bin"), 2nq)
    U = readmat(joinpath(indir, name * "_U.bin"), nu)
    B = size(X, 2)
    D = zeros(2nq, B); DX = zeros(2nq, 2nq, B); DU = zeros(2nq, nu, B); IT = fill(-1.0, 3, B); ST = fill(-1.0, 3, B)
    ipg = ip_of(im_dyn.grad_sim)
    nz = (ipg !== nothing && hasproperty(ipg, :z)) ? length(ipg.z) : 0
    ZG = fill(NaN, max(nz, 1), B)
    for b = 1:B
        d = zeros(2nq); dx = zeros(2nq, 2nq); du = zeros(2nq, nu)
        f(d, im_dyn, X[:, b], U[:, b], zeros(0));   IT[1, b] = iters_of(im_dyn.eval_sim); ST[1, b] = status_of(im_dyn.eval_sim)
        fx(dx, im_dyn, X[:, b], U[:, b], zeros(0)); IT[2, b] = iters_of(im_dyn.grad_sim); ST[2, b] = status_of(im_dyn.grad_sim)
        nz > 0 && (ZG[:, b] .= ipg.z)
        fu(du, im_dyn, X[:, b], U[:, b], zeros(0)); IT[3, b] = iters_of(im_dyn.grad_sim); ST[3, b] = status_of(im_dyn.grad_sim)
        D[:, b] = d; DX[:, :, b] = dx; DU[:, :, b] = du
    end
    writearr(joinpath(outdir, name * "_D.bin"), D)
    writearr(joinpath(outdir, name * "_DX.bin"), DX)
    writearr(joinpath(outdir, name * "_DU.bin"), DU)
    writearr(joinpath(outdir, name * "_IT.bin"), IT)
    writearr(joinpath(outdir, name * "_ST.bin"), ST)
    nz > 0 && writearr(joinpath(outdir, name * "_ZG.bin"), ZG)
end

# gradient! with the exported eta.  The constructor sizes q1η / q2η / u1η with the module globals nq, nu (the rocket's
# 12 and 3 after `using`, src/gradient_bundle.jl:79-81), so they are resized to the model's here; nothing else changes.
function run_bundle(name, im_dyn, model)
    p = joinpath(indir, "bundle_" * name * "_X.bin")
    isfile(p) || return
    nq, nu = model.nq, model.nu
    X = readmat(p, 2nq); U = readmat(joinpath(indir, "bundle_" * name * "_U.bin"), nu)
    eta = readmat(joinpath(indir, "bundle_" * name * "_eta.bin"), 2nq + nu)
    N = size(eta, 2); B = size(X, 2)
    gb = OptimizationDynamics.GradientBundle(model, N=N, ϵ=1.0e-4)
    resize!(gb.q1η, nq); resize!(gb.q2η, nq); resize!(gb.u1η, nu)
    for i = 1:N
        gb.ls.η[i] .= eta[:, i]
    end
    DZ = zeros(nq, 2nq + nu, B)
    for b = 1:B
        gb.ls.θ .= 0.0
        DZ[:, :, b] = OptimizationDynamics.gradient!(im_dyn.eval_sim, gb, X[1:nq, b], X[nq+1:2nq, b], U[:, b])
    end
    writearr(joinpath(outdir, "bundle_" * name * "_DZ.bin"), DZ)
end

configs = Dict(
    "acrobot_impact" => (acrobot_impact, 0.05, r_acrobot_impact_func, rz_acrobot_impact_func, rθ_acrobot_impact_func, 1.0e-4, 1.0e-3),
    "acrobot_nominal" => (acrobot_nominal, 0.05, r_acrobot_nominal_func, rz_acrobot_nominal_func, rθ_acrobot_nominal_func, 1.0, 1.0),
    "cartpole_friction" => (cartpole_friction, 0.05, r_cartpole_friction_func, rz_cartpole_friction_func, rθ_cartpole_friction_func, 1.0e-4, 1.0e-4),
    "cartpole_frictionless" => (cartpole_frictionless, 0.05, r_cartpole_frictionless_func, rz_cartpole_frictionless_func, rθ_cartpole_frictionless_func, 1.0, 1.0),
    "planar_push" => (planarpush, 0.1, r_pp_func, rz_pp_func, rθ_pp_func, 1.0e-4, 1.0e-2),
)

cartpole_friction.friction .= [0.35; 0.35]        # examples/cartpole.jl:21
for (name, (model, h, r, rz, rθ, κe, κg)) in configs
    im_dyn = ImplicitDynamics(model, h, eval(r), eval(rz), eval(rθ); r_tol=1.0e-8, κ_eval_tol=κe, κ_grad_tol=κg)
    run_mech(name, im_dyn, model.nq, model.nu)
    run_bundle(name, im_dyn, model)
end

# hopper: residual expressions come from RoboDojo (examples/hopper.jl:38-42)
let hopper = RoboDojo.hopper
    im_dyn = ImplicitDynamics(hopper, 0.05, eval(RoboDojo.residual_expr(hopper)), eval(RoboDojo.jacobian_var_expr(hopper)),
        eval(RoboDojo.jacobian_data_expr(hopper)); r_tol=1.0e-8, κ_eval_tol=1.0e-4, κ_grad_tol=1.0e-3, nc=4, nb=2)
    run_mech("hopper", im_dyn, hopper.nq, hopper.nu)
    run_bundle("hopper", im_dyn, hopper)
end

# rocket (examples/rocket.jl:16-23)
let
    info = RocketInfo(rocket, 12.5, 0.05,
        eval(r_rocket_func), eval(rz_rocket_func), eval(rθ_rocket_func),
        eval(r_proj_func), eval(rz_proj_func), eval(rθ_proj_func))
    X = readmat(joinpath(indir, "rocket_X.bin"), 12); U = readmat(joinpath(indir, "rocket_U.bin"), 3)
    B = size(X, 2)
    Y = zeros(12, B); DX = zeros(12, 12, B); DU = zeros(12, 3, B)
    Yp = zeros(12, B); DXp = zeros(12, 12, B); DUp = zeros(12, 3, B); UP = zeros(3, B); DP = zeros(3, 3, B)
    for b = 1:B
        x, u = X[:, b], U[:, b]
        d = zeros(12); dx = zeros(12, 12); du = zeros(12, 3)
        f_rocket(d, info, x, u, zeros(0)); fx_rocket(dx, info, x, u, zeros(0)); fu_rocket(du, info, x, u, zeros(0))
        Y[:, b] = d; DX[:, :, b] = dx; DU[:, :, b] = du
        f_rocket_proj(d, info, x, u, zeros(0)); fx_rocket_proj(dx, info, x, u, zeros(0)); fu_rocket_proj(du, info, x, u, zeros(0))
        Yp[:, b] = d; DXp[:, :, b] = dx; DUp[:, :, b] = du
        UP[:, b] = copy(OptimizationDynamics.soc_projection(u, info))
        DP[:, :, b] = copy(OptimizationDynamics.soc_projection_gradient(u, info))
    end
    for (n, a) in (("Y", Y), ("DX", DX), ("DU", DU), ("Yp", Yp), ("DXp", DXp), ("DUp", DUp), ("UP", UP), ("DP", DP))
        writearr(joinpath(outdir, "rocket_" * n * ".bin"), a)
    end
    # The projection's ITERATE PATH: z after k = 1 .. KP iterations of soc_projection's interior-point solve (the same solve cut
    # short by max_iter = k).  Its line search compares rounding noise from the first full step on (DESIGN.md section 5,
    # "line-search ties"): implementations can accept different step lengths and end on kappa_tol-level different points.  With
    # the path on file the accepted step of every iteration of the reference is known (z_k - z_{k-1} against the direction), and
    # tests/test_reference_golden.py compares paths instead of end points.  (Needs the solver options to be mutable, as
    # RoboDojo's InteriorPointOptions are; skipped with a note otherwise.)
    try
        KP = 14
        PATH = fill(NaN, 10, KP, B)
        keep = info.ip_proj.opts.max_iter
        for b = 1:B, k = 1:KP
            info.ip_proj.opts.max_iter = k
            OptimizationDynamics.soc_projection(U[:, b], info)
            PATH[:, k, b] .= info.ip_proj.z
        end
        info.ip_proj.opts.max_iter = keep
        writearr(joinpath(outdir, "rocket_PATH.bin"), PATH)
    catch e
        println("projection iterate path not written: ", e)
    end
end
# iLQR.solve! on the acrobot swing-up of examples/acrobot.jl (`:impact` model, its objective, terminal constraint, solver options
# :97-107 and initial controls :90-91) -- the per-iteration record that pins the decisions of an AL-iLQR implementation
# (oracle/ilqr_np.py::solve, od_ilqr_*): the solve is run K times, cut short after k = 1 .. K inner iterations of the first
# augmented-Lagrangian round (max_iter = k, max_al_iter = 1), and after each the objective (eval_obj: without multiplier terms),
# the merit as the solver holds it (s_data.obj[1]), the iteration count it reports, the trajectory's terminal violation and the
# controls are written:
#     acrobot_ilqr_U0.bin (1 x 100) the initial controls;  acrobot_ilqr_trace.bin (5 x K): k, s_data.iter[1], eval_obj, s_data.obj[1],
#     |x_T - goal|_inf;  acrobot_ilqr_U.bin (1 x 100 x K) the controls after k iterations;  and acrobot_ilqr_full.bin (4): iterations,
#     eval_obj, merit, violation of the uncut solve (max_iter = 50, max_al_iter = 20).
# From consecutive columns the accepted step size of iteration k follows (u_k - u_{k-1} against the feed-forward direction), the
# cost decrease is there directly.  Only call sites the reference's own example makes are used (examples/acrobot.jl:33-37,74-76,
# 85-87,92,97-121); if the installed IterativeLQR differs, the block is skipped with a note.
# Both modes of the file: `:impact` (joint limits; prefix acrobot_ilqr) and `:nominal` (:11-12, what the file ends up in: no contact,
# smooth dynamics -- two correct implementations of one rule set should not part at all there; prefix acrobot_nominal_ilqr).
for (prefix, mdl, rf, rzf, rθf, κe, κg) in (("acrobot_ilqr", acrobot_impact, r_acrobot_impact_func, rz_acrobot_impact_func, rθ_acrobot_impact_func, 1.0e-4, 1.0e-3),
                                              ("acrobot_nominal_ilqr", acrobot_nominal, r_acrobot_nominal_func, rz_acrobot_nominal_func, rθ_acrobot_nominal_func, 1.0, 1.0))
try
    iLQR = OptimizationDynamics.IterativeLQR
    h = 0.05; T = 101
    im_dyn = ImplicitDynamics(mdl, h, eval(rf), eval(rzf), eval(rθf); r_tol=1.0e-8, κ_eval_tol=κe, κ_grad_tol=κg, no_friction=true)
    nx = 2 * acrobot_impact.nq; nu = acrobot_impact.nu
    ilqr_dyn = iLQR.Dynamics((d, x, u, w) -> f(d, im_dyn, x, u, w), (dx, x, u, w) -> fx(dx, im_dyn, x, u, w),
                             (du, x, u, w) -> fu(du, im_dyn, x, u, w), nx, nx, nu)
    model = [ilqr_dyn for t = 1:T-1]
    x1 = zeros(4); xT = [π; 0.0; π; 0.0]
    objt(x, u, w) = 0.5 * 0.1 * sum(((x[3:4] - x[1:2]) ./ h) .^ 2) + 0.5 * sum(u .^ 2)
    objT(x, u, w) = 0.5 * 0.1 * sum(((x[3:4] - x[1:2]) ./ h) .^ 2)
    obj = [[iLQR.Cost(objt, nx, nu) for t = 1:T-1]..., iLQR.Cost(objT, nx, 0)]
    terminal_con(x, u, w) = x - xT
    cons = [[iLQR.Constraint() for t = 1:T-1]..., iLQR.Constraint(terminal_con, nx, 0)]
    Random.seed!(1)
    ū = [1.0e-3 * randn(nu) for t = 1:T-1]
    writearr(joinpath(outdir, prefix * "_U0.bin"), reshape(vcat(ū...), 1, T - 1))
    function run(max_iter, max_al_iter)
        x̄ = iLQR.rollout(model, x1, ū)
        solver = iLQR.solver(model, obj, cons, opts=iLQR.Options(linesearch=:armijo, α_min=1.0e-5, obj_tol=1.0e-5, grad_tol=1.0e-5,
            max_iter=max_iter, max_al_iter=max_al_iter, con_tol=0.001, ρ_init=1.0, ρ_scale=10.0, verbose=false))
        iLQR.initialize_controls!(solver, ū)
        iLQR.initialize_states!(solver, x̄)
        iLQR.reset!(solver.s_data)
        iLQR.solve!(solver)
        x_sol, u_sol = iLQR.get_trajectory(solver)
        J = iLQR.eval_obj(solver.m_data.obj.costs, solver.m_data.x, solver.m_data.u, solver.m_data.w)
        return Float64(solver.s_data.iter[1]), J, solver.s_data.obj[1], norm(x_sol[T] - xT, Inf), vcat(u_sol[1:T-1]...)
    end
    K = 30
    TR = zeros(5, K); UU = zeros(1, T - 1, K)
    for k = 1:K
        it, J, M, v, u = run(k, 1)
        TR[:, k] = [k, it, J, M, v]; UU[1, :, k] = u
    end
    writearr(joinpath(outdir, prefix * "_trace.bin"), TR)
    writearr(joinpath(outdir, prefix * "_U.bin"), UU)
    it, J, M, v, u = run(50, 20)
    writearr(joinpath(outdir, prefix * "_full.bin"), reshape([it, J, M, v], 4, 1))
catch e
    println(prefix, ": iLQR trace not written: ", e)
end
end
println("reference vectors written to ", outdir)
