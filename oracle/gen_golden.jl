# Produces TRUE reference vectors with the real reference (thowell/optimization_dynamics + RoboDojo.jl +
# IterativeLQR.jl as pinned by its Project.toml).  Cannot run in the build environment (no Julia);
# run it where Julia >= 1.6 and the packages exist:
#
#     julia --project=/path/to/optimization_dynamics oracle/gen_golden.jl tests/golden/inputs tests/golden/reference
#
# Inputs: little-endian Float64 files <model>_X.bin (2nq x B), <model>_U.bin (nu x B) written by
# `python tests/golden/export_inputs.py` from tests/golden/oracle_v1.npz.  Outputs, same layout as the
# npz arrays: <model>_D.bin (2nq x B), <model>_DX.bin (2nq x 2nq x B), <model>_DU.bin (2nq x nu x B).
using OptimizationDynamics
const RoboDojo = OptimizationDynamics.RoboDojo

indir, outdir = ARGS[1], ARGS[2]
mkpath(outdir)

readmat(path, n) = (v = reinterpret(Float64, read(path)); reshape(collect(v), n, :))
writearr(path, a) = write(path, reinterpret(UInt8, vec(Float64.(a))))

configs = Dict(
    "acrobot_impact" => (acrobot_impact, 0.05, r_acrobot_impact_func, rz_acrobot_impact_func, rθ_acrobot_impact_func, 1.0e-4, 1.0e-3),
    "acrobot_nominal" => (acrobot_nominal, 0.05, r_acrobot_nominal_func, rz_acrobot_nominal_func, rθ_acrobot_nominal_func, 1.0, 1.0),
    "cartpole_friction" => (cartpole_friction, 0.05, r_cartpole_friction_func, rz_cartpole_friction_func, rθ_cartpole_friction_func, 1.0e-4, 1.0e-4),
    "cartpole_frictionless" => (cartpole_frictionless, 0.05, r_cartpole_frictionless_func, rz_cartpole_frictionless_func, rθ_cartpole_frictionless_func, 1.0, 1.0),
    "planar_push" => (planarpush, 0.1, r_pp_func, rz_pp_func, rθ_pp_func, 1.0e-4, 1.0e-2),
)

function run_model(name, model, h, r, rz, rθ, κe, κg)
    nq, nu = model.nq, model.nu
    X = readmat(joinpath(indir, name * "_X.bin"), 2nq)
    U = readmat(joinpath(indir, name * "_U.bin"), nu)
    B = size(X, 2)
    im_dyn = ImplicitDynamics(model, h, eval(r), eval(rz), eval(rθ); r_tol=1.0e-8, κ_eval_tol=κe, κ_grad_tol=κg)
    D = zeros(2nq, B); DX = zeros(2nq, 2nq, B); DU = zeros(2nq, nu, B)
    for b = 1:B
        d = zeros(2nq); dx = zeros(2nq, 2nq); du = zeros(2nq, nu)
        f(d, im_dyn, X[:, b], U[:, b], zeros(0))
        fx(dx, im_dyn, X[:, b], U[:, b], zeros(0))
        fu(du, im_dyn, X[:, b], U[:, b], zeros(0))
        D[:, b] = d; DX[:, :, b] = dx; DU[:, :, b] = du
    end
    writearr(joinpath(outdir, name * "_D.bin"), D)
    writearr(joinpath(outdir, name * "_DX.bin"), DX)
    writearr(joinpath(outdir, name * "_DU.bin"), DU)
end

cartpole_friction.friction .= [0.35; 0.35]        # examples/cartpole.jl:21
for (name, c) in configs
    run_model(name, c...)
end

# hopper: residual expressions come from RoboDojo (examples/hopper.jl:38-42)
let hopper = RoboDojo.hopper
    nq, nu = hopper.nq, hopper.nu
    X = readmat(joinpath(indir, "hopper_X.bin"), 2nq); U = readmat(joinpath(indir, "hopper_U.bin"), nu)
    B = size(X, 2)
    im_dyn = ImplicitDynamics(hopper, 0.05, eval(RoboDojo.residual_expr(hopper)), eval(RoboDojo.jacobian_var_expr(hopper)),
        eval(RoboDojo.jacobian_data_expr(hopper)); r_tol=1.0e-8, κ_eval_tol=1.0e-4, κ_grad_tol=1.0e-3, nc=4, nb=2)
    D = zeros(2nq, B); DX = zeros(2nq, 2nq, B); DU = zeros(2nq, nu, B)
    for b = 1:B
        d = zeros(2nq); dx = zeros(2nq, 2nq); du = zeros(2nq, nu)
        f(d, im_dyn, X[:, b], U[:, b], zeros(0)); fx(dx, im_dyn, X[:, b], U[:, b], zeros(0)); fu(du, im_dyn, X[:, b], U[:, b], zeros(0))
        D[:, b] = d; DX[:, :, b] = dx; DU[:, :, b] = du
    end
    writearr(joinpath(outdir, "hopper_D.bin"), D); writearr(joinpath(outdir, "hopper_DX.bin"), DX); writearr(joinpath(outdir, "hopper_DU.bin"), DU)
end
