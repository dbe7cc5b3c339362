# ccall shim over libod_mi355x.so that keeps the reference's exported API (src/OptimizationDynamics.jl:32-34,75-79):
#
#   ImplicitDynamics(model, h, r, rz, rθ; T, r_tol, κ_eval_tol, κ_grad_tol, no_impact, no_friction, n, m, d, nc, nb, info)
#   f / fx / fu (src/dynamics.jl:81-128)             state_to_configuration (:131-145)
#   GradientBundle(model; N, ϵ), fx_gb / fu_gb (src/gradient_bundle.jl:26-147)
#   RocketInfo(rocket, u_max, h, r..., r_proj...), f_rocket / fx_rocket / fu_rocket, f/fx/fu_rocket_proj,
#   soc_projection, soc_projection_gradient (src/models/rocket/dynamics.jl:12-268)
#
# Every scalar callback is one ccall with host vectors (od_*_host: copy in, launch, copy out, synchronise), so neither
# AMDGPU.jl nor any device array is needed; the batched entry points at the end take device arrays.
# The generated residual functions the reference passes around (r, rz, rθ) are accepted and ignored: the device code
# was generated from the same residual statements (optimization_dynamics_amd/codegen).
# NOT exercised in the build environment (no Julia there); tests/test_julia_shim_calls.py replays, through ctypes,
# the exact call sequence of every function below.  See INTEGRATION.md.
module OptimizationDynamicsMI355X

using LinearAlgebra

export ImplicitDynamics, f, fx, fu, state_to_configuration, GradientBundle, fx_gb, fu_gb,
       RocketInfo, f_rocket, fx_rocket, fu_rocket, f_rocket_proj, fx_rocket_proj, fu_rocket_proj,
       soc_projection, soc_projection_gradient, ffxfu!,
       od_step_grad!, od_rollout!, od_rollout_compact!, od_rocket!, od_soc_project!, od_step_full!, model_indices,
       ILQRSolver, set_constraints!, set_parameter_stage!, set_gradient_bundle!, get_status!, get_trace!, initialize!, iterate!, al_update!,
       solve!, get_trajectory!,
       Communicator, comm_unique_id, allgather_compact!, allgather!

const LIB = get(ENV, "OD_MI355X_LIB", joinpath(@__DIR__, "..", "optimization_dynamics_amd", "libod_mi355x.so"))

# include/od_mi355x.h: OD_ABI_VERSION this shim was written against (the struct mirrors below, the defaults the wrappers rely on)
const ABI_VERSION = 101
function __init__()
    got = ccall((:od_version, LIB), Cint, ())
    got == ABI_VERSION || error("libod_mi355x reports ABI version $got, this shim was written against $ABI_VERSION: rebuild the library")
end

const MODEL_IDS = Dict(:acrobot_impact => 0, :acrobot_nominal => 1, :cartpole_friction => 2,
                       :cartpole_frictionless => 3, :planarpush => 4, :rocket => 5,
                       :rocket_projection => 6, :hopper => 7)

# built-in ids, or -- for a model added with `python -m optimization_dynamics_amd.codegen --add` -- the library's registry
function model_id(sym::Symbol)
    haskey(MODEL_IDS, sym) && return MODEL_IDS[sym]
    id = ccall((:od_model_id, LIB), Cint, (Cstring,), string(sym))
    id < 0 && error("unknown model $sym; built in: $(keys(MODEL_IDS))")
    return Int(id)
end

struct ODOptions                      # od_options (include/od_mi355x.h)
    r_tol::Cdouble; kappa_eval_tol::Cdouble; kappa_grad_tol::Cdouble
    max_iter::Cint; max_ls::Cint
    eps_min::Cdouble; kappa_reg::Cdouble; gamma_reg::Cdouble; undercut::Cdouble
end

function check(rc)
    rc == 0 && return nothing
    error("libod_mi355x: " * unsafe_string(ccall((:od_last_error, LIB), Cstring, ())))
end

# model argument: a Symbol of MODEL_IDS, or the reference's model object (matched by its type name / fields)
function model_symbol(model)
    model isa Symbol && return model
    n = lowercase(string(nameof(typeof(model))))
    occursin("hopper", n) && return :hopper
    occursin("planarpush", n) && return :planarpush
    occursin("rocket", n) && return :rocket
    if occursin("acrobot", n)
        return (hasproperty(model, :nc) && model.nc == 0) ? :acrobot_nominal : :acrobot_impact
    elseif occursin("cartpole", n)
        return (hasproperty(model, :friction) && length(model.friction) > 0) ? :cartpole_friction : :cartpole_frictionless
    end
    error("unknown model $(typeof(model)); pass one of $(keys(MODEL_IDS))")
end

mutable struct ImplicitDynamics{I}
    h::Ptr{Cvoid}
    model::Any
    nq::Int; nu::Int
    n::Int; m::Int; d::Int; nc::Int; nb::Int        # bookkeeping keywords of the reference constructor
    dx_buf::Matrix{Float64}; du_buf::Matrix{Float64}
    idx_q1::Vector{Int}; idx_q2::Vector{Int}; idx_u1::Vector{Int}
    info::I
end

"ImplicitDynamics(model, h, r, rz, rθ; ...) -- src/dynamics.jl:51-79 (option preset of get_simulator, :16-33)"
function ImplicitDynamics(model, h, r_func=nothing, rz_func=nothing, rθ_func=nothing;
        T=1, r_tol=1.0e-8, κ_eval_tol=1.0e-6, κ_grad_tol=1.0e-6,
        no_impact=false, no_friction=false,
        n=nothing, m=nothing, d=0, nc=nothing, nb=nothing, info=nothing)
    sym = model_symbol(model)
    id = model_id(sym)
    dims = [Ref{Cint}(0) for _ in 1:5]
    check(ccall((:od_model_dims, LIB), Cint, (Cint, Ref{Cint}, Ref{Cint}, Ref{Cint}, Ref{Cint}, Ref{Cint}), id, dims...))
    nq, nu = Int(dims[1][]), Int(dims[2][])
    o = Ref(ODOptions(r_tol, κ_eval_tol, κ_grad_tol, 100, 25, 0.25, 1.0e-3, 0.1, Inf))   # src/dynamics.jl:25-33
    hd = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:od_create, LIB), Cint, (Cint, Cint, Ref{ODOptions}, Cdouble, Ref{Ptr{Cvoid}}), id, 0, o, Float64(h), hd))
    # n, m default to 2nq, nu (:54); nc / nb only size the reference's contact-force buffers (:36-46, :58-59):
    # no_impact => nc = 0, no_friction => nb = 0.  They are kept for callers that read them back.
    nc_ = no_impact ? 0 : (nc === nothing ? (hasproperty(model, :nc) ? model.nc : 0) : nc)
    nb_ = no_friction ? 0 : (nb === nothing ? 0 : nb)
    im = ImplicitDynamics(hd[], model, nq, nu, n === nothing ? 2nq : n, m === nothing ? nu : m, d, nc_, nb_,
                          zeros(2nq, 2nq), zeros(2nq, nu),
                          collect(1:nq), collect(nq .+ (1:nq)), collect(2nq .+ (1:nu)), info)
    if hasproperty(model, :friction) && length(model.friction) > 0     # friction_coefficients(model), cartpole/simulator_friction.jl:1
        set_friction!(im, Float64.(collect(model.friction)))
    end
    finalizer(x -> ccall((:od_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), im)
    return im
end

set_friction!(im::ImplicitDynamics, μ::Vector{Float64}) =
    check(ccall((:od_set_friction, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint), im.h, μ, length(μ)))

# the reference mutates model.friction after construction (examples/cartpole.jl:21): re-read it on every call
function sync_friction!(im::ImplicitDynamics)
    if hasproperty(im.model, :friction) && length(im.model.friction) > 0
        set_friction!(im, Float64.(collect(im.model.friction)))
    end
end

"f(d, model, x, u, w) -- src/dynamics.jl:81-94"
function f(d, im::ImplicitDynamics, x, u, w)
    sync_friction!(im)
    check(ccall((:od_f_host, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), im.h, x, u, d))
    return d
end

function write_fx!(dx, im::ImplicitDynamics)      # the three blocks the reference writes (:105-111)
    for i = 1:im.nq
        dx[im.idx_q1[i], im.idx_q2[i]] = 1.0
    end
    dx[im.idx_q2, im.idx_q1] .= @view im.dx_buf[im.idx_q2, im.idx_q1]
    dx[im.idx_q2, im.idx_q2] .= @view im.dx_buf[im.idx_q2, im.idx_q2]
    return dx
end

"fx(dx, model, x, u, w) -- src/dynamics.jl:96-114"
function fx(dx, im::ImplicitDynamics, x, u, w)
    sync_friction!(im)
    check(ccall((:od_fx_host, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), im.h, x, u, im.dx_buf))
    return write_fx!(dx, im)
end

"fu(du, model, x, u, w) -- src/dynamics.jl:116-128"
function fu(du, im::ImplicitDynamics, x, u, w)
    sync_friction!(im)
    check(ccall((:od_fu_host, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), im.h, x, u, im.du_buf))
    du[im.idx_q2, :] .= @view im.du_buf[im.idx_q2, :]
    return du
end

"f, fx and fu of one knot from one solve (the three reference callbacks solve three times)"
function ffxfu!(d, dx, du, im::ImplicitDynamics, x, u, w)
    sync_friction!(im)
    check(ccall((:od_ffxfu_host, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                im.h, x, u, d, im.dx_buf, im.du_buf))
    write_fx!(dx, im)
    du[im.idx_q2, :] .= @view im.du_buf[im.idx_q2, :]
    return d, dx, du
end

"state_to_configuration -- src/dynamics.jl:131-145"
function state_to_configuration(x::Vector{Vector{T}}) where T
    nq = length(x[1]) ÷ 2
    q = Vector{T}[]
    for t = 1:length(x)
        t == 1 && push!(q, x[t][1:nq])
        push!(q, x[t][nq .+ (1:nq)])
    end
    return q
end

# ---- gradient bundle (src/gradient_bundle.jl) -------------------------------------------------------------------
struct GradientBundle
    N::Int
    η::Matrix{Float64}          # (2nq+nu) x N: gb.ls.η, one nonzero per column (:49-54)
    dz::Matrix{Float64}         # ny x nz (:102)
    ny::Int; nz::Int
end

"GradientBundle(model; N, ϵ) -- src/gradient_bundle.jl:26-85 (sizes from the model, not from module globals)"
function GradientBundle(model; N=100, ϵ=1.0e-4)
    nq, nu = model.nq, model.nu
    nz = 2nq + nu
    η = zeros(nz, N)
    for i = 1:N
        η[rand(1:nz), i] = ϵ * randn()
    end
    GradientBundle(N, η, zeros(nq, nz), nq, nz)
end

"gradient!(sim, gb, q1, q2, u1) -- :87-104, on the handle's eval simulator"
function gradient!(im::ImplicitDynamics, gb::GradientBundle, q1, q2, u1)
    sync_friction!(im)
    x = vcat(q1, q2)
    check(ccall((:od_bundle_grad_host, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}),
                im.h, gb.N, x, Float64.(collect(u1)), gb.η, gb.dz, C_NULL))
    return gb.dz
end

"fx_gb(dx, model, x, u, w) -- :109-126 (model.info::GradientBundle)"
function fx_gb(dx, im::ImplicitDynamics, x, u, w)
    nq = im.nq
    q1 = x[im.idx_q1]; q2 = x[im.idx_q2]
    for i = 1:nq
        dx[im.idx_q1[i], im.idx_q2[i]] = 1.0
    end
    gradient!(im, im.info, q1, q2, u)
    dx[im.idx_q2, im.idx_q1] .= @view im.info.dz[:, im.idx_q1]
    dx[im.idx_q2, im.idx_q2] .= @view im.info.dz[:, im.idx_q2]
    return dx
end

"fu_gb(du, model, x, u, w) -- :136-147"
function fu_gb(du, im::ImplicitDynamics, x, u, w)
    q1 = x[im.idx_q1]; q2 = x[im.idx_q2]
    gradient!(im, im.info, q1, q2, u)
    du[im.idx_q2, :] .= @view im.info.dz[:, im.idx_u1]
    return du
end

# ---- rocket (src/models/rocket/dynamics.jl): one OD_ROCKET_DYNAMICS handle plays ip_dyn and ip_proj -------------
mutable struct RocketInfo
    h::Ptr{Cvoid}
    u_max::Float64; dt::Float64
    up::Vector{Float64}; dp::Matrix{Float64}
end

"RocketInfo(rocket, u_max, h, r, rz, rθ, r_proj, rz_proj, rθ_proj) -- dynamics.jl:12-99 (the residual functions are ignored)"
function RocketInfo(rocket, u_max, h, funcs...)
    hd = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:od_create, LIB), Cint, (Cint, Cint, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}), MODEL_IDS[:rocket], 0, C_NULL, Float64(h), hd))
    check(ccall((:od_set_u_max, LIB), Cint, (Ptr{Cvoid}, Cdouble), hd[], Float64(u_max)))
    r = RocketInfo(hd[], u_max, h, zeros(3), zeros(3, 3))
    finalizer(x -> ccall((:od_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), r)
    return r
end

rocket_host(info::RocketInfo, project, x, u, y, dx, du) =
    check(ccall((:od_rocket_host, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}),
                info.h, project, x, u, y, dx, du, C_NULL, C_NULL))

"f_rocket(d, info, x, u, w) -- :101-114"
f_rocket(d, info::RocketInfo, x, u, w) = (rocket_host(info, 0, x, u, d, C_NULL, C_NULL); d)
"fx_rocket(dx, info, x, u, w) -- :134-147"
fx_rocket(dx, info::RocketInfo, x, u, w) = (rocket_host(info, 0, x, u, C_NULL, dx, C_NULL); dx)
"fu_rocket(du, info, x, u, w) -- :149-163"
fu_rocket(du, info::RocketInfo, x, u, w) = (rocket_host(info, 0, x, u, C_NULL, C_NULL, du); du)
"f_rocket_proj / fx_rocket_proj / fu_rocket_proj -- :215-268"
f_rocket_proj(d, info::RocketInfo, x, u, w) = (rocket_host(info, 1, x, u, d, C_NULL, C_NULL); d)
fx_rocket_proj(dx, info::RocketInfo, x, u, w) = (rocket_host(info, 1, x, u, C_NULL, dx, C_NULL); dx)
fu_rocket_proj(du, info::RocketInfo, x, u, w) = (rocket_host(info, 1, x, u, C_NULL, C_NULL, du); du)

"soc_projection(x, info) -- :168-186 (returns a vector owned by info, like the reference's view)"
function soc_projection(x, info::RocketInfo)
    check(ccall((:od_soc_project_host, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}), info.h, x, info.up, C_NULL, C_NULL))
    return info.up
end
"soc_projection_gradient(x, info) -- :190-210"
function soc_projection_gradient(x, info::RocketInfo)
    check(ccall((:od_soc_project_host, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}), info.h, x, info.up, info.dp, C_NULL))
    return info.dp
end

# ---- batched entry points on device arrays (e.g. AMDGPU.ROCArray): Julia n x B matrices = OD_LAYOUT_BATCH_MAJOR ----
function od_step_grad!(im::ImplicitDynamics, B, X, U, D, DX, DU)
    check(ccall((:od_set_layout, LIB), Cint, (Ptr{Cvoid}, Cint), im.h, 1))
    check(ccall((:od_step_grad, LIB), Cint, (Ptr{Cvoid}, Clong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}),
                im.h, B, pointer(X), pointer(U), pointer(D), pointer(DX), pointer(DU), C_NULL, C_NULL))
end

function od_rollout!(im::ImplicitDynamics, B, T, x1, U, X, A, Bm)
    check(ccall((:od_set_layout, LIB), Cint, (Ptr{Cvoid}, Cint), im.h, 1))
    check(ccall((:od_rollout, LIB), Cint, (Ptr{Cvoid}, Clong, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}),
                im.h, B, T, pointer(x1), pointer(U), pointer(X), pointer(A), pointer(Bm), C_NULL, C_NULL))
end

"rollout with the compact linearisation: G (nq*(2nq+nu)) x (T*B) = dq3/d(q1, q2, u1) per knot"
function od_rollout_compact!(im::ImplicitDynamics, B, T, x1, U, X, G)
    check(ccall((:od_set_layout, LIB), Cint, (Ptr{Cvoid}, Cint), im.h, 1))
    check(ccall((:od_rollout_compact, LIB), Cint, (Ptr{Cvoid}, Clong, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}),
                im.h, B, T, pointer(x1), pointer(U), pointer(X), pointer(G), C_NULL, C_NULL))
end

# ---- one Julia process per GPU (SURVEY.md 8(e)): rollouts sharded, the compact linearisation gathered over RCCL behind the C ABI ----------
# Rank 0 makes the id and hands its 128 bytes to the other processes (a file, a socket, Distributed.jl); every process then creates its
# communicator -- a collective call -- on the device its ImplicitDynamics handle lives on (INTEGRATION.md shows the eight-process loop).
mutable struct Communicator
    c::Ptr{Cvoid}
    world::Int
    rank::Int
end
function comm_unique_id()
    id = zeros(UInt8, 128)                         # OD_COMM_ID_BYTES
    check(ccall((:od_comm_unique_id, LIB), Cint, (Ptr{Cvoid},), id))
    return id
end
function Communicator(im::ImplicitDynamics, id::Vector{UInt8}, rank::Integer, world::Integer)
    length(id) == 128 || error("the id of comm_unique_id() has 128 bytes")
    hd = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:od_comm_create, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Ref{Ptr{Cvoid}}), im.h, id, rank, world, hd))
    w = Ref{Cint}(0); r = Ref{Cint}(0)
    check(ccall((:od_comm_info, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ptr{Cint}), hd[], w, r, C_NULL))
    comm = Communicator(hd[], w[], r[])
    finalizer(x -> ccall((:od_comm_destroy, LIB), Cint, (Ptr{Cvoid},), x.c), comm)
    return comm
end
"all-gather of od_rollout_compact!'s X and G (device arrays, every rank the same B and T): X_all / G_all hold `world` blocks, block r = rank r's array"
allgather_compact!(im::ImplicitDynamics, comm::Communicator, B, T, X, G, X_all, G_all) =
    check(ccall((:od_allgather_compact, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Clong, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                im.h, comm.c, B, T, pointer(X), pointer(G), pointer(X_all), pointer(G_all)))
"all-gather of any device buffer of `bytes` bytes per rank (e.g. the gains of a backward pass)"
allgather!(im::ImplicitDynamics, comm::Communicator, send, recv, bytes::Integer) =
    check(ccall((:od_comm_allgather, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), im.h, comm.c, pointer(send), pointer(recv), bytes))

"whole solution on device arrays (contact impulses and their sensitivities): Z nz x B, DZ (nz*(2nq+nu)) x B; rows via model_indices"
function od_step_full!(im::ImplicitDynamics, B, X, U, Z, DZ)
    check(ccall((:od_set_layout, LIB), Cint, (Ptr{Cvoid}, Cint), im.h, 1))
    check(ccall((:od_step_full, LIB), Cint, (Ptr{Cvoid}, Clong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}),
                im.h, B, pointer(X), pointer(U), pointer(Z), pointer(DZ), C_NULL, C_NULL))
end
"1-based z indices of (:q | :γ | :b) for a model"
function model_indices(model::Symbol, which::Symbol)
    buf = zeros(Cint, 16)
    n = ccall((:od_model_indices, LIB), Cint, (Cint, Cint, Ptr{Cint}, Cint), model_id(model), Dict(:q => 0, :γ => 1, :b => 2)[which], buf, 16)
    return Int.(buf[1:n]) .+ 1
end

"batched soc_projection / soc_projection_gradient on device arrays: U 3 x B, UP 3 x B, DP 9 x B"
function od_soc_project!(r::RocketInfo, B, U, UP, DP)
    check(ccall((:od_set_layout, LIB), Cint, (Ptr{Cvoid}, Cint), r.h, 1))
    check(ccall((:od_soc_project, LIB), Cint, (Ptr{Cvoid}, Clong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}),
                r.h, B, pointer(U), pointer(UP), pointer(DP), C_NULL))
end
"batched f_rocket(_proj) + fx + fu: X 12 x B, U 3 x B -> Y 12 x B, DX 144 x B, DU 36 x B"
function od_rocket!(r::RocketInfo, B, project::Bool, X, U, Y, DX, DU)
    check(ccall((:od_set_layout, LIB), Cint, (Ptr{Cvoid}, Cint), r.h, 1))
    check(ccall((:od_rocket, LIB), Cint, (Ptr{Cvoid}, Clong, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}),
                r.h, B, project ? 1 : 0, pointer(X), pointer(U), pointer(Y), pointer(DX), pointer(DU), C_NULL, C_NULL))
end

# ---- the iLQR iteration on the device (od_ilqr_*): iLQR.solver / solve! for B problems in lockstep ------------------------------
# Counterpart of the reference's use of IterativeLQR.jl (examples/acrobot.jl:97-113, examples/rocket.jl:118-139) for quadratic
# objectives and terminal equality constraints.  Device arrays are batch-minor: x1 is B x n, U is B x T x m, X is B x (T+1) x n
# (Julia column-major: element (b, t, i) at b + B (t + T i)), all Float64.
struct ILQROptions                    # od_ilqr_options
    reg::Cdouble; c1::Cdouble; obj_tol::Cdouble; con_tol::Cdouble; rho_init::Cdouble; rho_scale::Cdouble
    max_iter::Cint; max_al_iter::Cint; project::Cint; history::Cint
    proj_stall_exit::Cint; rho_max::Cdouble       # (C layout: four bytes of padding before the double, as in the header's struct)
end
struct ILQRInfo                       # od_ilqr_info
    iterations::Cint; al_iterations::Cint; done::Cint; al_done::Cint; bad_linearisations::Cint
    reg::Cdouble; rho::Cdouble; max_dJ::Cdouble; max_violation::Cdouble
end
mutable struct ILQRSolver
    s::Ptr{Cvoid}
    owner::Any                        # the ImplicitDynamics / RocketInfo whose handle the solver borrows (kept alive)
    B::Int; T::Int
end

"""
    ILQRSolver(dyn, B, T; alphas, Q, R, QT, xref, goal_idx, goal, obj_tol, con_tol, max_iter, max_al_iter, ρ_init, ρ_scale, project)

`dyn`: an ImplicitDynamics or a RocketInfo.  Objective Σ ½(x-xref)'Q(x-xref) + ½u'Ru + ½(x_T-xref)'QT(x_T-xref); terminal
constraints x_T[goal_idx] = goal by augmented Lagrangian (cf. iLQR.Options, examples/acrobot.jl:98-107).
"""
function ILQRSolver(dyn, B::Integer, T::Integer; alphas=[2.0^-i for i in 0:10], Q, R, QT, xref,
        goal_idx=Int[], goal=Float64[], reg=1.0e-6, c1=1.0e-4, obj_tol=1.0e-6, con_tol=1.0e-3, max_iter=50, max_al_iter=1,
        ρ_init=1.0, ρ_scale=10.0, project=true, history=0, proj_stall_exit=true, ρ_max=1.0e8)
    o = Ref(ILQROptions(reg, c1, obj_tol, con_tol, ρ_init, ρ_scale, max_iter, max_al_iter, project ? 1 : 0, history, proj_stall_exit ? 1 : 0, ρ_max))
    hd = Ref{Ptr{Cvoid}}(C_NULL)
    a = Float64.(collect(alphas))
    check(ccall((:od_ilqr_create, LIB), Cint, (Ptr{Cvoid}, Clong, Cint, Cint, Ptr{Cdouble}, Ref{ILQROptions}, Ref{Ptr{Cvoid}}),
                dyn.h, B, T, length(a), a, o, hd))
    gi = Cint.(collect(goal_idx) .- 1)                                 # 0-based rows of x_T
    check(ccall((:od_ilqr_set_objective, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cint}, Ptr{Cdouble}),
                hd[], Matrix{Float64}(Q), Matrix{Float64}(R), Matrix{Float64}(QT), Float64.(collect(xref)), length(gi), gi, Float64.(collect(goal))))
    s = ILQRSolver(hd[], dyn, B, T)
    # (Julia gives no order between this finaliser and the owner's od_destroy when both become unreachable in one sweep: the
    # library takes either order -- od_destroy releases what its live solvers hold and detaches them, od_ilqr_destroy of a
    # detached solver only frees the host object)
    finalizer(x -> ccall((:od_ilqr_destroy, LIB), Cint, (Ptr{Cvoid},), x.s), s)
    return s
end

"""
    set_constraints!(s; Cs, Ds, ds, n_stage_ineq, Ct, dt, n_terminal_ineq)

Affine constraints by augmented Lagrangian (iLQR.Constraint with idx_ineq, examples/rocket.jl:82-110): stage rows
`Cs x_t + Ds u_t - ds` (t < T) and terminal rows `Ct x_T - dt`; the first `n_*_ineq` rows of each are inequalities (<= 0).
"""
function set_constraints!(s::ILQRSolver; Cs=zeros(0, 0), Ds=zeros(0, 0), ds=Float64[], n_stage_ineq=0, Ct=zeros(0, 0), dt=Float64[], n_terminal_ineq=0)
    check(ccall((:od_ilqr_set_constraints, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}),
                s.s, length(ds), n_stage_ineq, Matrix{Float64}(Cs), Matrix{Float64}(Ds), Float64.(collect(ds)),
                length(dt), n_terminal_ineq, Matrix{Float64}(Ct), Float64.(collect(dt))))
end
struct ILQRParameterStage          # od_ilqr_parameter_stage
    constraint::Cint; n_p::Cint; p::Ptr{Cdouble}; w_theta::Ptr{Cdouble}; cost_const::Cdouble
    nt::Cint; nt_ineq::Cint; Ct_x::Ptr{Cdouble}; Ct_theta::Ptr{Cdouble}; dt::Ptr{Cdouble}
end
"""
    set_parameter_stage!(s; w_theta, constraint="", p, Ct_x, Ct_theta, dt, n_terminal_ineq, cost_const)

The first stage of examples/hopper.jl (:52-101,165-175,234-266): the initial configurations θ = [q1; q2] are optimised with the controls.
Slot 1 of every trajectory is θ (x1 of `initialize!` / `solve!` its initial value); it costs ½ θ'diag(w_theta)θ + cost_const, obeys the
generated constraint function `constraint` (c(θ; p) = 0, e.g. "hopper_foot": stage1_con without its control limits) and enters the terminal
rows `Ct_x x_T + Ct_theta θ - dt` (the first `n_terminal_ineq` inequalities ≤ 0: terminal_con).
"""
function set_parameter_stage!(s::ILQRSolver; w_theta, constraint="", p=Float64[], Ct_x=zeros(0, 0), Ct_theta=zeros(0, 0), dt=Float64[], n_terminal_ineq=0, cost_const=0.0)
    cid = isempty(constraint) ? Cint(-1) : ccall((:od_constraint_id, LIB), Cint, (Cstring,), constraint)
    (isempty(constraint) || cid >= 0) || error("constraint function $constraint is not in the library")
    w = Float64.(collect(w_theta)); pp = Float64.(collect(p)); cx = Matrix{Float64}(Ct_x); ct = Matrix{Float64}(Ct_theta); d = Float64.(collect(dt))
    GC.@preserve w pp cx ct d begin
        ps = Ref(ILQRParameterStage(cid, length(pp), pointer(pp), pointer(w), cost_const, length(d), n_terminal_ineq, pointer(cx), pointer(ct), pointer(d)))
        check(ccall((:od_ilqr_set_parameter_stage, LIB), Cint, (Ptr{Cvoid}, Ref{ILQRParameterStage}), s.s, ps))
    end
end
"""
    set_gradient_bundle!(s, gb)

examples/planar_push.jl with `GB = true` (:15,22,29-30): the solver linearises with fx_gb / fu_gb (src/gradient_bundle.jl:109-147) --
N + 1 eval-simulator steps per knot with the samples `gb.ls.η` and the least-squares fit -- instead of the implicit gradients.
`set_gradient_bundle!(s, nothing)` goes back to them.
"""
function set_gradient_bundle!(s::ILQRSolver, eta::Union{Nothing,AbstractMatrix})
    if eta === nothing
        return check(ccall((:od_ilqr_set_gradient_bundle, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}), s.s, 0, C_NULL))
    end
    e = Matrix{Float64}(eta)                       # (2nq + nu) x N, column-major
    GC.@preserve e check(ccall((:od_ilqr_set_gradient_bundle, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}), s.s, size(e, 2), pointer(e)))
end
"per problem: flags (bit 0 inner loop converged, bit 1 constraints met), violation, penalty -- device arrays of length B"
get_status!(s::ILQRSolver, flags, violation, penalty) =
    check(ccall((:od_ilqr_get_status, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), s.s, pointer(flags), pointer(violation), pointer(penalty)))

"initialize_controls! + rollout + first linearisation (examples/acrobot.jl:108-113); x1, U0 device arrays"
initialize!(s::ILQRSolver, x1, U0) = check(ccall((:od_ilqr_init, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), s.s, pointer(x1), pointer(U0)))
"n iterations, asynchronous (no host synchronisation; capturable in a HIP graph)"
iterate!(s::ILQRSolver, n::Integer=1) = check(ccall((:od_ilqr_iterate, LIB), Cint, (Ptr{Cvoid}, Cint), s.s, n))
al_update!(s::ILQRSolver) = check(ccall((:od_ilqr_al_update, LIB), Cint, (Ptr{Cvoid},), s.s))
"solve!(solver, x1, U0) -- examples/acrobot.jl:113"
solve!(s::ILQRSolver, x1, U0) = check(ccall((:od_ilqr_solve, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), s.s, pointer(x1), pointer(U0)))
"get_trajectory(solver) -- examples/acrobot.jl:121: X (B x (T+1) x n), U (B x T x m), J (B) device arrays, filled asynchronously"
get_trajectory!(s::ILQRSolver, X, U, J) =
    check(ccall((:od_ilqr_get, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), s.s, pointer(X), pointer(U), pointer(J), C_NULL, C_NULL))
"""
    get_trace!(s, step_index, reg, rho, cap) -> rows

The decisions of every iteration since `initialize!` (what IterativeLQR prints with `verbose = true`): device arrays of `cap x B`
(`Int32`, `Float64`, `Float64`); row i: index into `alphas` of the accepted step size (-1 none, -2 the problem had converged before),
the problem's regularisation after the iteration, its penalty.
"""
get_trace!(s::ILQRSolver, step_index, reg, rho, cap::Integer) =
    ccall((:od_ilqr_get_trace, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), s.s, pointer(step_index), pointer(reg), pointer(rho), cap)
function info(s::ILQRSolver)
    r = Ref(ILQRInfo(0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0))
    check(ccall((:od_ilqr_get_info, LIB), Cint, (Ptr{Cvoid}, Ref{ILQRInfo}), s.s, r))
    return r[]
end

end # module
