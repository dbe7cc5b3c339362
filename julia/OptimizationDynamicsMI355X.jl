# ccall shim over libod_mi355x.so keeping the reference's ImplicitDynamics f / fx / fu API
# (src/dynamics.jl).  NOT exercised in the build environment (no Julia there); the identical entry
# points are exercised through ctypes by tests/.  See INTEGRATION.md.
module OptimizationDynamicsMI355X

export ImplicitDynamicsMI355X, f, fx, fu, state_to_configuration, od_step_grad!, od_rollout!,
       RocketInfoMI355X, od_rocket!, od_soc_project!, od_step_full!, model_indices

const LIB = get(ENV, "OD_MI355X_LIB", joinpath(@__DIR__, "..", "optimization_dynamics_amd", "libod_mi355x.so"))

const MODEL_IDS = Dict(:acrobot_impact => 0, :acrobot_nominal => 1, :cartpole_friction => 2,
                       :cartpole_frictionless => 3, :planarpush => 4, :rocket => 5,
                       :rocket_projection => 6, :hopper => 7)

struct ODOptions                      # od_options (include/od_mi355x.h)
    r_tol::Cdouble; kappa_eval_tol::Cdouble; kappa_grad_tol::Cdouble
    max_iter::Cint; max_ls::Cint
    eps_min::Cdouble; kappa_reg::Cdouble; gamma_reg::Cdouble; undercut::Cdouble
end

function check(rc)
    rc == 0 && return nothing
    error("libod_mi355x: " * unsafe_string(ccall((:od_last_error, LIB), Cstring, ())))
end

mutable struct ImplicitDynamicsMI355X
    h::Ptr{Cvoid}
    nq::Int; nu::Int
    dx_buf::Matrix{Float64}; du_buf::Matrix{Float64}
    idx_q1::Vector{Int}; idx_q2::Vector{Int}
end

"ImplicitDynamics(model, h, …; r_tol, κ_eval_tol, κ_grad_tol) — src/dynamics.jl:51-79"
function ImplicitDynamicsMI355X(model::Symbol, h::Float64; r_tol=1.0e-8, κ_eval_tol=1.0e-6, κ_grad_tol=1.0e-6)
    id = MODEL_IDS[model]
    dims = [Ref{Cint}(0) for _ in 1:5]
    check(ccall((:od_model_dims, LIB), Cint, (Cint, Ref{Cint}, Ref{Cint}, Ref{Cint}, Ref{Cint}, Ref{Cint}), id, dims...))
    nq, nu = Int(dims[1][]), Int(dims[2][])
    o = Ref(ODOptions(r_tol, κ_eval_tol, κ_grad_tol, 100, 25, 0.25, 1.0e-3, 0.1, Inf))   # get_simulator preset, src/dynamics.jl:25-33
    hd = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:od_create, LIB), Cint, (Cint, Cint, Ref{ODOptions}, Cdouble, Ref{Ptr{Cvoid}}), id, 0, o, h, hd))
    m = ImplicitDynamicsMI355X(hd[], nq, nu, zeros(2nq, 2nq), zeros(2nq, nu), collect(1:nq), collect(nq .+ (1:nq)))
    finalizer(x -> ccall((:od_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), m)
    return m
end

set_friction!(m::ImplicitDynamicsMI355X, μ::Vector{Float64}) =
    check(ccall((:od_set_friction, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint), m.h, μ, length(μ)))

"f(d, model, x, u, w) — src/dynamics.jl:81-94"
function f(d, m::ImplicitDynamicsMI355X, x, u, w)
    check(ccall((:od_f_host, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), m.h, x, u, d))
    return d
end

"fx(dx, model, x, u, w) — src/dynamics.jl:96-114 (writes the same three blocks as the reference)"
function fx(dx, m::ImplicitDynamicsMI355X, x, u, w)
    check(ccall((:od_fx_host, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), m.h, x, u, m.dx_buf))
    for i = 1:m.nq
        dx[m.idx_q1[i], m.idx_q2[i]] = 1.0
    end
    dx[m.idx_q2, m.idx_q1] .= @view m.dx_buf[m.idx_q2, m.idx_q1]
    dx[m.idx_q2, m.idx_q2] .= @view m.dx_buf[m.idx_q2, m.idx_q2]
    return dx
end

"fu(du, model, x, u, w) — src/dynamics.jl:116-128"
function fu(du, m::ImplicitDynamicsMI355X, x, u, w)
    check(ccall((:od_fu_host, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), m.h, x, u, m.du_buf))
    du[m.idx_q2, :] .= @view m.du_buf[m.idx_q2, :]
    return du
end

"state_to_configuration — src/dynamics.jl:131-145"
function state_to_configuration(x::Vector{Vector{T}}) where T
    nq = length(x[1]) ÷ 2
    q = Vector{T}[]
    for t = 1:length(x)
        t == 1 && push!(q, x[t][1:nq])
        push!(q, x[t][nq .+ (1:nq)])
    end
    return q
end

# batched entry points on device arrays (e.g. AMDGPU.ROCArray): Julia n×B matrices = OD_LAYOUT_BATCH_MAJOR
function od_step_grad!(m::ImplicitDynamicsMI355X, B, X, U, D, DX, DU)
    check(ccall((:od_set_layout, LIB), Cint, (Ptr{Cvoid}, Cint), m.h, 1))
    check(ccall((:od_step_grad, LIB), Cint, (Ptr{Cvoid}, Clong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}),
                m.h, B, pointer(X), pointer(U), pointer(D), pointer(DX), pointer(DU), C_NULL, C_NULL))
end

function od_rollout!(m::ImplicitDynamicsMI355X, B, T, x1, U, X, A, Bm)
    check(ccall((:od_set_layout, LIB), Cint, (Ptr{Cvoid}, Cint), m.h, 1))
    check(ccall((:od_rollout, LIB), Cint, (Ptr{Cvoid}, Clong, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}),
                m.h, B, T, pointer(x1), pointer(U), pointer(X), pointer(A), pointer(Bm), C_NULL, C_NULL))
end

"whole solution on device arrays (contact impulses and their sensitivities): Z nz×B, DZ (nz*(2nq+nu))×B; rows via model_indices"
function od_step_full!(m::ImplicitDynamicsMI355X, B, X, U, Z, DZ)
    check(ccall((:od_set_layout, LIB), Cint, (Ptr{Cvoid}, Cint), m.h, 1))
    check(ccall((:od_step_full, LIB), Cint, (Ptr{Cvoid}, Clong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}),
                m.h, B, pointer(X), pointer(U), pointer(Z), pointer(DZ), C_NULL, C_NULL))
end
"1-based z indices of (:q | :γ | :b) for a model"
function model_indices(model::Symbol, which::Symbol)
    buf = zeros(Cint, 16)
    n = ccall((:od_model_indices, LIB), Cint, (Cint, Cint, Ptr{Cint}, Cint), MODEL_IDS[model], Dict(:q => 0, :γ => 1, :b => 2)[which], buf, 16)
    return Int.(buf[1:n]) .+ 1
end

# rocket (src/models/rocket/dynamics.jl): one OD_ROCKET_DYNAMICS handle plays ip_dyn and ip_proj
mutable struct RocketInfoMI355X
    h::Ptr{Cvoid}
end
function RocketInfoMI355X(u_max::Float64, h::Float64)
    hd = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:od_create, LIB), Cint, (Cint, Cint, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}), MODEL_IDS[:rocket], 0, C_NULL, h, hd))
    check(ccall((:od_set_u_max, LIB), Cint, (Ptr{Cvoid}, Cdouble), hd[], u_max))
    r = RocketInfoMI355X(hd[])
    finalizer(x -> ccall((:od_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), r)
    return r
end
"batched soc_projection / soc_projection_gradient (dynamics.jl:168-214) on device arrays: U 3×B, UP 3×B, DP 9×B"
function od_soc_project!(r::RocketInfoMI355X, B, U, UP, DP)
    check(ccall((:od_set_layout, LIB), Cint, (Ptr{Cvoid}, Cint), r.h, 1))
    check(ccall((:od_soc_project, LIB), Cint, (Ptr{Cvoid}, Clong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}),
                r.h, B, pointer(U), pointer(UP), pointer(DP), C_NULL))
end
"batched f_rocket(_proj) + fx + fu (dynamics.jl:101-164, 215-268): X 12×B, U 3×B -> Y 12×B, DX 144×B, DU 36×B"
function od_rocket!(r::RocketInfoMI355X, B, project::Bool, X, U, Y, DX, DU)
    check(ccall((:od_set_layout, LIB), Cint, (Ptr{Cvoid}, Cint), r.h, 1))
    check(ccall((:od_rocket, LIB), Cint, (Ptr{Cvoid}, Clong, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}),
                r.h, B, project ? 1 : 0, pointer(X), pointer(U), pointer(Y), pointer(DX), pointer(DU), C_NULL, C_NULL))
end

end # module
