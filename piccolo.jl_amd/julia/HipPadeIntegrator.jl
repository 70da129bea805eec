# HipPadeIntegrator.jl -- Julia-side glue of the drop-in: `DirectTrajOpt.AbstractIntegrator`s whose arithmetic runs in
# libpiccolo_hip.so (HIP kernels on an MI355X) through `ccall`.
#
# STATUS: written against the interface Piccolo.jl v2.0.2 *uses* (call sites cited below).  The abstract type and the
# generic functions live in the un-vendored DirectTrajOpt.jl (compat 0.9.5 / 0.10) and there is no Julia in the build
# container, so this file has NOT been executed.  `selftest.jl` (next to this file) is the script a maintainer runs first:
# it checks this glue against DirectTrajOpt's own `test_integrator` and against `BilinearIntegrator`, and lists the
# DirectTrajOpt generics the glue bound itself to at load time (`HipPade.BOUND_GENERICS`, section "adaptive binding" below).  All arithmetic is in the C
# library (include/piccolo_hip.h); this file only marshals arguments.  See INTEGRATION.md.
#
#   plug-in points in the reference:
#     SmoothPulseProblem(qtraj, N; integrator = HipPadeIntegrator(qtraj, N))
#         src/control/templates/smooth_pulse_problem.jl:123,213-233
#     SamplingProblem(qcp, systems; integrator = (sq, N) -> HipPadeIntegrator(sq, N))     # returns a Vector, one per member
#         src/control/templates/sampling_problem.jl:190-237,292
#     Specs.register_integrator!(:hip_pade, RegistryEntry(factory = (qtraj, N; alg) -> HipPadeIntegrator(qtraj, N)))
#         src/specs/registries.jl:82-86,112,119 ; src/specs/materialize.jl:216-222
module HipPade

using LinearAlgebra, SparseArrays
using NamedTrajectories
using DirectTrajOpt
import DirectTrajOpt: AbstractIntegrator
using Piccolo: get_system, state_name, state_names, drive_name, sampling_member_states,
               UnitaryTrajectory, KetTrajectory, MultiKetTrajectory, DensityTrajectory, MultiDensityTrajectory, SamplingTrajectory,
               compact_lindbladian_generators

const LIB = get(ENV, "PICCOLO_HIP_LIB", "libpiccolo_hip.so")

# mirror of `pcl_desc` (include/piccolo_hip.h) -- field order and widths must match
struct PclDesc
    struct_size::Int32; d::Int32; n_drives::Int32; N::Int32; z_dim::Int32
    u_off::Int32; dt_off::Int32; batch::Int32; batch_mode::Int32; pade_order::Int32
    device_id::Int32; index_base::Int32; per_member_G0::Int32; state_cols::Int32
    global_dim::Int64
    G0::Ptr{Float64}; Gj::Ptr{Float64}; x_offs::Ptr{Int32}
end

_lasterr(ctx) = unsafe_string(ccall((:pcl_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))
check(ctx, rc) = rc == 0 || error("libpiccolo_hip: ", _lasterr(ctx), " (code $rc)")

# One `pcl_ctx` (one GPU, one stream).  A SamplingTrajectory's M member integrators share ONE batched context: the first
# member asked about a trajectory launches the fused kernels for ALL members, the others slice their rows out of the
# cached result (the cache key is the trajectory buffer itself, compared by value, so a stale result is never served).
mutable struct PclCore
    ctx::Ptr{Cvoid}
    n_members::Int
    x_dim::Int                      # per member
    rows_per::Int                   # x_dim * (N-1)
    jac_per::Int                    # Jacobian values per member
    hess_per::Int
    z::Vector{Float64}              # trajectory buffer the cached results belong to
    δ::Vector{Float64}              # all members, member-major
    vals::Vector{Float64}
    have_δ::Bool
    have_vals::Bool
    launches::Int
end

function _destroy!(c::PclCore)
    c.ctx == C_NULL || ccall((:pcl_destroy, LIB), Cvoid, (Ptr{Cvoid},), c.ctx)
    c.ctx = C_NULL
    return nothing
end

mutable struct HipPadeIntegrator <: AbstractIntegrator
    core::PclCore
    member::Int               # 1-based member of the core's ensemble (1 for a single-state integrator)
    x_name::Symbol
    x_names::Vector{Symbol}
    u_name::Symbol
    x_dim::Int
    dim::Int                  # == x_dim * (N - 1)                                    [REF integrators.jl:307-309]
    jac_rows::Vector{Int32}   # 1-based, in value order, numbered inside this integrator's block (never assumed, always queried)
    jac_cols::Vector{Int32}
    hess_rows::Vector{Int32}
    hess_cols::Vector{Int32}
    n_vars::Int
    G0::Matrix{Float64}       # this member's generators (for `f`)
    Gj::Vector{Matrix{Float64}}
    state_cols::Int
    pade_order::Int
    fcore::Union{Nothing,PclCore}   # lazily created 2-knot context behind `B.f`
end

function _create_core(G0s::Vector{Matrix{Float64}}, Gjs::Vector{Matrix{Float64}}, N::Int, z_dim::Int, u_off::Int, dt_off::Int,
                      x_offs::Vector{Int32}, global_dim::Int; device::Integer = 0, pade_order::Integer = 4, state_cols::Integer = 0)
    # state_cols: 0 unitary (n = 2d), 1 ket, -1 = PCL_STATE_VECTOR (general n x n generator on one real column, d := n)
    n = size(G0s[1], 1); d = state_cols == -1 ? n : n ÷ 2; m = length(Gjs)
    G0 = reduce(vcat, [vec(G) for G in G0s])                       # column-major, one block per member
    Gj = m == 0 ? zeros(1) : reduce(vcat, [vec(G) for G in Gjs])
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve G0 Gj x_offs begin
        desc = PclDesc(sizeof(PclDesc), d, m, N, z_dim, u_off, dt_off,
                       length(x_offs), 0 #= PCL_BATCH_MEMBERS =#, pade_order #= 2, 4, 6, 8 or 10 =#, device, 1 #= 1-based =#,
                       length(G0s) > 1 ? 1 : 0, state_cols, global_dim,
                       pointer(G0), pointer(Gj), pointer(x_offs))
        rc = ccall((:pcl_create, LIB), Cint, (Ref{PclDesc}, Ref{Ptr{Cvoid}}), desc, ctx)
        rc == 0 || error("pcl_create: ", _lasterr(C_NULL))
    end
    c = ctx[]
    xd = Ref{Int64}(0); nr = Ref{Int64}(0); ncol = Ref{Int64}(0); nnz = Ref{Int64}(0); per = Ref{Int64}(0)
    check(c, ccall((:pcl_constraint_dim, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}), c, xd, nr, ncol))
    M = length(x_offs)
    check(c, ccall((:pcl_jac_nnz, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), c, nnz, per))
    jac_per = Int(nnz[]) ÷ M
    check(c, ccall((:pcl_hess_nnz, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), c, nnz, per))
    core = PclCore(c, M, Int(xd[]), Int(nr[]) ÷ M, jac_per, Int(nnz[]) ÷ M, Float64[], Float64[], Float64[], false, false, 0)
    finalizer(_destroy!, core)
    return core, Int(ncol[])
end

_window!(core::PclCore, first0::Integer, count::Integer) =
    check(core.ctx, ccall((:pcl_set_member_window, LIB), Cint, (Ptr{Cvoid}, Int32, Int32), core.ctx, first0, count))

# structure of ONE member (rows numbered inside the member's block, columns = global variable indices), 1-based
function _member_structure(core::PclCore, member::Int)
    _window!(core, member - 1, 1)
    jr = Vector{Int32}(undef, core.jac_per); jc = similar(jr)
    check(core.ctx, ccall((:pcl_jac_structure, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}), core.ctx, jr, jc))
    hr = Vector{Int32}(undef, core.hess_per); hc = similar(hr)
    check(core.ctx, ccall((:pcl_hess_structure, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}), core.ctx, hr, hc))
    _window!(core, 0, core.n_members)
    return jr, jc, hr, hc
end

function _generators(sys)
    sys.time_dependent && error("HipPadeIntegrator: time-dependent systems use TimeDependentBilinearIntegrator")
    m = sys.n_drives
    e(j) = (u = zeros(m); u[j] = 1.0; u)
    G0 = Matrix{Float64}(sys.G(zeros(m), 0.0))
    Gj = [Matrix{Float64}(sys.G(e(j), 0.0)) - G0 for j in 1:m]      # linear drives: G(u) = G0 + sum u_j G_j
    return G0, Gj
end

# The order in use, asked of the context (never a cached copy of a constructor argument): 0 = not decided yet.
function _order_in_use(core::PclCore)
    v = Ref{Int64}(0)
    check(core.ctx, ccall((:pcl_get_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), core.ctx, "pade_order", v))
    return Int(v[])
end

# pade_order = 0 (the default): pcl_set_order_policy with dt_max and |u|_max from traj.bounds when both bounds exist, else
# pcl_set_order_from_trajectory on the trajectory the integrator is constructed with (theta = 1.5 max_k |dt_k G(u_k)|).  Either way the
# order is decided HERE, so B.f, evaluate! and eval_jacobian evaluate one and the same constraint from the first call on.
function _decide_order!(core::PclCore, traj::NamedTrajectory, u_name::Symbol, m::Int, tol::Float64)
    order = _order_from_bounds!(core, traj, u_name, m, tol)
    order != 0 && return order
    Z = Vector{Float64}(traj.datavec)
    out = Ref{Int32}(0)
    GC.@preserve Z check(core.ctx, ccall((:pcl_set_order_from_trajectory, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64, Ref{Int32}),
                                         core.ctx, Z, tol, out))
    return Int(out[])
end

function _order_from_bounds!(core::PclCore, traj::NamedTrajectory, u_name::Symbol, m::Int, tol::Float64)
    (m > 0 && haskey(traj.bounds, u_name) && haskey(traj.bounds, traj.timestep)) || return 0
    ub = traj.bounds[u_name]; tb = traj.bounds[traj.timestep]
    umax = Float64[max(abs(ub[1][j]), abs(ub[2][j])) for j in 1:m]
    dtmax = Float64(maximum(abs, vcat(collect(tb[1]), collect(tb[2]))))
    order = Ref{Int32}(0)
    GC.@preserve umax check(core.ctx, ccall((:pcl_set_order_policy, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Float64, Ref{Int32}),
                                           core.ctx, dtmax, umax, tol, order))
    return Int(order[])
end

function _integrators(G0s, Gj, traj::NamedTrajectory, names::Vector{Symbol}, u_name::Symbol;
                      state_cols::Integer = 0, pade_order::Integer = 0, order_tol::Float64 = 1e-10, kwargs...)
    x_offs = Int32[traj.components[nm][1] - 1 for nm in names]
    core, n_vars = _create_core(G0s, Gj, traj.N, traj.dim, traj.components[u_name][1] - 1,
                                traj.components[traj.timestep][1] - 1, x_offs, traj.global_dim;
                                state_cols = state_cols, pade_order = pade_order, kwargs...)
    # pade_order = 0 (default): the smallest order that matches the reference's exp constraint [REF docs/src/concepts/index.md:21] to
    # order_tol over the trajectory's bounds (else over the trajectory itself); 2 ... 10 pins the order (BASELINE's metric is quoted on 4)
    pade_order == 0 && _decide_order!(core, traj, u_name, length(Gj), order_tol)
    pade_order = _order_in_use(core)   # read back: the struct never holds a stale 0
    pade_order == 0 && error("HipPadeIntegrator: the Pade order could not be decided")
    Bs = HipPadeIntegrator[]
    for (i, nm) in enumerate(names)
        jr, jc, hr, hc = _member_structure(core, i)
        push!(Bs, HipPadeIntegrator(core, i, nm, [nm], u_name, core.x_dim, core.rows_per, jr, jc, hr, hc, n_vars,
                                    G0s[length(G0s) > 1 ? i : 1], Gj, state_cols, pade_order, nothing))
    end
    return Bs
end

# BilinearIntegrator(qtraj::UnitaryTrajectory, N)                              [REF src/control/integrators.jl:35-51]
function HipPadeIntegrator(qtraj::UnitaryTrajectory, N::Int; kwargs...)
    G0, Gj = _generators(get_system(qtraj))
    return only(_integrators([G0], Gj, NamedTrajectory(qtraj, N), [state_name(qtraj)], drive_name(qtraj); kwargs...))
end

# BilinearIntegrator(qtraj::KetTrajectory, N) [REF integrators.jl:58-74]
function HipPadeIntegrator(qtraj::KetTrajectory, N::Int; kwargs...)
    G0, Gj = _generators(get_system(qtraj))
    return only(_integrators([G0], Gj, NamedTrajectory(qtraj, N), [state_name(qtraj)], drive_name(qtraj); state_cols = 1, kwargs...))
end

# BilinearIntegrator(qtraj::MultiKetTrajectory, N) -> Vector, one per ket, all under the same system [REF integrators.jl:103-117]
function HipPadeIntegrator(qtraj::MultiKetTrajectory, N::Int; kwargs...)
    G0, Gj = _generators(get_system(qtraj))
    return _integrators([G0], Gj, NamedTrajectory(qtraj, N), collect(state_names(qtraj)), drive_name(qtraj); state_cols = 1, kwargs...)
end

# BilinearIntegrator(qtraj::DensityTrajectory, N) [REF integrators.jl:82-95]: compact Lindbladian generators, linear
# drives and constant dissipation rates (compact_lindbladian_generators folds the dissipators into the drift)
function HipPadeIntegrator(qtraj::DensityTrajectory, N::Int; kwargs...)
    Gc_drift, Gc_drives = compact_lindbladian_generators(get_system(qtraj))
    return only(_integrators([Matrix{Float64}(Gc_drift)], [Matrix{Float64}(G) for G in Gc_drives], NamedTrajectory(qtraj, N),
                             [state_name(qtraj)], drive_name(qtraj); state_cols = -1, kwargs...))
end

# BilinearIntegrator(qtraj::SamplingTrajectory, N) -> Vector{<:AbstractIntegrator}, ONE PER MEMBER, in member order
# [REF integrators.jl:134-146]; SamplingProblem rejects anything else [REF sampling_problem.jl:195-223].
# Members that share the drive generators (perturbed drifts: the robust-control use) evaluate through one batched
# context; members whose drive generators differ too (each member uses its full sys.G, [REF integrators.jl:149-162])
# get a context each.
function HipPadeIntegrator(qtraj::SamplingTrajectory, N::Int; kwargs...)
    # Every base the reference samples [REF src/control/integrators.jl:149-226 (_sampling_integrator)]:
    #   Unitary / Ket       one state per member                     -> one integrator per member
    #   MultiKet            a Vector of ket names per member         -> one integrator per ket, all on the member's system
    #   Density             compact Lindbladian of the member        -> one integrator per member (state vector of length levels^2)
    #   MultiDensity        a Vector of density names per member     -> one integrator per density sub-state
    # The result is flat, member-major, sub-states in order: the reference's `reduce(vcat, ...)`.
    base = qtraj.base_trajectory
    dens = base isa DensityTrajectory || base isa MultiDensityTrajectory
    base isa UnitaryTrajectory || base isa KetTrajectory || base isa MultiKetTrajectory || dens ||
        error("HipPadeIntegrator(::SamplingTrajectory): unsupported base trajectory $(typeof(base))")
    sc = base isa UnitaryTrajectory ? 0 : (dens ? -1 : 1)
    traj = NamedTrajectory(qtraj, N)
    u = drive_name(qtraj)
    per = [s isa AbstractVector ? collect(Symbol, s) : Symbol[s] for s in sampling_member_states(qtraj)]   # sub-state names per member
    gens = dens ? [begin
                       Gd, Gs = compact_lindbladian_generators(sys)
                       (Matrix{Float64}(Gd), [Matrix{Float64}(G) for G in Gs])
                   end for sys in qtraj.systems] : [_generators(sys) for sys in qtraj.systems]
    flat = reduce(vcat, per)
    owner = reduce(vcat, [fill(i, length(p)) for (i, p) in enumerate(per)])
    if all(g -> g[2] == gens[1][2], gens)       # exact equality: the same drive generators in every member -> ONE batched context
        return _integrators([gens[i][1] for i in owner], gens[1][2], traj, flat, u; state_cols = sc, kwargs...)
    end
    # members that differ in their drive generators too: a context per member, shared by the member's sub-states
    return reduce(vcat, [_integrators(fill(gens[i][1], length(per[i])), gens[i][2], traj, per[i], u; state_cols = sc, kwargs...) for i in eachindex(per)])
end

# ---- cached fused evaluation -------------------------------------------------------------------------------------
function _fresh!(core::PclCore, z::AbstractVector{Float64})
    if length(core.z) != length(z) || core.z != z
        core.z = copy(z)
        core.have_δ = core.have_vals = false
    end
    return core.z
end

function _all_δ!(core::PclCore, z)
    zc = _fresh!(core, z)
    if !core.have_δ
        length(core.δ) == core.rows_per * core.n_members || (core.δ = Vector{Float64}(undef, core.rows_per * core.n_members))
        δ = core.δ
        GC.@preserve zc δ check(core.ctx, ccall((:pcl_eval, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), core.ctx, zc, δ))
        core.have_δ = true; core.launches += 1
    end
    return core.δ
end

function _all_vals!(core::PclCore, z)
    zc = _fresh!(core, z)
    if !core.have_vals
        length(core.δ) == core.rows_per * core.n_members || (core.δ = Vector{Float64}(undef, core.rows_per * core.n_members))
        length(core.vals) == core.jac_per * core.n_members || (core.vals = Vector{Float64}(undef, core.jac_per * core.n_members))
        δ = core.δ; v = core.vals
        GC.@preserve zc δ v check(core.ctx, ccall((:pcl_eval_jac, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), core.ctx, zc, δ, v))
        core.have_δ = core.have_vals = true; core.launches += 1
    end
    return core.vals
end

_core(B::HipPadeIntegrator) = getfield(B, :core)
_rows(B::HipPadeIntegrator) = ((getfield(B, :member) - 1) * _core(B).rows_per + 1):(getfield(B, :member) * _core(B).rows_per)
_jrng(B::HipPadeIntegrator) = ((getfield(B, :member) - 1) * _core(B).jac_per + 1):(getfield(B, :member) * _core(B).jac_per)

# evaluate!(delta, B, traj)                                                   [REF integrators.jl:311,777]
function DirectTrajOpt.evaluate!(δ::AbstractVector{Float64}, B::HipPadeIntegrator, traj::NamedTrajectory)
    length(δ) == getfield(B, :dim) || throw(DimensionMismatch("δ has length $(length(δ)), integrator dim is $(getfield(B, :dim))"))
    copyto!(δ, view(_all_δ!(_core(B), traj.datavec), _rows(B)))
    return δ
end

# eval_jacobian(B, traj) -> sparse (B.dim, traj.dim*traj.N + traj.global_dim)  [REF integrators.jl:780-783]
function DirectTrajOpt.eval_jacobian(B::HipPadeIntegrator, traj::NamedTrajectory)
    vals = _all_vals!(_core(B), traj.datavec)[_jrng(B)]
    return sparse(getfield(B, :jac_rows), getfield(B, :jac_cols), vals, getfield(B, :dim), getfield(B, :n_vars))
end

# ---- what DirectTrajOpt's MOI evaluator needs per IPM iteration, in the shapes this library produces them: triplet structure
#      queried once, values in that order.  DirectTrajOpt is not vendored with Piccolo and its generic names differ between
#      the 0.9 and 0.10 lines, so the methods are defined under this module's own names AND -- at load time -- as methods of
#      whichever of the candidate generics the INSTALLED DirectTrajOpt defines (`_bind_to_directtrajopt!` below); selftest.jl
#      prints what was bound.
jacobian_structure(B::HipPadeIntegrator) = collect(zip(Int.(getfield(B, :jac_rows)), Int.(getfield(B, :jac_cols))))
hessian_structure(B::HipPadeIntegrator) = collect(zip(Int.(getfield(B, :hess_rows)), Int.(getfield(B, :hess_cols))))

function eval_constraint_and_jacobian!(δ::AbstractVector{Float64}, vals::AbstractVector{Float64}, B::HipPadeIntegrator, z::Vector{Float64})
    v = _all_vals!(_core(B), z)
    copyto!(vals, view(v, _jrng(B)))
    copyto!(δ, view(_core(B).δ, _rows(B)))
    return nothing
end

# Jacobian values alone, in structure order (the in-place filler shape)
function eval_jacobian_values!(vals::AbstractVector{Float64}, B::HipPadeIntegrator, z::AbstractVector{Float64})
    copyto!(vals, view(_all_vals!(_core(B), z), _jrng(B)))
    return vals
end

# Hessian of the Lagrangian of THIS integrator's rows: μ is the block's multiplier slice (length B.dim); runs on a
# one-member window of the shared context (every Pade order)
function eval_hessian_of_lagrangian!(vals::Vector{Float64}, B::HipPadeIntegrator, z::Vector{Float64}, μ::Vector{Float64})
    core = _core(B)
    length(μ) == getfield(B, :dim) || throw(DimensionMismatch("μ has length $(length(μ)), integrator dim is $(getfield(B, :dim))"))
    length(vals) == core.hess_per || throw(DimensionMismatch("vals has length $(length(vals)), expected $(core.hess_per)"))
    _window!(core, getfield(B, :member) - 1, 1)
    try
        GC.@preserve z μ vals check(core.ctx, ccall((:pcl_hess, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), core.ctx, z, μ, vals))
    finally
        _window!(core, 0, core.n_members)
    end
    return nothing
end

# Adaptive binding.  For every candidate generic the installed DirectTrajOpt defines, a method for HipPadeIntegrator is added
# that forwards to the functions above; the argument shapes tried are the ones DirectTrajOpt's own BilinearIntegrator methods
# use in the 0.9 / 0.10 lines (structure: (B) or (B, traj); in-place values: (vals, B, traj) / (B, traj) -> values;
# Hessian of the Lagrangian: (vals, B, traj, μ) / (B, traj, μ)).  Names the installed version does not define are skipped --
# the module-local functions stay available either way.  Returns the list of names bound (selftest.jl prints it).
const BOUND_GENERICS = Symbol[]
function _bind_to_directtrajopt!()
    empty!(BOUND_GENERICS)
    z_of(traj) = traj isa NamedTrajectory ? traj.datavec : traj
    for nm in (:jacobian_structure, :get_jacobian_structure)
        isdefined(DirectTrajOpt, nm) || continue
        @eval DirectTrajOpt.$nm(B::HipPadeIntegrator, args...) = jacobian_structure(B)
        push!(BOUND_GENERICS, nm)
    end
    for nm in (:hessian_structure, :hessian_of_lagrangian_structure, :get_hessian_structure)
        isdefined(DirectTrajOpt, nm) || continue
        @eval DirectTrajOpt.$nm(B::HipPadeIntegrator, args...) = hessian_structure(B)
        push!(BOUND_GENERICS, nm)
    end
    for nm in (:jacobian!, :eval_jacobian!, :jacobian_values!)
        isdefined(DirectTrajOpt, nm) || continue
        @eval DirectTrajOpt.$nm(vals::AbstractVector{Float64}, B::HipPadeIntegrator, traj, args...) =
            eval_jacobian_values!(vals, B, $z_of(traj))
        push!(BOUND_GENERICS, nm)
    end
    for nm in (:hessian_of_lagrangian!, :eval_hessian_of_lagrangian!, :hessian_of_lagrangian_values!)
        isdefined(DirectTrajOpt, nm) || continue
        @eval function DirectTrajOpt.$nm(vals::AbstractVector{Float64}, B::HipPadeIntegrator, traj, μ::AbstractVector{Float64}, args...)
            # the C entry point fills a dense Vector{Float64}: a view (or any other AbstractVector) is filled through a temporary
            if vals isa Vector{Float64}
                eval_hessian_of_lagrangian!(vals, B, Vector{Float64}($z_of(traj)), Vector{Float64}(μ))
            else
                tmp = Vector{Float64}(undef, length(vals))
                eval_hessian_of_lagrangian!(tmp, B, Vector{Float64}($z_of(traj)), Vector{Float64}(μ))
                copyto!(vals, tmp)
            end
            return nothing
        end
        push!(BOUND_GENERICS, nm)
    end
    for nm in (:hessian_of_lagrangian, :eval_hessian_of_lagrangian)
        isdefined(DirectTrajOpt, nm) || continue
        @eval function DirectTrajOpt.$nm(B::HipPadeIntegrator, traj, μ::AbstractVector{Float64}, args...)
            vals = Vector{Float64}(undef, _core(B).hess_per)
            eval_hessian_of_lagrangian!(vals, B, Vector{Float64}($z_of(traj)), Vector{Float64}(μ))
            return sparse(Int.(getfield(B, :hess_rows)), Int.(getfield(B, :hess_cols)), vals, getfield(B, :n_vars), getfield(B, :n_vars))
        end
        push!(BOUND_GENERICS, nm)
    end
    return BOUND_GENERICS
end
__init__() = _bind_to_directtrajopt!()

# ---- B.f(x_next, x, u, Δt): the scalar one-interval form the reference reads [REF integrators.jl:518-525,552;
#      src/control/display/inspect.jl:630-636] -- a cached 2-knot context with layout [x | Δt | u] -----------------------
function _f(B::HipPadeIntegrator, x_next::AbstractVector, x::AbstractVector, u::AbstractVector, Δt::Real)
    xd = getfield(B, :x_dim); Gj = getfield(B, :Gj); m = length(Gj)
    fc = getfield(B, :fcore)
    if fc === nothing
        fc, _ = _create_core([getfield(B, :G0)], Gj, 2, xd + 1 + m, xd + 1, xd, Int32[0], 0;
                             state_cols = getfield(B, :state_cols), pade_order = _order_in_use(getfield(B, :core)))   # the main context's order, read back
        setfield!(B, :fcore, fc)
    end
    z = zeros(2 * (xd + 1 + m))
    z[1:xd] .= x; z[xd+1] = Δt; z[(xd+2):(xd+1+m)] .= u[1:m]
    z[(xd+1+m+1):(xd+1+m+xd)] .= x_next
    δ = Vector{Float64}(undef, xd)
    GC.@preserve z δ check(fc.ctx, ccall((:pcl_eval, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), fc.ctx, z, δ))
    return δ
end

function Base.getproperty(B::HipPadeIntegrator, s::Symbol)
    s === :f && return (x_next, x, u, Δt) -> _f(B, x_next, x, u, Δt)
    s === :ctx && return getfield(B, :core).ctx
    s === :pade_order && return _order_in_use(getfield(B, :core))   # asked of the context: never a stale copy
    return getfield(B, s)
end
Base.propertynames(B::HipPadeIntegrator, private::Bool = false) = (fieldnames(HipPadeIntegrator)..., :f, :ctx)

export HipPadeIntegrator
end # module
