# HipPadeIntegrator.jl -- Julia-side glue of the drop-in: a `DirectTrajOpt.AbstractIntegrator`
# whose arithmetic runs in libpiccolo_hip.so (HIP kernels on an MI355X) through `ccall`.
#
# STATUS: written against the interface Piccolo.jl v2.0.2 *uses* (call sites cited below); the
# abstract type and the generic functions live in the un-vendored DirectTrajOpt.jl (compat 0.9.5 / 0.10)
# and there is no Julia in the build container, so this file has NOT been executed.  All logic is in
# the C library (include/piccolo_hip.h); this file only marshals arguments.  See INTEGRATION.md.
#
#   plug-in points in the reference:
#     SmoothPulseProblem(qtraj, N; integrator = HipPadeIntegrator(qtraj, N))
#         src/control/templates/smooth_pulse_problem.jl:123,213-233
#     SamplingProblem(qcp, systems; integrator = (sq, N) -> HipPadeIntegrator(sq, N))
#         src/control/templates/sampling_problem.jl:190-237,292
#     Specs.register_integrator!(:hip_pade, RegistryEntry(factory = (qtraj, N; alg) -> HipPadeIntegrator(qtraj, N)))
#         src/specs/registries.jl:82-86,112,119 ; src/specs/materialize.jl:216-222
module HipPade

using LinearAlgebra, SparseArrays
using NamedTrajectories
using DirectTrajOpt
import DirectTrajOpt: AbstractIntegrator
using Piccolo: get_system, state_name, state_names, drive_name, UnitaryTrajectory, SamplingTrajectory

const LIB = get(ENV, "PICCOLO_HIP_LIB", "libpiccolo_hip.so")

# mirror of `pcl_desc` (include/piccolo_hip.h) -- field order and widths must match
struct PclDesc
    struct_size::Int32; d::Int32; n_drives::Int32; N::Int32; z_dim::Int32
    u_off::Int32; dt_off::Int32; batch::Int32; batch_mode::Int32; pade_order::Int32
    device_id::Int32; index_base::Int32; per_member_G0::Int32; state_cols::Int32
    global_dim::Int64
    G0::Ptr{Float64}; Gj::Ptr{Float64}; x_offs::Ptr{Int32}
end

check(ctx, rc) = rc == 0 || error("libpiccolo_hip: ",
    unsafe_string(ccall((:pcl_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx)), " (code $rc)")

mutable struct HipPadeIntegrator <: AbstractIntegrator
    ctx::Ptr{Cvoid}
    x_name::Symbol            # first (or only) state component
    x_names::Vector{Symbol}
    u_name::Symbol
    x_dim::Int
    dim::Int                  # == x_dim * (N - 1) * length(x_names)        [REF integrators.jl:309]
    jac_rows::Vector{Int32}   # 1-based, in value order (never assumed, always queried)
    jac_cols::Vector{Int32}
    hess_rows::Vector{Int32}
    hess_cols::Vector{Int32}
    n_vars::Int
end

function _create(G0s::Vector{<:AbstractMatrix}, Gjs::Vector{<:AbstractMatrix}, traj::NamedTrajectory,
                 x_names::Vector{Symbol}, u_name::Symbol; device::Integer = 0, pade_order::Integer = 4,
                 state_cols::Integer = 0)
    # state_cols: 0 unitary (n = 2d), 1 ket, -1 = PCL_STATE_VECTOR (general n x n generator on one real column, d := n)
    n = size(G0s[1], 1); d = state_cols == -1 ? n : n ÷ 2; m = length(Gjs)
    G0 = reduce(vcat, [vec(Matrix{Float64}(G)) for G in G0s])          # column-major, one block per member
    Gj = m == 0 ? zeros(1) : reduce(vcat, [vec(Matrix{Float64}(G)) for G in Gjs])
    x_offs = Int32[traj.components[nm][1] - 1 for nm in x_names]
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve G0 Gj x_offs begin
        desc = PclDesc(sizeof(PclDesc), d, m, traj.N, traj.dim,
                       traj.components[u_name][1] - 1, traj.components[traj.timestep][1] - 1,
                       length(x_names), 0 #= PCL_BATCH_MEMBERS =#, pade_order #= 2, 4, 6, 8 or 10 =#, device, 1 #= 1-based =#,
                       length(G0s) > 1 ? 1 : 0, state_cols, traj.global_dim,
                       pointer(G0), pointer(Gj), pointer(x_offs))
        rc = ccall((:pcl_create, LIB), Cint, (Ref{PclDesc}, Ref{Ptr{Cvoid}}), desc, ctx)
        rc == 0 || error("pcl_create: ", unsafe_string(ccall((:pcl_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    end
    c = ctx[]
    nnz = Ref{Int64}(0); per = Ref{Int64}(0); xd = Ref{Int64}(0); nr = Ref{Int64}(0); ncol = Ref{Int64}(0)
    check(c, ccall((:pcl_constraint_dim, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}), c, xd, nr, ncol))
    check(c, ccall((:pcl_jac_nnz, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), c, nnz, per))
    jr = Vector{Int32}(undef, nnz[]); jc = similar(jr)
    check(c, ccall((:pcl_jac_structure, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}), c, jr, jc))
    check(c, ccall((:pcl_hess_nnz, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), c, nnz, per))
    hr = Vector{Int32}(undef, nnz[]); hc = similar(hr)
    check(c, ccall((:pcl_hess_structure, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}), c, hr, hc))
    B = HipPadeIntegrator(c, x_names[1], x_names, u_name, Int(xd[]) * length(x_names), Int(nr[]), jr, jc, hr, hc, Int(ncol[]))
    finalizer(b -> (b.ctx == C_NULL || ccall((:pcl_destroy, LIB), Cvoid, (Ptr{Cvoid},), b.ctx); b.ctx = C_NULL), B)
    return B
end

# BilinearIntegrator(qtraj::UnitaryTrajectory, N)                              [REF src/control/integrators.jl:35-51]
function HipPadeIntegrator(qtraj::UnitaryTrajectory, N::Int; kwargs...)
    sys = get_system(qtraj)
    sys.time_dependent && error("HipPadeIntegrator: time-dependent systems use TimeDependentBilinearIntegrator")
    traj = NamedTrajectory(qtraj, N)
    m = sys.n_drives
    e(j) = (u = zeros(m); u[j] = 1.0; u)
    G0 = Matrix(sys.G(zeros(m), 0.0))
    Gj = [Matrix(sys.G(e(j), 0.0)) - G0 for j in 1:m]          # linear drives: G(u) = G0 + sum u_j G_j
    return _create([G0], Gj, traj, [state_name(qtraj)], drive_name(qtraj); kwargs...)
end

# BilinearIntegrator(qtraj::SamplingTrajectory, N): one member per system, shared controls [REF integrators.jl:134-162]
function HipPadeIntegrator(qtraj::SamplingTrajectory, N::Int; kwargs...)
    traj = NamedTrajectory(qtraj, N)
    m = qtraj.systems[1].n_drives
    e(j) = (u = zeros(m); u[j] = 1.0; u)
    G0s = [Matrix(s.G(zeros(m), 0.0)) for s in qtraj.systems]
    Gj = [Matrix(qtraj.systems[1].G(e(j), 0.0)) - G0s[1] for j in 1:m]
    return _create(G0s, Gj, traj, state_names(qtraj), drive_name(qtraj); kwargs...)
end

# BilinearIntegrator(qtraj::KetTrajectory, N) [REF integrators.jl:58-74]
function HipPadeIntegrator(qtraj::KetTrajectory, N::Int; kwargs...)
    sys = get_system(qtraj); traj = NamedTrajectory(qtraj, N); m = sys.n_drives
    e(j) = (u = zeros(m); u[j] = 1.0; u)
    G0 = Matrix(sys.G(zeros(m), 0.0))
    Gj = [Matrix(sys.G(e(j), 0.0)) - G0 for j in 1:m]
    return _create([G0], Gj, traj, [state_name(qtraj)], drive_name(qtraj); state_cols = 1, kwargs...)
end

# BilinearIntegrator(qtraj::DensityTrajectory, N) [REF integrators.jl:82-95]: compact Lindbladian generators, linear
# drives and constant dissipation rates (compact_lindbladian_generators folds the dissipators into the drift)
function HipPadeIntegrator(qtraj::DensityTrajectory, N::Int; kwargs...)
    sys = get_system(qtraj); traj = NamedTrajectory(qtraj, N)
    Gc_drift, Gc_drives = compact_lindbladian_generators(sys)
    return _create([Matrix(Gc_drift)], [Matrix(G) for G in Gc_drives], traj, [state_name(qtraj)], drive_name(qtraj);
                   state_cols = -1, kwargs...)
end

_z(traj::NamedTrajectory) = traj.datavec    # knot-major flat buffer, passed as is

# evaluate!(delta, B, traj)                                                   [REF integrators.jl:311,777]
function DirectTrajOpt.evaluate!(δ::AbstractVector{Float64}, B::HipPadeIntegrator, traj::NamedTrajectory)
    length(δ) == B.dim || throw(DimensionMismatch("δ has length $(length(δ)), integrator dim is $(B.dim)"))
    z = _z(traj)
    GC.@preserve z δ check(B.ctx, ccall((:pcl_eval, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), B.ctx, z, δ))
    return δ
end

# eval_jacobian(B, traj) -> sparse (B.dim, traj.dim*traj.N + traj.global_dim)  [REF integrators.jl:780-783]
function DirectTrajOpt.eval_jacobian(B::HipPadeIntegrator, traj::NamedTrajectory)
    vals = Vector{Float64}(undef, length(B.jac_rows)); z = _z(traj)
    GC.@preserve z vals check(B.ctx, ccall((:pcl_jac, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), B.ctx, z, vals))
    return sparse(B.jac_rows, B.jac_cols, vals, B.dim, B.n_vars)
end

# what DirectTrajOpt's MOI evaluator needs per IPM iteration (names to be matched to the installed DTO version):
jacobian_structure(B::HipPadeIntegrator) = collect(zip(Int.(B.jac_rows), Int.(B.jac_cols)))
hessian_structure(B::HipPadeIntegrator) = collect(zip(Int.(B.hess_rows), Int.(B.hess_cols)))

function eval_constraint_and_jacobian!(δ::Vector{Float64}, vals::Vector{Float64}, B::HipPadeIntegrator, z::Vector{Float64})
    GC.@preserve z δ vals check(B.ctx, ccall((:pcl_eval_jac, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), B.ctx, z, δ, vals))
    return nothing
end

function eval_hessian_of_lagrangian!(vals::Vector{Float64}, B::HipPadeIntegrator, z::Vector{Float64}, μ::Vector{Float64})
    GC.@preserve z μ vals check(B.ctx, ccall((:pcl_hess, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), B.ctx, z, μ, vals))
    return nothing
end

# B.f(x_next, x, u, Δt): scalar form used by the reference's cross-integrator test [REF integrators.jl:518-525]
function Base.getproperty(B::HipPadeIntegrator, s::Symbol)
    s === :f || return getfield(B, s)
    return (x_next, x, u, Δt) -> error("HipPadeIntegrator.f: build a 2-knot trajectory and call evaluate! (see INTEGRATION.md)")
end

export HipPadeIntegrator
end # module
