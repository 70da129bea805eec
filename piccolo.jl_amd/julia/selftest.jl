# selftest.jl -- first thing to run on a box that has Julia + Piccolo 2.0.2 + DirectTrajOpt and an MI355X:
#
#     PICCOLO_HIP_LIB=/path/to/libpiccolo_hip.so julia --project=<env with Piccolo> selftest.jl
#
# NOT EXECUTED in the build container (no Julia there).  What it does, in order:
#   1. prints the DirectTrajOpt version and every DirectTrajOpt name that looks like part of the integrator interface, then CHECKS the
#      installed interface against dto_interface_expected.jl and FAILS (error, exit code 1) when
#        * a required role (evaluate!, eval_jacobian, test_integrator) has no generic or HipPadeIntegrator no method of the listed shape,
#        * an optional role's generic exists in the installed DirectTrajOpt but the binder attached nothing to it,
#        * DirectTrajOpt implements a function for its own BilinearIntegrator that has no method accepting a HipPadeIntegrator in the
#          same argument position (constructors and Base.show aside) -- the list of exactly what the glue still has to provide;
#   2. checks the glue with DirectTrajOpt's own `test_integrator` (the finite-difference gate the reference applies to
#      its integrators, [REF src/control/integrators.jl:359,382,413]);
#   3. compares `evaluate!` / `eval_jacobian` of the glue with DirectTrajOpt's BilinearIntegrator on the same trajectory:
#      the two differ by the Pade truncation error only (order 10: ~1e-11 at these step sizes), see DESIGN.md section 1;
#   4. checks `B.f` against one interval of `evaluate!` [REF integrators.jl:518-525];
#   5. checks the SamplingTrajectory shape: a Vector with one integrator per member that `SamplingProblem` accepts
#      [REF src/control/templates/sampling_problem.jl:190-237].
using Piccolo, DirectTrajOpt, NamedTrajectories, LinearAlgebra, SparseArrays, Test

include(joinpath(@__DIR__, "HipPadeIntegrator.jl"))
using .HipPade

println("DirectTrajOpt ", pkgversion(DirectTrajOpt), "   Piccolo ", pkgversion(Piccolo))
println("\n-- DirectTrajOpt names that look like integrator interface --")
for nm in sort(names(DirectTrajOpt; all = true))
    s = string(nm)
    if occursin(r"jacobian|hessian|evaluate|integrator|structure"i, s) && !startswith(s, "#")
        println("  ", s, isdefined(DirectTrajOpt, nm) && getfield(DirectTrajOpt, nm) isa Function ?
                "   (" * string(length(methods(getfield(DirectTrajOpt, nm)))) * " methods)" : "")
    end
end
println("\n-- methods that accept a DirectTrajOpt.BilinearIntegrator --")
for mth in methodswith(DirectTrajOpt.BilinearIntegrator; supertypes = true)
    println("  ", mth)
end
println("\n-- fields of DirectTrajOpt.BilinearIntegrator --\n  ", fieldnames(DirectTrajOpt.BilinearIntegrator))

# ---- the interface check: red / green in one command ------------------------------------------------------------------------------
include(joinpath(@__DIR__, "dto_interface_expected.jl"))
println("\n-- generics the binder attached methods to --\n  ", HipPade.BOUND_GENERICS)
problems = String[]
_shape(sh) = Tuple{map(a -> a === :B ? HipPade.HipPadeIntegrator : (a === :traj ? NamedTrajectory : a), sh)...}
for e in DTO_EXPECTED_INTERFACE
    defined = [nm for nm in e.names if isdefined(DirectTrajOpt, nm)]
    if isempty(defined)
        e.required && push!(problems, "role $(e.role): DirectTrajOpt $(pkgversion(DirectTrajOpt)) defines none of $(e.names)  [pin: $(e.pin)]")
        continue
    end
    ok = any(nm -> hasmethod(getfield(DirectTrajOpt, nm), _shape(e.shape)) ||
                   (e.role == :hessian_values && hasmethod(getfield(DirectTrajOpt, nm), Tuple{AbstractVector{Float64},HipPade.HipPadeIntegrator,NamedTrajectory,AbstractVector{Float64}})),
             defined)
    ok || push!(problems, "role $(e.role): $(defined) exist in DirectTrajOpt but none has a method of shape $(e.shape) for HipPadeIntegrator  [pin: $(e.pin)]")
end
# every function DirectTrajOpt implements for its own BilinearIntegrator must answer a HipPadeIntegrator in the same position
for mth in methodswith(DirectTrajOpt.BilinearIntegrator; supertypes = false)
    mth.name in (:BilinearIntegrator, :show, :print, :summary) && continue
    f = try getfield(mth.module, mth.name) catch; nothing end
    f isa Function || continue
    sig = Base.unwrap_unionall(mth.sig)
    params = collect(sig.parameters)[2:end]
    any(p -> p isa Type && p <: DirectTrajOpt.BilinearIntegrator, params) || continue
    swapped = map(p -> (p isa Type && p <: DirectTrajOpt.BilinearIntegrator) ? HipPade.HipPadeIntegrator : (p isa TypeVar ? Any : p), params)
    hasmethod(f, Tuple{swapped...}) || push!(problems, "$(mth.module).$(mth.name)$(Tuple(params)) has no counterpart accepting a HipPadeIntegrator")
end
let B0 = HipPadeIntegrator(UnitaryTrajectory(QuantumSystem(GATES[:Z], [GATES[:X], GATES[:Y]], [1.0, 1.0]),
                                              ZeroOrderPulse(zeros(2, 5), collect(range(0, 1.0, length = 5))), GATES[:X]), 5)
    for pr in DTO_EXPECTED_PROPERTIES
        hasproperty(B0, pr) || push!(problems, "property $pr is read by the reference and missing on HipPadeIntegrator")
    end
end
if !isempty(problems)
    println("\n== INTERFACE CHECK FAILED ==")
    foreach(p -> println("  * ", p), problems)
    error("HipPadeIntegrator does not yet answer the installed DirectTrajOpt's integrator interface ($(length(problems)) item(s) above)")
end
println("\n== interface check passed: every role of dto_interface_expected.jl is answered ==")

@testset "HipPadeIntegrator vs DirectTrajOpt" begin
    # the reference's own dispatch test case [REF src/control/integrators.jl:335-360]
    sys = QuantumSystem(GATES[:Z], [GATES[:X], GATES[:Y]], [1.0, 1.0])
    N = 11
    times = collect(range(0, 1.0, length = N))
    pulse = ZeroOrderPulse(0.1 * randn(2, N), times)
    qtraj = UnitaryTrajectory(sys, pulse, GATES[:X])
    traj = NamedTrajectory(qtraj, N)
    Bref = BilinearIntegrator(qtraj, N)
    for p in (4, 10)
        B = HipPadeIntegrator(qtraj, N; pade_order = p)
        @test B isa DirectTrajOpt.AbstractIntegrator
        @test B.dim == Bref.dim && B.x_dim == Bref.x_dim && B.x_name == Bref.x_name
        δ = zeros(B.dim); δref = zeros(Bref.dim)
        DirectTrajOpt.evaluate!(δ, B, traj); DirectTrajOpt.evaluate!(δref, Bref, traj)
        # B^- (x_{k+1} - R_p x_k) vs x_{k+1} - exp x_k: equal up to O(residual) + the truncation error of R_p
        println("pade_order $p: |δ - δ_ref|_inf = ", norm(δ - δref, Inf), "   |δ_ref|_inf = ", norm(δref, Inf))
        J = DirectTrajOpt.eval_jacobian(B, traj); Jref = DirectTrajOpt.eval_jacobian(Bref, traj)
        @test size(J) == size(Jref) == (B.dim, traj.dim * traj.N + traj.global_dim)
        println("pade_order $p: |J - J_ref|_inf = ", norm(Matrix(J) - Matrix(Jref), Inf))
        p == 10 && @test norm(Matrix(J) - Matrix(Jref), Inf) < 1e-4      # the residual itself is O(1e-1) on this random trajectory
        # the finite-difference gate of DirectTrajOpt (uses whatever interface its version defines: a MethodError here
        # names exactly the method the glue still has to provide)
        test_integrator(B, traj; atol = 1e-3)
        # B.f == one interval of evaluate!
        k = 3
        xk = traj[k][B.x_name]; xn = traj[k+1][B.x_name]
        @test B.f(xn, xk, traj[k].u, traj[k].Δt[1]) ≈ δ[((k-1)*B.x_dim+1):(k*B.x_dim)] atol = 1e-13
    end
    # ensemble: one integrator per member, accepted by SamplingProblem
    systems = [QuantumSystem(s * GATES[:Z], [GATES[:X], GATES[:Y]], [1.0, 1.0]) for s in (1.0, 1.05, 0.95)]
    sq = SamplingTrajectory(qtraj, systems)
    Bs = HipPadeIntegrator(sq, N)
    Brefs = BilinearIntegrator(sq, N)
    @test Bs isa AbstractVector && length(Bs) == length(Brefs) == 3
    trajS = NamedTrajectory(sq, N)
    for (B, Br) in zip(Bs, Brefs)
        @test B.dim == Br.dim && B.x_name == Br.x_name
        δ = zeros(B.dim); δr = zeros(Br.dim)
        DirectTrajOpt.evaluate!(δ, B, trajS); DirectTrajOpt.evaluate!(δr, Br, trajS)
        @test norm(δ - δr, Inf) < 1e-6
    end
    @test Bs[1].core.launches == 1      # one fused launch served the three members
    qcp = SmoothPulseProblem(qtraj, N; integrator = HipPadeIntegrator(qtraj, N))
    sp = SamplingProblem(qcp, systems; integrator = (sq_, n_) -> HipPadeIntegrator(sq_, n_))
    @test sp isa QuantumControlProblem
end
