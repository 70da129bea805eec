# reference_cpu.jl -- comparator B1 of SURVEY.md section 8(d) / BASELINE.md section 2: the REAL reference path on the host.
#
#     JULIA_NUM_THREADS=<cores of one socket> julia --project=<env with Piccolo 2.0.2> bench/reference_cpu.jl [reps]
#
# NOT EXECUTED in the build container or on the GPU boxes (no Julia there, nothing can be installed): it ships so that a
# maintainer with Julia + Piccolo + DirectTrajOpt can put the reference's own number next to bench.py's `cpu_baseline`
# (which is this repository's analytic C restatement -- a far faster CPU path than ForwardDiff through expv).
#
# Problem = BASELINE.json config 3, built exactly as bench.py builds it (SURVEY.md section 8(d) "Synthetic inputs"):
#   MultiTransmonSystem(ωs = [4.0, 4.1, 4.2], δs = [0.2, 0.21, 0.22], gs = [0 .01 .02; .01 0 .03; .02 .03 0];
#                       levels_per_transmon = 3, drive_bounds = 0.1)      [REF src/quantum/templates/transmons/transmon_system.jl:341-343]
#   N = 100 knots, Δt = 0.1, u ~ 0.02 N(0,1) clipped to ±0.1, states near the exact rollout.
# Timed: one `evaluate!` + one `eval_jacobian` of the dynamics integrator [REF src/control/integrators.jl:311,780] = one
# "constraint+Jacobian eval" of BASELINE.json's metric.  Prints ONE JSON line.
using Piccolo, DirectTrajOpt, NamedTrajectories, LinearAlgebra, Random, Printf

reps = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 20
sys = MultiTransmonSystem([4.0, 4.1, 4.2], [0.2, 0.21, 0.22], [0 0.01 0.02; 0.01 0 0.03; 0.02 0.03 0];
                          levels_per_transmon = 3, drive_bounds = 0.1)
N = 100
times = collect(range(0, 0.1 * (N - 1), length = N))
Random.seed!(20260929 + 3)
controls = clamp.(0.02 .* randn(sys.n_drives, N), -0.1, 0.1)
pulse = ZeroOrderPulse(controls, times)
U_goal = Matrix{ComplexF64}(I, sys.levels, sys.levels)
qtraj = UnitaryTrajectory(sys, pulse, U_goal)
traj = NamedTrajectory(qtraj, N)
# near-feasible states, like an interior-point iterate
traj.datavec .+= 1e-3 .* randn(length(traj.datavec)) .* (repeat([i in traj.components[:Ũ⃗] for i in 1:traj.dim], N))
B = BilinearIntegrator(qtraj, N)
δ = zeros(B.dim)
DirectTrajOpt.evaluate!(δ, B, traj); J = DirectTrajOpt.eval_jacobian(B, traj)     # warm-up / compilation
t = @elapsed for _ in 1:reps
    DirectTrajOpt.evaluate!(δ, B, traj)
    DirectTrajOpt.eval_jacobian(B, traj)
end
@printf("{\"metric\": \"constraint+Jacobian evals/sec, 3-transmon d=27 unitary, T=100 knots\", \"value\": %.6g, \"unit\": \"evals/s\", \"kind\": \"reference\", \"julia_threads\": %d, \"cpu_threads\": %d, \"reps\": %d, \"s_per_eval\": %.6g, \"jac_nnz\": %d, \"DirectTrajOpt\": \"%s\"}\n",
        reps / t, Threads.nthreads(), Sys.CPU_THREADS, reps, t / reps, nnz(J), string(pkgversion(DirectTrajOpt)))
