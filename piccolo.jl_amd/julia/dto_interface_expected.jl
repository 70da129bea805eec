# dto_interface_expected.jl -- the DirectTrajOpt integrator interface the glue is written against, AS DATA.
#
# DirectTrajOpt is not vendored with Piccolo (Project.toml compat "0.9.5, 0.10"; no Manifest), so what is listed here is what the
# reference ITSELF pins about the interface -- the call sites and tests under /root/reference -- plus the candidate generic names of the
# 0.9 / 0.10 lines the binder looks for.  selftest.jl compares the installed DirectTrajOpt with this table and FAILS (does not print)
# when a required role has no generic, when a bound generic has no method of the listed shape for HipPadeIntegrator, or when the
# installed DirectTrajOpt implements a generic for its own BilinearIntegrator that HipPadeIntegrator does not answer.
#
# role       : what the MOI evaluator / Piccolo needs
# required   : true = pinned by the reference's own code (a drop-in without it is not one); false = needed only when the installed
#              DirectTrajOpt defines one of the names (Hessian rows: `eval_hessian = true`, spline_pulse_problem.jl:96)
# names      : candidate generics, first match wins
# shape      : argument tuple the method must accept with B::HipPadeIntegrator (checked with hasmethod)
# pin        : reference file:line that fixes the role
const DTO_EXPECTED_INTERFACE = [
    (role = :evaluate,           required = true,  names = [:evaluate!],
     shape = (AbstractVector{Float64}, :B, :traj),                  pin = "src/control/integrators.jl:311,777; smooth_pulse_problem.jl:783"),
    (role = :eval_jacobian,      required = true,  names = [:eval_jacobian],
     shape = (:B, :traj),                                           pin = "src/control/integrators.jl:780-783 (size (B.dim, traj.dim*traj.N + traj.global_dim))"),
    (role = :test_integrator,    required = true,  names = [:test_integrator],
     shape = (:B, :traj),                                           pin = "src/control/integrators.jl:341-359 (atol = 1e-3)"),
    (role = :jacobian_structure, required = false, names = [:jacobian_structure, :get_jacobian_structure],
     shape = (:B,),                                                 pin = "test/aqua.jl:6-9 (exported thrice: ambiguity exclusions)"),
    (role = :hessian_structure,  required = false, names = [:hessian_structure, :hessian_of_lagrangian_structure, :get_hessian_structure],
     shape = (:B,),                                                 pin = "test/aqua.jl:6-9"),
    (role = :jacobian_values,    required = false, names = [:jacobian!, :eval_jacobian!, :jacobian_values!],
     shape = (AbstractVector{Float64}, :B, :traj),                  pin = "DirectTrajOpt's in-place filler (MOI.eval_constraint_jacobian)"),
    (role = :hessian_values,     required = false, names = [:hessian_of_lagrangian!, :eval_hessian_of_lagrangian!, :hessian_of_lagrangian_values!,
                                                             :hessian_of_lagrangian, :eval_hessian_of_lagrangian],
     shape = (:B, :traj, AbstractVector{Float64}),                  pin = "spline_pulse_problem.jl:96 (eval_hessian = true)"),
]
# properties the reference reads off an integrator object [REF src/control/integrators.jl:307-309,525,552; src/control/display/inspect.jl:630-636]
const DTO_EXPECTED_PROPERTIES = (:dim, :x_dim, :x_name, :x_names, :f)
