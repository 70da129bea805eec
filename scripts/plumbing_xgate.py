#!/usr/bin/env python3
"""BASELINE.json configs[0] ("plumbing"): single-qubit X gate, T=50 knots, 4th-order Pade collocation, solved with a
CPU NLP solver whose constraint callbacks (residual, sparse Jacobian, Hessian of the Lagrangian) are served by the GPU
evaluator -- the shape of `solve!(SmoothPulseProblem(qtraj, N))` [REF src/control/templates/smooth_pulse_problem.jl:
240-295, src/control/problems.jl:409-429] with scipy's trust-constr standing in for Ipopt (no Ipopt in the image).
Constraint list = prob.integrators order: [dynamics, DerivativeIntegrator(u,du), DerivativeIntegrator(du,ddu)] + time
consistency [REF smooth_pulse_problem.jl:264-277]."""
import os
import sys

import numpy as np
import scipy.linalg
import scipy.sparse as sp
from scipy.optimize import BFGS, Bounds, NonlinearConstraint, minimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piccolo_jl_amd as pa


def solve(N=50, T=10.0, Q=100.0, R=1e-2, seed=0, max_iter=300, verbose=0, exact_hessian=False, callbacks_only=False):
    """exact_hessian: the objective's Hessian from the device too (pcl_objective_hess) -- with the constraints' term (pcl_hess) the solver
    then works on the exact Hessian of the Lagrangian, the reference's `eval_hessian = true` [REF spline_pulse_problem.jl:96]; otherwise a
    BFGS model of the objective.  (scipy's trust-constr has no inertia correction: on the exact, indefinite Hessian of Q |1 - F| it stops
    at an infeasible stationary point of this problem -- fidelity 1, violation 0.8 -- where Ipopt would regularise; the BFGS model is what
    the plumbing test solves with.  callbacks_only: return the callbacks instead of solving: tests/test_plumbing_gpu.py checks that the
    device's Hessian of the Lagrangian is the derivative of the device's gradient of the Lagrangian.)"""
    system = pa.QuantumSystem(0.5 * pa.PAULIS["Z"], [pa.PAULIS["X"], pa.PAULIS["Y"]], [1.0, 1.0])  # first_gate.jl:42-48
    U_goal = pa.GATES["X"]
    rng = np.random.default_rng(seed)
    times = np.linspace(0, T, N)
    u0 = 0.1 * rng.standard_normal((2, N))
    u0[:, 0] = u0[:, -1] = 0.0
    # rollout for the initial states (piecewise-constant exact propagation)
    states, U = [], np.eye(2, dtype=complex)
    for k in range(N):
        states.append(U)
        if k + 1 < N:
            U = scipy.linalg.expm(-1j * (times[k + 1] - times[k]) * system.H(u0[:, k])) @ U
    traj = pa.unitary_trajectory(system, u0, times, U_goal, states=states)
    B = pa.BilinearIntegrator(system, traj, pade_order=4)
    rows = [B, pa.DerivativeIntegrator("u", "du", traj, like=B), pa.DerivativeIntegrator("du", "ddu", traj, like=B),
            pa.DerivativeIntegrator("t", None, traj, like=B)]  # fmt: skip
    nv = traj.dim * traj.N
    structs = [pa.jacobian_structure(r) for r in rows]
    offs = np.cumsum([0] + [r.dim for r in rows])
    comp = traj.components

    def cons(z):
        traj.update(z)
        return np.concatenate([pa.evaluate_(np.zeros(r.dim), r, traj) for r in rows])

    def cons_jac(z):
        traj.update(z)
        mats = []
        for r, (rr, cc), o in zip(rows, structs, offs):
            mats.append(sp.csr_matrix((r.ctx.jac(traj.datavec), (rr, cc)), shape=(r.dim, nv)))
        return sp.vstack(mats).tocsr()

    def cons_hess(z, v):
        traj.update(z)
        H = pa.eval_hessian_of_lagrangian(B, traj, v[: B.dim])
        # derivative rows: d^2/(d dt_k d dx_k[r]) = -1
        ii, jj, vv = [], [], []
        for r, o in zip(rows[1:3], offs[1:3]):
            mu = v[o : o + r.dim].reshape(N - 1, r.x_dim)
            for k in range(N - 1):
                a = k * traj.dim + comp["Δt"].start
                b = k * traj.dim + r.dx_off + np.arange(r.x_dim)
                ii += [np.full(r.x_dim, a), b]
                jj += [b, np.full(r.x_dim, a)]
                vv += [-mu[k], -mu[k]]
        return H + sp.csr_matrix((np.concatenate(vv), (np.concatenate(ii), np.concatenate(jj))), shape=(nv, nv))

    # objective on the GPU: UnitaryInfidelityObjective + 3 x QuadraticRegularizer [REF smooth_pulse_problem.jl:240-251]
    J = pa.UnitaryInfidelityObjective(U_goal, "Ũ⃗", traj, Q=Q)
    for c_ in ("u", "du", "ddu"):
        J = J + pa.QuadraticRegularizer(c_, traj, R, dt_power=0)
    J.bind(B)

    def obj(z):
        return J.value_and_gradient(z)

    hr, hc = J.hessian_structure()

    def obj_hess(z):  # sigma grad^2 f, lower triangle from the device -> symmetric sparse matrix
        L = sp.csr_matrix((J.hessian(z, 1.0), (hr, hc)), shape=(nv, nv))
        return L + sp.tril(L, -1).T

    if callbacks_only:
        return dict(z0=traj.datavec.copy(), obj=obj, obj_hess=obj_hess, cons=cons, cons_jac=cons_jac, cons_hess=cons_hess, n_rows=int(offs[-1]), close=B.close)
    lb, ub = np.full(nv, -np.inf), np.full(nv, np.inf)
    for k in range(N):
        o = k * traj.dim
        lb[o + comp["Ũ⃗"].start : o + comp["Ũ⃗"].stop], ub[o + comp["Ũ⃗"].start : o + comp["Ũ⃗"].stop] = -1.0, 1.0
        lb[o + comp["u"].start : o + comp["u"].stop], ub[o + comp["u"].start : o + comp["u"].stop] = -1.0, 1.0
        lb[o + comp["ddu"].start : o + comp["ddu"].stop], ub[o + comp["ddu"].start : o + comp["ddu"].stop] = -2.0, 2.0
        lb[o + comp["Δt"].start] = ub[o + comp["Δt"].start] = traj.datavec[o + comp["Δt"].start]  # timesteps_all_equal
    z0 = traj.datavec.copy()
    x1 = slice(comp["Ũ⃗"].start, comp["Ũ⃗"].stop)
    lb[x1] = ub[x1] = z0[x1]  # initial condition
    for k in (0, N - 1):  # u(0) = u(T) = 0
        s = slice(k * traj.dim + comp["u"].start, k * traj.dim + comp["u"].stop)
        lb[s] = ub[s] = 0.0
    z0 = np.clip(z0, lb, ub)
    nc_rows = int(offs[-1])
    res = minimize(obj, z0, jac=True, method="trust-constr", hess=obj_hess if exact_hessian else BFGS(), bounds=Bounds(lb, ub, keep_feasible=False),
                   constraints=[NonlinearConstraint(cons, np.zeros(nc_rows), np.zeros(nc_rows), jac=cons_jac, hess=cons_hess)],
                   options=dict(maxiter=max_iter, gtol=1e-8, xtol=1e-12, verbose=verbose, sparse_jacobian=True))  # fmt: skip
    traj.update(res.x)
    viol = np.abs(cons(res.x)).max()
    fid = 1.0 - pa.Objective([pa.UnitaryInfidelityObjective(U_goal, "Ũ⃗", traj, Q=1.0)]).bind(B).value_and_gradient(res.x, want_grad=False)[0]
    B.close()
    return dict(fidelity=float(fid), max_violation=float(viol), iterations=int(res.nit), n_vars=nv, n_rows=nc_rows, traj=traj)


if __name__ == "__main__":
    r = solve(verbose=1)
    print({k: v for k, v in r.items() if k != "traj"})
