"""BASELINE configs[0]: the evaluator driving a whole NLP solve through the solver's callbacks (GPU residual, Jacobian,
Hessian of the Lagrangian; CPU trust-constr instead of Ipopt).  Outcome asserts follow the reference's integration tests:
`fidelity > 0.9`, `norm(delta, Inf) < 1e-2` after a few hundred iterations [REF src/control/templates/
smooth_pulse_problem.jl:745-785]."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))

pytestmark = pytest.mark.gpu


def test_xgate_plumbing_solve():
    import plumbing_xgate

    r = plumbing_xgate.solve(N=50, max_iter=300, seed=0)
    assert r["n_vars"] == 16 * 50 and r["n_rows"] == (8 + 2 + 2 + 1) * 49
    assert r["fidelity"] > 0.9, r
    assert r["max_violation"] < 1e-2, r


def test_hessian_of_the_lagrangian_is_complete_on_the_device():
    """MOI.eval_hessian_lagrangian = sigma grad^2 f + sum_i mu_i grad^2 g_i [REF spline_pulse_problem.jl:96 `eval_hessian = true`]: both
    terms come from the device (pcl_objective_hess: terminal infidelity block + regulariser entries; pcl_hess + the derivative rows) and
    their sum is the derivative of the device's own gradient of the Lagrangian, grad f + J' mu, along random directions."""
    import numpy as np

    import plumbing_xgate

    cb = plumbing_xgate.solve(N=12, callbacks_only=True)
    rng = np.random.default_rng(3)
    z = cb["z0"] + 0.05 * rng.standard_normal(cb["z0"].size)
    mu = rng.standard_normal(cb["n_rows"])
    gradL = lambda zz: cb["obj"](zz)[1] + cb["cons_jac"](zz).T @ mu
    H = (cb["obj_hess"](z) + cb["cons_hess"](z, mu)).toarray()
    assert np.abs(H - H.T).max() < 1e-12 * np.abs(H).max()
    for _ in range(6):
        e = rng.standard_normal(z.size)
        e /= np.linalg.norm(e)
        fd = (gradL(z + 1e-6 * e) - gradL(z - 1e-6 * e)) / 2e-6
        assert np.abs(H @ e - fd).max() < 1e-6 * max(1.0, np.abs(fd).max())
    cb["close"]()
