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
