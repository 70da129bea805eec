"""Host-side (numpy) terminal objective used by the plumbing solve: the reference's unitary infidelity
``Q * |1 - |tr(U_goal' U_N)|^2 / n^2|`` [REF src/control/objectives.jl:330-356] and its gradient with respect to
the terminal iso-vec.  SURVEY.md section 8(f) lists the on-device version as the next row after the constraint
path; this small host function exists so that a whole NLP can be driven through the GPU evaluator's callbacks."""
import numpy as np

from .quantum import iso_vec_to_operator


def unitary_fidelity_loss(x, U_goal):
    """|tr(U_goal' U)|^2 / n^2 (the reference calls this the *fidelity loss*; it is the fidelity)."""
    U = iso_vec_to_operator(x)
    n = U.shape[0]
    return abs(np.trace(np.asarray(U_goal).conj().T @ U)) ** 2 / n**2


def unitary_infidelity(x, U_goal, Q=100.0):
    """(value, gradient w.r.t. the iso-vec x) of Q * |1 - F(x)|."""
    Ug = np.asarray(U_goal, dtype=complex)
    U = iso_vec_to_operator(x)
    n = U.shape[0]
    t = np.trace(Ug.conj().T @ U)
    F = abs(t) ** 2 / n**2
    W = np.conj(t) * np.conj(Ug)  # d|t|^2/dRe U = 2 Re W ; d|t|^2/dIm U = -2 Im W
    gU = np.vstack((2 * W.real, -2 * W.imag)) / n**2  # [Re; Im] blocks, n x ... -> (2n, n)
    sign = 1.0 if 1 - F >= 0 else -1.0
    return Q * abs(1 - F), (-sign * Q) * gU.T.reshape(-1)
