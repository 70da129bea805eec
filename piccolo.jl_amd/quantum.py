"""Host-side producers of the evaluator's inputs: real isomorphisms, linear-drive
systems and the transmon templates that generate BASELINE.json's configurations.

Only what the hot path consumes is mirrored here:
the generator pieces ``G_drift``, ``G_drives`` and the iso-vec state layout.
Reference (harmoniqs/Piccolo.jl v2.0.2):
  src/quantum/primitives/isomorphisms.jl:74-82,110-118,350,359
  src/quantum/systems/quantum_systems.jl:190-227
  src/quantum/systems/composite_quantum_systems.jl:92-154
  src/quantum/templates/transmons/transmon_system.jl:34-96,139-171,199-263
  src/quantum/operators/lifted_operators.jl:22-31, src/quantum/object_utils.jl:154
Time-dependent / nonlinear drives are out of scope (they go to
TimeDependentBilinearIntegrator in the reference, integrators.jl:38-46).
"""
from functools import reduce

import numpy as np

__all__ = [
    "PAULIS", "GATES", "annihilate", "create", "lift_operator",
    "ket_to_iso", "iso_to_ket", "operator_to_iso_vec", "iso_vec_to_operator",
    "iso_vec_to_iso_operator", "iso_operator_to_iso_vec", "iso", "G", "H",
    "QuantumSystem", "CompositeQuantumSystem", "TransmonSystem",
    "TransmonDipoleCoupling", "MultiTransmonSystem",
]  # fmt: skip

_c = np.complex128
PAULIS = {
    "I": np.array([[1, 0], [0, 1]], _c),
    "X": np.array([[0, 1], [1, 0]], _c),
    "Y": np.array([[0, -1j], [1j, 0]], _c),
    "Z": np.array([[1, 0], [0, -1]], _c),
}
GATES = dict(
    PAULIS,
    H=np.array([[1, 1], [1, -1]], _c) / np.sqrt(2),
    CX=np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], _c),
    CZ=np.diag([1, 1, 1, -1]).astype(_c),
    XI=np.array([[0, 0, -1j, 0], [0, 0, 0, -1j], [-1j, 0, 0, 0], [0, -1j, 0, 0]], _c),
    sqrtiSWAP=np.array(
        [[1, 0, 0, 0], [0, 1 / np.sqrt(2), 1j / np.sqrt(2), 0], [0, 1j / np.sqrt(2), 1 / np.sqrt(2), 0], [0, 0, 0, 1]], _c
    ),
)
GATES["CNOT"] = GATES["CX"]


# ---------------------------------------------------------------------------
# isomorphisms
# ---------------------------------------------------------------------------
def ket_to_iso(psi):
    psi = np.asarray(psi, _c)
    return np.concatenate((psi.real, psi.imag))


def iso_to_ket(v):
    v = np.asarray(v, float)
    return v[: v.size // 2] + 1j * v[v.size // 2 :]


def operator_to_iso_vec(U):
    """vec over columns of [Re U; Im U] (a 2d x d real matrix)."""
    U = np.asarray(U, _c)
    return np.vstack((U.real, U.imag)).T.reshape(-1).copy()


def iso_vec_to_operator(v):
    v = np.asarray(v, float)
    d = int(round(np.sqrt(v.size / 2)))
    X = v.reshape(d, 2 * d).T
    return X[:d] + 1j * X[d:]


def iso_vec_to_iso_operator(v):
    U = iso_vec_to_operator(v)
    return np.block([[U.real, -U.imag], [U.imag, U.real]])


def iso_operator_to_iso_vec(Ut):
    Ut = np.asarray(Ut, float)
    return Ut[:, : Ut.shape[0] // 2].T.reshape(-1).copy()


def iso(Hm):
    Hm = np.asarray(Hm, _c)
    return np.block([[Hm.real, -Hm.imag], [Hm.imag, Hm.real]])


def G(Hm):
    """iso(-i H): the real generator of  dx/dt = G x  for  dpsi/dt = -i H psi."""
    return iso(-1j * np.asarray(Hm, _c))


def H(Gm):
    Gm = np.asarray(Gm, float)
    d = Gm.shape[0] // 2
    return -Gm[d:, :d] + 1j * Gm[:d, :d]


# ---------------------------------------------------------------------------
# operators
# ---------------------------------------------------------------------------
def annihilate(levels):
    return np.diag(np.sqrt(np.arange(1.0, levels)), 1).astype(_c)


def create(levels):
    return annihilate(levels).conj().T


def lift_operator(op, i, subsystem_levels):
    """``op`` on (1-based) subsystem ``i``, identity elsewhere."""
    op = np.asarray(op, _c)
    if op.shape[0] != subsystem_levels[i - 1]:
        raise ValueError("Operator must match subsystem level.")
    factors = [np.eye(l, dtype=_c) for l in subsystem_levels]
    factors[i - 1] = op
    return reduce(np.kron, factors)


def _is_hermitian(A, atol=1e-12):
    return np.allclose(A, A.conj().T, atol=atol)


def _normalize_bounds(bounds):
    return [tuple(map(float, b)) if isinstance(b, (tuple, list)) else (-float(b), float(b)) for b in bounds]


# ---------------------------------------------------------------------------
# systems
# ---------------------------------------------------------------------------
class QuantumSystem:
    """H(u) = H_drift + sum_j u_j H_drives[j];  G(u) = G_drift + sum_j u_j G_drives[j]."""

    time_dependent = False

    def __init__(self, H_drift, H_drives=None, drive_bounds=None, hermitian=True):
        if H_drives is None and isinstance(H_drift, (list, tuple)):  # QuantumSystem(H_drives, bounds)
            H_drift, H_drives = None, H_drift
        H_drives = [np.asarray(Hd, _c) for Hd in (H_drives or [])]
        if H_drift is None:
            H_drift = np.zeros_like(H_drives[0])
        self.H_drift = np.asarray(H_drift, _c)
        self.H_drives = H_drives
        if hermitian and not _is_hermitian(self.H_drift):
            raise AssertionError("Drift Hamiltonian H_drift is not Hermitian")
        for i, Hd in enumerate(H_drives, 1):
            if not _is_hermitian(Hd):
                raise AssertionError("Drive Hamiltonian H_drives[%d] is not Hermitian" % i)
        self.drive_bounds = _normalize_bounds(drive_bounds if drive_bounds is not None else [1.0] * len(H_drives))
        if len(self.drive_bounds) != len(H_drives):
            raise ValueError("drive_bounds must have one entry per drive")
        self.G_drift = G(self.H_drift)
        self.G_drives = [G(Hd) for Hd in H_drives]
        self.levels = self.H_drift.shape[0]
        self.n_drives = len(H_drives)
        self.subsystem_levels = [self.levels]

    def H(self, u, t=0.0):
        return self.H_drift + sum((uj * Hd for uj, Hd in zip(u, self.H_drives)), np.zeros_like(self.H_drift))

    def G(self, u, t=0.0):
        return self.G_drift + sum((uj * Gd for uj, Gd in zip(u, self.G_drives)), np.zeros_like(self.G_drift))

    def G_drives_array(self):
        n = 2 * self.levels
        return np.array(self.G_drives).reshape(self.n_drives, n, n)


class CompositeQuantumSystem(QuantumSystem):
    """Coupling drift + lifted subsystem drifts; coupling drives then lifted subsystem drives."""

    def __init__(self, H_coupling, subsystems, coupling_drives=(), coupling_bounds=()):
        levels = [s.levels for s in subsystems]
        Hd = np.asarray(H_coupling, _c).copy()
        drives = [np.asarray(Hc, _c) for Hc in coupling_drives]
        bounds = list(coupling_bounds)
        for i, s in enumerate(subsystems, 1):
            Hd = Hd + lift_operator(s.H_drift, i, levels)
            drives += [lift_operator(Hs, i, levels) for Hs in s.H_drives]
            bounds += s.drive_bounds
        super().__init__(Hd, drives, bounds)
        self.subsystems = list(subsystems)
        self.subsystem_levels = levels


def TransmonSystem(omega=4.0, delta=0.2, levels=3, lab_frame=False, frame_omega=None, multiply_by_2pi=True,
                   drives=True, drive_bounds=(1.0, 1.0)):  # fmt: skip
    """Duffing transmon; rotating at its own frequency unless ``lab_frame``."""
    if frame_omega is None:
        frame_omega = 0.0 if lab_frame else omega
    a = annihilate(levels)
    ad = a.conj().T
    detuning = omega if lab_frame else omega - frame_omega
    H_drift = detuning * (ad @ a) - 0.5 * delta * (ad @ ad @ a @ a)
    H_drives = [a + ad, 1j * (a - ad)] if drives else []
    scale = 2 * np.pi if multiply_by_2pi else 1.0
    return QuantumSystem(scale * H_drift, [scale * Hd for Hd in H_drives], list(drive_bounds) if drives else [])


def TransmonDipoleCoupling(g_ij, pair, subsystem_levels, lab_frame=False, multiply_by_2pi=True):
    i, j = pair
    a_i = lift_operator(annihilate(subsystem_levels[i - 1]), i, subsystem_levels)
    a_j = lift_operator(annihilate(subsystem_levels[j - 1]), j, subsystem_levels)
    if lab_frame:
        op = (a_i + a_i.conj().T) @ (a_j + a_j.conj().T)
    else:
        op = a_i @ a_j.conj().T + a_i.conj().T @ a_j
    return g_ij * op * (2 * np.pi if multiply_by_2pi else 1.0)


def MultiTransmonSystem(omegas, deltas, gs, levels_per_transmon=3, drive_bounds=1.0, lab_frame=False):
    gs = np.asarray(gs, float)
    if gs.shape != (len(omegas), len(omegas)) or len(deltas) != len(omegas):
        raise AssertionError("gs must be n x n and deltas of length n")
    db = [drive_bounds, drive_bounds] if np.isscalar(drive_bounds) else list(drive_bounds)
    subs = [TransmonSystem(omega=w, delta=dl, levels=levels_per_transmon, lab_frame=lab_frame, drive_bounds=db)
            for w, dl in zip(omegas, deltas)]  # fmt: skip
    lv = [s.levels for s in subs]
    dim = int(np.prod(lv))
    Hc = np.zeros((dim, dim), _c)
    for i in range(1, len(subs)):
        for j in range(i + 1, len(subs) + 1):
            Hc += TransmonDipoleCoupling(gs[i - 1, j - 1], (i, j), lv, lab_frame=lab_frame)
    return CompositeQuantumSystem(Hc, subs)


# ---------------------------------------------------------------------------
# open systems: compact density isomorphism + compact Lindbladian generators
# ---------------------------------------------------------------------------
def ad_vec(Hm, anti=False):
    """I (x) H - (-1)^anti conj(H)' (x) I  [REF isomorphisms.jl:378-381]."""
    Hm = np.asarray(Hm, _c)
    Id = np.eye(Hm.shape[0])
    return np.kron(Id, Hm) - (-1.0) ** int(anti) * np.kron(Hm.T, Id)


def density_to_compact_iso(rho):
    """n^2 reals of a Hermitian rho: Re upper triangle (column-major), then Im strict upper triangle
    [REF isomorphisms.jl:176-192]."""
    rho = np.asarray(rho, _c)
    n = rho.shape[0]
    return np.array([rho[j, k].real for k in range(n) for j in range(k + 1)] + [rho[j, k].imag for k in range(1, n) for j in range(k)])


def compact_iso_to_density(x):
    """[REF isomorphisms.jl:201-222]"""
    x = np.asarray(x, float)
    n = int(round(np.sqrt(x.size)))
    rho = np.zeros((n, n), _c)
    iu = [(j, k) for k in range(n) for j in range(k + 1)]
    for (j, k), v in zip(iu, x[: len(iu)]):
        rho[j, k] = v
        rho[k, j] = v
    for (j, k), v in zip([(j, k) for k in range(1, n) for j in range(k)], x[len(iu) :]):
        rho[j, k] += 1j * v
        rho[k, j] -= 1j * v
    return rho


def density_lift_matrix(n):
    """L (2n^2 x n^2), compact -> iso_vec [REF isomorphisms.jl:236-276]."""
    L = np.zeros((2 * n * n, n * n))
    col = 0
    for k in range(n):
        for j in range(k + 1):
            L[k * n + j, col] = 1.0
            if j != k:
                L[j * n + k, col] = 1.0
            col += 1
    for k in range(1, n):
        for j in range(k):
            L[n * n + k * n + j, col] = 1.0
            L[n * n + j * n + k, col] = -1.0
            col += 1
    return L


def density_projection_matrix(n):
    """P (n^2 x 2n^2), iso_vec -> compact; P L = I [REF isomorphisms.jl:294-324]."""
    P = np.zeros((n * n, 2 * n * n))
    row = 0
    for k in range(n):
        for j in range(k + 1):
            P[row, k * n + j] = 1.0
            row += 1
    for k in range(1, n):
        for j in range(k):
            P[row, n * n + k * n + j] = 1.0
            row += 1
    return P


def iso_D(Lop):
    """Isomorphic Lindblad dissipator [REF isomorphisms.jl:394-396]."""
    Lop = np.asarray(Lop, _c)
    return iso(np.kron(Lop.conj(), Lop) - 0.5 * ad_vec(Lop.conj().T @ Lop, anti=True))


class OpenQuantumSystem:
    """Linear drives, constant-rate dissipators.  ``G_drift`` / ``G_drives`` are the COMPACT Lindbladian generators
    (levels^2 x levels^2, real): d/dt x = (G_drift + sum_j u_j G_drives[j]) x for x = density_to_compact_iso(rho)
    [REF open_quantum_systems.jl:541-588 (compact_lindbladian_generators); integrators.jl:82-95]."""

    time_dependent = False

    def __init__(self, H_drift, H_drives=(), drive_bounds=None, dissipation_operators=()):
        self.H_drift = np.asarray(H_drift, _c)
        self.H_drives = [np.asarray(Hd, _c) for Hd in H_drives]
        self.dissipation_operators = [np.asarray(Lo, _c) for Lo in dissipation_operators]
        self.levels = self.H_drift.shape[0]
        self.n_drives = len(self.H_drives)
        self.drive_bounds = _normalize_bounds(drive_bounds if drive_bounds is not None else [1.0] * self.n_drives)
        n = self.levels
        P, L = density_projection_matrix(n), density_lift_matrix(n)
        self.G_drift = P @ G(ad_vec(self.H_drift)) @ L
        for Lo in self.dissipation_operators:
            self.G_drift = self.G_drift + P @ iso_D(Lo) @ L
        self.G_drives = [P @ G(ad_vec(Hd)) @ L for Hd in self.H_drives]

    def G(self, u, t=0.0):
        return self.G_drift + sum((uj * Gd for uj, Gd in zip(u, self.G_drives)), np.zeros_like(self.G_drift))

    def G_drives_array(self):
        n2 = self.levels**2
        return np.array(self.G_drives).reshape(self.n_drives, n2, n2)
