"""CPU oracle for the collocation-constraint hot path (numpy/scipy).

TEST INFRASTRUCTURE ONLY.  Nothing under ``piccolo.jl_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

What it restates (reference = harmoniqs/Piccolo.jl v2.0.2, paths relative to
``/root/reference``):

* isomorphisms                     src/quantum/primitives/isomorphisms.jl:74-82,110-118,350,359
* annihilate / lift_operator       src/quantum/object_utils.jl:154, src/quantum/operators/lifted_operators.jl:22-31
* QuantumSystem (linear drives)    src/quantum/systems/quantum_systems.jl:212-227
* CompositeQuantumSystem           src/quantum/systems/composite_quantum_systems.jl:104-133
* TransmonSystem / coupling / MultiTransmonSystem
                                   src/quantum/templates/transmons/transmon_system.jl:34-96,139-171,199-263
* knot-major NamedTrajectory layout  src/quantum/trajectories/named_trajectory_conversion.jl:289-352,
                                   src/quantum/trajectories/sampling_trajectory.jl:181-238
* generator closure  Ghat(u) = I_d (x) G(u)   src/control/integrators.jl:35-51,134-162
* the constraint  x_{k+1} = exp(dt_k Ghat(u_k)) x_k   docs/src/concepts/index.md:17-34,62

The arithmetic of the reference's evaluator lives in the un-vendored dependency
DirectTrajOpt.jl (compat 0.9.5/0.10, Project.toml:8,48) and is an exp-action
constraint with ForwardDiff Jacobians.  BASELINE.json's north_star mandates the
diagonal-Pade residual instead, for which the reference holds NO code and NO
golden values:  **parity unpinned** for the Pade delta/Jacobian/Hessian values.
What IS pinned here (tests/test_oracle_pins.py): every known-answer literal the
reference's tests hold for the iso maps / generators / operators, and the
semantics (layout, G = iso(-iH), drive order, 2*pi, (u_k, dt_k) convention) via
exp-residuals on trajectories solved by the reference itself (docs/data/*.jld2).
Pade-p is then tied to exp by its truncation order (p>=6 reproduces the exp
residual floor on those trajectories) and to itself by finite differences.

Pade formulas (diagonal Pade approximant of exp):
    B^{+-}_p(A) = sum_{j=0}^{p/2} (+-1)^j c_j A^j ,  c_0 = 1
    delta_k = B^-_p(h G) X_{k+1} - B^+_p(h G) X_k ,  h = dt_k, G = G(u_k)
            = sum_j c_j h^j G^j Y_j ,  Y_j = (-1)^j X_{k+1} - X_k
"""
from __future__ import annotations

import dataclasses
import itertools
import struct
from typing import List, Optional, Sequence

import numpy as np
import scipy.linalg

# --------------------------------------------------------------------------- #
# Pade coefficients c_1..c_{p/2} (c_0 = 1): B^+ is the numerator of the (q,q)
# diagonal Pade approximant of exp, q = p/2:  c_j = (2q-j)! q! / ((2q)! j! (q-j)!)
# --------------------------------------------------------------------------- #


def pade_coeffs(order: int) -> np.ndarray:
    if order % 2 or order < 2:
        raise ValueError("Pade order must be even and >= 2")
    from math import factorial as f

    q = order // 2
    return np.array(
        [f(2 * q - j) * f(q) / (f(2 * q) * f(j) * f(q - j)) for j in range(q + 1)]
    )


# --------------------------------------------------------------------------- #
# Isomorphisms  [REF isomorphisms.jl]
# --------------------------------------------------------------------------- #

_IM2 = np.array([[0.0, -1.0], [1.0, 0.0]])


def ket_to_iso(psi):
    """[REF isomorphisms.jl:55]"""
    psi = np.asarray(psi, dtype=complex)
    return np.concatenate([psi.real, psi.imag])


def iso_to_ket(v):
    """[REF isomorphisms.jl:62-63]"""
    v = np.asarray(v, dtype=float)
    h = v.size // 2
    return v[:h] + 1j * v[h:]


def operator_to_iso_vec(U):
    """Column c of U -> [Re U[:,c]; Im U[:,c]], columns concatenated.
    [REF isomorphisms.jl:110-118]"""
    U = np.asarray(U, dtype=complex)
    d = U.shape[0]
    out = np.empty(2 * d * d)
    for c in range(d):
        out[c * 2 * d : c * 2 * d + d] = U[:, c].real
        out[c * 2 * d + d : (c + 1) * 2 * d] = U[:, c].imag
    return out


def iso_vec_to_operator(v):
    """[REF isomorphisms.jl:74-82]"""
    v = np.asarray(v, dtype=float)
    d = int(round(np.sqrt(v.size // 2)))
    U = np.empty((d, d), dtype=complex)
    for c in range(d):
        U[:, c] = v[c * 2 * d : c * 2 * d + d] + 1j * v[c * 2 * d + d : (c + 1) * 2 * d]
    return U


def iso_vec_to_iso_operator(v):
    """[REF isomorphisms.jl:89-103]"""
    U = iso_vec_to_operator(v)
    return np.block([[U.real, -U.imag], [U.imag, U.real]])


def iso_operator_to_iso_vec(Ut):
    """[REF isomorphisms.jl:125-132]"""
    Ut = np.asarray(Ut, dtype=float)
    d = Ut.shape[0] // 2
    return np.concatenate([Ut[:, c] for c in range(d)])


def iso(H):
    """iso(H) = I2 (x) Re H + [[0,-1],[1,0]] (x) Im H.  [REF isomorphisms.jl:350]"""
    H = np.asarray(H, dtype=complex)
    return np.kron(np.eye(2), H.real) + np.kron(_IM2, H.imag)


def G_of_H(H):
    """G(H) = iso(-i H).  [REF isomorphisms.jl:359]"""
    return iso(-1j * np.asarray(H, dtype=complex))


def H_of_G(G):
    """[REF isomorphisms.jl:368-373]"""
    G = np.asarray(G, dtype=float)
    d = G.shape[0] // 2
    return -G[d:, :d] + 1j * G[:d, :d]


def var_G(G, G_vars):
    """[REF isomorphisms.jl:410-422]"""
    n, m = G.shape
    v = len(G_vars)
    out = np.kron(np.eye(v + 1), G)
    for i, Gv in enumerate(G_vars, start=1):
        out[i * n : (i + 1) * n, :m] += Gv
    return out


def ad_vec(H, anti=False):
    """kron(I, H) - (-1)^anti kron(conj(H)', I);  conj(H)' is the plain transpose.
    [REF isomorphisms.jl:384-387]"""
    H = np.asarray(H, dtype=complex)
    Id = np.eye(H.shape[0])
    return np.kron(Id, H) - (-1) ** int(anti) * np.kron(H.T, Id)


# --------------------------------------------------------------------------- #
# Open systems: compact density isomorphism and compact Lindbladian generators
# [REF isomorphisms.jl:163-330,390-396; open_quantum_systems.jl:541-588]
# --------------------------------------------------------------------------- #


def density_to_iso_vec(rho):
    """ket_to_iso(vec(rho)) [REF isomorphisms.jl:150-160]."""
    return ket_to_iso(np.asarray(rho, complex).T.reshape(-1))


def density_to_compact_iso(rho):
    """n^2 reals of a Hermitian rho: Re of the upper triangle (column-major), then Im of the strict upper triangle
    (column-major) [REF isomorphisms.jl:176-192]."""
    rho = np.asarray(rho, complex)
    n = rho.shape[0]
    x = [rho[j, k].real for k in range(n) for j in range(k + 1)]
    x += [rho[j, k].imag for k in range(1, n) for j in range(k)]
    return np.array(x)


def compact_iso_to_density(x):
    """[REF isomorphisms.jl:201-222]"""
    x = np.asarray(x, float)
    n = int(round(np.sqrt(x.size)))
    rho = np.zeros((n, n), complex)
    idx = 0
    for k in range(n):
        for j in range(k + 1):
            rho[j, k] = x[idx]
            if j != k:
                rho[k, j] = x[idx]
            idx += 1
    for k in range(1, n):
        for j in range(k):
            rho[j, k] += 1j * x[idx]
            rho[k, j] -= 1j * x[idx]
            idx += 1
    return rho


def density_lift_matrix(n):
    """L (2n^2 x n^2): compact -> iso_vec [REF isomorphisms.jl:236-276]."""
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
    """P (n^2 x 2n^2): iso_vec -> compact, P L = I [REF isomorphisms.jl:294-324]."""
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
    """iso(conj(L) (x) L - 1/2 ad_vec(L'L, anti)) [REF isomorphisms.jl:394-396]."""
    Lop = np.asarray(Lop, complex)
    return iso(np.kron(Lop.conj(), Lop) - 0.5 * ad_vec(Lop.conj().T @ Lop, anti=True))


def compact_lindbladian_generators(H_drift, H_drives, dissipators=()):
    """(Gc_drift, [Gc_drive_j]) with Gc = P G(ad_vec(H)) L (+ sum of P iso_D(L_j) L for constant-rate dissipators
    folded into the drift) [REF open_quantum_systems.jl:541-588]: d/dt x = (Gc_drift + sum u_j Gc_j) x for the compact
    density vector x."""
    H_drift = np.asarray(H_drift, complex)
    n = H_drift.shape[0]
    P, L = density_projection_matrix(n), density_lift_matrix(n)
    drift = P @ G_of_H(ad_vec(H_drift)) @ L
    for Lop in dissipators:
        drift = drift + P @ iso_D(Lop) @ L
    drives = [P @ G_of_H(ad_vec(np.asarray(H, complex))) @ L for H in H_drives]
    return drift, drives



# --------------------------------------------------------------------------- #
# Operators and systems
# --------------------------------------------------------------------------- #

PAULIS = {
    "I": np.eye(2, dtype=complex),
    "X": np.array([[0, 1], [1, 0]], dtype=complex),
    "Y": np.array([[0, -1j], [1j, 0]], dtype=complex),
    "Z": np.array([[1, 0], [0, -1]], dtype=complex),
}


def annihilate(levels: int):
    """diagm(1 => sqrt.(1:levels-1)).  [REF object_utils.jl:154]"""
    return np.diag(np.sqrt(np.arange(1, levels)), k=1).astype(complex)


def lift_operator(op, i: int, subsystem_levels: Sequence[int]):
    """Kronecker lift of ``op`` onto (1-based) subsystem ``i``.
    [REF lifted_operators.jl:22-31]"""
    assert op.shape[0] == subsystem_levels[i - 1]
    out = np.eye(1, dtype=complex)
    for j, l in enumerate(subsystem_levels, start=1):
        out = np.kron(out, op if j == i else np.eye(l, dtype=complex))
    return out


@dataclasses.dataclass
class System:
    """Linear-drive system:  H(u) = H_drift + sum_j u_j H_drives[j];
    G(u) = G_drift + sum_j u_j G_drives[j] with G_* = iso(-i H_*).
    [REF quantum_systems.jl:212-227, composite_quantum_systems.jl:124-133]"""

    H_drift: np.ndarray
    H_drives: List[np.ndarray]
    drive_bounds: List[tuple]
    subsystem_levels: Optional[List[int]] = None

    @property
    def levels(self):
        return self.H_drift.shape[0]

    @property
    def n_drives(self):
        return len(self.H_drives)

    @property
    def G_drift(self):
        return G_of_H(self.H_drift)

    @property
    def G_drives(self):
        return [G_of_H(H) for H in self.H_drives]

    def G(self, u):
        out = self.G_drift.copy()
        for uj, Gj in zip(u, self.G_drives):
            out += uj * Gj
        return out


def _norm_bounds(b):
    return [x if isinstance(x, tuple) else (-float(x), float(x)) for x in b]


def quantum_system(H_drift, H_drives, drive_bounds):
    """[REF quantum_systems.jl:190-227]"""
    return System(
        np.asarray(H_drift, dtype=complex),
        [np.asarray(H, dtype=complex) for H in H_drives],
        _norm_bounds(drive_bounds),
    )


def transmon_system(
    omega=4.0,
    delta=0.2,
    levels=3,
    lab_frame=False,
    frame_omega=None,
    multiply_by_2pi=True,
    drives=True,
    drive_bounds=(1.0, 1.0),
):
    """Duffing transmon in its own rotating frame by default (frame_omega = omega).
    [REF transmon_system.jl:34-96]"""
    if frame_omega is None:
        frame_omega = 0.0 if lab_frame else omega
    a = annihilate(levels)
    ad = a.conj().T
    if lab_frame:
        H_drift = omega * ad @ a - delta / 2 * ad @ ad @ a @ a
    else:
        H_drift = (omega - frame_omega) * ad @ a - delta / 2 * ad @ ad @ a @ a
    H_drives = [a + ad, 1j * (a - ad)] if drives else []
    if multiply_by_2pi:
        H_drift = H_drift * 2 * np.pi
        H_drives = [H * 2 * np.pi for H in H_drives]
    return quantum_system(H_drift, H_drives, list(drive_bounds) if drives else [])


def transmon_dipole_coupling(g_ij, pair, subsystem_levels, lab_frame=False, multiply_by_2pi=True):
    """[REF transmon_system.jl:139-171]"""
    i, j = pair
    a_i = lift_operator(annihilate(subsystem_levels[i - 1]), i, subsystem_levels)
    a_j = lift_operator(annihilate(subsystem_levels[j - 1]), j, subsystem_levels)
    if lab_frame:
        op = g_ij * (a_i + a_i.conj().T) @ (a_j + a_j.conj().T)
    else:
        op = g_ij * (a_i @ a_j.conj().T + a_i.conj().T @ a_j)
    if multiply_by_2pi:
        op = op * 2 * np.pi
    return op


def composite_system(H_coupling, subsystems: Sequence[System], coupling_drives=(), coupling_bounds=()):
    """Drift = coupling + lifted subsystem drifts; drives = coupling drives then
    lifted subsystem drives in subsystem order.
    [REF composite_quantum_systems.jl:104-133]"""
    levels = [s.levels for s in subsystems]
    H_drift = np.asarray(H_coupling, dtype=complex).copy()
    for i, s in enumerate(subsystems, start=1):
        H_drift = H_drift + lift_operator(s.H_drift, i, levels)
    H_drives = [np.asarray(H, dtype=complex) for H in coupling_drives]
    bounds = _norm_bounds(list(coupling_bounds))
    for i, s in enumerate(subsystems, start=1):
        for H in s.H_drives:
            H_drives.append(lift_operator(H, i, levels))
        bounds.extend(s.drive_bounds)
    return System(H_drift, H_drives, bounds, subsystem_levels=levels)


def multi_transmon_system(omegas, deltas, gs, levels_per_transmon=3, drive_bounds=1.0, lab_frame=False):
    """[REF transmon_system.jl:199-263]"""
    gs = np.asarray(gs, dtype=float)
    ns = len(omegas)
    db = [drive_bounds, drive_bounds] if np.isscalar(drive_bounds) else list(drive_bounds)
    subs = [
        transmon_system(omega=w, delta=dl, levels=levels_per_transmon, lab_frame=lab_frame, drive_bounds=db)
        for w, dl in zip(omegas, deltas)
    ]
    levels = [s.levels for s in subs]
    dim = int(np.prod(levels))
    H = np.zeros((dim, dim), dtype=complex)
    for i in range(1, ns):
        for j in range(i + 1, ns + 1):
            H = H + transmon_dipole_coupling(gs[i - 1, j - 1], (i, j), levels, lab_frame=lab_frame)
    return composite_system(H, subs)


# --------------------------------------------------------------------------- #
# Knot-major trajectory layout  (NamedTrajectory.datavec, z_dim x N column-major)
# --------------------------------------------------------------------------- #


@dataclasses.dataclass
class Layout:
    """0-based component offsets inside one knot column of ``datavec``.
    SmoothPulseProblem order is [Utilde(x_dim), dt, t, u(m), du(m), ddu(m)]
    [REF smooth_pulse_problem.jl:187-201; SURVEY section 0.3 (measured)]."""

    d: int
    m: int
    N: int
    z_dim: int
    x_off: int
    u_off: int
    dt_off: int
    cols: Optional[int] = None  # state columns: None/d = unitary (X is n x d), 1 = ket [REF integrators.jl:58-74]
    gen: Optional[int] = None  # generator dimension when it is not 2d: compact density vectors, gen = levels^2, one
    #                            column [REF integrators.jl:82-95] (then `d` is not used)

    @property
    def n(self):
        return 2 * self.d if self.gen is None else self.gen

    @property
    def C(self):
        return self.d if self.cols is None else self.cols

    @property
    def x_dim(self):
        return self.n * self.C

    @property
    def K(self):
        return self.N - 1

    @staticmethod
    def smooth_pulse(d, m, N, n_members=1):
        x_dim = 2 * d * d
        xs = n_members * x_dim
        return Layout(d=d, m=m, N=N, z_dim=xs + 2 + 3 * m, x_off=0, u_off=xs + 2, dt_off=xs)

    def X(self, Z, k, x_off=None):
        """n x d real matrix [Re U; Im U] of knot k (column c = iso-vec slice c)."""
        o = self.x_off if x_off is None else x_off
        return Z[k, o : o + self.x_dim].reshape(self.C, self.n).T

    def u(self, Z, k):
        return Z[k, self.u_off : self.u_off + self.m]

    def dt(self, Z, k):
        return Z[k, self.dt_off]


def as_knots(datavec, z_dim, N):
    """datavec (length z_dim*N, knot-major) -> array [N, z_dim] (row k = knot k)."""
    return np.asarray(datavec, dtype=float).reshape(N, z_dim)


# --------------------------------------------------------------------------- #
# Residuals
# --------------------------------------------------------------------------- #


def exp_residual(Z, lay: Layout, G0, Gj, x_off=None):
    """Reference constraint  delta_k = x_{k+1} - exp(dt_k (I_d (x) G(u_k))) x_k.
    [REF docs/src/concepts/index.md:21; integrators.jl:48]  Returns [K, x_dim]."""
    out = np.empty((lay.K, lay.x_dim))
    for k in range(lay.K):
        G = G0 + np.tensordot(lay.u(Z, k), Gj, axes=1) if lay.m else G0
        E = scipy.linalg.expm(lay.dt(Z, k) * G)
        R = lay.X(Z, k + 1, x_off) - E @ lay.X(Z, k, x_off)
        out[k] = R.T.reshape(-1)
    return out


def exact_rollout(Z, lay: Layout, G0, Gj, x_off=None):
    """X_{k+1} = expm(dt_k G(u_k)) X_k from the knot-0 state: what the reference's
    unitary_rollout(...; interpolation = :constant) integrates [REF src/quantum/dynamics.jl:631-667].
    Returns [N, x_dim] iso-vec rows."""
    from scipy.linalg import expm

    X = lay.X(Z, 0, x_off)
    out = np.empty((lay.N, lay.x_dim))
    out[0] = X.T.reshape(-1)
    for k in range(lay.K):
        G = G0 + np.tensordot(lay.u(Z, k), Gj, axes=1) if lay.m else G0
        X = expm(lay.dt(Z, k) * G) @ X
        out[k + 1] = X.T.reshape(-1)
    return out


def _powers(G, q):
    P = [np.eye(G.shape[0])]
    for _ in range(q):
        P.append(P[-1] @ G)
    return P


def pade_residual(Z, lay: Layout, G0, Gj, order=4, x_off=None):
    """delta_k = B^-_p(h G) X_{k+1} - B^+_p(h G) X_k.  Returns [K, x_dim]
    (row k = interval k, entries in iso-vec order: column c of the n x d block
    at [c*n, (c+1)*n))."""
    c = pade_coeffs(order)
    q = order // 2
    out = np.empty((lay.K, lay.x_dim))
    for k in range(lay.K):
        G = G0 + np.tensordot(lay.u(Z, k), Gj, axes=1) if lay.m else G0
        h = lay.dt(Z, k)
        Xn, Xc = lay.X(Z, k + 1, x_off), lay.X(Z, k, x_off)
        P = _powers(G, q)
        R = np.zeros_like(Xc)
        for j in range(q + 1):
            Y = ((-1) ** j) * Xn - Xc
            R += c[j] * h**j * (P[j] @ Y)
        out[k] = R.T.reshape(-1)
    return out


# --------------------------------------------------------------------------- #
# Jacobian (analytic).  Triplet order per interval k, fixed by this project
# (the C ABI reports it through pcl_jac_structure):
#   seg 0  d delta / d X_k      : for c in 0..d-1, for j in 0..n-1, for i in 0..n-1 : -B^+[i,j]
#   seg 1  d delta / d X_{k+1}  : same loop                                          :  B^-[i,j]
#   tail   for c in 0..C-1 (state column):  for l in 0..m-1, for i in 0..n-1 : d delta[c*n+i] / d u_l ;
#                                           then for i in 0..n-1            : d delta[c*n+i] / d dt
#          (column-major: everything one state column produces is contiguous, (m+1)*n doubles)
# row(delta_k[r]) = k*x_dim + r ;  col(comp i of knot k) = k*z_dim + i   (0-based)
# --------------------------------------------------------------------------- #


def jac_nnz_per_interval(lay: Layout):
    return 2 * lay.C * lay.n * lay.n + lay.x_dim * (lay.m + 1)


def jac_structure(lay: Layout, x_off=None, index_base=0):
    o = lay.x_off if x_off is None else x_off
    d, n, m, xd, zd = lay.C, lay.n, lay.m, lay.x_dim, lay.z_dim
    per = jac_nnz_per_interval(lay)
    rows = np.empty(lay.K * per, dtype=np.int64)
    cols = np.empty_like(rows)
    c_, j_, i_ = np.meshgrid(np.arange(d), np.arange(n), np.arange(n), indexing="ij")
    blk_r = (c_ * n + i_).reshape(-1)
    blk_c = (c_ * n + j_).reshape(-1)
    r_all = np.arange(xd)
    for k in range(lay.K):
        p = k * per
        nb = d * n * n
        rows[p : p + nb] = k * xd + blk_r
        cols[p : p + nb] = k * zd + o + blk_c
        p += nb
        rows[p : p + nb] = k * xd + blk_r
        cols[p : p + nb] = (k + 1) * zd + o + blk_c
        p += nb
        for cc in range(d):
            for l in range(m + 1):
                rows[p : p + n] = k * xd + cc * n + np.arange(n)
                cols[p : p + n] = k * zd + (lay.u_off + l if l < m else lay.dt_off)
                p += n
    return rows + index_base, cols + index_base


def pade_jacobian_values(Z, lay: Layout, G0, Gj, order=4, x_off=None):
    """Jacobian values in the fixed triplet order above.  Returns [K, nnz_per_interval]."""
    c = pade_coeffs(order)
    q = order // 2
    d, n, m, xd = lay.C, lay.n, lay.m, lay.x_dim
    per = jac_nnz_per_interval(lay)
    dt_ = np.result_type(np.asarray(Z).dtype, np.float64)  # complex Z: the complex-step derivative of this function pins the Hessian
    out = np.empty((lay.K, per), dtype=dt_)
    for k in range(lay.K):
        G = G0 + np.tensordot(lay.u(Z, k), Gj, axes=1) if m else G0
        h = lay.dt(Z, k)
        Xn, Xc = lay.X(Z, k + 1, x_off), lay.X(Z, k, x_off)
        P = _powers(G, q)
        Bp = sum(c[j] * h**j * P[j] for j in range(q + 1))
        Bm = sum(c[j] * (-h) ** j * P[j] for j in range(q + 1))
        p = 0
        nb = d * n * n
        out[k, p : p + nb] = np.tile((-Bp).T.reshape(-1), d)  # column-major flat of -B^+
        p += nb
        out[k, p : p + nb] = np.tile(Bm.T.reshape(-1), d)
        p += nb
        Y = [((-1) ** j) * Xn - Xc for j in range(q + 1)]
        tail = np.empty((d, m + 1, n), dtype=dt_)  # [state column][drive l | dt][row]
        for l in range(m):
            R = np.zeros_like(Xc)
            for j in range(1, q + 1):
                dGj = sum(P[a] @ Gj[l] @ P[j - 1 - a] for a in range(j))
                R += c[j] * h**j * (dGj @ Y[j])
            tail[:, l, :] = R.T
        R = np.zeros_like(Xc)
        for j in range(1, q + 1):
            R += j * c[j] * h ** (j - 1) * (P[j] @ Y[j])
        tail[:, m, :] = R.T
        out[k, p:] = tail.reshape(-1)
    return out


def pade_jacobian_dense(Z, lay: Layout, G0, Gj, order=4, x_off=None):
    """Dense (x_dim*K) x (z_dim*N) Jacobian assembled from the triplets (small cases)."""
    rows, cols = jac_structure(lay, x_off)
    vals = pade_jacobian_values(Z, lay, G0, Gj, order, x_off).reshape(-1)
    J = np.zeros((lay.x_dim * lay.K, lay.z_dim * lay.N), dtype=vals.dtype)
    np.add.at(J, (rows, cols), vals)
    return J


def exp_jacobian_values(Z, lay: Layout, G0, Gj, x_off=None):
    """Jacobian of the REFERENCE's constraint  delta_k = x_{k+1} - exp(dt_k Ghat(u_k)) x_k
    [REF docs/src/concepts/index.md:21; integrators.jl:48] in the SAME triplet order as ``pade_jacobian_values``:
    d/dX_k = -(I_d (x) E), d/dX_{k+1} = I, d/du_l = -L(hG; h G_l) X_k (L = Frechet derivative of expm,
    scipy.linalg.expm_frechet, Al-Mohy & Higham), d/dh = -G E X_k.  The reference itself obtains these numbers by
    ForwardDiff through expv [REF integrators.jl:282-285]; this is what the Pade Jacobian is pinned against."""
    d, n, m = lay.C, lay.n, lay.m
    per = jac_nnz_per_interval(lay)
    out = np.empty((lay.K, per))
    for k in range(lay.K):
        G = G0 + np.tensordot(lay.u(Z, k), Gj, axes=1) if m else G0
        h = lay.dt(Z, k)
        Xc = lay.X(Z, k, x_off)
        E = scipy.linalg.expm(h * G)
        nb = d * n * n
        out[k, :nb] = np.tile((-E).T.reshape(-1), d)
        out[k, nb : 2 * nb] = np.tile(np.eye(n).T.reshape(-1), d)
        tail = np.empty((d, m + 1, n))
        for l in range(m):
            L = scipy.linalg.expm_frechet(h * G, h * Gj[l], compute_expm=False)
            tail[:, l, :] = (-(L @ Xc)).T
        tail[:, m, :] = (-(G @ E @ Xc)).T
        out[k, 2 * nb :] = tail.reshape(-1)
    return out


def pade_jacobian_in_exp_form(Z, lay: Layout, G0, Gj, order, x_off=None):
    """The Pade-p Jacobian premultiplied by (B^-_p)^{-1} per interval:  delta^P = B^- (X_{k+1} - R_p X_k) with
    R_p = (B^-)^{-1} B^+ the (p/2, p/2) Pade approximant of exp, so (B^-)^{-1} J^P is the Jacobian of
    X_{k+1} - R_p X_k up to terms proportional to the residual itself.  On a trajectory that satisfies the constraint
    this equals ``exp_jacobian_values`` to the truncation order of R_p.  Same triplet order."""
    c = pade_coeffs(order)
    q = order // 2
    d, n, m = lay.C, lay.n, lay.m
    V = pade_jacobian_values(Z, lay, G0, Gj, order, x_off)
    out = np.empty_like(V)
    nb = d * n * n
    for k in range(lay.K):
        G = G0 + np.tensordot(lay.u(Z, k), Gj, axes=1) if m else G0
        h = lay.dt(Z, k)
        P = _powers(G, q)
        Bm = sum(c[j] * (-h) ** j * P[j] for j in range(q + 1))
        Bi = np.linalg.inv(Bm)
        for seg in range(2):
            blocks = V[k, seg * nb : (seg + 1) * nb].reshape(d, n, n)  # [copy][col j][row i]
            out[k, seg * nb : (seg + 1) * nb] = np.stack([(Bi @ b.T).T for b in blocks]).reshape(-1)
        tail = V[k, 2 * nb :].reshape(d, m + 1, n)
        out[k, 2 * nb :] = np.einsum("ij,clj->cli", Bi, tail).reshape(-1)
    return out


# --------------------------------------------------------------------------- #
# Objectives of the unitary problems.
#   unitary_fidelity_loss(Utilde, U_goal)                [REF src/control/objectives.jl:330-337]
#   unitary_fidelity_loss(Utilde, op::EmbeddedOperator)   [REF src/control/objectives.jl:339-345]
#   UnitaryInfidelityObjective = TerminalObjective(|1 - F|; Q)   [REF :347-356]  (Q * l(x_N), DirectTrajOpt [EXT])
#   embed / unembed / get_subspace_indices                [REF src/quantum/operators/embedded_operators.jl:24-40,116-131,345-369]
#   weighted ensemble sum  sum_i w_i Q l_i + regularisers  [REF src/control/templates/sampling_problem.jl:381-387]
#   QuadraticRegularizer(name, traj, R)                   [EXT DirectTrajOpt; used at smooth_pulse_problem.jl:249-251]
# --------------------------------------------------------------------------- #


def get_subspace_indices(subspaces, subsystem_levels):
    """0-based indices of the product basis states whose every subsystem level lies in that subsystem's subspace
    (``subspaces`` 0-based level lists).  [REF embedded_operators.jl:352-364; literal [1:2,1:2],[3,3] -> [1,2,4,5] (1-based) :629]"""
    idx = []
    for flat, lv in enumerate(itertools.product(*[range(L) for L in subsystem_levels])):
        if all(l in sub for l, sub in zip(lv, subspaces)):
            idx.append(flat)
    return idx


def embed(op, subspace, levels):
    """Place the subspace operator into a levels x levels zero matrix.  [REF embedded_operators.jl:24-32]"""
    out = np.zeros((levels, levels), dtype=complex)
    out[np.ix_(subspace, subspace)] = np.asarray(op)
    return out


def unembed(op_embedded, subspace):
    """[REF embedded_operators.jl:34-40]"""
    return np.asarray(op_embedded)[np.ix_(subspace, subspace)]


def unitary_fidelity_loss(Uvec, U_goal, subspace=None):
    """F: |tr(U_goal' U)|^2 / n^2 for a matrix goal; for an embedded goal (``subspace`` given, ``U_goal`` the embedded
    levels x levels operator) the reference's subspace formula (tr(M'M) + |tr M|^2) / (n (n+1)), M = U_goal_sub' U_sub."""
    U = iso_vec_to_operator(np.asarray(Uvec, dtype=float))
    Ug = np.asarray(U_goal, dtype=complex)
    if subspace is None:
        n = U.shape[0]
        return abs(np.trace(Ug.conj().T @ U)) ** 2 / n**2
    Ugs = unembed(Ug, subspace)
    Us = U[np.ix_(subspace, subspace)]
    n = len(subspace)
    M = Ugs.conj().T @ Us
    return (abs(np.trace(M.conj().T @ M)) + abs(np.trace(M)) ** 2) / (n * (n + 1))


def unitary_infidelity(Uvec, U_goal, Q=100.0, subspace=None):
    """Q * |1 - F|  (UnitaryInfidelityObjective's terminal term)."""
    return Q * abs(1.0 - unitary_fidelity_loss(Uvec, U_goal, subspace))


def unitary_infidelity_gradient(Uvec, U_goal, Q=100.0, subspace=None):
    """Gradient of ``unitary_infidelity`` w.r.t. the iso-vec (analytic; the reference differentiates with ForwardDiff)."""
    U = iso_vec_to_operator(np.asarray(Uvec, dtype=float))
    Ug = np.asarray(U_goal, dtype=complex)
    lv = U.shape[0]
    dF = np.zeros((lv, lv), dtype=complex)  # dF/dRe(U) + i dF/dIm(U)
    if subspace is None:
        t = np.trace(Ug.conj().T @ U)
        dF = 2.0 * t * Ug / lv**2  # t = sum conj(g) u:  dF/dRe(u) = 2 Re(t g), dF/dIm(u) = 2 Im(t g)
        F = abs(t) ** 2 / lv**2
    else:
        Ugs = unembed(Ug, subspace)
        Us = U[np.ix_(subspace, subspace)]
        n = len(subspace)
        M = Ugs.conj().T @ Us
        t = np.trace(M)
        W = Ugs @ M  # d tr(M'M) / d conj(U_s)
        dsub = (2.0 * W + 2.0 * t * Ugs) / (n * (n + 1))
        dF[np.ix_(subspace, subspace)] = dsub
        F = (abs(np.trace(M.conj().T @ M)) + abs(t) ** 2) / (n * (n + 1))
    sgn = 1.0 if 1.0 - F >= 0.0 else -1.0
    g = -sgn * Q * dF
    return operator_to_iso_vec_parts(g.real, g.imag)


# ---- terminal losses of the other state types (complex arithmetic, as the reference writes them) --------------------------------
def ket_fidelity_loss(psi_iso, psi_goal):
    """|<goal|psi>|^2  [REF src/control/objectives.jl:24-27]."""
    return abs(np.vdot(np.asarray(psi_goal, complex), iso_to_ket(psi_iso))) ** 2


def coherent_ket_fidelity(psi_isos, psi_goals, weights=None):
    """|sum_i w_i <g_i|psi_i> / sum w|^2; uniform (or no) weights take the unweighted path |sum / n|^2  [REF objectives.jl:96-121]."""
    n = len(psi_isos)
    ov = [np.vdot(np.asarray(g, complex), iso_to_ket(x)) for g, x in zip(psi_goals, psi_isos)]
    if weights is None or len(set(float(w) for w in weights)) == 1:
        return abs(sum(ov) / n) ** 2
    return abs(sum(w * o for w, o in zip(weights, ov)) / sum(weights)) ** 2


def density_matrix_infidelity_loss(rho_compact, rho_goal):
    """|1 - Re tr(rho rho_goal)|, rho from the compact iso vector  [REF objectives.jl:387-395]."""
    return abs(1.0 - np.trace(compact_iso_to_density(rho_compact) @ np.asarray(rho_goal, complex)).real)


def density_matrix_pure_state_infidelity_loss(rho_compact, psi):
    """|1 - Re <psi|rho|psi>|  [REF objectives.jl:416-424]."""
    psi = np.asarray(psi, complex)
    return abs(1.0 - np.vdot(psi, compact_iso_to_density(rho_compact) @ psi).real)


def numerical_gradient(f, x, h=1e-6):
    """Central differences (test infrastructure for the losses above: the reference differentiates them with ForwardDiff)."""
    x = np.asarray(x, float)
    g = np.zeros_like(x)
    for i in range(x.size):
        e = np.zeros_like(x)
        e[i] = h
        g[i] = (f(x + e) - f(x - e)) / (2 * h)
    return g


def quadratic_hessian(F, L):
    """Hessian of a function that is at most quadratic, from its values at 0, e_i, e_i + e_j: exact up to rounding."""
    F0 = F(np.zeros(L))
    Fi = np.array([F(np.eye(L)[i]) for i in range(L)])
    H = np.zeros((L, L))
    for i in range(L):
        for j in range(i + 1):
            e = np.zeros(L)
            e[i] += 1.0
            e[j] += 1.0
            H[i, j] = H[j, i] = F(e) - Fi[i] - Fi[j] + F0
    return H


def operator_to_iso_vec_parts(re, im):
    """iso-vec layout (column c = [Re U[:,c]; Im U[:,c]]) from separate real / imaginary parts."""
    lv = re.shape[0]
    out = np.empty(2 * lv * lv)
    for c in range(lv):
        out[c * 2 * lv : c * 2 * lv + lv] = re[:, c]
        out[c * 2 * lv + lv : (c + 1) * 2 * lv] = im[:, c]
    return out


def quadratic_regularizer(Z, off, dim, R, dt_off, dt_power=2):
    """DirectTrajOpt's QuadraticRegularizer(name, traj, R) [EXT; un-vendored -- the QuantumCollocation-lineage form]:
        J = 1/2 sum_k (dt_k^p) * sum_i R_i v_{k,i}^2 ,   v = Z[k, off:off+dim]
    with p = dt_power = 2 (r_k = dt_k v_k, J += r_k' R r_k / 2); p = 0 drops the time-step weighting (knot-point form).
    Nothing in the reference pins the value (its tests only check the term exists and is zero for R = 0,
    [REF spline_pulse_problem.jl:1570-1578]); ``dt_power`` is a parameter of the C ABI for that reason."""
    Z = np.asarray(Z)
    R = np.broadcast_to(np.asarray(R, dtype=float), (dim,))
    v = Z[:, off : off + dim]
    w = Z[:, dt_off] ** dt_power if dt_power else np.ones(Z.shape[0])
    return 0.5 * float(np.sum(w[:, None] * R[None, :] * v * v))


def quadratic_regularizer_gradient(Z, off, dim, R, dt_off, dt_power=2):
    """Gradient of ``quadratic_regularizer`` w.r.t. the whole knot array (same shape as Z)."""
    Z = np.asarray(Z)
    R = np.broadcast_to(np.asarray(R, dtype=float), (dim,))
    g = np.zeros_like(Z)
    v = Z[:, off : off + dim]
    h = Z[:, dt_off]
    w = h**dt_power if dt_power else np.ones(Z.shape[0])
    g[:, off : off + dim] += w[:, None] * R[None, :] * v
    if dt_power:
        g[:, dt_off] += 0.5 * dt_power * h ** (dt_power - 1) * np.sum(R[None, :] * v * v, axis=1)
    return g


def sampling_objective(Z, lay: Layout, x_offs, goal, weights, Q, regs=(), subspace=None):
    """SamplingProblem objective  sum_i (w_i Q) |1 - F_i(x_N^{(i)})| + sum regs  [REF sampling_problem.jl:381-387]
    (regs: tuples (off, dim, R, dt_power)); returns (value, gradient [N, z_dim])."""
    Z = np.asarray(Z)
    J = 0.0
    g = np.zeros_like(Z)
    for xo, w in zip(x_offs, weights):
        xN = Z[-1, xo : xo + lay.x_dim]
        J += unitary_infidelity(xN, goal, w * Q, subspace)
        g[-1, xo : xo + lay.x_dim] += unitary_infidelity_gradient(xN, goal, w * Q, subspace)
    for off, dim, R, pw in regs:
        J += quadratic_regularizer(Z, off, dim, R, lay.dt_off, pw)
        g += quadratic_regularizer_gradient(Z, off, dim, R, lay.dt_off, pw)
    return J, g


# --------------------------------------------------------------------------- #
# Hessian of the Lagrangian  sum_k mu_k^T delta_k  (Pade-4 analytic).
# Per-interval triplet order (lower triangle of the symmetric matrix, row >= col
# in GLOBAL variable index):
#   seg 0  (u_i,u_j), i>=j       : for i in 0..m-1, for j in 0..i
#   seg 1  (dt,u_j) or (u_j,dt)  : for j in 0..m-1
#   seg 2  (dt,dt)
#   seg 3  (u_l, X_k[r])         : for l, for r        (m*x_dim)
#   seg 4  (dt , X_k[r])         : for r
#   seg 5  (X_{k+1}[r], u_l)     : for l, for r
#   seg 6  (X_{k+1}[r], dt)      : for r
# Each pair is emitted once with (row, col) = (max index, min index).
# --------------------------------------------------------------------------- #


def hess_nnz_per_interval(lay: Layout):
    m = lay.m
    return (m + 1) * (m + 2) // 2 + 2 * lay.x_dim * (m + 1)


def hess_structure(lay: Layout, x_off=None, index_base=0):
    o = lay.x_off if x_off is None else x_off
    m, xd, zd = lay.m, lay.x_dim, lay.z_dim
    per = hess_nnz_per_interval(lay)
    rows = np.empty(lay.K * per, dtype=np.int64)
    cols = np.empty_like(rows)
    r_all = np.arange(xd)
    for k in range(lay.K):
        a, b = [], []
        uk = k * zd + lay.u_off
        hk = k * zd + lay.dt_off
        for i in range(m):
            for j in range(i + 1):
                a.append(np.array([uk + i]))
                b.append(np.array([uk + j]))
        for j in range(m):
            a.append(np.array([hk]))
            b.append(np.array([uk + j]))
        a.append(np.array([hk]))
        b.append(np.array([hk]))
        for l in range(m):
            a.append(np.full(xd, uk + l))
            b.append(k * zd + o + r_all)
        a.append(np.full(xd, hk))
        b.append(k * zd + o + r_all)
        for l in range(m):
            a.append((k + 1) * zd + o + r_all)
            b.append(np.full(xd, uk + l))
        a.append((k + 1) * zd + o + r_all)
        b.append(np.full(xd, hk))
        a = np.concatenate(a)
        b = np.concatenate(b)
        rows[k * per : (k + 1) * per] = np.maximum(a, b)
        cols[k * per : (k + 1) * per] = np.minimum(a, b)
    return rows + index_base, cols + index_base


def pade4_hessian_values(Z, mu, lay: Layout, G0, Gj, x_off=None):
    """Values of grad^2 (sum_k mu_k^T delta_k) in the order above.  ``mu`` is
    [K, x_dim].  Returns [K, hess_nnz_per_interval]."""
    d, n, m, xd = lay.C, lay.n, lay.m, lay.x_dim
    per = hess_nnz_per_interval(lay)
    out = np.empty((lay.K, per))
    ip = lambda A, B: float(np.sum(A * B))
    for k in range(lay.K):
        G = G0 + np.tensordot(lay.u(Z, k), Gj, axes=1) if m else G0
        h = lay.dt(Z, k)
        Xn, Xc = lay.X(Z, k + 1, x_off), lay.X(Z, k, x_off)
        S, D = Xn + Xc, Xn - Xc
        M = mu[k].reshape(d, n).T
        G2 = G @ G
        p = 0
        for i in range(m):
            for j in range(i + 1):
                out[k, p] = h * h / 12 * ip(M, (Gj[i] @ Gj[j] + Gj[j] @ Gj[i]) @ D)
                p += 1
        for j in range(m):
            out[k, p] = -0.5 * ip(M, Gj[j] @ S) + h / 6 * ip(M, (Gj[j] @ G + G @ Gj[j]) @ D)
            p += 1
        out[k, p] = ip(M, G2 @ D) / 6
        p += 1
        Kl = [Gj[l] @ G + G @ Gj[l] for l in range(m)]
        for l in range(m):  # (u_l, X_k): d delta/du_l is linear in X_k with coefficient below
            A = (-h / 2 * Gj[l] - h * h / 12 * Kl[l]).T @ M
            out[k, p : p + xd] = A.T.reshape(-1)
            p += xd
        A = (-G / 2 - h / 6 * G2).T @ M
        out[k, p : p + xd] = A.T.reshape(-1)
        p += xd
        for l in range(m):
            A = (-h / 2 * Gj[l] + h * h / 12 * Kl[l]).T @ M
            out[k, p : p + xd] = A.T.reshape(-1)
            p += xd
        A = (-G / 2 + h / 6 * G2).T @ M
        out[k, p : p + xd] = A.T.reshape(-1)
    return out


def pade_hessian_values(Z, mu, lay: Layout, G0, Gj, order=4, x_off=None):
    """grad^2 (sum_k mu_k^T delta_k) for ANY diagonal Pade order p = 2q, same value order as ``pade4_hessian_values``.
    With T_j = c_j h^j, T'_j = j c_j h^(j-1), T''_j = j (j-1) c_j h^(j-2), Y_j = (-1)^j X_{k+1} - X_k and M = mu_k (n x d):
        W_0 = M,  W_j = G^T W_{j-1}                                         ((G^j)^T M)
        V_{l,0} = 0,  V_{l,j} = G^T V_{l,j-1} + G_l^T W_{j-1}               ((d_l G^j)^T M)
        U_{il,j} = G^T U_{il,j-1} + G_l^T V_{i,j-1} + G_i^T V_{l,j-1}       ((d_i d_l G^j)^T M;  0 for j < 2)
        (u_i,u_l): sum_j T_j <U_{il,j}, Y_j>     (h,u_l): sum_j T'_j <V_{l,j}, Y_j>     (h,h): sum_j T''_j <W_j, Y_j>
        d2/du_l dX_{k+1} = sum_j T_j (-1)^j V_{l,j},  d2/du_l dX_k = -sum_j T_j V_{l,j}
        d2/dh dX_{k+1}   = sum_j T'_j (-1)^j W_j,      d2/dh dX_k   = -sum_j T'_j W_j
    (delta is linear in X: no X-X block).  Order 4 reproduces ``pade4_hessian_values`` (tests/test_oracle_pins.py)."""
    c = pade_coeffs(order)
    q = order // 2
    d, n, m, xd = lay.C, lay.n, lay.m, lay.x_dim
    per = hess_nnz_per_interval(lay)
    out = np.empty((lay.K, per))
    ip = lambda A, B: float(np.sum(A * B))
    for k in range(lay.K):
        G = G0 + np.tensordot(lay.u(Z, k), Gj, axes=1) if m else G0
        h = lay.dt(Z, k)
        Xn, Xc = lay.X(Z, k + 1, x_off), lay.X(Z, k, x_off)
        M = mu[k].reshape(d, n).T
        Y = [((-1) ** j) * Xn - Xc for j in range(q + 1)]
        T = [c[j] * h**j for j in range(q + 1)]
        T1 = [j * c[j] * h ** (j - 1) if j >= 1 else 0.0 for j in range(q + 1)]
        T2 = [j * (j - 1) * c[j] * h ** (j - 2) if j >= 2 else 0.0 for j in range(q + 1)]
        W = [M]
        V = [[np.zeros_like(M) for _ in range(m)]]
        U = [{(i, l): np.zeros_like(M) for i in range(m) for l in range(i + 1)}]
        for j in range(1, q + 1):
            W.append(G.T @ W[j - 1])
            V.append([G.T @ V[j - 1][l] + Gj[l].T @ W[j - 1] for l in range(m)])
            U.append({(i, l): G.T @ U[j - 1][(i, l)] + Gj[l].T @ V[j - 1][i] + Gj[i].T @ V[j - 1][l] for i in range(m) for l in range(i + 1)})
        p = 0
        for i in range(m):
            for l in range(i + 1):
                out[k, p] = sum(T[j] * ip(U[j][(i, l)], Y[j]) for j in range(2, q + 1))
                p += 1
        for l in range(m):
            out[k, p] = sum(T1[j] * ip(V[j][l], Y[j]) for j in range(1, q + 1))
            p += 1
        out[k, p] = sum(T2[j] * ip(W[j], Y[j]) for j in range(2, q + 1))
        p += 1
        for l in range(m):
            A = -sum(T[j] * V[j][l] for j in range(1, q + 1))
            out[k, p : p + xd] = A.T.reshape(-1)
            p += xd
        A = -sum(T1[j] * W[j] for j in range(1, q + 1))
        out[k, p : p + xd] = A.T.reshape(-1)
        p += xd
        for l in range(m):
            A = sum(T[j] * (-1) ** j * V[j][l] for j in range(1, q + 1))
            out[k, p : p + xd] = A.T.reshape(-1)
            p += xd
        A = sum(T1[j] * (-1) ** j * W[j] for j in range(1, q + 1))
        out[k, p : p + xd] = A.T.reshape(-1)
    return out


def hessian_dense(vals, lay: Layout, x_off=None):
    rows, cols = hess_structure(lay, x_off)
    nv = lay.z_dim * lay.N
    Hm = np.zeros((nv, nv))
    np.add.at(Hm, (rows, cols), np.asarray(vals).reshape(-1))
    return Hm + np.tril(Hm, -1).T


# --------------------------------------------------------------------------- #
# Derivative / time-consistency rows -- host-side only
# --------------------------------------------------------------------------- #


def derivative_residual(Z, off_x, off_dx, m, dt_off):
    """u_{k+1} - u_k - dt_k * du_k  [REF smooth_pulse_problem.jl:267-275]"""
    return Z[1:, off_x : off_x + m] - Z[:-1, off_x : off_x + m] - Z[:-1, dt_off : dt_off + 1] * Z[:-1, off_dx : off_dx + m]


def derivative_jacobian(Z, z_dim, off_x, off_dx, m, dt_off, index_base=0):
    """Triplets of the rows above in the library's order: per interval [-1 | +1 | -dt_k | -dx_k]
    (off_dx < 0: time consistency, dx == 1, segments [-1 | +1 | -1])."""
    N = Z.shape[0]
    rows, cols, vals = [], [], []
    r = np.arange(m)
    for k in range(N - 1):
        r0, v0 = k * m + index_base, k * z_dim + index_base
        rows += [r0 + r, r0 + r]
        cols += [v0 + off_x + r, v0 + z_dim + off_x + r]
        vals += [-np.ones(m), np.ones(m)]
        if off_dx >= 0:
            rows += [r0 + r, r0 + r]
            cols += [v0 + off_dx + r, np.full(m, v0 + dt_off)]
            vals += [np.full(m, -Z[k, dt_off]), -Z[k, off_dx : off_dx + m]]
        else:
            rows += [r0 + r]
            cols += [np.full(m, v0 + dt_off)]
            vals += [-np.ones(m)]
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


def time_consistency_residual(Z, t_off, dt_off):
    """t_{k+1} - t_k - dt_k  [REF smooth_pulse_problem.jl:277]"""
    return (Z[1:, t_off] - Z[:-1, t_off] - Z[:-1, dt_off])[:, None]


# --------------------------------------------------------------------------- #
# Synthetic benchmark inputs
# --------------------------------------------------------------------------- #


def config_system(config: int) -> System:
    if config == 1:
        return quantum_system(0.5 * PAULIS["Z"], [PAULIS["X"], PAULIS["Y"]], [1.0, 1.0])
    if config == 2:
        return multi_transmon_system([4.0, 4.1], [0.2, 0.2], [[0, 0.1], [0.1, 0]], levels_per_transmon=2, drive_bounds=0.1)
    if config in (3, 4, 5):
        return multi_transmon_system(
            [4.0, 4.1, 4.2],
            [0.2, 0.21, 0.22],
            [[0, 0.01, 0.02], [0.01, 0, 0.03], [0.02, 0.03, 0]],
            levels_per_transmon=3,
            drive_bounds=0.1,
        )
    raise ValueError(config)


def synthetic_trajectory(sys: System, N: int, seed: int, dt=0.1, u_scale=0.02, u_clip=0.1, noise=1e-3):
    """Near-feasible iterate: X_1 = iso(I), X_{k+1} = expm(dt G(u_k)) X_k + noise.
    Returns (Z [N, z_dim], Layout)."""
    rng = np.random.default_rng(seed)
    d, m = sys.levels, sys.n_drives
    lay = Layout.smooth_pulse(d, m, N)
    Z = np.zeros((N, lay.z_dim))
    u = np.clip(u_scale * rng.standard_normal((N, m)), -u_clip, u_clip)
    du = 0.01 * rng.standard_normal((N, m))
    ddu = 0.01 * rng.standard_normal((N, m))
    Z[:, lay.dt_off] = dt
    Z[:, lay.dt_off + 1] = dt * np.arange(N)
    Z[:, lay.u_off : lay.u_off + m] = u
    Z[:, lay.u_off + m : lay.u_off + 2 * m] = du
    Z[:, lay.u_off + 2 * m : lay.u_off + 3 * m] = ddu
    G0, Gj = sys.G_drift, np.array(sys.G_drives)
    X = np.vstack([np.eye(d), np.zeros((d, d))])
    for k in range(N):
        Z[k, : lay.x_dim] = X.T.reshape(-1)
        if k + 1 < N:
            G = G0 + np.tensordot(u[k], Gj, axes=1)
            X = scipy.linalg.expm(dt * G) @ X + noise * rng.standard_normal(X.shape)
    return Z, lay


# --------------------------------------------------------------------------- #
# Reading a NamedTrajectory.datavec out of a reference JLD2 cache
# (SURVEY Appendix A): JLD2 stores Vector{Float64} raw; datavec starts with the
# iso-vec of the identity (the initial condition of every unitary trajectory).
# --------------------------------------------------------------------------- #


def read_jld2_datavec(path, d, z_dim, N, n_members=1):
    b = open(path, "rb").read()
    x0 = operator_to_iso_vec(np.eye(d))
    x0 = np.tile(x0, n_members)
    off = b.find(struct.pack("<%dd" % x0.size, *x0))
    if off < 0:
        raise ValueError("iso(I) prefix not found in %s" % path)
    Z = np.frombuffer(b[off : off + 8 * z_dim * N], "<f8").reshape(N, z_dim).copy()
    return Z
