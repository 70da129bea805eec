"""CPU tests: pin the oracle (numpy + C restatement) on everything the reference holds for
this path -- known-answer literals of its own tests and trajectories solved by the reference
itself -- then on itself (finite differences, numpy vs C, committed vectors)."""
import numpy as np
import pytest

from helpers import ref_case
from oracle import pade_oracle as po
from oracle import ref_lib


# ---- known-answer literals from the reference's own tests --------------------------------
def test_iso_literals():
    # [REF src/quantum/primitives/isomorphisms.jl:471-509]
    assert np.allclose(po.ket_to_iso([1.0, 2.0]), [1, 2, 0, 0])
    assert np.allclose(po.ket_to_iso([-1j, 2 + 3j]), [0, 2, -1, 3])
    assert np.allclose(po.iso_to_ket([0, 2, -1, 3]), [-1j, 2 + 3j])
    I = [1.0, 0, 0, 0, 0, 1.0, 0, 0]
    assert np.allclose(po.iso_vec_to_operator(I), np.eye(2))
    assert np.allclose(po.iso_vec_to_iso_operator(I), np.eye(4))
    assert np.allclose(po.operator_to_iso_vec(np.eye(2)), I)
    XY = [0, 1, 0, 1, 1, 0, -1, 0]
    U = np.array([[0, 1 - 1j], [1 + 1j, 0]])
    assert np.allclose(po.iso_vec_to_operator(XY), U)
    assert np.allclose(po.iso_vec_to_iso_operator(XY), [[0, 1, 0, 1], [1, 0, -1, 0], [0, -1, 0, 1], [1, 0, 1, 0]])
    assert np.allclose(po.operator_to_iso_vec(U), XY)
    assert np.allclose(po.iso_operator_to_iso_vec(po.iso_vec_to_iso_operator(XY)), XY)


def test_generator_literals():
    # [REF isomorphisms.jl:620-643]
    Hc = np.array([[1.0, 2.0], [3.0, 4.0]]) + 1j * np.array([[0.0, 1.0], [1.0, 0.0]])
    GH = po.G_of_H(Hc)
    assert np.allclose(po.H_of_G(GH), Hc)
    assert np.allclose(GH, [[0, 1, 1, 2], [1, 0, 3, 4], [-1, -2, 0, 1], [-3, -4, 1, 0]])
    assert np.allclose(po.iso(Hc), [[1, 2, 0, -1], [3, 4, -1, 0], [0, 1, 1, 2], [1, 0, 3, 4]])
    assert np.allclose(po.iso(-1j * Hc), GH)
    assert np.allclose(po.ad_vec(np.array([[0, 1], [1, 0]])), [[0, 1, -1, 0], [1, 0, 0, -1], [-1, 0, 0, 1], [0, -1, 1, 0]])
    assert np.allclose(po.ad_vec(np.array([[0, -1j], [1j, 0]])), np.array([[0, -1j, -1j, 0], [1j, 0, 0, -1j], [1j, 0, 0, -1j], [0, 1j, 1j, 0]]))


def test_var_G_literals():
    # [REF isomorphisms.jl:645-671]
    G = np.array([[1.0, 2.0], [3.0, 4.0]])
    v1, v2 = np.array([[0.0, 1.0], [1.0, 0.0]]), np.array([[0.0, 0.0], [1.0, 1.0]])
    assert np.allclose(po.var_G(G, [v1]), [[1, 2, 0, 0], [3, 4, 0, 0], [0, 1, 1, 2], [1, 0, 3, 4]])
    assert np.allclose(
        po.var_G(G, [v1, v2]),
        [[1, 2, 0, 0, 0, 0], [3, 4, 0, 0, 0, 0], [0, 1, 1, 2, 0, 0], [1, 0, 3, 4, 0, 0], [0, 0, 0, 0, 1, 2], [1, 1, 0, 0, 3, 4]],
    )


def test_operator_literals():
    # [REF src/quantum/object_utils.jl:174-176]
    assert np.allclose(po.annihilate(2), [[0, 1], [0, 0]])
    assert np.allclose(po.annihilate(3), [[0, 1, 0], [0, 0, np.sqrt(2)], [0, 0, 0]])
    # [REF src/quantum/operators/lifted_operators.jl:22-31] lift = kron with identities
    X = po.PAULIS["X"]
    assert np.allclose(po.lift_operator(X, 1, [2, 2]), np.kron(X, np.eye(2)))
    assert np.allclose(po.lift_operator(X, 2, [2, 3][::-1][::-1][:1] + [2]), np.kron(np.eye(2), X))


def test_bilinear_fixture_generators():
    # the 4x4 Gx, Gy, Gz of test/test_utils.jl:113-133 are iso(-i sigma/...) of the Paulis (up to the
    # fixture's own sign/scale conventions): Gz is G(-Z)/..; check the structural identity G = iso(-iH)
    Gx = np.array([[0, 0, 0, 1], [0, 0, 1, 0], [0, -1, 0, 0], [-1, 0, 0, 0]], float)
    Gy = np.array([[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 0, -1], [0, 0, 1, 0]], float)
    Gz = np.array([[0, 0, 1, 0], [0, 0, 0, -1], [-1, 0, 0, 0], [0, 1, 0, 0]], float)
    assert np.allclose(po.G_of_H(po.PAULIS["X"]), Gx)
    assert np.allclose(po.G_of_H(po.PAULIS["Y"]), Gy)
    assert np.allclose(po.G_of_H(po.PAULIS["Z"]), Gz)


def test_hadamard_fixture_is_iso_consistent():
    # test/test_utils.jl:56-100: the literal Hadamard trajectory's initial/goal iso-vecs
    init = [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0]
    goal_cols_last_knot = np.array([0.707107, 0.707107, 1.38778e-17, -1.52656e-16, 0.707107, -0.707107, -1.249e-16, 4.16334e-16])
    assert np.allclose(po.operator_to_iso_vec(np.eye(2)), init)
    Hgate = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
    assert np.allclose(po.operator_to_iso_vec(Hgate), goal_cols_last_knot, atol=1e-6)


# ---- semantics pinned on trajectories solved by the reference itself ----------------------
# thresholds from SURVEY.md section 0.3 / 8(c)
@pytest.mark.parametrize(
    "name,exp_tol,pade",
    [
        ("two_qubit_zoh", 1e-11, {4: (1e-9, 3e-9), 6: (0, 1e-11), 8: (0, 1e-11)}),
        ("multilevel_transmon", 1e-9, {10: (0, 1e-7)}),
        ("first_gate", 3e-5, {}),
    ],
)
def test_reference_trajectories_single(name, exp_tol, pade, golden, golden_meta):
    systems, lay, _ = ref_case(name, golden_meta)
    Z = golden("ref_" + name)["Z"]
    s = systems[0]
    G0, Gj = s.G_drift, np.array(s.G_drives)
    r = np.abs(po.exp_residual(Z, lay, G0, Gj)).max()
    assert r < exp_tol, r
    for order, (lo, hi) in pade.items():
        rp = np.abs(po.pade_residual(Z, lay, G0, Gj, order)).max()
        assert lo <= rp < hi, (order, rp)


def test_first_gate_index_convention_tripwire(golden, golden_meta):
    """(u_k, dt_k) -- not u_{k+1} -- drives interval k: the wrong choice is 100x worse."""
    systems, lay, _ = ref_case("first_gate", golden_meta)
    Z = golden("ref_first_gate")["Z"]
    s = systems[0]
    G0, Gj = s.G_drift, np.array(s.G_drives)
    good = np.abs(po.exp_residual(Z, lay, G0, Gj)).max()
    Zs = Z.copy()
    Zs[:-1, lay.u_off : lay.u_off + lay.m] = Z[1:, lay.u_off : lay.u_off + lay.m]
    bad = np.abs(po.exp_residual(Zs, lay, G0, Gj)).max()
    assert good < 3e-5 and bad > 1e-3 and bad > 50 * good


@pytest.mark.parametrize("name,own_tol,nominal_idx", [("sampling_robust", 1e-4, 0), ("robust_sampling", 5e-3, 2)])
def test_reference_trajectories_ensemble(name, own_tol, nominal_idx, golden, golden_meta):
    """Member-major state order, member -> system mapping, per-knot (free) dt."""
    systems, lay, x_offs = ref_case(name, golden_meta)
    Z = golden("ref_" + name)["Z"]
    dts = Z[:, lay.dt_off]
    assert dts.max() / dts.min() > 5  # free, non-uniform timesteps
    assert np.abs(Z[1:, lay.dt_off + 1] - Z[:-1, lay.dt_off + 1] - dts[:-1]).max() < 1e-13  # time consistency rows
    nom = systems[nominal_idx]
    for i, (s, xo) in enumerate(zip(systems, x_offs)):
        own = np.abs(po.exp_residual(Z, lay, s.G_drift, np.array(s.G_drives), x_off=xo)).max()
        assert own < own_tol, (i, own)
        if i != nominal_idx:
            wrong = np.abs(po.exp_residual(Z, lay, nom.G_drift, np.array(nom.G_drives), x_off=xo)).max()
            assert wrong > 4 * own, (i, own, wrong)


def test_derivative_rows_on_reference_trajectory(golden, golden_meta):
    # u_{k+1} - u_k - dt_k du_k = 0 and the same for (du, ddu)  [REF smooth_pulse_problem.jl:267-275]
    systems, lay, _ = ref_case("two_qubit_zoh", golden_meta)
    Z = golden("ref_two_qubit_zoh")["Z"]
    m = lay.m
    assert np.abs(po.derivative_residual(Z, lay.u_off, lay.u_off + m, m, lay.dt_off)).max() < 1e-6
    assert np.abs(po.derivative_residual(Z, lay.u_off + m, lay.u_off + 2 * m, m, lay.dt_off)).max() < 1e-6


# ---- the oracle against itself ---------------------------------------------------------------
def test_pade_coefficients():
    assert np.allclose(po.pade_coeffs(4), [1, 1 / 2, 1 / 12])
    assert np.allclose(po.pade_coeffs(6), [1, 1 / 2, 1 / 10, 1 / 120])
    assert np.allclose(po.pade_coeffs(8), [1, 1 / 2, 3 / 28, 1 / 84, 1 / 1680])
    assert np.allclose(po.pade_coeffs(10), [1, 1 / 2, 1 / 9, 1 / 72, 1 / 1008, 1 / 30240])


def test_pade_converges_to_exp():
    s = po.config_system(2)
    # exp-feasible trajectory (noise = 0): the Pade residual is then pure truncation error
    Z, lay = po.synthetic_trajectory(s, 8, seed=11, dt=0.4, u_scale=0.1, noise=0.0)
    G0, Gj = s.G_drift, np.array(s.G_drives)
    assert np.abs(po.exp_residual(Z, lay, G0, Gj)).max() < 1e-13
    errs = [np.abs(po.pade_residual(Z, lay, G0, Gj, p)).max() for p in (4, 6, 8, 10)]
    assert errs[0] > 10 * errs[1] > 100 * errs[2] > 1000 * errs[3] and errs[3] < 1e-11, errs


def _fd_jac(f, z0, eps=1e-6):
    J = np.zeros((f(z0).size, z0.size))
    for i in range(z0.size):
        zp, zm = z0.copy(), z0.copy()
        zp[i] += eps
        zm[i] -= eps
        J[:, i] = (f(zp) - f(zm)) / (2 * eps)
    return J


@pytest.mark.parametrize("order", [4, 6])
def test_oracle_jacobian_finite_differences(order):
    s = po.config_system(1)
    Z, lay = po.synthetic_trajectory(s, 5, seed=3, noise=1e-2)
    Z[:, lay.dt_off] = 0.1 + 0.05 * np.random.default_rng(0).random(5)
    G0, Gj = s.G_drift, np.array(s.G_drives)
    J = po.pade_jacobian_dense(Z, lay, G0, Gj, order)
    Jfd = _fd_jac(lambda z: po.pade_residual(z.reshape(Z.shape), lay, G0, Gj, order).reshape(-1), Z.reshape(-1).copy())
    assert np.abs(J - Jfd).max() < 1e-8


def test_oracle_hessian_finite_differences():
    s = po.config_system(1)
    Z, lay = po.synthetic_trajectory(s, 4, seed=4, noise=1e-2)
    Z[:, lay.dt_off] = 0.1 + 0.05 * np.random.default_rng(1).random(4)
    G0, Gj = s.G_drift, np.array(s.G_drives)
    mu = np.random.default_rng(2).standard_normal((lay.K, lay.x_dim))
    Hd = po.hessian_dense(po.pade4_hessian_values(Z, mu, lay, G0, Gj), lay)
    g = lambda z: po.pade_jacobian_dense(z.reshape(Z.shape), lay, G0, Gj, 4).T @ mu.reshape(-1)
    Hfd = _fd_jac(g, Z.reshape(-1).copy())
    assert np.abs(Hd - Hfd).max() < 1e-7
    assert np.abs(Hd - Hd.T).max() == 0


@pytest.mark.parametrize("cfg,order", [(1, 4), (2, 4), (2, 8), (1, 10)])
def test_oracle_hessian_complex_step_of_the_pinned_jacobian(cfg, order):
    """The tightest pin the Hessian of the Lagrangian can have here: the COMPLEX-STEP derivative of the Jacobian the tests above pin on
    the Frechet derivative of the reference's exp constraint.  g(z) = J(z)^T mu is a polynomial in (X, u, h) evaluated by the
    same oracle code on complex input, so H[:, j] = Im g(z + i eps e_j) / eps is exact to rounding (no step-size error:
    eps = 1e-30), 1e-12 instead of the 1e-7 of central differences -- for order 4 (both restatements) and one higher order."""
    s = po.config_system(cfg)
    N = 3
    Z, lay = po.synthetic_trajectory(s, N, seed=40 + cfg, noise=1e-2)
    Z[:, lay.dt_off] = 0.1 + 0.05 * np.random.default_rng(1).random(N)
    G0, Gj = s.G_drift, np.array(s.G_drives)
    mu = np.random.default_rng(2).standard_normal((lay.K, lay.x_dim))
    Hd = po.hessian_dense(po.pade_hessian_values(Z, mu, lay, G0, Gj, order), lay)
    if order == 4:
        assert np.abs(Hd - po.hessian_dense(po.pade4_hessian_values(Z, mu, lay, G0, Gj), lay)).max() < 1e-12 * max(1.0, np.abs(Hd).max())
    nv, eps = Z.size, 1e-30
    Hcs = np.empty((nv, nv))
    z0 = Z.reshape(-1).astype(complex)
    for j in range(nv):
        z = z0.copy()
        z[j] += 1j * eps
        Hcs[:, j] = (po.pade_jacobian_dense(z.reshape(Z.shape), lay, G0, Gj, order).T @ mu.reshape(-1)).imag / eps
    scale = max(1.0, np.abs(Hcs).max())
    assert np.abs(Hd - Hcs).max() < 1e-12 * scale, np.abs(Hd - Hcs).max() / scale
    assert np.abs(Hcs - Hcs.T).max() < 1e-12 * scale
    # tripwire: a wrong index convention (u_{k+1} instead of u_k) is caught at this tolerance
    Zs = Z.copy()
    Zs[:-1, lay.u_off : lay.u_off + lay.m] = Z[1:, lay.u_off : lay.u_off + lay.m]
    Hs = po.hessian_dense(po.pade_hessian_values(Zs, mu, lay, G0, Gj, order), lay)
    assert np.abs(Hs - Hcs).max() > 1e-6 * scale


@pytest.mark.parametrize("cfg,N", [(1, 9), (2, 10), (3, 4)])
def test_c_restatement_matches_numpy(cfg, N):
    s = po.config_system(cfg)
    Z, lay = po.synthetic_trajectory(s, N, seed=100 + cfg)
    rng = np.random.default_rng(cfg)
    Z[:, lay.dt_off] = 0.1 + 0.05 * rng.random(N)
    G0, Gj = s.G_drift, np.array(s.G_drives)
    d1, j1 = ref_lib.eval_jac(Z, lay, G0, Gj, nthreads=2)
    assert np.abs(d1 - po.pade_residual(Z, lay, G0, Gj, 4)).max() < 1e-14
    assert np.abs(j1 - po.pade_jacobian_values(Z, lay, G0, Gj, 4)).max() < 1e-13
    mu = rng.standard_normal((lay.K, lay.x_dim))
    h1 = ref_lib.hess(Z, mu, lay, G0, Gj, nthreads=2)
    h0 = po.pade4_hessian_values(Z, mu, lay, G0, Gj)
    assert np.abs(h1 - h0).max() < 1e-12 * max(1.0, np.abs(h0).max())


@pytest.mark.parametrize("name", ["config1", "config2", "config3"])
def test_committed_vectors_reproduce(name, golden, golden_meta):
    """The committed golden vectors are what the oracle computes today (guards silent drift)."""
    v = golden("vec_" + name)
    m = golden_meta["oracle_vectors"][name]
    lay = po.Layout(d=m["d"], m=m["m"], N=m["N"], z_dim=m["z_dim"], x_off=m["x_off"], u_off=m["u_off"], dt_off=m["dt_off"])
    assert np.abs(po.pade_residual(v["Z"], lay, v["G0"], v["Gj"], 4) - v["delta"]).max() < 1e-14
    d1, j1 = ref_lib.eval_jac(v["Z"], lay, v["G0"], v["Gj"])
    assert np.abs(d1 - v["delta"]).max() < 1e-14 and np.abs(j1 - v["jac"]).max() < 1e-13
    s = po.config_system(m["config"])
    assert np.allclose(s.G_drift, v["G0"], rtol=0, atol=1e-14) and np.allclose(np.array(s.G_drives), v["Gj"], rtol=0, atol=1e-14)


def test_structure_counts_match_survey_table():
    # SURVEY.md section 8 config table
    for cfg, (jn, hn) in {1: (88, 54), 2: (672, 335), 3: (167670, 20440)}.items():
        s = po.config_system(cfg)
        lay = po.Layout.smooth_pulse(s.levels, s.n_drives, 3)
        assert po.jac_nnz_per_interval(lay) == jn and po.hess_nnz_per_interval(lay) == hn
        assert ref_lib.lib().pade_ref_jac_nnz_per_interval(lay.d, lay.m) == jn
        assert ref_lib.lib().pade_ref_hess_nnz_per_interval(lay.d, lay.m) == hn


def test_derivative_rows_jacobian_finite_differences():
    rng = np.random.default_rng(8)
    N, z_dim, m = 5, 12, 3
    Z = rng.standard_normal((N, z_dim))
    off_x, off_dx, dt_off = 2, 6, 0
    r, c, v = po.derivative_jacobian(Z, z_dim, off_x, off_dx, m, dt_off)
    J = np.zeros(((N - 1) * m, N * z_dim))
    np.add.at(J, (r, c), v)
    f = lambda z: po.derivative_residual(z.reshape(N, z_dim), off_x, off_dx, m, dt_off).reshape(-1)
    assert np.abs(J - _fd_jac(f, Z.reshape(-1).copy())).max() < 1e-8
    r, c, v = po.derivative_jacobian(Z, z_dim, 1, -1, 1, dt_off)  # time consistency on component 1
    J = np.zeros((N - 1, N * z_dim))
    np.add.at(J, (r, c), v)
    f = lambda z: po.time_consistency_residual(z.reshape(N, z_dim), 1, dt_off).reshape(-1)
    assert np.abs(J - _fd_jac(f, Z.reshape(-1).copy())).max() < 1e-8


def test_oracle_rollout_reproduces_reference_solved_states(golden, golden_meta):
    """Pin of `exact_rollout` (the checker of the GPU rollout): propagating the reference's own converged trajectory from
    its knot-0 state with exp(dt_k G(u_k)) must land on the reference's own states (it solved x_{k+1} = expv(dt G) x_k to
    6e-12 per knot), at every knot."""
    systems, lay, _ = ref_case("two_qubit_zoh", golden_meta)
    Z = golden("ref_two_qubit_zoh")["Z"]
    so = systems[0]
    X = po.exact_rollout(Z, lay, so.G_drift, np.array(so.G_drives))
    states = np.stack([lay.X(Z, k).T.reshape(-1) for k in range(lay.N)])
    assert np.abs(X - states).max() < 1e-9
    assert np.array_equal(X[0], states[0])


def test_compact_density_isomorphism_literals_and_lindblad_rhs():
    """The reference's own checks of the compact density isomorphism [REF isomorphisms.jl:539-620]: the explicit 2x2
    ordering literal, sizes / nnz of L and P, P L = I, L x = iso_vec; and the physics pin of the compact Lindbladian
    [REF open_quantum_systems.jl:541-588]: Gc x == compact(-i[H, rho] + sum L rho L' - 1/2 {L'L, rho})."""
    a, b, c, d = 0.6, 0.4, 0.2, 0.1
    rho = np.array([[a, c + d * 1j], [c - d * 1j, b]])
    assert np.allclose(po.density_to_compact_iso(rho), [a, c, b, d])
    rng = np.random.default_rng(0)
    for n in (2, 3, 4, 7):
        L, P = po.density_lift_matrix(n), po.density_projection_matrix(n)
        assert L.shape == (2 * n * n, n * n) and P.shape == (n * n, 2 * n * n)
        assert np.count_nonzero(L) == n * (2 * n - 1) and np.count_nonzero(P) == n * n
        assert np.allclose(P @ L, np.eye(n * n))
        A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        r = (A + A.conj().T) / 2
        assert np.allclose(L @ po.density_to_compact_iso(r), po.density_to_iso_vec(r))
        assert np.allclose(P @ po.density_to_iso_vec(r), po.density_to_compact_iso(r))
        assert np.allclose(po.compact_iso_to_density(po.density_to_compact_iso(r)), r)
        LP = L @ P
        assert np.allclose(LP @ LP, LP)
    n = 3
    H = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    H = H + H.conj().T
    Ls = [rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) for _ in range(2)]
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    r = (A + A.conj().T) / 2
    drift, drives = po.compact_lindbladian_generators(H, [H @ H], Ls)
    rhs = -1j * (H @ r - r @ H) + sum(Lo @ r @ Lo.conj().T - 0.5 * (Lo.conj().T @ Lo @ r + r @ Lo.conj().T @ Lo) for Lo in Ls)
    assert np.abs(drift @ po.density_to_compact_iso(r) - po.density_to_compact_iso(rhs)).max() < 1e-13
    HH = H @ H
    assert np.abs(drives[0] @ po.density_to_compact_iso(r) - po.density_to_compact_iso(-1j * (HH @ r - r @ HH))).max() < 1e-12


def test_oracle_general_generator_dimension_finite_differences():
    """Layout(gen=n): the oracle's residual / Jacobian / Hessian for a general real n x n generator on one column (odd n),
    against finite differences."""
    rng = np.random.default_rng(3)
    n, m, N = 9, 2, 4
    lay = po.Layout(d=0, m=m, N=N, z_dim=n + 2 + m, x_off=0, u_off=n + 2, dt_off=n, cols=1, gen=n)
    G0 = rng.standard_normal((n, n))
    Gj = rng.standard_normal((m, n, n))
    Z = 0.3 * rng.standard_normal((N, lay.z_dim))
    Z[:, lay.dt_off] = 0.1
    assert lay.x_dim == n and po.jac_nnz_per_interval(lay) == 2 * n * n + n * (m + 1)
    J = po.pade_jacobian_dense(Z, lay, G0, Gj, 4)
    eps = 1e-6
    for col in (0, n - 1, lay.dt_off, lay.u_off + 1, lay.z_dim + 3):
        Zp, Zm = Z.reshape(-1).copy(), Z.reshape(-1).copy()
        Zp[col] += eps
        Zm[col] -= eps
        fd = (po.pade_residual(Zp.reshape(N, -1), lay, G0, Gj, 4) - po.pade_residual(Zm.reshape(N, -1), lay, G0, Gj, 4)).reshape(-1) / (2 * eps)
        assert np.abs(fd - J[:, col]).max() < 1e-7


# ---- the Pade Jacobian pinned against the reference's OWN constraint function ---------------------------------
@pytest.mark.parametrize(
    "name,orders",
    [
        ("two_qubit_zoh", {4: (1e-11, 2e-7), 6: (0, 1e-9), 8: (0, 1e-9), 10: (0, 1e-9)}),
        ("multilevel_transmon", {10: (0, 1e-6)}),  # large steps (Pade-10 residual 1e-7 there, see above): looser band
    ],
)
def test_pade_jacobian_vs_frechet_derivative_of_the_exp_constraint(name, orders, golden, golden_meta):
    """The reference's constraint is x_{k+1} - exp(dt G(u_k)) x_k and its Jacobian is the Frechet derivative of expm
    (it gets it by ForwardDiff through expv).  On trajectories SOLVED BY THE REFERENCE (residual ~1e-11) the oracle's
    analytic Pade-p Jacobian, premultiplied by (B^-_p)^{-1}, must equal that Jacobian: to 1e-9 at orders 6..10 (the
    high orders are the exp constraint's own Jacobian to rounding), and at order 4 to its known truncation order
    (||dt G|| <= 0.076: x^5/720-sized terms, i.e. nonzero but below 2e-7).  This pins every Jacobian segment -- the B^{+-}
    blocks, d/du_l, d/ddt, their signs, the (u_k, dt_k) index convention and the triplet order -- on numbers the
    reference's constraint function defines, not on finite differences of the oracle itself."""
    systems, lay, _ = ref_case(name, golden_meta)
    Z = golden("ref_" + name)["Z"]
    s = systems[0]
    G0, Gj = s.G_drift, np.array(s.G_drives)
    if lay.K > 60:  # every interval is an independent check; a spread subset keeps the CPU suite fast
        keep = np.unique(np.linspace(0, lay.K - 1, 40).astype(int))
    else:
        keep = np.arange(lay.K)
    Je = po.exp_jacobian_values(Z, lay, G0, Gj)[keep]
    scale = np.abs(Je).max()
    for order, (lo, hi) in orders.items():
        Jp = po.pade_jacobian_in_exp_form(Z, lay, G0, Gj, order)[keep]
        err = np.abs(Jp - Je).max() / scale
        assert lo <= err < hi, (order, err)
    # tripwire: the same comparison with the drive of the NEXT knot (the wrong index convention) is far off
    Zs = Z.copy()
    Zs[:-1, lay.u_off : lay.u_off + lay.m] = Z[1:, lay.u_off : lay.u_off + lay.m]
    order = max(orders)
    bad = np.abs(po.pade_jacobian_in_exp_form(Zs, lay, G0, Gj, order)[keep] - Je).max() / scale
    assert bad > 1e-4, bad


def test_exp_jacobian_finite_differences(golden, golden_meta):
    """The Frechet-derivative Jacobian above really is the Jacobian of the oracle's exp residual (central differences)."""
    systems, lay, _ = ref_case("two_qubit_zoh", golden_meta)
    Z = golden("ref_two_qubit_zoh")["Z"][:4].copy()
    lay = po.Layout(d=lay.d, m=lay.m, N=4, z_dim=lay.z_dim, x_off=lay.x_off, u_off=lay.u_off, dt_off=lay.dt_off)
    s = systems[0]
    G0, Gj = s.G_drift, np.array(s.G_drives)
    rows, cols = po.jac_structure(lay)
    J = np.zeros((lay.x_dim * lay.K, lay.z_dim * lay.N))
    np.add.at(J, (rows, cols), po.exp_jacobian_values(Z, lay, G0, Gj).reshape(-1))
    f = lambda z: po.exp_residual(z.reshape(lay.N, lay.z_dim), lay, G0, Gj).reshape(-1)
    Jfd = _fd_jac(f, Z.reshape(-1))
    assert np.abs(J - Jfd).max() < 1e-8


# ---- objectives: known answers of the reference's own tests, then finite differences ---------------------------------
def test_fidelity_literals_of_the_reference():
    """[REF src/quantum/dynamics.jl:1305-1315] unitary_fidelity(X, X) = 1, (X, Z) = 0, the subspace restriction;
    [REF src/quantum/operators/embedded_operators.jl:627-633] get_subspace_indices literals (1-based there)."""
    X, Zp, I2 = po.PAULIS["X"], po.PAULIS["Z"], np.eye(2)
    xv = po.operator_to_iso_vec(X)
    assert abs(po.unitary_fidelity_loss(xv, X) - 1.0) < 1e-15
    assert abs(po.unitary_fidelity_loss(xv, Zp)) < 1e-15
    # global-phase invariance, F(U, U) = 1 for a random unitary, 0 <= F <= 1
    rng = np.random.default_rng(1)
    for lv in (2, 3, 4, 9):
        U = np.linalg.qr(rng.standard_normal((lv, lv)) + 1j * rng.standard_normal((lv, lv)))[0]
        V = np.linalg.qr(rng.standard_normal((lv, lv)) + 1j * rng.standard_normal((lv, lv)))[0]
        uv = po.operator_to_iso_vec(U)
        assert abs(po.unitary_fidelity_loss(uv, U) - 1.0) < 1e-13
        assert abs(po.unitary_fidelity_loss(po.operator_to_iso_vec(np.exp(0.7j) * U), U) - 1.0) < 1e-13
        F = po.unitary_fidelity_loss(uv, V)
        assert 0.0 <= F <= 1.0
        assert abs(F - abs(np.trace(V.conj().T @ U)) ** 2 / lv**2) < 1e-14
    # hand-computed 2x2: U = diag(1, i), goal = I: tr = 1 + i, |tr|^2 = 2, F = 2/4
    assert abs(po.unitary_fidelity_loss(po.operator_to_iso_vec(np.diag([1, 1j])), I2) - 0.5) < 1e-15
    # subspace literals
    assert [i + 1 for i in po.get_subspace_indices([[0, 1], [0, 1]], [3, 3])] == [1, 2, 4, 5]
    assert [i + 1 for i in po.get_subspace_indices([[1], [1]], [3, 3])] == [5]
    assert [i + 1 for i in po.get_subspace_indices([[1], [0, 1]], [3, 3])] == [4, 5]
    # a 3-level embedding of X scored on the qubit subspace: the embedded-goal formula gives 1 for the goal itself,
    # whatever the leakage block holds, and reduces to (n + |tr M|^2) / (n (n + 1)) for a unitary subspace block
    U3 = np.zeros((3, 3), dtype=complex)
    U3[:2, :2] = X
    U3[2, 2] = np.exp(0.3j)
    goal = po.embed(X, [0, 1], 3)
    assert np.array_equal(po.unembed(goal, [0, 1]), X.astype(complex))
    assert abs(po.unitary_fidelity_loss(po.operator_to_iso_vec(U3), goal, [0, 1]) - 1.0) < 1e-15
    U3b = np.zeros((3, 3), dtype=complex)
    U3b[:2, :2] = np.diag([1, 1j])  # M = X' diag(1, i): tr M = 0, tr M'M = 2 -> F = 2 / 6
    U3b[2, 2] = 1.0
    assert abs(po.unitary_fidelity_loss(po.operator_to_iso_vec(U3b), goal, [0, 1]) - 1.0 / 3.0) < 1e-15


@pytest.mark.parametrize("sub", [None, [0, 1, 3, 4]])
def test_objective_gradients_finite_differences(sub):
    rng = np.random.default_rng(7)
    lv = 9 if sub else 4
    U = np.linalg.qr(rng.standard_normal((lv, lv)) + 1j * rng.standard_normal((lv, lv)))[0]
    U = U + 0.05 * (rng.standard_normal((lv, lv)) + 1j * rng.standard_normal((lv, lv)))
    ns = len(sub) if sub else lv
    Gs = np.linalg.qr(rng.standard_normal((ns, ns)) + 1j * rng.standard_normal((ns, ns)))[0]
    goal = po.embed(Gs, sub, lv) if sub else Gs
    x = po.operator_to_iso_vec(U)
    g = po.unitary_infidelity_gradient(x, goal, 3.0, sub)
    fd = np.array([(po.unitary_infidelity(x + e, goal, 3.0, sub) - po.unitary_infidelity(x - e, goal, 3.0, sub)) / 2e-6 for e in 1e-6 * np.eye(x.size)])
    assert np.abs(g - fd).max() < 1e-8
    # regulariser + weighted ensemble sum
    Z = rng.standard_normal((5, 12))
    Z[:, 3] = 0.1 + 0.1 * rng.random(5)
    for pw in (0, 1, 2):
        gz = po.quadratic_regularizer_gradient(Z, 5, 4, [1.0, 2.0, 0.5, 0.0], 3, pw)
        f = lambda z: po.quadratic_regularizer(z.reshape(5, 12), 5, 4, [1.0, 2.0, 0.5, 0.0], 3, pw)
        fdz = np.array([(f(Z.reshape(-1) + e) - f(Z.reshape(-1) - e)) / 2e-6 for e in 1e-6 * np.eye(60)])
        assert np.abs(gz.reshape(-1) - fdz).max() < 1e-8
    assert po.quadratic_regularizer(Z, 5, 4, 0.0, 3) == 0.0  # [REF spline_pulse_problem.jl:1570-1578]: R = 0 -> exactly zero


@pytest.mark.parametrize("order", [2, 4, 6, 8, 10])
def test_oracle_general_order_hessian(order):
    """The general-order Hessian of the Lagrangian: equal to the order-4 closed form at p = 4, symmetric, and equal to central
    differences of J^T mu of the (Frechet-pinned) analytic Jacobian at every order."""
    s = po.config_system(1)
    Z, lay = po.synthetic_trajectory(s, 4, seed=4, noise=1e-2)
    Z[:, lay.dt_off] = 0.3 + 0.2 * np.random.default_rng(1).random(4)  # large steps: the high-order terms matter
    G0, Gj = s.G_drift, np.array(s.G_drives)
    mu = np.random.default_rng(2).standard_normal((lay.K, lay.x_dim))
    vals = po.pade_hessian_values(Z, mu, lay, G0, Gj, order)
    if order == 4:
        assert np.abs(vals - po.pade4_hessian_values(Z, mu, lay, G0, Gj)).max() < 1e-13
    Hd = po.hessian_dense(vals, lay)
    g = lambda z: po.pade_jacobian_dense(z.reshape(Z.shape), lay, G0, Gj, order).T @ mu.reshape(-1)
    Hfd = _fd_jac(g, Z.reshape(-1).copy())
    assert np.abs(Hd - Hfd).max() < 2e-7
    assert np.abs(Hd - Hd.T).max() == 0


def test_ket_and_density_loss_literals_of_the_reference():
    """The terminal losses of the other state types, on the literals of the reference's own tests: coherent ket fidelity with weights
    [REF src/control/objectives.jl:636-667: |0.9 + 0.1/2|^2 = 0.9025, |0.1 + 0.9/2|^2 = 0.3025, only ratios matter, uniform weights are the
    unweighted value bit for bit, also where 1/n is not exact], the objective values 100 (1 - F) [REF :585-603], F = 1 on the goals
    [REF :616-634]; ket fidelity [REF :24-27]; density losses on pure states [REF :387-424]."""
    k = po.ket_to_iso
    psi0, psi1 = np.array([1.0, 0.0], complex), np.array([0.0, 1.0], complex)
    goals = [psi1, psi0]
    xs = [k(psi1), k(0.5 * psi0)]
    assert abs(po.coherent_ket_fidelity(xs, goals, [0.9, 0.1]) - 0.9025) < 1e-15
    assert abs(po.coherent_ket_fidelity(xs, goals, [0.1, 0.9]) - 0.3025) < 1e-15
    assert abs(po.coherent_ket_fidelity(xs, goals, [9.0, 1.0]) - po.coherent_ket_fidelity(xs, goals, [0.9, 0.1])) < 1e-15
    assert abs(100.0 * abs(1 - po.coherent_ket_fidelity(xs, goals, [0.9, 0.1])) - 100.0 * (1 - 0.9025)) < 1e-12
    xs3, goals3 = [k(psi1), k(0.5 * psi0), k(0.25 * psi1)], [psi1, psi0, psi1]
    plain = po.coherent_ket_fidelity(xs3, goals3)
    assert po.coherent_ket_fidelity(xs3, goals3, None) == plain == po.coherent_ket_fidelity(xs3, goals3, [1 / 3] * 3) == po.coherent_ket_fidelity(xs3, goals3, [1.0] * 3)
    assert abs(po.coherent_ket_fidelity([k(psi1), k(psi0)], goals) - 1.0) < 1e-15
    assert po.ket_fidelity_loss(k(psi1), psi1) == 1.0 and po.ket_fidelity_loss(k(psi0), psi1) == 0.0
    plus = np.array([1.0, 1.0j]) / np.sqrt(2)
    assert abs(po.ket_fidelity_loss(k(psi0), plus) - 0.5) < 1e-15 and abs(po.ket_fidelity_loss(k(np.exp(0.7j) * plus), plus) - 1.0) < 1e-15
    rho = np.outer(plus, plus.conj())
    x = po.density_to_compact_iso(rho)
    assert po.density_matrix_pure_state_infidelity_loss(x, plus) < 1e-15 and abs(po.density_matrix_pure_state_infidelity_loss(x, psi0) - 0.5) < 1e-15
    assert abs(po.density_matrix_infidelity_loss(x, np.outer(psi1, psi1.conj())) - 0.5) < 1e-15


def test_terminal_loss_forms_of_the_host_mirror():
    """Every terminal loss of the reference is Q |1 - F|, F = c'x + sum_r (A_r'x)^2: the rows and linear parts piccolo.jl_amd/objectives.py
    hands to pcl_set_goal_form reproduce the oracle's restatement of the reference's formulas on random states (no GPU involved)."""
    import piccolo_jl_amd as pa

    rng = np.random.default_rng(8)
    d = 5
    F_of = lambda A, c, x: (0.0 if c is None else c @ x) + (0.0 if A is None else ((A @ x) ** 2).sum())
    g = rng.standard_normal(d) + 1j * rng.standard_normal(d)
    g /= np.linalg.norm(g)
    scope, A, c = pa.KetInfidelityObjective(g, "ψ̃").form(2 * d, 1)
    for _ in range(3):
        x = rng.standard_normal(2 * d)
        assert scope == 0 and abs(F_of(A, c, x) - po.ket_fidelity_loss(x, g)) < 1e-13
    goals = [rng.standard_normal(d) + 1j * rng.standard_normal(d) for _ in range(3)]
    for w in (None, [1.0, 1.0, 1.0], [0.9, 0.1, 0.4]):
        scope, A, c = pa.CoherentKetInfidelityObjective(goals, ["a", "b", "c"], weights=w).form(2 * d, 3)
        xs = [rng.standard_normal(2 * d) for _ in range(3)]
        assert scope == 1 and abs(F_of(A, c, np.concatenate(xs)) - po.coherent_ket_fidelity(xs, goals, w)) < 1e-12
    n = 3
    M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    rho_goal = M @ M.conj().T
    rho_goal /= np.trace(rho_goal).real
    scope, A, c = pa.DensityMatrixInfidelityObjective("ρ", rho_goal).form(n * n, 1)
    x = rng.standard_normal(n * n)
    assert A is None and abs(abs(1 - F_of(A, c, x)) - po.density_matrix_infidelity_loss(x, rho_goal)) < 1e-13
    psi = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    scope, A, c = pa.DensityMatrixPureStateInfidelityObjective("ρ", psi).form(n * n, 1)
    assert abs(abs(1 - F_of(A, c, x)) - po.density_matrix_pure_state_infidelity_loss(x, psi)) < 1e-12


def test_reference_algorithm_port_expv_with_forward_mode_duals():
    """oracle/expv_ref.c -- what bench.py reports as cpu_baseline.reference_algorithm: the reference's constraint delta = x_{k+1} - expv(dt, Ghat(u), x_k)
    [REF docs/src/concepts/index.md:21; src/control/integrators.jl:48] by Al-Mohy & Higham's truncated Taylor action (ExponentialAction.jl's algorithm) and
    its Jacobian by forward-mode dual numbers pushed through expv in chunks, as ForwardDiff does [REF integrators.jl:282-285].  Against scipy's expm /
    expm_frechet (the oracle's exp_jacobian_values, the function the Pade Jacobian is pinned on) to 1e-13, every chunk size, entries outside the block
    structure exactly zero, a sub-range of intervals leaves the others untouched."""
    from oracle import ref_lib

    so = po.config_system(2)
    Z, lay = po.synthetic_trajectory(so, 7, seed=11)
    Z[:, lay.dt_off] = 0.05 + 0.1 * np.random.default_rng(2).random(7)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    d_ref, j_ref = po.exp_residual(Z, lay, G0, Gj), po.exp_jacobian_values(Z, lay, G0, Gj)
    for chunk in (1, 5, 12, 64):
        dl, jc, info = ref_lib.expv_eval_jac(Z, lay, G0, Gj, nthreads=2, chunk=chunk)
        assert np.abs(dl - d_ref).max() < 1e-13 and np.abs(jc - j_ref).max() < 1e-13 and info["off_structure_max"] == 0.0, chunk
        assert info["taylor_terms"] >= lay.K * 5
    dl, jc, _ = ref_lib.expv_eval_jac(Z, lay, G0, Gj, nthreads=1, k_first=2, k_count=3)
    assert np.isnan(jc[:2]).all() and np.isnan(jc[5:]).all() and np.abs(jc[2:5] - j_ref[2:5]).max() < 1e-13 and np.abs(dl[2:5] - d_ref[2:5]).max() < 1e-13
    # BASELINE config 3, one interval of a longer time step (|dt G|_1 ~ 2: two scaling steps)
    so3 = po.config_system(3)
    Z3, lay3 = po.synthetic_trajectory(so3, 3, seed=5)
    Z3[:, lay3.dt_off] = 0.2
    G03, Gj3 = so3.G_drift, np.array(so3.G_drives)
    dl, jc, info = ref_lib.expv_eval_jac(Z3, lay3, G03, Gj3, nthreads=4, k_first=0, k_count=1)
    assert np.abs(jc[0] - po.exp_jacobian_values(Z3, lay3, G03, Gj3)[0]).max() < 1e-12 and np.abs(dl[0] - po.exp_residual(Z3, lay3, G03, Gj3)[0]).max() < 1e-13
