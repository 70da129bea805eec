"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the
C ABI, against the CPU oracle on identical inputs.

Tolerance: fp64 throughout.  GPU and oracle evaluate the same polynomial with different
summation orders (MFMA 4-wide k-blocks / FMA contraction), so agreement is to rounding:
    |gpu - oracle| <= TOL * max(1, max|oracle|),  TOL = 1e-12
(observed ~1e-15 .. 1e-14).  The Pade-vs-exp deviation is a modelling difference, not an
error of the kernel, and is asserted separately per order in tests/test_oracle_pins.py.
"""
import os

import numpy as np
import pytest

import piccolo_jl_amd as pa
from helpers import ref_case, traj_from_Z
from oracle import pade_oracle as po
from oracle import ref_lib

pytestmark = pytest.mark.gpu
TOL = 1e-12


def close(a, b, tol=TOL):
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= tol * max(1.0, np.abs(b).max() if b.size else 0.0), err
    return err


def product_system(cfg):
    if cfg == 1:
        return pa.QuantumSystem(0.5 * pa.PAULIS["Z"], [pa.PAULIS["X"], pa.PAULIS["Y"]], [1.0, 1.0])
    if cfg == 2:
        return pa.MultiTransmonSystem([4.0, 4.1], [0.2, 0.2], [[0, 0.1], [0.1, 0]], levels_per_transmon=2, drive_bounds=0.1)
    return pa.MultiTransmonSystem([4.0, 4.1, 4.2], [0.2, 0.21, 0.22], [[0, 0.01, 0.02], [0.01, 0, 0.03], [0.02, 0.03, 0]], drive_bounds=0.1)


def _auto_large_launch(c):
    """What `auto` picks at order 4 for sparse exact-iso generators: the pattern-compiled kernel 4 (the matrix-core kernel 3 keeps the
    payload-fused call and contexts that ask for it)."""
    assert c.get_option("last_kernel") == 42


def make_ctx(lay, G0, Gj, **kw):
    args = dict(d=lay.d, m=lay.m, N=lay.N, z_dim=lay.z_dim, u_off=lay.u_off, dt_off=lay.dt_off, x_offs=[lay.x_off], G0=G0,
                Gj=Gj, batch=1, batch_mode=pa._lib.PCL_BATCH_MEMBERS)  # fmt: skip
    host_path = kw.pop("host_path", 1)  # 1: full values over PCIe (the launch under test writes every replicated block)
    args.update(kw)
    c = pa.integrators._PclContext(**args)
    c.set_option("host_path", host_path)
    return c


# ---- committed golden vectors ----------------------------------------------------------------
# kernel variants: (kernel_version, use_mfma).  (3,1) is the default: one persistent, wave-specialised,
# software-pipelined workgroup per CU; (2,1) persistent, 2 workgroups per CU (fallback); (1,1) the
# single-role MFMA kernel, (1,0) the plain-VALU kernel.
# (0, 1) is `auto`: what ships -- the pattern-compiled kernel 4 at config 3, the small-system kernel at configs 1 and 2.
VARIANTS = [(2, 1), (3, 1), (1, 1), (1, 0), (0, 1)]


def set_variant(c, variant):
    c.set_option("kernel_version", variant[0])
    c.set_option("use_mfma", variant[1])


@pytest.mark.parametrize("name", ["config1", "config2", "config3"])
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("host_path", [1, 2])  # full values over PCIe | the default: compact values + host expansion
def test_golden_vectors(name, variant, host_path, golden, golden_meta):
    v = golden("vec_" + name)
    m = golden_meta["oracle_vectors"][name]
    lay = po.Layout(d=m["d"], m=m["m"], N=m["N"], z_dim=m["z_dim"], x_off=m["x_off"], u_off=m["u_off"], dt_off=m["dt_off"])
    c = make_ctx(lay, v["G0"], v["Gj"], host_path=host_path)
    set_variant(c, variant)
    assert c.get_option("iso_structured") == 1
    delta, vals = c.eval_jac(v["Z"])
    close(delta, v["delta"])
    close(vals, v["jac"])
    close(c.eval(v["Z"]), v["delta"])
    close(c.jac(v["Z"]), v["jac"])
    close(c.hess(v["Z"], v["mu"]), v["hess"], 1e-11)
    c.close()


@pytest.mark.parametrize("name", ["config1", "config2", "config3"])
@pytest.mark.parametrize("order", [8, 10])
@pytest.mark.parametrize("host_path", [1, 2])
def test_golden_vectors_orders_8_and_10(name, order, host_path, golden, golden_meta):
    """The committed order-8 and order-10 vectors (the orders that reach the reference's exp constraint at config 3) through the kernels
    `auto` launches: pattern-compiled residual + Jacobian 44 / 45 and Hessian 84 / 85 at config 3, the small-system kernel at 1 and 2."""
    v = golden("vec_" + name)
    m = golden_meta["oracle_vectors"][name]
    lay = po.Layout(d=m["d"], m=m["m"], N=m["N"], z_dim=m["z_dim"], x_off=m["x_off"], u_off=m["u_off"], dt_off=m["dt_off"])
    c = make_ctx(lay, v["G0"], v["Gj"], host_path=host_path, pade_order=order)
    delta, vals = c.eval_jac(v["Z"])
    if name == "config3":
        assert c.get_option("last_kernel") == 40 + order // 2
    close(delta, v["delta%d" % order], 1e-12)
    close(vals, v["jac%d" % order], 1e-12)
    close(c.eval(v["Z"]), v["delta%d" % order], 1e-12)
    close(c.jac(v["Z"]), v["jac%d" % order], 1e-12)
    close(c.hess(v["Z"], v["mu"]), v["hess%d" % order], 1e-11)
    if name == "config3":
        assert c.get_option("last_hess_kernel") == 80 + order // 2  # the column-group kernel
        c.set_option("hess_kernel", 7)  # ... and the chain-per-wave kernel it replaced as the `auto` choice
        close(c.hess(v["Z"], v["mu"]), v["hess%d" % order], 1e-11)
        assert c.get_option("last_hess_kernel") == 70 + order // 2
    c.close()


# ---- seeded inputs vs the oracle, every slicing of the state columns ------------------------------
@pytest.mark.parametrize("cfg,N", [(1, 50), (2, 100), (3, 5)])
def test_seeded_vs_oracle_all_slicings(cfg, N):
    so = po.config_system(cfg)
    Z, lay = po.synthetic_trajectory(so, N, seed=20260929 + cfg)
    Z[:, lay.dt_off] = 0.1 + 0.05 * np.random.default_rng(cfg).random(N)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    d_ref, j_ref = ref_lib.eval_jac(Z, lay, G0, Gj)
    c = make_ctx(lay, G0, Gj)
    for variant in VARIANTS:
        set_variant(c, variant)
        for nc in sorted({0, 1, 2, 3, 5, lay.d}):
            if nc > lay.d:
                continue
            c.set_option("cols_per_slice", nc)
            delta, vals = c.eval_jac(Z)
            close(delta, d_ref)
            close(vals, j_ref)
            close(c.eval(Z), d_ref)
    c.close()


def test_structure_matches_oracle():
    so = po.config_system(2)
    lay = po.Layout.smooth_pulse(so.levels, so.n_drives, 6)
    for base in (0, 1):
        c = make_ctx(lay, so.G_drift, np.array(so.G_drives), index_base=base)
        r, cc = c.jac_structure()
        r0, c0 = po.jac_structure(lay, index_base=base)
        assert np.array_equal(r, r0) and np.array_equal(cc, c0)
        r32, c32 = c.jac_structure(np.int32)
        assert np.array_equal(r32, r0) and np.array_equal(c32, c0)
        hr, hc = c.hess_structure()
        hr0, hc0 = po.hess_structure(lay, index_base=base)
        assert np.array_equal(hr, hr0) and np.array_equal(hc, hc0)
        assert (hr >= hc).all()
        assert c.n_rows == lay.x_dim * lay.K and c.n_cols == lay.z_dim * lay.N
        c.close()


# ---- the reference-style interface ------------------------------------------------------------------
def test_integrator_interface_pins():
    """Structural pins of the reference's integrator tests [REF src/control/integrators.jl:306-317,780-783]."""
    s = product_system(2)
    N = 10
    rng = np.random.default_rng(0)
    t = pa.unitary_trajectory(s, 0.02 * rng.standard_normal((4, N)), np.linspace(0, 1, N), pa.GATES["CX"])
    B = pa.BilinearIntegrator(s, t, pade_order=4)
    assert B.x_dim == 32 and B.dim == 32 * (N - 1) and B.x_name == "Ũ⃗" and B.x_names == ["Ũ⃗"]
    delta = np.zeros(B.dim)
    pa.evaluate_(delta, B, t)
    assert np.isfinite(delta).all() and np.linalg.norm(delta) > 0
    J = pa.eval_jacobian(B, t)
    assert J.shape == (B.dim, t.dim * t.N + t.global_dim)
    # scalar form f(x_next, x, u, dt) == the matching rows of evaluate!
    k = 3
    f = B.f(t["Ũ⃗"][:, k + 1], t["Ũ⃗"][:, k], t["u"][:, k], t["Δt"][0, k])
    close(f, delta[k * 32 : (k + 1) * 32])
    with pytest.raises(ValueError):
        pa.evaluate_(np.zeros(3), B, t)
    B.close()


def test_test_integrator_style_finite_differences():
    """DTO's ``test_integrator(B, traj; atol=1e-3)`` is an analytic-vs-finite-difference check of the
    Jacobian and the Hessian of the Lagrangian [REF integrators.jl:341-359]; same check, tighter."""
    s = product_system(1)
    N = 6
    rng = np.random.default_rng(5)
    states = [np.linalg.qr(rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2)))[0] for _ in range(N)]
    t = pa.unitary_trajectory(s, 0.3 * rng.standard_normal((2, N)), np.cumsum(0.1 + 0.05 * rng.random(N)), pa.GATES["X"], states=states)
    B = pa.BilinearIntegrator(s, t, pade_order=4)
    J = pa.eval_jacobian(B, t).toarray()
    z0 = t.datavec.copy()

    def f(z):
        t.update(z)
        return pa.evaluate_(np.zeros(B.dim), B, t).copy()

    eps = 1e-6
    Jfd = np.zeros_like(J)
    for i in range(z0.size):
        zp, zm = z0.copy(), z0.copy()
        zp[i] += eps
        zm[i] -= eps
        Jfd[:, i] = (f(zp) - f(zm)) / (2 * eps)
    t.update(z0)
    assert np.abs(J - Jfd).max() < 1e-8
    mu = rng.standard_normal(B.dim)
    Hm = pa.eval_hessian_of_lagrangian(B, t, mu).toarray()

    def g(z):
        t.update(z)
        return pa.eval_jacobian(B, t).T @ mu

    Hfd = np.zeros_like(Hm)
    for i in range(z0.size):
        zp, zm = z0.copy(), z0.copy()
        zp[i] += eps
        zm[i] -= eps
        Hfd[:, i] = (g(zp) - g(zm)) / (2 * eps)
    t.update(z0)
    assert np.abs(Hm - Hfd).max() < 1e-7
    B.close()


# ---- trajectories solved by the reference itself -----------------------------------------------------------
@pytest.mark.parametrize("name", ["two_qubit_zoh", "multilevel_transmon", "first_gate"])
def test_reference_solved_trajectories(name, golden, golden_meta):
    systems, lay, _ = ref_case(name, golden_meta)
    Z = golden("ref_" + name)["Z"]
    so = systems[0]
    G0, Gj = so.G_drift, np.array(so.G_drives)
    c = make_ctx(lay, G0, Gj)
    delta, vals = c.eval_jac(Z)
    close(delta, po.pade_residual(Z, lay, G0, Gj, 4))
    close(vals, po.pade_jacobian_values(Z, lay, G0, Gj, 4))
    if name == "two_qubit_zoh":  # Pade-4 residual of a converged exp-solution ~1.5e-9
        assert 1e-9 < np.abs(delta).max() < 3e-9
    c.close()


@pytest.mark.parametrize("name", ["sampling_robust", "robust_sampling"])
def test_reference_solved_ensembles(name, golden, golden_meta):
    """SamplingTrajectory layout [U1..UM, dt, t, u], per-member drift, shared controls, free dt, on the reference's own
    solved ensembles -- through the boundary the reference defines: ``BilinearIntegrator(qtraj::SamplingTrajectory, N)``
    returns ONE INTEGRATOR PER MEMBER [REF src/control/integrators.jl:134-146] (``SamplingProblem`` rejects anything else,
    [REF sampling_problem.jl:195-223]); each has dim = x_dim (N-1), its own x_name and rows numbered inside its block, and
    the problem concatenates them in member order [REF integrators.jl:316-317]."""
    systems, lay, x_offs = ref_case(name, golden_meta)
    Z = golden("ref_" + name)["Z"]
    M = len(systems)
    psys = [pa.QuantumSystem(s.H_drift, s.H_drives, [1.0, 1.0]) for s in systems]
    traj = traj_from_Z(pa, Z, lay, n_members=M)
    Bs = pa.BilinearIntegrator(psys, traj, pade_order=4)
    assert isinstance(Bs, list) and len(Bs) == M
    core = Bs[0].ensemble
    per_d, per = lay.x_dim * lay.K, po.jac_nnz_per_interval(lay) * lay.K
    hper = po.hess_nnz_per_interval(lay) * lay.K
    mu = np.random.default_rng(3).standard_normal(M * per_d)
    deltas, jvals = [], []
    for i, (B, s, xo) in enumerate(zip(Bs, systems, x_offs)):
        G0, Gj = s.G_drift, np.array(s.G_drives)
        assert B.dim == per_d and B.x_dim == lay.x_dim and B.x_name == "Ũ⃗%d" % (i + 1) and B.ensemble is core
        delta = pa.evaluate_(np.zeros(B.dim), B, traj)
        close(delta, po.pade_residual(Z, lay, G0, Gj, 4, x_off=xo))
        J = pa.eval_jacobian(B, traj)
        assert J.shape == (B.dim, traj.dim * traj.N + traj.global_dim)  # [REF integrators.jl:780-783]
        vals = B.ctx.jac(traj.datavec)
        close(vals, po.pade_jacobian_values(Z, lay, G0, Gj, 4, x_off=xo))
        rows, cols = pa.jacobian_structure(B)
        r0, c0 = po.jac_structure(lay, x_off=xo)
        assert np.array_equal(rows, r0) and np.array_equal(cols, c0)  # rows inside the member's own block
        h0 = po.pade4_hessian_values(Z, mu[i * per_d : (i + 1) * per_d].reshape(lay.K, -1), lay, G0, Gj, x_off=xo)
        close(B.ctx.hess(traj.datavec, mu[i * per_d : (i + 1) * per_d]), h0, 1e-11)
        hr, hc = pa.hessian_structure(B)
        hr0, hc0 = po.hess_structure(lay, x_off=xo)
        assert np.array_equal(hr, hr0) and np.array_equal(hc, hc0)
        # B.f: the scalar one-interval form of THIS member's system [REF integrators.jl:518-525]
        k = 3
        fk = B.f(Z[k + 1, xo : xo + lay.x_dim], Z[k, xo : xo + lay.x_dim], Z[k, lay.u_off : lay.u_off + lay.m], Z[k, lay.dt_off])
        close(fk, delta[k * lay.x_dim : (k + 1) * lay.x_dim])
        deltas.append(delta)
        jvals.append(vals)
    assert core.launches == 2  # ONE fused residual launch and ONE fused residual+Jacobian launch served all M members
    # a changed trajectory invalidates the cache
    Z2 = Z.copy()
    Z2[:, lay.u_off] += 1e-3
    traj2 = traj_from_Z(pa, Z2, lay, n_members=M)
    d2 = pa.evaluate_(np.zeros(per_d), Bs[-1], traj2)
    close(d2, po.pade_residual(Z2, lay, systems[-1].G_drift, np.array(systems[-1].G_drives), 4, x_off=x_offs[-1]))
    assert core.launches == 3
    # the members' blocks in order are the fused context's output (what the device-resident path hands to the solver)
    fd, fv = core.ctx.eval_jac(traj.datavec)
    assert np.array_equal(fd, np.concatenate(deltas)) and np.array_equal(fv, np.concatenate(jvals))
    # member windows through the C ABI give the same numbers as the slices
    core.ctx.set_member_window(1, M - 1)
    wd, wv = core.ctx.eval_jac(traj.datavec)
    assert np.array_equal(wd, fd[per_d:]) and np.array_equal(wv, fv[per:])
    core.ctx.set_member_window(0, M)
    with pytest.raises(pa.PclError):
        core.ctx.set_member_window(1, M)
    for B in Bs:
        B.close()


def test_ensemble_with_per_member_drive_generators():
    """Members that differ in their DRIVE generators too (each member uses its full sys.G, [REF integrators.jl:149-162]):
    one integrator per member, each with its own context."""
    rng = np.random.default_rng(8)
    M, N, xd = 3, 6, 8
    lay = po.Layout(d=2, m=2, N=N, z_dim=M * xd + 2 + 2, x_off=0, u_off=M * xd + 2, dt_off=M * xd)
    Z = 0.3 * rng.standard_normal((N, lay.z_dim))
    Z[:, lay.dt_off] = 0.1 + 0.1 * rng.random(N)
    osys = [po.quantum_system(0.5 * po.PAULIS["Z"], [(1 + 0.02 * i) * po.PAULIS["X"], po.PAULIS["Y"]], [1.0, 1.0]) for i in range(M)]
    psys = [pa.QuantumSystem(s.H_drift, s.H_drives, [1.0, 1.0]) for s in osys]
    traj = traj_from_Z(pa, Z, lay, n_members=M)
    Bs = pa.BilinearIntegrator(psys, traj, pade_order=4)
    assert isinstance(Bs, list) and len(Bs) == M
    for i, (B, s) in enumerate(zip(Bs, osys)):
        assert B.x_name == "Ũ⃗%d" % (i + 1) and B.dim == lay.x_dim * lay.K
        close(pa.evaluate_(np.zeros(B.dim), B, traj), po.pade_residual(Z, lay, s.G_drift, np.array(s.G_drives), 4, x_off=i * xd))
        close(B.ctx.jac(traj.datavec), po.pade_jacobian_values(Z, lay, s.G_drift, np.array(s.G_drives), 4, x_off=i * xd))
        B.close()


# ---- BASELINE.json's full sizes ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config3_full():
    so = po.config_system(3)
    Z, lay = po.synthetic_trajectory(so, 100, seed=20260929 + 3)
    return so, Z, lay


def test_config3_full_size_vs_c_oracle(config3_full):
    so, Z, lay = config3_full
    G0, Gj = so.G_drift, np.array(so.G_drives)
    d_ref, j_ref = ref_lib.eval_jac(Z, lay, G0, Gj)
    c = make_ctx(lay, G0, Gj)
    assert c.jac_per == 167670 and c.jac_nnz == 16599330 and c.hess_per == 20440
    for variant in VARIANTS:
        set_variant(c, variant)
        for spec in (1, 0):  # shape-specialised (d=27, m=6) and run-time-shape instances of the same kernel
            c.set_option("specialize", spec)
            delta, vals = c.eval_jac(Z)
            close(delta, d_ref)
            close(vals, j_ref)
    mu = np.random.default_rng(9).standard_normal((lay.K, lay.x_dim))
    close(c.hess(Z, mu), ref_lib.hess(Z, mu, lay, G0, Gj), 1e-11)
    c.close()


def test_config3_size_independent_properties(config3_full):
    """Properties that hold at any size: the d diagonal blocks are identical copies; B^- - (-(-B^+)) = ... ;
    the residual is linear in the states; Jacobian * state-part reproduces the residual (delta is linear in X)."""
    so, Z, lay = config3_full
    G0, Gj = so.G_drift, np.array(so.G_drives)
    c = make_ctx(lay, G0, Gj)
    delta, vals = c.eval_jac(Z)
    d, n, K, xd = lay.d, lay.n, lay.K, lay.x_dim
    V = vals.reshape(K, -1)
    blk = V[:, : 2 * d * n * n].reshape(K, 2, d, n * n)
    assert np.array_equal(blk, np.broadcast_to(blk[:, :, :1], blk.shape))  # bit-identical replicas
    Bp, Bm = -blk[:, 0, 0].reshape(K, n, n).transpose(0, 2, 1), blk[:, 1, 0].reshape(K, n, n).transpose(0, 2, 1)
    # B^+ + B^- = 2 (I + h^2/12 G^2),  B^+ - B^- = h G : skew part of a Hermitian system
    hG = Bp - Bm
    assert np.abs(hG + hG.transpose(0, 2, 1)).max() < 1e-14
    assert np.abs(Bm.transpose(0, 2, 1) - Bp).max() < 1e-13  # SURVEY: G^T = -G  =>  (B^-)^T = B^+
    # delta is linear in X: delta_k = B^- X_{k+1} - B^+ X_k
    X = Z[:, :xd].reshape(lay.N, d, n).transpose(0, 2, 1)
    recon = np.einsum("kij,kjc->kic", Bm, X[1:]) - np.einsum("kij,kjc->kic", Bp, X[:-1])
    close(delta.reshape(K, d, n).transpose(0, 2, 1), recon)
    # scaling all states by a scales delta by a (bitwise for a power of two), on both entry points
    Z2 = Z.copy()
    Z2[:, :xd] *= 2.0
    close(c.eval(Z2), 2.0 * c.eval(Z), 0.0)
    close(c.eval_jac(Z2)[0], 2.0 * delta, 0.0)
    close(c.eval(Z), delta)  # evaluate! alone forms G(GD); the fused call forms (G^2)D: equal to rounding
    c.close()


def test_compact_and_expand(config3_full):
    import torch

    so, Z, lay = config3_full
    G0, Gj = so.G_drift, np.array(so.G_drives)
    c = make_ctx(lay, G0, Gj)
    delta, vals = c.eval_jac(Z)
    Zd = torch.from_numpy(np.ascontiguousarray(Z)).cuda()
    dd = torch.zeros(c.n_rows, dtype=torch.float64, device="cuda")
    cd = torch.zeros(c.compact_nnz, dtype=torch.float64, device="cuda")
    fd = torch.zeros(c.jac_nnz, dtype=torch.float64, device="cuda")
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.eval_jac_compact_dev(Zd, dd, cd)
    c.jac_expand_dev(cd, fd)
    torch.cuda.synchronize()
    assert c.compact_per == 2 * lay.n**2 + lay.x_dim * (lay.m + 1)
    # the compact producer and the default fused kernel are different code paths: equal to rounding
    close(dd.cpu().numpy(), delta)
    close(fd.cpu().numpy(), vals)
    # the expansion itself is exact replication of the unique tiles and a verbatim copy of the tail
    nn, d, K = lay.n**2, lay.d, lay.K
    F = fd.cpu().numpy().reshape(K, -1)
    Cc = cd.cpu().numpy().reshape(K, -1)
    assert np.array_equal(F[:, : 2 * d * nn].reshape(K, 2, d, nn), np.broadcast_to(Cc[:, : 2 * nn].reshape(K, 2, 1, nn), (K, 2, d, nn)))
    assert np.array_equal(F[:, 2 * d * nn :], Cc[:, 2 * nn :])
    c.close()


def test_multistart_batch_matches_single():
    """BASELINE config 5: B independent seeds in one launch == B single evaluations."""
    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Bn, N = 3, 6
    Zs, lay = [], None
    for s in range(Bn):
        Z, lay = po.synthetic_trajectory(so, N, seed=1000 + s)
        Zs.append(Z)
    t = traj_from_Z(pa, Zs[0], lay)
    ms = pa.HipPadeMultistart(G0, Gj, t, Bn, pade_order=4)
    delta, vals = ms.ctx.eval_jac(np.stack(Zs))
    per_d, per_j = lay.x_dim * lay.K, po.jac_nnz_per_interval(lay) * lay.K
    for s in range(Bn):
        d_ref, j_ref = ref_lib.eval_jac(Zs[s], lay, G0, Gj)
        close(delta[s * per_d : (s + 1) * per_d], d_ref)
        close(vals[s * per_j : (s + 1) * per_j], j_ref)
    r, cc = ms.ctx.jac_structure()
    r0, c0 = po.jac_structure(lay)
    assert np.array_equal(r[per_j : 2 * per_j], r0 + per_d) and np.array_equal(cc[per_j : 2 * per_j], c0 + lay.z_dim * lay.N)
    ms.close()


# ---- edge cases ------------------------------------------------------------------------------------------
def _random_case(d, m, N, rng, x_off=0, pad=3):
    n = 2 * d
    xd = 2 * d * d
    z_dim = x_off + xd + pad + m + 1
    lay = po.Layout(d=d, m=m, N=N, z_dim=z_dim, x_off=x_off, u_off=x_off + xd + 1, dt_off=x_off + xd)
    G0 = rng.standard_normal((n, n))
    Gj = rng.standard_normal((m, n, n)) * (rng.random((m, n, n)) < 0.3) if m else np.zeros((0, n, n))
    Z = rng.standard_normal((N, z_dim))
    Z[:, lay.dt_off] = 0.05 + 0.1 * rng.random(N)
    return lay, G0, Gj, Z


@pytest.mark.parametrize(
    "d,m,N,x_off",
    [(1, 1, 2, 0), (1, 0, 3, 0), (2, 0, 4, 0), (3, 2, 2, 5), (4, 7, 3, 1), (8, 1, 3, 0), (16, 2, 3, 0), (27, 6, 2, 3), (32, 3, 2, 0)],
)
def test_edge_shapes_general_dense_generators(d, m, N, x_off):
    """Minimum sizes (N=2 -> one interval, d=1), no drives, odd/nonzero state offsets, dense non-skew G,
    dense-ish drives, the maximum supported d."""
    rng = np.random.default_rng(1000 * d + 10 * m + N)
    lay, G0, Gj, Z = _random_case(d, m, N, rng, x_off)
    c = make_ctx(lay, G0, Gj)
    d_ref, j_ref = ref_lib.eval_jac(Z, lay, G0, Gj)
    assert c.get_option("iso_structured") == 0  # random dense generators: the general G^2 path
    for variant in VARIANTS:
        set_variant(c, variant)
        delta, vals = c.eval_jac(Z)
        close(delta, d_ref, 1e-11)
        close(vals, j_ref, 1e-11)
        close(c.eval(Z), d_ref, 1e-11)
    mu = rng.standard_normal((lay.K, lay.x_dim))
    close(c.hess(Z, mu), ref_lib.hess(Z, mu, lay, G0, Gj), 1e-10)
    c.close()


def test_unsupported_requests_fail_loudly():
    rng = np.random.default_rng(0)
    lay, G0, Gj, Z = _random_case(2, 1, 3, rng)
    for bad in (3, 5, 12, -2):  # (0 is the order policy: test_order_policy_...)
        with pytest.raises(pa.PclError) as ei:
            make_ctx(lay, G0, Gj, pade_order=bad)
        assert ei.value.code == pa._lib.PCL_ENOTIMPL
    c = make_ctx(lay, G0, Gj)
    with pytest.raises(ValueError):
        c.eval(Z[:-1])
    with pytest.raises(pa.PclError):
        c.set_option("no_such_option", 1)
    c.close()


def test_deterministic_bitwise_repeatability(config3_full):
    so, Z, lay = config3_full
    c = make_ctx(lay, so.G_drift, np.array(so.G_drives))
    a = c.eval_jac(Z)
    b = c.eval_jac(Z)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    c.close()


def test_ensemble_merit_and_shared_gradient_on_device():
    """Config-4 style: per-member drifts, shared controls; merit + shared-control gradient from the device
    buffers (piccolo.jl_amd/distributed.py) against finite differences of the oracle's merit."""
    import torch

    from piccolo_jl_amd import distributed as pd

    rng = np.random.default_rng(4)
    base = po.config_system(2)
    d, m, M, N = base.levels, base.n_drives, 3, 6
    xd = 2 * d * d
    lay = po.Layout(d=d, m=m, N=N, z_dim=M * xd + 2 + m, x_off=0, u_off=M * xd + 2, dt_off=M * xd)
    Z = 0.2 * rng.standard_normal((N, lay.z_dim))
    Z[:, lay.dt_off] = 0.1 + 0.05 * rng.random(N)
    osys = [po.System(base.H_drift + 0.01 * i * np.diag(np.arange(d)).astype(complex), base.H_drives, base.drive_bounds) for i in range(M)]
    psys = [pa.QuantumSystem(s.H_drift, s.H_drives, [b[1] for b in s.drive_bounds]) for s in osys]
    traj = traj_from_Z(pa, Z, lay, n_members=M)
    B = pa.BilinearIntegrator(psys, traj, pade_order=4)[0].ensemble.fused  # the batched context the per-member integrators share
    Zd = torch.from_numpy(traj.datavec).cuda()
    dd = torch.empty(B.dim, dtype=torch.float64, device="cuda")
    vd = torch.empty(B.ctx.jac_nnz, dtype=torch.float64, device="cuda")
    B.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    B.ctx.eval_jac_dev(Zd, dd, vd)
    # the payload kernel of the C ABI (pcl_merit_grad_dev) ...
    ln, sets = B.ctx.merit_grad_len()
    assert (ln, sets) == (1 + lay.K * m + lay.K, 1)
    out = torch.empty(ln, dtype=torch.float64, device="cuda")
    B.ctx.merit_grad_dev(dd, None, vd, out)
    torch.cuda.synchronize()
    phi, gu, gdt = out[0], out[1 : 1 + lay.K * m].view(lay.K, m), out[1 + lay.K * m :]
    # ... equals the torch restatement of distributed.py (used by the gloo tests on CPU)
    phi_t, gu_t, gdt_t = pd.constraint_merit_and_shared_gradient(dd, vd, M, lay.K, d, m)
    assert abs(float(phi) - float(phi_t)) < 1e-12 * max(1.0, float(phi_t))
    close(gu.cpu().numpy(), gu_t.cpu().numpy())
    close(gdt.cpu().numpy(), gdt_t.cpu().numpy())
    # J^T lam for arbitrary multipliers and weights against the oracle's dense Jacobian
    lam = torch.from_numpy(rng.standard_normal(B.dim)).cuda()
    w = 1.0 + 0.1 * np.arange(M)
    B.ctx.set_weights(w)
    out2 = torch.empty(ln, dtype=torch.float64, device="cuda")
    B.ctx.merit_grad_dev(dd, lam, vd, out2)
    torch.cuda.synchronize()
    B.ctx.set_weights(None)
    g_ref = np.zeros(lay.z_dim * lay.N)
    phi_ref = 0.0
    for i, s in enumerate(osys):
        Ji = po.pade_jacobian_dense(Z, lay, s.G_drift, np.array(s.G_drives), 4, x_off=i * xd)
        li = lam[i * lay.x_dim * lay.K : (i + 1) * lay.x_dim * lay.K].cpu().numpy()
        g_ref += w[i] * (Ji.T @ li)
        phi_ref += w[i] * float(li @ po.pade_residual(Z, lay, s.G_drift, np.array(s.G_drives), 4, x_off=i * xd).reshape(-1))
    g_ref = g_ref.reshape(lay.N, lay.z_dim)
    o2 = out2.cpu().numpy()
    assert abs(o2[0] - phi_ref) < 1e-12 * max(1.0, abs(phi_ref))
    close(o2[1 : 1 + lay.K * m].reshape(lay.K, m), g_ref[: lay.K, lay.u_off : lay.u_off + m], 1e-11)
    close(o2[1 + lay.K * m :], g_ref[: lay.K, lay.dt_off], 1e-11)

    def merit(Zp):
        return sum(0.5 * (po.pade_residual(Zp, lay, s.G_drift, np.array(s.G_drives), 4, x_off=i * xd) ** 2).sum() for i, s in enumerate(osys))

    assert abs(float(phi) - merit(Z)) < 1e-12 * max(1.0, merit(Z))
    eps = 1e-6
    for (k, l) in [(0, 0), (3, 2), (lay.K - 1, m - 1)]:
        Zp, Zm = Z.copy(), Z.copy()
        Zp[k, lay.u_off + l] += eps
        Zm[k, lay.u_off + l] -= eps
        assert abs((merit(Zp) - merit(Zm)) / (2 * eps) - float(gu[k, l])) < 1e-6
    Zp, Zm = Z.copy(), Z.copy()
    Zp[2, lay.dt_off] += eps
    Zm[2, lay.dt_off] -= eps
    assert abs((merit(Zp) - merit(Zm)) / (2 * eps) - float(gdt[2])) < 1e-6
    B.close()


def test_profiling_hooks_are_not_in_the_shipped_library():
    """The shipped kernels carry no cycle stamps or ablation switches: `debug_timing` is refused (PCL_ENOTIMPL) and the
    round-1 experiment keys are unknown options (PCL_EINVAL)."""
    so = po.config_system(2)
    Z, lay = po.synthetic_trajectory(so, 6, seed=5)
    c = make_ctx(lay, so.G_drift, np.array(so.G_drives))
    with pytest.raises(pa.PclError) as e:
        c.set_option("debug_timing", 1)
    assert e.value.code == pa._lib.PCL_ENOTIMPL
    for key in ("debug_ablate", "stream_xcds", "stream_dynamic", "stream_piece_cols", "aligned_stream", "copies_per_piece"):
        with pytest.raises(pa.PclError) as e:
            c.set_option(key, 1)
        assert e.value.code == pa._lib.PCL_EINVAL
    for v in (6, 7):
        with pytest.raises(pa.PclError):
            c.set_option("kernel_version", v)
    c.close()


def test_derivative_and_time_consistency_rows(golden, golden_meta):
    """the other members of prob.integrators, on a trajectory solved by the reference
    (residuals ~1e-7 there) and on random data, values + structure against the oracle."""
    systems, lay, _ = ref_case("two_qubit_zoh", golden_meta)
    Z = golden("ref_two_qubit_zoh")["Z"]
    traj = traj_from_Z(pa, Z, lay)
    B = pa.BilinearIntegrator(product_system(2), traj, pade_order=4)
    m = lay.m
    for x, dx in (("u", "du"), ("du", "ddu")):
        D = pa.DerivativeIntegrator(x, dx, traj, like=B)
        assert D.dim == m * lay.K
        delta = pa.evaluate_(np.zeros(D.dim), D, traj)
        ox, odx = traj.components[x].start, traj.components[dx].start
        close(delta, po.derivative_residual(Z, ox, odx, m, lay.dt_off))
        assert np.abs(delta).max() < 1e-6  # the reference solved these rows
        J = pa.eval_jacobian(D, traj)
        r, c, v = po.derivative_jacobian(Z, lay.z_dim, ox, odx, m, lay.dt_off)
        rr, cc = pa.jacobian_structure(D)
        assert np.array_equal(rr, r) and np.array_equal(cc, c)
        close(D.jac(traj.datavec), v)
        assert J.shape == (D.dim, traj.dim * traj.N)
    T = pa.DerivativeIntegrator("t", None, traj, like=B)
    delta = pa.evaluate_(np.zeros(T.dim), T, traj)
    close(delta, po.time_consistency_residual(Z, lay.dt_off + 1, lay.dt_off))
    assert np.abs(delta).max() < 1e-12
    r, c, v = po.derivative_jacobian(Z, lay.z_dim, lay.dt_off + 1, -1, 1, lay.dt_off)
    rr, cc = pa.jacobian_structure(T)
    assert np.array_equal(rr, r) and np.array_equal(cc, c)
    close(T.jac(traj.datavec), v)
    with pytest.raises(pa.PclError):
        B.ctx.deriv_eval_jac(lay.z_dim - 1, -1, 4, traj.datavec)
    B.close()


def test_rccl_reduce_through_the_c_abi_single_rank():
    """pcl_comm_* / pcl_reduce_sum_dev (RCCL, dlopen'ed lazily): with one rank the all-reduce is the identity; errors are loud."""
    import torch

    rng = np.random.default_rng(0)
    lay, G0, Gj, Z = _random_case(2, 1, 3, rng)
    c = make_ctx(lay, G0, Gj)
    buf = torch.arange(700, dtype=torch.float64, device="cuda")
    with pytest.raises(pa.PclError) as ei:
        c.reduce_sum_dev(buf)
    assert ei.value.code == pa._lib.PCL_ERCCL
    uid = c.comm_unique_id()
    assert len(uid) == 128
    c.comm_init(uid, 0, 1)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.reduce_sum_dev(buf)
    torch.cuda.synchronize()
    assert torch.equal(buf.cpu(), torch.arange(700, dtype=torch.float64))
    hb = np.arange(811, dtype=np.float64)  # host-buffer variant (what a Julia host holds): staged, synchronous
    assert np.array_equal(c.reduce_sum(hb), np.arange(811, dtype=np.float64))
    with pytest.raises(pa.PclError):
        c.comm_init(uid, 0, 1)  # already initialised
    c.close()


def test_terminal_infidelity_objective_on_device():
    """Q |1 - |tr(Ug' U_N)|^2/d^2| and its gradient, per seed, against the ORACLE's restatement of
    [REF src/control/objectives.jl:330-337,347-356] (pinned on the reference's literals in tests/test_oracle_pins.py)."""
    import torch

    rng = np.random.default_rng(12)
    so = po.config_system(2)
    d, Bn, N = so.levels, 3, 5
    Zs, lay = [], None
    for s_ in range(Bn):
        Z, lay = po.synthetic_trajectory(so, N, seed=50 + s_, noise=5e-2)
        Zs.append(Z)
    t = traj_from_Z(pa, Zs[0], lay)
    ms = pa.HipPadeMultistart(so.G_drift, np.array(so.G_drives), t, Bn, pade_order=4)
    Ug = po.PAULIS["X"]
    Ug = np.kron(np.diag([1, 0]), np.eye(2)) + np.kron(np.diag([0, 1]), Ug)  # CX
    Ug = Ug @ np.diag(np.exp(1j * rng.random(d)))
    ms.ctx.set_goal(po.operator_to_iso_vec(Ug))
    Zd = torch.from_numpy(np.stack(Zs)).cuda()
    val = torch.zeros(Bn, dtype=torch.float64, device="cuda")
    grad = torch.zeros(Bn * lay.x_dim, dtype=torch.float64, device="cuda")
    ms.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ms.ctx.infidelity_dev(Zd, 100.0, val, grad)
    torch.cuda.synchronize()
    for s_ in range(Bn):
        x = Zs[s_][N - 1, : lay.x_dim]
        f = po.unitary_infidelity(x, Ug, 100.0)
        assert abs(val[s_].item() - f) < 1e-12 * max(1.0, f)
        close(grad[s_ * lay.x_dim : (s_ + 1) * lay.x_dim].cpu().numpy(), po.unitary_infidelity_gradient(x, Ug, 100.0))
    with pytest.raises(ValueError):
        ms.ctx.set_goal(np.zeros(3))
    ms.close()


def test_subspace_fidelity_regularisers_and_weighted_ensemble_objective():
    """The rest of SURVEY 8(f) row 1: the EmbeddedOperator (subspace) fidelity a multilevel-transmon gate uses
    [REF objectives.jl:339-345], the three quadratic regularisers [REF smooth_pulse_problem.jl:249-251] and the weighted
    ensemble sum [REF sampling_problem.jl:381-387] -- value and full gradient on the device against the oracle."""
    rng = np.random.default_rng(21)
    # two 3-level transmons, CZ-like goal on the qubit subspace, three perturbed-drift members with weights
    base = po.multi_transmon_system([4.0, 4.1], [0.2, 0.2], [[0, 0.05], [0.05, 0]], levels_per_transmon=3, drive_bounds=0.1)
    d, m, M, N = base.levels, base.n_drives, 3, 6
    assert d == 9
    xd = 2 * d * d
    sub = po.get_subspace_indices([[0, 1], [0, 1]], [3, 3])
    assert sub == [0, 1, 3, 4] and pa.get_subspace_indices([[0, 1], [0, 1]], [3, 3]) == sub
    Gs = np.diag([1, 1, 1, -1]).astype(complex) @ np.diag(np.exp(1j * rng.random(4)))
    goal = po.embed(Gs, sub, d)
    lay = po.Layout(d=d, m=m, N=N, z_dim=M * xd + 2 + 3 * m, x_off=0, u_off=M * xd + 2, dt_off=M * xd)
    Z = 0.2 * rng.standard_normal((N, lay.z_dim))
    Z[:, lay.dt_off] = 0.1 + 0.05 * rng.random(N)
    for i in range(M):  # terminal states near a unitary, so that F is in the interesting range
        U = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0]
        Z[-1, i * xd : (i + 1) * xd] = po.operator_to_iso_vec(U) + 0.02 * rng.standard_normal(xd)
    osys = [po.System(base.H_drift + 0.01 * i * np.diag(np.arange(d)).astype(complex), base.H_drives, base.drive_bounds) for i in range(M)]
    psys = [pa.QuantumSystem(s.H_drift, s.H_drives, [b[1] for b in s.drive_bounds]) for s in osys]
    comps = {"Ũ⃗%d" % (i + 1): Z[:, i * xd : (i + 1) * xd].T for i in range(M)}
    o = M * xd
    comps["Δt"], comps["t"] = Z[:, o][None], Z[:, o + 1][None]
    comps["u"], comps["du"], comps["ddu"] = Z[:, o + 2 : o + 2 + m].T, Z[:, o + 2 + m : o + 2 + 2 * m].T, Z[:, o + 2 + 2 * m :].T
    traj = pa.NamedTrajectory(comps, controls=("ddu", "Δt"), timestep="Δt")
    assert np.array_equal(traj.datavec, Z.reshape(-1))
    Bs = pa.BilinearIntegrator(psys, traj, pade_order=4)
    w = np.array([0.5, 0.3, 0.2])
    Q = 100.0
    Ru, Rdu, Rddu = 1e-2, np.linspace(0.5, 2.0, m), 3.0
    names = [B.x_name for B in Bs]
    for pw in (2, 0, 1):
        J = pa.UnitaryInfidelityObjective(pa.EmbeddedOperator(Gs, sub, [3, 3]), names, traj, Q=Q, weights=w)
        J = J + pa.QuadraticRegularizer("u", traj, Ru, pw) + pa.QuadraticRegularizer("du", traj, Rdu, pw) + pa.QuadraticRegularizer("ddu", traj, Rddu, pw)
        val, grad = J.bind(Bs).value_and_gradient(traj)
        regs = [(lay.u_off, m, Ru, pw), (lay.u_off + m, m, Rdu, pw), (lay.u_off + 2 * m, m, Rddu, pw)]
        v_ref, g_ref = po.sampling_objective(Z, lay, [i * xd for i in range(M)], goal, w, Q, regs, subspace=sub)
        assert abs(val - v_ref) < 1e-12 * max(1.0, abs(v_ref)), (pw, val, v_ref)
        close(grad, g_ref.reshape(-1), 1e-11)
    # a regulariser-only objective binds through any member (or sub-list) of the ensemble: the shared context
    Jr = pa.QuadraticRegularizer("u", traj, Ru, 2) + pa.QuadraticRegularizer("ddu", traj, Rddu, 2)
    v_ref, g_ref = po.sampling_objective(Z, lay, [], goal, np.zeros(0), Q, [(lay.u_off, m, Ru, 2), (lay.u_off + 2 * m, m, Rddu, 2)], subspace=sub)
    for sel in (Bs[1], Bs[:2], Bs):
        val, grad = Jr.bind(sel).value_and_gradient(traj)
        assert abs(val - v_ref) < 1e-12 * max(1.0, abs(v_ref))
        close(grad, g_ref.reshape(-1), 1e-11)
    with pytest.raises(ValueError):  # ... an infidelity term needs every member
        pa.Objective([pa.UnitaryInfidelityObjective(pa.EmbeddedOperator(Gs, sub, [3, 3]), names, traj, Q=Q)]).bind(Bs[:2])
    # plain (full-space) goal through the same entry point, unit weights; repeated calls are bitwise identical
    Ufull = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0]
    J = pa.Objective([pa.UnitaryInfidelityObjective(Ufull, names, traj, Q=Q)]).bind(Bs)
    val, grad = J.value_and_gradient(traj)
    v_ref, g_ref = po.sampling_objective(Z, lay, [i * xd for i in range(M)], Ufull, np.ones(M), Q)
    assert abs(val - v_ref) < 1e-12 * max(1.0, abs(v_ref))
    close(grad, g_ref.reshape(-1), 1e-11)
    val2, grad2 = J.value_and_gradient(traj)
    assert val2 == val and np.array_equal(grad2, grad)
    # multistart context: one objective value and one gradient per seed
    Zs = [Z[:, : xd + 2 + 3 * m].copy() for _ in range(2)]
    for q, Zq in enumerate(Zs):
        Zq[:, xd:] = Z[:, M * xd :]
        Zq[:, :xd] = Z[:, q * xd : (q + 1) * xd]
    lay1 = po.Layout.smooth_pulse(d, m, N)
    t1 = traj_from_Z(pa, Zs[0], lay1)
    ms = pa.HipPadeMultistart(osys[0].G_drift, np.array(osys[0].G_drives), t1, 2, pade_order=4)
    J1 = (pa.UnitaryInfidelityObjective(pa.EmbeddedOperator(Gs, sub, [3, 3]), "Ũ⃗", t1, Q=Q) + pa.QuadraticRegularizer("u", t1, Ru)).bind(ms)
    vals, grads = J1.value_and_gradient(np.stack(Zs))
    for q, Zq in enumerate(Zs):
        v_ref, g_ref = po.sampling_objective(Zq, lay1, [0], goal, [1.0], Q, [(lay1.u_off, m, Ru, 2)], subspace=sub)
        assert abs(vals[q] - v_ref) < 1e-12 * max(1.0, abs(v_ref))
        close(grads[q * lay1.z_dim * N : (q + 1) * lay1.z_dim * N], g_ref.reshape(-1), 1e-11)
    ms.close()
    for B in Bs:
        B.close()


@pytest.mark.parametrize("d,m,N", [(2, 2, 8), (5, 2, 6), (27, 6, 4)])
def test_ket_variant(d, m, N):
    """BilinearIntegrator(qtraj::KetTrajectory, N) [REF src/control/integrators.jl:58-74]: the generator acts on one
    iso-ket column (x_dim = 2d), no I_d (x) replication; every output against the oracle with cols = 1."""
    rng = np.random.default_rng(100 + d)
    n = 2 * d
    Hd = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    Hd = Hd + Hd.conj().T
    Hs = []
    for _ in range(m):
        A = (rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))) * (rng.random((d, d)) < 0.4)
        Hs.append(A + A.conj().T)
    so = po.quantum_system(0.3 * Hd, Hs, [1.0] * m)
    xd = n
    z_dim = xd + 2 + 3 * m
    lay = po.Layout(d=d, m=m, N=N, z_dim=z_dim, x_off=0, u_off=xd + 2, dt_off=xd, cols=1)
    Z = 0.5 * rng.standard_normal((N, z_dim))
    Z[:, lay.dt_off] = 0.05 + 0.05 * rng.random(N)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    psys = pa.QuantumSystem(so.H_drift, so.H_drives, [1.0] * m)
    comps = {"ψ̃": Z[:, :xd].T, "Δt": Z[:, xd][None], "t": Z[:, xd + 1][None], "u": Z[:, xd + 2 : xd + 2 + m].T,
             "du": Z[:, xd + 2 + m : xd + 2 + 2 * m].T, "ddu": Z[:, xd + 2 + 2 * m :].T}  # fmt: skip
    traj = pa.NamedTrajectory(comps, controls=("ddu", "Δt"), timestep="Δt")
    assert np.array_equal(traj.datavec, Z.reshape(-1))
    B = pa.BilinearIntegrator(psys, traj, x_name="ψ̃", pade_order=4)
    assert B.x_dim == n and B.dim == n * (N - 1) and B.ctx.jac_per == 2 * n * n + n * (m + 1)
    delta = pa.evaluate_(np.zeros(B.dim), B, traj)
    close(delta, po.pade_residual(Z, lay, G0, Gj, 4), 1e-11)
    d2, vals = B.ctx.eval_jac(traj.datavec)
    close(d2, po.pade_residual(Z, lay, G0, Gj, 4), 1e-11)
    close(vals, po.pade_jacobian_values(Z, lay, G0, Gj, 4), 1e-11)
    r, c = pa.jacobian_structure(B)
    r0, c0 = po.jac_structure(lay)
    assert np.array_equal(r, r0) and np.array_equal(c, c0)
    mu = rng.standard_normal((lay.K, lay.x_dim))
    close(B.ctx.hess(traj.datavec, mu), po.pade4_hessian_values(Z, mu, lay, G0, Gj), 1e-10)
    close(B.f(Z[2, :xd], Z[1, :xd], Z[1, xd + 2 : xd + 2 + m], Z[1, xd]), delta[n : 2 * n], 1e-11)
    B.close()


# ---- Hessian kernels ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,N,batch", [(1, 50, 1), (2, 100, 2), (3, 6, 1), (3, 5, 3)])
def test_hessian_kernel_variants(cfg, N, batch):
    """The Hessian kernels (1: one workgroup per interval; 2: persistent, wave-synchronous, column slices whose scalar
    entries are combined by the last slice to arrive; 3: one workgroup per interval, jobs split by drive, no
    cross-workgroup step) against the oracle, for every slicing of the state columns / any grid; `auto` runs kernel 3 for
    d >= 12 (static instance at BASELINE config 3's shape) and kernel 2 below; all are bitwise repeatable."""
    so = po.config_system(cfg)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Zs, lay = [], None
    for s in range(batch):
        Z, lay = po.synthetic_trajectory(so, N, seed=77 + s)
        Zs.append(Z)
    rng = np.random.default_rng(5 + cfg)
    mus = [rng.standard_normal((lay.K, lay.x_dim)) for _ in range(batch)]
    ref = np.concatenate([po.pade4_hessian_values(Z, mu, lay, G0, Gj).reshape(-1) for Z, mu in zip(Zs, mus)])
    c = make_ctx(lay, G0, Gj, batch=batch, batch_mode=pa._lib.PCL_BATCH_TRAJ, x_offs=[lay.x_off])
    Zb, mub = np.stack(Zs), np.concatenate([m_.reshape(-1) for m_ in mus])
    h_auto = c.hess(Zb, mub)
    # 6: the pattern-compiled order-4 kernel (sparse iso generators, 9 <= d <= 32); 82: the column-group kernel, `auto` for launches of at most n_cu / 2 intervals
    assert c.get_option("last_hess_kernel") in ((6, 82) if cfg == 3 else (2,))
    close(h_auto, ref, 1e-11)
    if cfg == 3:
        c.set_option("hess_kernel", 4)
        for grid in (0, 1, 2, 3, 7, 1000):
            c.set_option("grid", grid)
            h = c.hess(Zb, mub)
            assert c.get_option("last_hess_kernel") == 6
            close(h, ref, 1e-11)
            assert np.array_equal(h, c.hess(Zb, mub))
        c.set_option("grid", 0)
    else:
        c.set_option("hess_kernel", 4)  # not a sparse iso system with 9 <= d <= 32: loud failure, no silent substitute
        with pytest.raises(pa.PclError):
            c.hess(Zb, mub)
    for hk in (1, 2):
        c.set_option("hess_kernel", hk)
        for cps in (0, 1, 2, 3, 5, 16):
            c.set_option("cols_per_slice", cps)
            h = c.hess(Zb, mub)
            assert c.get_option("last_hess_kernel") == hk
            close(h, ref, 1e-11)
            assert np.array_equal(h, c.hess(Zb, mub))
    c.set_option("cols_per_slice", 0)
    c.set_option("hess_kernel", 3)
    for grid in (0, 1, 2, 3, 7, 1000):
        c.set_option("grid", grid)
        h = c.hess(Zb, mub)
        assert c.get_option("last_hess_kernel") == (4 if cfg == 3 else 5)  # 5: the shape's instance was compiled on first use
        close(h, ref, 1e-11)
        assert np.array_equal(h, c.hess(Zb, mub))
    c.close()


def test_contiguous_column_ranges_any_grid():
    """Default work split of kernel 3: the batch*K*d state columns are cut into `grid` equal contiguous ranges, a
    workgroup's items are the pieces of its range inside one interval.  Any grid (also ones that cut intervals at
    odd places, one workgroup only, more workgroups than intervals) must give the oracle's values, bitwise equal to
    the round-robin split."""
    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Bn, N = 2, 5
    Zs, lay = [], None
    for s in range(Bn):
        Z, lay = po.synthetic_trajectory(so, N, seed=300 + s)
        Zs.append(Z)
    ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Zs[0], lay), Bn, pade_order=4)
    c = ms.ctx
    refs = [ref_lib.eval_jac(Z, lay, G0, Gj) for Z in Zs]
    d_ref = np.concatenate([r[0].reshape(-1) for r in refs])
    j_ref = np.concatenate([r[1].reshape(-1) for r in refs])
    c.set_option("kernel_version", 3)
    c.set_option("contiguous", 0)
    d_rr, j_rr = c.eval_jac(np.stack(Zs))
    close(d_rr, d_ref)
    close(j_rr, j_ref)
    c.set_option("contiguous", 1)
    c.set_option("stream_workgroups", 0)
    assert c.get_option("effective_cols_per_slice") == lay.d
    for grid in (0, 1, 2, 3, 7, 8, 13, 31, 100, 215, 216, 217, 256, 1000):
        c.set_option("grid", grid)
        delta, vals = c.eval_jac(np.stack(Zs))
        assert c.get_option("last_kernel") // 10 == 3
        close(delta, d_ref)
        close(vals, j_ref)
        assert np.array_equal(vals, j_rr) and np.array_equal(delta, d_rr), grid
    # role split: `stream_workgroups` workgroups stream the blocks of all columns, the others (eight matrix waves each,
    # single-buffered G) do the column work of all columns
    for grid, ns in ((0, 128), (0, 1), (0, 255), (2, 1), (7, 3), (100, 37), (256, 100), (256, 200)):
        c.set_option("grid", grid)
        c.set_option("stream_workgroups", ns)
        delta, vals = c.eval_jac(np.stack(Zs))
        assert np.array_equal(vals, j_rr) and np.array_equal(delta, d_rr), (grid, ns)
    ms.close()


@pytest.mark.parametrize("Bn", [4, 8])
def test_config5_share_default_path(Bn):
    """BASELINE config 5's per-GPU share (8 seeds of the full-size config-3 problem in one launch: the shipped share; 4: a narrower one):
    the path `auto` picks at this size - contiguous column ranges: kernel 3 with half the workgroups streaming the blocks at 4
    trajectories, the pattern-compiled kernel 4 at 8 - against the C oracle, and bitwise against the round-robin split."""
    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    N = 100
    Zs, lay = [], None
    for s in range(Bn):
        Z, lay = po.synthetic_trajectory(so, N, seed=1000 + s)
        Zs.append(Z)
    ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Zs[0], lay), Bn, pade_order=4)
    c = ms.ctx
    c.set_option("host_path", 1)  # the full-values launch (the default host delivery launches the compact kernel)
    delta, vals = c.eval_jac(np.stack(Zs))
    _auto_large_launch(c)
    per_d, per_j = lay.x_dim * lay.K, po.jac_nnz_per_interval(lay) * lay.K
    for s in range(Bn):
        d_ref, j_ref = ref_lib.eval_jac(Zs[s], lay, G0, Gj)
        close(delta[s * per_d : (s + 1) * per_d], d_ref)
        close(vals[s * per_j : (s + 1) * per_j], j_ref)
    c.set_option("kernel_version", c.get_option("last_kernel") // 10)  # the same kernel on the other split: bitwise the same values
    c.set_option("contiguous", 0)
    d2, v2 = c.eval_jac(np.stack(Zs))
    assert c.get_option("last_stream_workgroups") == 0
    assert np.array_equal(d2, delta) and np.array_equal(v2, vals)
    ms.close()


# ---- the pattern-compiled fused kernel (kernel_version 4; auto for every Pade order but 4) ------------------------------------
@pytest.mark.parametrize("order", [2, 4, 6, 8, 10])
def test_pattern_compiled_fused_kernel(order):
    """pcl_fused_sparse_kernel (generated per system and order): residual + Jacobian at BASELINE config 3 against the oracle, every
    work split (round-robin slices, contiguous ranges, explicit slice widths, odd grids), both tail-store modes, the compact
    layout, full size -- and bitwise equal across all of them: a lane's arithmetic depends on its state column only."""
    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    for N in (5, 100):
        Z, lay = po.synthetic_trajectory(so, N, seed=77)
        Z[:, lay.dt_off] = 0.1 + 0.05 * np.random.default_rng(3).random(N)
        d_ref = po.pade_residual(Z, lay, G0, Gj, order)
        j_ref = po.pade_jacobian_values(Z, lay, G0, Gj, order)
        c = make_ctx(lay, G0, Gj, pade_order=order)
        c.set_option("kernel_version", 4)
        first = None
        splits = ((-1, 0, 0, 1, 3), (0, 0, 0, 1, 3), (1, 0, 0, 1, 3), (0, 5, 0, 1, 3), (0, 27, 7, 1, 3), (0, 1, 0, 1, 3), (1, 0, 5, 1, 0), (0, 5, 0, 1, 0),
                  (0, 0, 0, 1, 2), (-1, 0, 0, 2, 3))
        if N == 100:  # full size: the default split, contiguous ranges (several items per workgroup), the writer wave
            splits = (splits[0], splits[2], splits[7]) if order in (4, 8) else (splits[0],)
        for contig, cps, grid, hp, tm in splits:
            c.set_option("contiguous", contig)
            c.set_option("cols_per_slice", cps)
            c.set_option("grid", grid)
            c.set_option("v4_tail_mode", tm)
            c.set_option("host_path", hp)
            delta, vals = c.eval_jac(Z)
            assert c.get_option("last_kernel") == 40 + order // 2
            close(delta, d_ref, 1e-12)
            close(vals, j_ref, 1e-12)
            if first is None:
                first = (delta, vals)
            assert np.array_equal(delta, first[0]) and np.array_equal(vals, first[1])
        # the cooperative first item (store-stream waves build the item's powers of G, a quarter of the rows each): same bits without
        # it (4), without the chains waiting behind it (16), and with every LDS tile NaN at kernel start (8: the tiles are not
        # initialised -- nothing may read an entry its item has not written)
        c.set_option("contiguous", -1), c.set_option("cols_per_slice", 0), c.set_option("grid", 0)
        c.set_option("v4_power_tiles", 0), c.set_option("v4_tail_mode", 3), c.set_option("host_path", 1)
        for flags in (4, 16, 8, 8 | 4, 32):
            c.set_option("v4_flags", flags)
            delta, vals = c.eval_jac(Z)
            assert np.array_equal(delta, first[0]) and np.array_equal(vals, first[1]), flags
        c.set_option("v4_flags", 0)
        close(c.jac(Z), j_ref, 1e-12)  # eval_jacobian alone: no residual is written
        c.close()


def test_slice_tickets_equal_the_static_split():
    """Kernel 4, launches of several trajectories: groups of workgroups + slice tickets (`v4_ticket`; `auto` at orders 2 and 4 wherever the
    static split would hand out contiguous ranges) against the static split -- the same bits for every group size, slice width and
    request time, at an order `auto` leaves static (8), on TRAJ batches and on an ensemble with per-member drifts and the fused reduce
    payload; against the C oracle once; 60 launches in a row all equal (the counters are re-zeroed by the last pipeline out); a launch
    with fewer intervals than groups."""
    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    for Bn, order in ((8, 4), (4, 2), (4, 8)):
        Zs = [po.synthetic_trajectory(so, 100, seed=900 + s)[0] for s in range(Bn)]
        lay = po.synthetic_trajectory(so, 100, seed=900)[1]
        c = make_ctx(lay, G0, Gj, batch=Bn, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        Zd = torch.from_numpy(np.stack(Zs)).cuda()
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        out = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")

        def run(**opts):
            for k, v in dict(v4_ticket=-1, v4_group=0, v4_ticket_cols=0, v4_ticket_ahead=2, contiguous=-1).items():
                c.set_option(k, v)
            for k, v in opts.items():
                c.set_option(k, v)
            dd.fill_(float("nan")), out.fill_(float("nan"))
            c.eval_jac_dev(Zd, dd, out)
            c.sync()
            assert c.get_option("last_kernel") == 40 + order // 2
            return dd.clone(), out.clone()

        ref = run(v4_ticket=0)
        assert c.get_option("last_v4_ticket") == 0
        assert bool(torch.isfinite(ref[0]).all()) and bool(torch.isfinite(ref[1]).all())
        if order == 4:  # against the C oracle, one trajectory of the batch
            d_ref, j_ref = ref_lib.eval_jac(Zs[Bn - 1], lay, G0, Gj)
            close(ref[0].cpu().numpy().reshape(Bn, -1)[-1], np.asarray(d_ref).reshape(-1))
            close(ref[1].cpu().numpy().reshape(Bn, -1)[-1], np.asarray(j_ref).reshape(-1))
        got = run()  # auto
        assert (c.get_option("last_v4_ticket") > 0) == (order <= 4 and Bn * 99 * 27 >= 28 * c.get_option("n_cu"))
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
        for opts in (dict(v4_ticket=1), dict(v4_ticket=1, v4_group=16), dict(v4_ticket=1, v4_group=4, v4_ticket_cols=5), dict(v4_ticket=1, v4_ticket_cols=1),
                     dict(v4_ticket=1, v4_ticket_cols=27, v4_group=2), dict(v4_ticket=1, v4_ticket_ahead=0), dict(v4_ticket=1, v4_ticket_ahead=1, v4_ticket_cols=2)):
            got = run(**opts)
            assert c.get_option("last_v4_ticket") == opts.get("v4_ticket_cols", 4)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (Bn, order, opts)
        if order == 4:
            run(v4_ticket=1)
            bad = torch.zeros((), dtype=torch.int64, device="cuda")
            for _ in range(60):
                c.eval_jac_dev(Zd, dd, out)
                bad += (out != ref[1]).sum() + (dd != ref[0]).sum()
            assert int(bad.item()) == 0
            vals = torch.full_like(out, float("nan"))
            c.jac_dev(Zd, vals)  # eval_jacobian alone
            c.sync()
            assert torch.equal(vals, ref[1])
            # a switch of streams with a ticket launch in flight waits for it (its counters are zero only once it has finished)
            s2 = torch.cuda.Stream()
            out2 = torch.full_like(out, float("nan"))
            for _ in range(5):
                c.set_stream(torch.cuda.current_stream().cuda_stream)
                c.eval_jac_dev(Zd, dd, out)
                c.set_stream(s2.cuda_stream)
                c.eval_jac_dev(Zd, dd, out2)
            torch.cuda.synchronize()
            assert torch.equal(out, ref[1]) and torch.equal(out2, ref[1])
            c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.close()
    # fewer intervals than groups (2 trajectories of 5 knots: 8 intervals, 32 groups), tickets forced
    Zs = [po.synthetic_trajectory(so, 5, seed=950 + s)[0] for s in range(2)]
    lay = po.synthetic_trajectory(so, 5, seed=950)[1]
    c = make_ctx(lay, G0, Gj, batch=2, batch_mode=pa._lib.PCL_BATCH_TRAJ)
    Zh = np.stack(Zs)
    c.set_option("v4_ticket", 0)
    d0, v0 = c.eval_jac(Zh)
    c.set_option("v4_ticket", 1)
    d1, v1 = c.eval_jac(Zh)
    assert c.get_option("last_v4_ticket") == 4 and np.array_equal(d0, d1) and np.array_equal(v0, v1)
    for b in range(2):
        close(d1.reshape(2, -1)[b], po.pade_residual(Zs[b], lay, G0, Gj, 4).reshape(-1), 1e-12)
        close(v1.reshape(2, -1)[b], po.pade_jacobian_values(Zs[b], lay, G0, Gj, 4).reshape(-1), 1e-12)
    c.close()
    # an ensemble (per-member drifts: the less used drift classes stream through the chunk registers) with the reduce payload formed
    # by the writer wave of whichever workgroup takes an interval's chains
    osys, psys, lay, Z, traj = _config4_share(8, 100)
    B = _fused_ensemble(psys, traj)
    ce = B.ctx
    ce.set_stream(torch.cuda.current_stream().cuda_stream)
    Zd = torch.from_numpy(traj.datavec).cuda()
    ln, _ = ce.merit_grad_len()
    res = []
    for tk in (0, 1, -1):
        ce.set_option("v4_ticket", tk)
        dd = torch.full((ce.n_rows,), float("nan"), dtype=torch.float64, device="cuda")
        out = torch.full((ce.jac_nnz,), float("nan"), dtype=torch.float64, device="cuda")
        pay = torch.full((ln,), float("nan"), dtype=torch.float64, device="cuda")
        ce.eval_jac_merit_dev(Zd, None, dd, out, pay)
        ce.sync()
        assert ce.get_option("last_kernel") == 42 and ce.get_option("last_merit_fused") == 1 and (ce.get_option("last_v4_ticket") > 0) == (tk != 0)
        res.append((dd, out, pay))
    for r in res[1:]:
        assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1]) and torch.equal(r[2], res[0][2])
    B.close()



def test_multistart_64_equals_64_single_launches():
    """BASELINE config 5 whole on one GPU -- 64 seeds in ONE launch (8.5 GB of values; the N = 1 point of the scaling curve, slice
    tickets) -- against 64 launches of one trajectory each (static split), compared on the device, bitwise: a lane's arithmetic depends
    on its state column only.  One seed against the C oracle."""
    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    T = 64
    Zs = [po.synthetic_trajectory(so, 100, seed=1000 + s)[0] for s in range(T)]
    lay = po.synthetic_trajectory(so, 100, seed=1000)[1]
    c64 = make_ctx(lay, G0, Gj, batch=T, batch_mode=pa._lib.PCL_BATCH_TRAJ)
    c1 = make_ctx(lay, G0, Gj, batch=1, batch_mode=pa._lib.PCL_BATCH_TRAJ)
    st = torch.cuda.current_stream().cuda_stream
    c64.set_stream(st), c1.set_stream(st)
    Zd = torch.from_numpy(np.stack(Zs)).cuda()
    dd = torch.full((c64.n_rows,), float("nan"), dtype=torch.float64, device="cuda")
    out = torch.full((c64.jac_nnz,), float("nan"), dtype=torch.float64, device="cuda")
    c64.eval_jac_dev(Zd, dd, out)
    c64.sync()
    assert c64.get_option("last_kernel") == 42 and c64.get_option("last_v4_ticket") > 0
    d1 = torch.empty(c1.n_rows, dtype=torch.float64, device="cuda")
    o1 = torch.empty(c1.jac_nnz, dtype=torch.float64, device="cuda")
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    for s_ in range(T):
        c1.eval_jac_dev(Zd[s_], d1, o1)
        bad += (out.view(T, -1)[s_] != o1).sum() + (dd.view(T, -1)[s_] != d1).sum()
    c1.sync()
    assert c1.get_option("last_v4_ticket") == 0 and int(bad.item()) == 0
    d_ref, j_ref = ref_lib.eval_jac(Zs[37], lay, G0, Gj)
    close(dd.view(T, -1)[37].cpu().numpy(), np.asarray(d_ref).reshape(-1))
    close(out.view(T, -1)[37].cpu().numpy(), np.asarray(j_ref).reshape(-1))
    c64.close(), c1.close()



def test_code_objects_persist_across_processes(tmp_path):
    """The run-time compiled modules are kept on disk under the hash of their sources: a fresh process finds the BASELINE config-3 modules
    in csrc/prebuilt (written by __graft_entry__.build(), no hiprtc at run time) and anything else in the user's cache once ONE process
    has compiled it -- `jit_compiles` == 0 in the second process, same values; with the cache switched off it compiles again."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import json, sys, time, numpy as np
sys.path.insert(0, %r)
import piccolo_jl_amd as pa
from oracle import pade_oracle as po
order = int(sys.argv[1])
so = po.config_system(3)
Z, lay = po.synthetic_trajectory(so, 6, seed=5)
c = pa.integrators._PclContext(d=lay.d, m=lay.m, N=lay.N, z_dim=lay.z_dim, u_off=lay.u_off, dt_off=lay.dt_off, x_offs=[lay.x_off], G0=so.G_drift,
                               Gj=np.array(so.G_drives), batch=1, batch_mode=pa._lib.PCL_BATCH_MEMBERS, pade_order=order)
c.set_option("host_path", 1)
c.set_option("require_jit", 1)
t0 = time.perf_counter()
delta, vals = c.eval_jac(Z)
t1 = time.perf_counter() - t0
h = c.hess(Z, np.ones(c.n_rows))
print(json.dumps(dict(compiles=c.get_option("jit_compiles"), hits=c.get_option("jit_cache_hits"), kernel=c.get_option("last_kernel"), first_call_ms=t1 * 1e3,
                      checksum=float(np.abs(vals).sum() + np.abs(delta).sum() + np.abs(h).sum()))))
""" % root

    def run(order, **env):
        e = dict(os.environ, PCL_JIT_CACHE_DIR=str(tmp_path / "cache"), **env)
        out = subprocess.run([sys.executable, "-c", code, str(order)], env=e, capture_output=True, text=True, check=True).stdout
        return json.loads(out.strip().splitlines()[-1])

    pre = os.path.join(root, "piccolo.jl_amd", "csrc", "prebuilt")
    if os.path.isdir(pre) and any(f.endswith(".hsaco") for f in os.listdir(pre)):
        a = run(4)  # config 3, order 4: prebuilt
        assert a["kernel"] == 42 and a["compiles"] == 0 and a["hits"] >= 2, a
    first, second = run(6), run(6)  # order 6 is not prebuilt: compiled once, then read back
    assert first["kernel"] == 43 and first["compiles"] >= 2 and second["compiles"] == 0 and second["hits"] >= 2, (first, second)
    assert first["checksum"] == second["checksum"]
    assert second["first_call_ms"] < first["first_call_ms"]
    off = run(6, PCL_JIT_CACHE="0")
    assert off["compiles"] >= 2 and off["hits"] == 0 and off["checksum"] == first["checksum"]



def test_order_policy_picks_the_order_that_matches_the_exp_constraint():
    """pade_order = 0: the smallest diagonal Pade order whose deviation from the reference's constraint x_{k+1} = exp(dt G(u_k)) x_k
    [REF docs/src/concepts/index.md:21] stays below a tolerance -- from the problem's bounds (pcl_set_order_policy) or from the first
    trajectory.  On an exp-feasible config-3 trajectory the chosen order's residual is below the tolerance and the next lower order's is
    not; the values are those of a context created with that order; a device-pointer call before an order exists is refused."""
    import math

    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    N = 12
    Z, lay = po.synthetic_trajectory(so, N, seed=11, noise=0.0)  # X_{k+1} = expm(dt G(u_k)) X_k exactly: the reference's feasible set
    theta = max(abs(Z[k, lay.dt_off]) * np.linalg.norm(G0 + np.tensordot(Z[k, lay.u_off : lay.u_off + lay.m], Gj, axes=1), 2) for k in range(N - 1))
    kappa = lambda q: math.factorial(q) ** 2 / (math.factorial(2 * q) * math.factorial(2 * q + 1))
    want = lambda th, tol: next((2 * q for q in range(1, 6) if kappa(q) * th ** (2 * q + 1) <= tol), 10)
    for tol in (1e-4, 1e-7, 1e-10, 1e-13):
        c = make_ctx(lay, G0, Gj, pade_order=0)
        assert c.pade_order == 0
        with pytest.raises(pa._lib.PclError) as ei:  # nothing to look at yet
            c.eval_dev(torch.from_numpy(Z).cuda(), torch.empty(c.n_rows, dtype=torch.float64, device="cuda"))
        assert ei.value.code == pa._lib.PCL_EINVAL
        # bounds: u in [-0.1, 0.1] (drive_bounds of the system), dt <= 0.1: theta = dt_max x the maximum of |G(u)|_2 over the box, at a vertex
        import itertools

        th_b = 0.1 * max(np.linalg.norm(G0 + np.tensordot(0.1 * np.array(sg), Gj, axes=1), 2) for sg in itertools.product((-1.0, 1.0), repeat=lay.m))
        order = c.set_order_policy(0.1, np.full(lay.m, 0.1), tol)
        assert order == want(th_b, tol) == c.pade_order, (tol, order, th_b)
        assert abs(c.get_option("order_theta_1e9") * 1e-9 - th_b) < 1e-6 * th_b
        delta, vals = c.eval_jac(Z)
        assert bool(c.get_option("order_tol_met")) == (kappa(5) * th_b**11 <= tol)
        if c.get_option("order_tol_met"):
            assert np.abs(delta).max() <= tol  # (the bound holds with theta over the bounds >= theta on this trajectory)
        cx = make_ctx(lay, G0, Gj, pade_order=order)
        d2, v2 = cx.eval_jac(Z)
        assert np.array_equal(delta, d2) and np.array_equal(vals, v2)
        close(c.hess(Z, np.ones(c.n_rows)), cx.hess(Z, np.ones(c.n_rows)), 0.0)
        cx.close(), c.close()
        # from the first trajectory (no policy call): theta = 1.5 x the trajectory's maximum
        c = make_ctx(lay, G0, Gj, pade_order=0)
        delta, _ = c.eval_jac(Z)
        assert c.pade_order == want(1.5 * theta, 1e-10) and np.abs(delta).max() <= 1e-10
        if c.pade_order > 2:  # the next lower order misses the tolerance on this trajectory's own theta by the factor the bound predicts
            lo = make_ctx(lay, G0, Gj, pade_order=c.pade_order - 2)
            assert kappa(c.pade_order // 2 - 1) * (1.5 * theta) ** (c.pade_order - 1) > 1e-10
            lo.close()
        c.close()
    # the high-level constructor reads the bounds of the trajectory
    system = pa.MultiTransmonSystem([4.0, 4.1, 4.2], [0.2, 0.21, 0.22], [[0, 0.01, 0.02], [0.01, 0, 0.03], [0.02, 0.03, 0]], levels_per_transmon=3, drive_bounds=0.1)
    traj = traj_from_Z(pa, Z, lay)
    traj.bounds["u"] = (-0.1 * np.ones(lay.m), 0.1 * np.ones(lay.m))
    traj.bounds["Δt"] = (np.array([0.05]), np.array([0.1]))
    B = pa.HipPadeIntegrator(system.G_drift, system.G_drives_array(), traj, pade_order=0, order_tol=1e-10)
    assert B.pade_order == want(th_b, 1e-10)
    B.close()



def _dense_from_triplets(rows, cols, vals, nvar):
    H = np.zeros((nvar, nvar))
    np.add.at(H, (rows, cols), vals)
    return H + np.tril(H, -1).T


def test_ket_coherent_ket_and_density_objectives_on_the_device():
    """The terminal losses of the other state types [REF src/control/objectives.jl:24-60, 96-200, 387-435] through pcl_set_goal_form:
    value and gradient of the whole objective (loss + regularisers) against the oracle's restatement of the reference's formulas
    (gradients: central differences of it, as the reference takes them with ForwardDiff), the reference's literal 100 (1 - 0.9025), and
    the objective's Hessian against the exact second differences of the oracle's (at most quadratic) fidelity and of the regularisers."""
    rng = np.random.default_rng(31)
    d, m, N, Kk = 3, 2, 4, 3
    n = 2 * d
    Hd = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    Hs = [(lambda A: A + A.conj().T)(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))) for _ in range(m)]
    so = po.quantum_system(0.3 * (Hd + Hd.conj().T), Hs, [1.0] * m)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    z_dim = Kk * n + 2 + m
    Z = 0.4 * rng.standard_normal((N, z_dim))
    Z[:, Kk * n] = 0.05 + 0.05 * rng.random(N)
    comps = {"ψ̃%d" % (i + 1): Z[:, i * n : (i + 1) * n].T for i in range(Kk)}
    comps["Δt"], comps["t"], comps["u"] = Z[:, Kk * n][None], Z[:, Kk * n + 1][None], Z[:, Kk * n + 2 :].T
    traj = pa.NamedTrajectory(comps, controls=("u", "Δt"), timestep="Δt")
    names = ["ψ̃%d" % (i + 1) for i in range(Kk)]
    B = pa.HipPadeIntegrator(G0, Gj, traj, names, pade_order=4)
    goals = [(lambda v: v / np.linalg.norm(v))(rng.standard_normal(d) + 1j * rng.standard_normal(d)) for _ in range(Kk)]
    Q, Ru = 100.0, np.array([0.3, 0.7])
    u_off, dt_off = Kk * n + 2, Kk * n
    nvar = N * z_dim

    def reg(Zf, pw):
        Zm = Zf.reshape(N, z_dim)
        return po.quadratic_regularizer(Zm, u_off, m, Ru, dt_off, pw)

    def check(J, loss, pw):
        """loss(Z [N, z_dim]) -> the terminal term; the regulariser on u with dt_power pw rides along."""
        total = lambda zf: loss(zf.reshape(N, z_dim)) + reg(zf, pw)
        val, grad = J.value_and_gradient(traj)
        z0 = traj.datavec.copy()
        assert abs(val - total(z0)) < 1e-11 * max(1.0, abs(val))
        close(grad, po.numerical_gradient(total, z0, 1e-6), 2e-6)
        rows, cols = J.hessian_structure()
        assert (rows >= cols).all() and len(set(zip(rows.tolist(), cols.tolist()))) >= len(rows) - N  # ((dt, dt) repeats per regulariser only)
        Hn = np.zeros((nvar, nvar))
        for i in range(nvar):  # central differences of the oracle's (numerical) gradient, column by column
            e = np.zeros(nvar)
            e[i] = 1e-4
            Hn[:, i] = (po.numerical_gradient(total, z0 + e, 1e-4) - po.numerical_gradient(total, z0 - e, 1e-4)) / 2e-4
        for sigma in (1.0, 0.37):
            H = _dense_from_triplets(rows, cols, J.hessian(traj, sigma), nvar)
            assert np.abs(H - sigma * 0.5 * (Hn + Hn.T)).max() < 2e-4 * max(1.0, np.abs(H).max()), (sigma, np.abs(H - sigma * Hn).max())
        return val

    last = lambda Zm, i: Zm[-1, i * n : (i + 1) * n]
    for pw in (2, 1):
        # one ket term per member (a SamplingProblem-style weighted sum over the kets)
        w = np.array([0.5, 0.3, 0.2])
        J = (pa.KetInfidelityObjective(goals[0], names, traj, Q=Q, weights=w) + pa.QuadraticRegularizer("u", traj, Ru, pw)).bind(B)
        check(J, lambda Zm: sum(w[i] * Q * abs(1 - po.ket_fidelity_loss(last(Zm, i), goals[0])) for i in range(Kk)), pw)
        # ONE coherent term over the three kets, weighted and unweighted
        for cw in (None, [0.9, 0.1, 0.4]):
            J = (pa.CoherentKetInfidelityObjective(goals, names, traj, Q=Q, weights=cw) + pa.QuadraticRegularizer("u", traj, Ru, pw)).bind(B)
            check(J, lambda Zm: Q * abs(1 - po.coherent_ket_fidelity([last(Zm, i) for i in range(Kk)], goals, cw)), pw)
    # exact Hessian of the coherent term: -s Q times the second differences of the quadratic fidelity
    J = pa.Objective([pa.CoherentKetInfidelityObjective(goals, names, traj, Q=Q, weights=[0.9, 0.1, 0.4])]).bind(B)
    rows, cols = J.hessian_structure()
    H = _dense_from_triplets(rows, cols, J.hessian(traj, 1.0), nvar)
    t0 = (N - 1) * z_dim
    Fq = lambda x: po.coherent_ket_fidelity([x[i * n : (i + 1) * n] for i in range(Kk)], goals, [0.9, 0.1, 0.4])
    sgn = 1.0 if Fq(Z[-1, : Kk * n]) <= 1 else -1.0
    close(H[t0 : t0 + Kk * n, t0 : t0 + Kk * n], -sgn * Q * po.quadratic_hessian(Fq, Kk * n), 1e-11)
    assert np.abs(H).sum() == np.abs(H[t0 : t0 + Kk * n, t0 : t0 + Kk * n]).sum()
    B.close()
    # the reference's own literal [REF objectives.jl:585-603]: <psi1|psi1> = 1, <psi0|psi0/2> = 1/2, weights (0.9, 0.1): 100 (1 - 0.9025)
    psi0, psi1 = np.array([1.0, 0.0], complex), np.array([0.0, 1.0], complex)
    Nl = 4
    c2 = {"ψ̃1": np.tile(po.ket_to_iso(psi1)[:, None], (1, Nl)), "ψ̃2": np.tile(po.ket_to_iso(0.5 * psi0)[:, None], (1, Nl)),
          "u": rng.standard_normal((1, Nl)), "Δt": np.full((1, Nl), 0.1)}  # fmt: skip
    t2 = pa.NamedTrajectory(c2, controls=("u", "Δt"), timestep="Δt")
    s2 = po.quantum_system(np.diag([0.0, 1.0]).astype(complex), [np.array([[0, 1], [1, 0]], complex)], [1.0])
    B2 = pa.HipPadeIntegrator(s2.G_drift, np.array(s2.G_drives), t2, ["ψ̃1", "ψ̃2"], pade_order=4)
    for wts, F in (([0.9, 0.1], 0.9025), ([0.1, 0.9], 0.3025)):
        v, _ = pa.Objective([pa.CoherentKetInfidelityObjective([psi1, psi0], ["ψ̃1", "ψ̃2"], t2, Q=100.0, weights=wts)]).bind(B2).value_and_gradient(t2)
        assert abs(v - 100.0 * (1 - F)) < 1e-12
    vu, _ = pa.Objective([pa.CoherentKetInfidelityObjective([psi1, psi0], ["ψ̃1", "ψ̃2"], t2, Q=100.0)]).bind(B2).value_and_gradient(t2)
    vw, _ = pa.Objective([pa.CoherentKetInfidelityObjective([psi1, psi0], ["ψ̃1", "ψ̃2"], t2, Q=100.0, weights=[0.5, 0.5])]).bind(B2).value_and_gradient(t2)
    assert vu == vw  # uniform weights are the unweighted path
    B2.close()
    # a density: compact iso vector under the compact Lindbladian; both density losses are linear in the state
    nl = 3
    H0 = rng.standard_normal((nl, nl)) + 1j * rng.standard_normal((nl, nl))
    a = pa.annihilate(nl)
    sys_ = pa.OpenQuantumSystem(0.5 * (H0 + H0.conj().T), [a + a.conj().T], [1.0], [0.3 * a])
    psi = rng.standard_normal(nl) + 1j * rng.standard_normal(nl)
    psi /= np.linalg.norm(psi)
    times = np.cumsum(np.concatenate(([0.0], 0.05 + 0.05 * rng.random(N - 1))))
    trd = pa.density_trajectory(sys_, 0.5 * rng.standard_normal((1, N)), times, np.outer(psi, psi.conj()), np.outer(psi, psi.conj()))
    Bd = pa.BilinearIntegrator(sys_, trd, pade_order=4)
    gpsi = rng.standard_normal(nl) + 1j * rng.standard_normal(nl)
    gpsi /= np.linalg.norm(gpsi)
    Mg = rng.standard_normal((nl, nl)) + 1j * rng.standard_normal((nl, nl))
    rho_g = Mg @ Mg.conj().T / np.trace(Mg @ Mg.conj().T).real
    xo = trd.components[Bd.x_name].start
    for J, loss in ((pa.DensityMatrixPureStateInfidelityObjective(Bd.x_name, gpsi, trd, Q=Q), lambda x: Q * po.density_matrix_pure_state_infidelity_loss(x, gpsi)),
                    (pa.DensityMatrixInfidelityObjective(Bd.x_name, rho_g, trd, Q=Q), lambda x: Q * po.density_matrix_infidelity_loss(x, rho_g))):
        Jb = pa.Objective([J]).bind(Bd)
        val, grad = Jb.value_and_gradient(trd)
        xN = trd.datavec.reshape(N, trd.dim)[-1, xo : xo + nl * nl]
        assert abs(val - loss(xN)) < 1e-12 * max(1.0, abs(val))
        gref = np.zeros((N, trd.dim))
        gref[-1, xo : xo + nl * nl] = po.numerical_gradient(loss, xN, 1e-6)
        close(grad, gref.reshape(-1), 1e-7)
        assert Jb.hessian(trd).size == 0  # linear in the state: no second derivative
    Bd.close()


@pytest.mark.parametrize("sub", [False, True])
def test_objective_hessian_of_the_unitary_problem(sub):
    """sigma grad^2 f of the unitary templates' objective (terminal infidelity, plain or on an EmbeddedOperator's subspace, + the three
    quadratic regularisers) -- the part of eval_hessian_lagrangian that pcl_hess leaves to the objective [REF spline_pulse_problem.jl:96,
    objectives.jl:330-356]: the terminal block against the exact second differences of the oracle's (quadratic) fidelity, the
    regulariser entries against their closed forms, the structure's index pairs each once (lower triangle); an ensemble with weights
    and a multistart batch."""
    rng = np.random.default_rng(17)
    so = po.config_system(2)
    d, m, N = so.levels, so.n_drives, 5
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Z, lay = po.synthetic_trajectory(so, N, seed=3)
    Z[:, lay.dt_off] = 0.1 + 0.05 * rng.random(N)
    traj = traj_from_Z(pa, Z, lay)
    B = pa.HipPadeIntegrator(G0, Gj, traj, pade_order=4)
    if sub:
        idx = pa.get_subspace_indices([[0, 1], [0]], [2, 2])
        Us = np.linalg.qr(rng.standard_normal((len(idx), len(idx))) + 1j * rng.standard_normal((len(idx), len(idx))))[0]
        goal = pa.EmbeddedOperator(Us, idx, [2, 2])
        Ufull, subspace = goal.operator, idx
    else:
        goal = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0]
        Ufull, subspace = goal, None
    Q, regs = 100.0, (("u", 1e-2, 2), ("du", np.linspace(0.5, 2.0, m), 1), ("ddu", 3.0, 0))
    J = pa.UnitaryInfidelityObjective(goal, "Ũ⃗", traj, Q=Q)
    for nm, R, pw in regs:
        J = J + pa.QuadraticRegularizer(nm, traj, R, pw)
    J.bind(B)
    rows, cols = J.hessian_structure()
    nvar = N * lay.z_dim
    assert (rows >= cols).all() and rows.max() < nvar
    xd = lay.x_dim
    Fq = lambda x: po.unitary_fidelity_loss(x, Ufull, subspace)
    for sigma in (1.0, 2.5):
        H = _dense_from_triplets(rows, cols, J.hessian(traj, sigma), nvar)
        t0 = (N - 1) * lay.z_dim + lay.x_off
        sgn = 1.0 if Fq(Z[-1, lay.x_off : lay.x_off + xd]) <= 1 else -1.0
        close(H[t0 : t0 + xd, t0 : t0 + xd], -sgn * sigma * Q * po.quadratic_hessian(Fq, xd), 1e-10)
        Href = np.zeros((nvar, nvar))
        Href[t0 : t0 + xd, t0 : t0 + xd] = H[t0 : t0 + xd, t0 : t0 + xd]
        for k in range(N):
            z0, h = k * lay.z_dim, Z[k, lay.dt_off]
            for (nm, R, pw), off in zip(regs, (lay.u_off, lay.u_off + m, lay.u_off + 2 * m)):
                Rv = np.broadcast_to(np.asarray(R, float), (m,))
                v = Z[k, off : off + m]
                for i in range(m):
                    Href[z0 + off + i, z0 + off + i] += sigma * h**pw * Rv[i]
                    if pw >= 1:
                        c = sigma * pw * h ** (pw - 1) * Rv[i] * v[i]
                        Href[z0 + off + i, z0 + lay.dt_off] += c
                        Href[z0 + lay.dt_off, z0 + off + i] += c
                if pw == 2:
                    Href[z0 + lay.dt_off, z0 + lay.dt_off] += sigma * (Rv * v * v).sum()
        close(H, Href, 1e-11)
    # the Hessian is the derivative of the gradient the same context returns
    z0 = traj.datavec.copy()
    H = _dense_from_triplets(rows, cols, J.hessian(traj, 1.0), nvar)
    for i in rng.choice(nvar, 12, replace=False):
        e = np.zeros(nvar)
        e[i] = 1e-6
        gp, gm = J.value_and_gradient(z0 + e)[1], J.value_and_gradient(z0 - e)[1]
        assert np.abs((gp - gm) / 2e-6 - H[:, i]).max() < 1e-5 * max(1.0, np.abs(H[:, i]).max())
    B.close()
    # a multistart batch: one block per seed, offset by the seed's variables
    S = 3
    Zs = [po.synthetic_trajectory(so, N, seed=40 + q)[0] for q in range(S)]
    ms = pa.HipPadeMultistart(G0, Gj, traj, S, pade_order=4)
    Jm = (pa.UnitaryInfidelityObjective(goal, "Ũ⃗", traj, Q=Q) + pa.QuadraticRegularizer("u", traj, 1e-2, 2)).bind(ms)
    rows, cols = Jm.hessian_structure()
    vals = Jm.hessian(np.stack(Zs), 1.0)
    Hm = _dense_from_triplets(rows, cols, vals, S * nvar)
    for q in range(S):
        t0 = q * nvar + (N - 1) * lay.z_dim + lay.x_off
        sgn = 1.0 if Fq(Zs[q][-1, lay.x_off : lay.x_off + xd]) <= 1 else -1.0
        close(Hm[t0 : t0 + xd, t0 : t0 + xd], -sgn * Q * po.quadratic_hessian(Fq, xd), 1e-10)
    assert np.abs(Hm[:nvar, nvar:]).max() == 0.0
    ms.close()



def test_pattern_compiled_fused_kernel_soak():
    """Kernel 4 synchronises its waves through LDS counters with bounded waits (a wait that gives up writes NaN): 300 launches per
    shape of launch (one trajectory, four, compact, contiguous compact) -- every launch bitwise equal to the first, nothing NaN."""
    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    for Bn, order in ((1, 4), (4, 8)):
        Zs = [po.synthetic_trajectory(so, 100, seed=500 + s)[0] for s in range(Bn)]
        lay = po.synthetic_trajectory(so, 100, seed=500)[1]
        c = make_ctx(lay, G0, Gj, batch=Bn, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        Zd = torch.from_numpy(np.stack(Zs)).cuda()
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        for compact, contig in ((False, -1), (True, -1), (True, 1)):
            out = torch.empty(c.compact_nnz if compact else c.jac_nnz, dtype=torch.float64, device="cuda")
            c.set_option("contiguous", contig)
            call = (lambda: c.eval_jac_compact_dev(Zd, dd, out)) if compact else (lambda: c.eval_jac_dev(Zd, dd, out))
            call()
            torch.cuda.synchronize()
            assert c.get_option("last_kernel") == 40 + order // 2
            first, firstd = out.clone(), dd.clone()
            assert bool(torch.isfinite(first).all()) and bool(torch.isfinite(firstd).all())
            bad = torch.zeros((), dtype=torch.int64, device="cuda")
            for _ in range(300):
                call()
                bad += (out != first).sum() + (dd != firstd).sum()
            assert int(bad.item()) == 0, (Bn, order, compact, contig)
        c.close()


def test_pattern_compiled_fused_kernel_store_modes_and_first_item():
    """Kernel 4: the store modes of the block stream (plain, write-through, write-through on every other XCD's workgroups -- `auto` at
    orders 2 and 4 for launches that fit the infinity cache) and the two ways the first item's powers of G come about (built by waves
    0-3 together while the stream waves fold them, or by the P wave alone) are performance choices: the same bits, at the orders where
    the ring of power tiles is as long as q (2, 4, 6), shorter (8) and where `auto` leaves the cooperative start off (10); one
    trajectory per launch and three."""
    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    for Bn in (1, 3):
        Zs = [po.synthetic_trajectory(so, 100, seed=700 + s)[0] for s in range(Bn)]
        lay = po.synthetic_trajectory(so, 100, seed=700)[1]
        for order in (2, 4, 6, 8, 10):
            c = make_ctx(lay, G0, Gj, batch=Bn, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
            c.set_stream(torch.cuda.current_stream().cuda_stream)
            Zd = torch.from_numpy(np.stack(Zs)).cuda()
            dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
            out = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
            ref = None
            for nt, flags in ((-1, 0), (0, 0), (2, 0), (3, 0), (-1, 4), (3, 4)):
                c.set_option("nt_stores", nt)
                c.set_option("v4_flags", flags)
                dd.fill_(float("nan")), out.fill_(float("nan"))
                c.eval_jac_dev(Zd, dd, out)
                torch.cuda.synchronize()
                assert c.get_option("last_kernel") == 40 + order // 2
                if ref is None:
                    ref = (dd.clone(), out.clone())
                    assert bool(torch.isfinite(ref[0]).all()) and bool(torch.isfinite(ref[1]).all())
                    if Bn == 1 and order == 4:  # ... and against the C oracle once
                        d_ref, j_ref = ref_lib.eval_jac(Zs[0], lay, G0, Gj)
                        close(ref[0].cpu().numpy(), np.asarray(d_ref).reshape(-1))
                        close(ref[1].cpu().numpy(), np.asarray(j_ref).reshape(-1))
                assert torch.equal(dd, ref[0]) and torch.equal(out, ref[1]), (Bn, order, nt, flags)
            c.close()


def test_pattern_compiled_fused_kernel_ensemble_and_shapes():
    """Kernel 4 on per-member drifts (BASELINE config 4's members: 27 drift value classes, the less used ones streamed through the
    chunk registers), on a member window, on TRAJ batches, and on other sparse shapes (two 5-level transmons, d = 25, m = 4)."""
    osys, psys, lay, Z, traj = _config4_share(3, 9)
    xd = lay.x_dim
    B = _fused_ensemble(psys, traj)
    c = B.ctx
    c.set_option("host_path", 1)
    per_d, per_j = xd * lay.K, po.jac_nnz_per_interval(lay) * lay.K
    refs = [ref_lib.eval_jac(Z, lay, s.G_drift, np.array(s.G_drives), x_off=i * xd) for i, s in enumerate(osys)]
    d_ref = np.concatenate([r[0].reshape(-1) for r in refs])
    j_ref = np.concatenate([r[1].reshape(-1) for r in refs])
    c.set_option("kernel_version", 4)
    for contig, grid in ((-1, 0), (1, 0), (0, 3), (1, 7)):
        c.set_option("contiguous", contig)
        c.set_option("grid", grid)
        delta, vals = c.eval_jac(traj.datavec)
        assert c.get_option("last_kernel") == 42
        close(delta, d_ref)
        close(vals, j_ref)
    c.set_option("grid", 0)
    c.set_member_window(1, 2)
    delta, vals = c.eval_jac(traj.datavec)
    close(delta, d_ref[per_d:])
    close(vals, j_ref[per_j:])
    B.close()
    # multistart batch (TRAJ mode), order 8
    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Zs = [po.synthetic_trajectory(so, 6, seed=500 + i)[0] for i in range(3)]
    lay3 = po.Layout.smooth_pulse(so.levels, so.n_drives, 6)
    c = make_ctx(lay3, G0, Gj, batch=3, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=8)
    delta, vals = c.eval_jac(np.stack(Zs))
    assert c.get_option("last_kernel") == 44  # auto at order 8
    close(delta, np.concatenate([po.pade_residual(z, lay3, G0, Gj, 8).reshape(-1) for z in Zs]), 1e-12)
    close(vals, np.concatenate([po.pade_jacobian_values(z, lay3, G0, Gj, 8).reshape(-1) for z in Zs]), 1e-12)
    c.close()
    # another sparse shape: two 5-level transmons
    s2 = po.multi_transmon_system([4.0, 4.1], [0.2, 0.21], [[0, 0.02], [0.02, 0]], levels_per_transmon=5, drive_bounds=0.1)
    G0, Gj = s2.G_drift, np.array(s2.G_drives)
    Z, lay = po.synthetic_trajectory(s2, 7, seed=3)
    for order in (4, 6):
        c = make_ctx(lay, G0, Gj, pade_order=order)
        c.set_option("kernel_version", 4)
        delta, vals = c.eval_jac(Z)
        assert c.get_option("last_kernel") == 40 + order // 2
        close(delta, po.pade_residual(Z, lay, G0, Gj, order), 1e-12)
        close(vals, po.pade_jacobian_values(Z, lay, G0, Gj, order), 1e-12)
        c.close()


# ---- higher Pade orders ---------------------------------------------------------------------------------
@pytest.mark.parametrize("order", [2, 4, 6, 8, 10])
@pytest.mark.parametrize("cfg,N", [(1, 12), (2, 20), (3, 4)])
def test_general_pade_orders_vs_oracle(order, cfg, N):
    """delta and the Jacobian of the order-p residual B^-_p X_{k+1} - B^+_p X_k (general-order kernel; for p = 4 it is
    forced with `general_pade_kernel` and must agree with the tuned kernels too), every slicing, compact form."""
    so = po.config_system(cfg)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Z, lay = po.synthetic_trajectory(so, N, seed=40 + cfg)
    c = make_ctx(lay, G0, Gj, pade_order=order)
    d_ref = po.pade_residual(Z, lay, G0, Gj, order)
    j_ref = po.pade_jacobian_values(Z, lay, G0, Gj, order)
    if order == 4:
        d4, j4 = c.eval_jac(Z)
        c.set_option("general_pade_kernel", 1)
    c.set_option("general_kernel_version", 1)  # the reference formulation: three Horner chains, one workgroup per slice
    for cps in (0, 1, 2, 5):
        c.set_option("cols_per_slice", cps)
        delta, vals = c.eval_jac(Z)
        assert c.get_option("last_kernel") == 90 + order // 2
        close(delta, d_ref, 1e-11)
        close(vals, j_ref, 1e-11)
        close(c.eval(Z), d_ref, 1e-11)
    if order == 4:
        close(delta, d4)
        close(vals, j4)
    # the lock-step kernel (default where the shape fits): every slicing, full and compact layout
    c.set_option("general_kernel_version", 2)
    delta1 = vals1 = None
    for slices in (0, 1, 2, 3, lay.d):
        c.set_option("general_slices", slices)
        delta, vals = c.eval_jac(Z)
        assert c.get_option("last_kernel") == 190 + order // 2
        close(delta, d_ref, 1e-11)
        close(vals, j_ref, 1e-11)
        if delta1 is None:
            delta1, vals1 = delta, vals
        assert np.array_equal(delta, delta1) and np.array_equal(vals, vals1)  # the slicing does not change the arithmetic
    c.set_option("general_kernel_version", 0)
    c.set_option("general_slices", 0)
    delta, vals = c.eval_jac(Z)
    if cfg == 3 and order != 4:  # auto: the pattern-compiled fused kernel (sparse iso generators; test_pattern_compiled_fused_kernel)
        assert c.get_option("last_kernel") == 40 + order // 2
        close(delta, d_ref, 1e-11)
        close(vals, j_ref, 1e-11)
    elif cfg in (1, 2) and order != 4:  # auto: the small-system kernel (n <= 8 rows), every order; full and compact values (order 4 runs with general_pade_kernel = 1 here)
        assert c.get_option("last_kernel") == 50 + order // 2
        close(delta, d_ref, 1e-11)
        close(vals, j_ref, 1e-11)
        close(c.eval(Z), d_ref, 1e-11)
        assert c.get_option("last_kernel") == 50 + order // 2
        c.set_option("host_path", 2)
        d2, v2 = c.eval_jac(Z)
        assert np.array_equal(d2, delta) and np.array_equal(v2, vals)
        c.set_option("host_path", 1)
    else:
        assert c.get_option("last_kernel") == 190 + order // 2 and np.array_equal(vals, vals1)
    c.set_option("general_kernel_version", 2)  # (the device-pointer checks below: the lock-step kernel)
    import torch

    Zd = torch.from_numpy(np.ascontiguousarray(Z).reshape(-1)).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    cv = torch.empty(c.compact_nnz, dtype=torch.float64, device="cuda")
    fv = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    c.eval_jac_compact_dev(Zd, dd, cv)
    assert c.get_option("last_kernel") == 190 + order // 2
    c.jac_expand_dev(cv, fv)
    c.sync()
    assert np.array_equal(fv.cpu().numpy(), vals1) and np.array_equal(dd.cpu().numpy(), delta1)
    c.close()


def test_higher_orders_reach_the_reference_exp_floor(golden, golden_meta):
    """The reference's constraint is x_{k+1} = expv(dt G) x_k.  On its own solved trajectory (two_qubit_zoh, exp-residual
    6e-12) the GPU residual of order 4 sits at 1.5e-9 (Pade truncation), orders 6, 8, 10 at the reference's floor."""
    systems, lay, _ = ref_case("two_qubit_zoh", golden_meta)
    Z = golden("ref_two_qubit_zoh")["Z"]
    so = systems[0]
    G0, Gj = so.G_drift, np.array(so.G_drives)
    r_exp = np.abs(po.exp_residual(Z, lay, G0, Gj)).max()
    assert r_exp < 1e-11
    got = {}
    for order in (2, 4, 6, 8, 10):
        c = make_ctx(lay, G0, Gj, pade_order=order)
        delta = c.eval(Z)
        close(delta, po.pade_residual(Z, lay, G0, Gj, order), 1e-12)
        got[order] = np.abs(delta).max()
        c.close()
    assert got[2] > 1e-6 and 1e-9 < got[4] < 3e-9
    for order in (6, 8, 10):
        assert got[order] < 2e-11, got
    # large steps (multilevel_transmon, ||dt G|| ~ 1.6, exp-residual 3.2e-10): the truncation error falls by ~100x per
    # order step but is still 1.5e-8 at order 10 -- the Pade constraint is a different discretisation there, by design
    systems, lay, _ = ref_case("multilevel_transmon", golden_meta)
    Z = golden("ref_multilevel_transmon")["Z"]
    so = systems[0]
    G0, Gj = so.G_drift, np.array(so.G_drives)
    prev = None
    for order in (2, 4, 6, 8, 10):
        c = make_ctx(lay, G0, Gj, pade_order=order)
        delta = c.eval(Z)
        close(delta, po.pade_residual(Z, lay, G0, Gj, order), 1e-12)
        r = np.abs(delta).max()
        assert prev is None or r < prev / 20, (order, r, prev)
        prev = r
        c.close()
    assert 1e-9 < prev < 5e-8


def test_general_order_ket_ensemble_and_integrator_interface():
    """Order 8 through the plug-in interface: a ket integrator and a 3-member ensemble with per-member drift."""
    rng = np.random.default_rng(8)
    d, m, N = 3, 2, 6
    n = 2 * d
    Hd = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    Hd = Hd + Hd.conj().T
    Hs = []
    for _ in range(m):
        A = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        Hs.append(A + A.conj().T)
    so = po.quantum_system(0.3 * Hd, Hs, [1.0] * m)
    z_dim = n + 2 + 3 * m
    lay = po.Layout(d=d, m=m, N=N, z_dim=z_dim, x_off=0, u_off=n + 2, dt_off=n, cols=1)
    Z = 0.5 * rng.standard_normal((N, z_dim))
    Z[:, lay.dt_off] = 0.05 + 0.05 * rng.random(N)
    comps = {"ψ̃": Z[:, :n].T, "Δt": Z[:, n][None], "t": Z[:, n + 1][None], "u": Z[:, n + 2 : n + 2 + m].T,
             "du": Z[:, n + 2 + m : n + 2 + 2 * m].T, "ddu": Z[:, n + 2 + 2 * m :].T}  # fmt: skip
    traj = pa.NamedTrajectory(comps, controls=("ddu", "Δt"), timestep="Δt")
    B = pa.BilinearIntegrator(pa.QuantumSystem(so.H_drift, so.H_drives, [1.0] * m), traj, x_name="ψ̃", pade_order=8)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    d2, vals = B.ctx.eval_jac(traj.datavec)
    close(d2, po.pade_residual(Z, lay, G0, Gj, 8), 1e-11)
    close(vals, po.pade_jacobian_values(Z, lay, G0, Gj, 8), 1e-11)
    close(B.f(Z[2, :n], Z[1, :n], Z[1, n + 2 : n + 2 + m], Z[1, n]), d2[n : 2 * n], 1e-11)
    B.close()
    # ensemble
    systems = [po.quantum_system(s_ * 0.5 * po.PAULIS["Z"], [po.PAULIS["X"], po.PAULIS["Y"]], [1.0, 1.0]) for s_ in (1.0, 1.05, 0.95)]
    M, dd, xd = 3, 2, 8
    layE = po.Layout(d=dd, m=2, N=7, z_dim=M * xd + 2 + 2, x_off=0, u_off=M * xd + 2, dt_off=M * xd)
    ZE = 0.3 * rng.standard_normal((7, layE.z_dim))
    ZE[:, layE.dt_off] = 0.1 + 0.1 * rng.random(7)
    trajE = traj_from_Z(pa, ZE, layE, n_members=M)
    BEs = pa.BilinearIntegrator([pa.QuantumSystem(s_.H_drift, s_.H_drives, [1.0, 1.0]) for s_ in systems], trajE, pade_order=6)
    BE = BEs[0].ensemble.fused
    delta, vals = BE.ctx.eval_jac(trajE.datavec)
    per_d, per_j = layE.x_dim * layE.K, po.jac_nnz_per_interval(layE) * layE.K
    for i, s_ in enumerate(systems):
        close(delta[i * per_d : (i + 1) * per_d], po.pade_residual(ZE, layE, s_.G_drift, np.array(s_.G_drives), 6, x_off=i * xd), 1e-11)
        close(vals[i * per_j : (i + 1) * per_j], po.pade_jacobian_values(ZE, layE, s_.G_drift, np.array(s_.G_drives), 6, x_off=i * xd), 1e-11)
    BE.close()


# ---- rollout ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["two_qubit_zoh", "multilevel_transmon", "first_gate"])
def test_rollout_on_reference_solved_trajectories(name, golden, golden_meta):
    """Exact piecewise-constant propagation on the GPU vs scipy expm, and vs the reference's own solved states: a
    trajectory that satisfies x_{k+1} = expv(dt G) x_k to 1e-11 per knot IS its own rollout (to the accumulated
    constraint residual)."""
    systems, lay, _ = ref_case(name, golden_meta)
    Z = golden("ref_" + name)["Z"]
    so = systems[0]
    G0, Gj = so.G_drift, np.array(so.G_drives)
    c = make_ctx(lay, G0, Gj)
    X = c.rollout(Z)[0]
    close(X, po.exact_rollout(Z, lay, G0, Gj), 1e-11)
    states = np.stack([lay.X(Z, k).T.reshape(-1) for k in range(lay.N)])
    dev = np.abs(X - states).max()
    res = np.abs(po.exp_residual(Z, lay, G0, Gj)).max()
    assert dev <= 2 * lay.N * res + 1e-12, (dev, res)
    if name == "two_qubit_zoh":
        assert dev < 1e-9
    # unitarity of the propagated state (G is skew: exp(hG) orthogonal)
    Xk = X[-1].reshape(lay.d, lay.n).T
    U = Xk[: lay.d] + 1j * Xk[lay.d :]
    assert np.abs(U.conj().T @ U - np.eye(lay.d)).max() < 1e-11
    c.close()


def test_rollout_interface_ensemble_ket_and_large_steps():
    """unitary_rollout / unitary_rollout_fidelity through the integrator objects: config 3 synthetic (d = 27), an ensemble
    with per-member drift, a ket, and steps with ||dt G|| >> 1 (several squarings)."""
    so = po.config_system(3)
    Z, lay = po.synthetic_trajectory(so, 12, seed=5)
    Z[:, lay.dt_off] *= np.linspace(1.0, 40.0, lay.N)  # up to ||dt G||_1 ~ 25
    G0, Gj = so.G_drift, np.array(so.G_drives)
    psys = product_system(3)
    traj = traj_from_Z(pa, Z, lay)
    B = pa.BilinearIntegrator(psys, traj, pade_order=4)
    X = pa.unitary_rollout(B, traj)
    assert X.shape == (lay.x_dim, lay.N)
    close(X.T, po.exact_rollout(Z, lay, G0, Gj), 1e-10)
    Ug = pa.iso_vec_to_operator(X[:, -1])
    assert abs(pa.unitary_rollout_fidelity(B, traj, Ug) - 1.0) < 1e-10
    B.close()
    # ensemble: members propagate under their own drift from their own knot-0 state
    rng = np.random.default_rng(2)
    systems = [po.quantum_system(s_ * 0.5 * po.PAULIS["Z"], [po.PAULIS["X"], po.PAULIS["Y"]], [1.0, 1.0]) for s_ in (1.0, 1.05, 0.95)]
    M, xd = 3, 8
    layE = po.Layout(d=2, m=2, N=9, z_dim=M * xd + 2 + 2, x_off=0, u_off=M * xd + 2, dt_off=M * xd)
    ZE = 0.3 * rng.standard_normal((9, layE.z_dim))
    ZE[:, layE.dt_off] = 0.1 + 0.1 * rng.random(9)
    trajE = traj_from_Z(pa, ZE, layE, n_members=M)
    BEs = pa.BilinearIntegrator([pa.QuantumSystem(s_.H_drift, s_.H_drives, [1.0, 1.0]) for s_ in systems], trajE, pade_order=4)
    XE = pa.unitary_rollout(BEs[0].ensemble.fused, trajE)
    for i, s_ in enumerate(systems):
        close(XE[i].T, po.exact_rollout(ZE, layE, s_.G_drift, np.array(s_.G_drives), x_off=i * xd), 1e-11)
        close(pa.unitary_rollout(BEs[i], trajE).T, po.exact_rollout(ZE, layE, s_.G_drift, np.array(s_.G_drives), x_off=i * xd), 1e-11)
    for B_ in BEs:
        B_.close()
    # ket
    d, m, N = 5, 2, 7
    n = 2 * d
    Hd = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    Hs = [(lambda A: A + A.conj().T)(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))) for _ in range(m)]
    sk = po.quantum_system(0.3 * (Hd + Hd.conj().T), Hs, [1.0] * m)
    layK = po.Layout(d=d, m=m, N=N, z_dim=n + 2 + m, x_off=0, u_off=n + 2, dt_off=n, cols=1)
    ZK = 0.5 * rng.standard_normal((N, layK.z_dim))
    ZK[:, layK.dt_off] = 0.05 + 0.05 * rng.random(N)
    ck = make_ctx(layK, sk.G_drift, np.array(sk.G_drives), state_cols=1)
    close(ck.rollout(ZK)[0], po.exact_rollout(ZK, layK, sk.G_drift, np.array(sk.G_drives)), 1e-11)
    ck.close()


def test_multi_ket_integrator():
    """BilinearIntegrator(qtraj::MultiKetTrajectory, N) [REF src/control/integrators.jl:98-117]: one ket integrator per
    state component psi1..psiK, same system, shared controls; rows concatenated ket-major.  Here: one context, K members."""
    rng = np.random.default_rng(21)
    d, m, N, Kk = 4, 2, 8, 3
    n = 2 * d
    Hd = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    Hs = [(lambda A: A + A.conj().T)(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))) for _ in range(m)]
    so = po.quantum_system(0.3 * (Hd + Hd.conj().T), Hs, [1.0] * m)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    z_dim = Kk * n + 2 + m
    lay = po.Layout(d=d, m=m, N=N, z_dim=z_dim, x_off=0, u_off=Kk * n + 2, dt_off=Kk * n, cols=1)
    Z = 0.5 * rng.standard_normal((N, z_dim))
    Z[:, lay.dt_off] = 0.05 + 0.05 * rng.random(N)
    comps = {}
    for i in range(Kk):
        comps["ψ̃%d" % (i + 1)] = Z[:, i * n : (i + 1) * n].T
    comps["Δt"], comps["t"], comps["u"] = Z[:, Kk * n][None], Z[:, Kk * n + 1][None], Z[:, Kk * n + 2 :].T
    traj = pa.NamedTrajectory(comps, controls=("u", "Δt"), timestep="Δt")
    assert np.array_equal(traj.datavec, Z.reshape(-1))
    names = ["ψ̃%d" % (i + 1) for i in range(Kk)]
    B = pa.HipPadeIntegrator(G0, Gj, traj, names, pade_order=4)
    assert B.dim == Kk * n * (N - 1)
    delta, vals = B.ctx.eval_jac(traj.datavec)
    rows, cols = pa.jacobian_structure(B)
    per_d, per_j = n * lay.K, po.jac_nnz_per_interval(lay) * lay.K
    for i in range(Kk):
        close(delta[i * per_d : (i + 1) * per_d], po.pade_residual(Z, lay, G0, Gj, 4, x_off=i * n), 1e-11)
        close(vals[i * per_j : (i + 1) * per_j], po.pade_jacobian_values(Z, lay, G0, Gj, 4, x_off=i * n), 1e-11)
        r0, c0 = po.jac_structure(lay, x_off=i * n)
        assert np.array_equal(rows[i * per_j : (i + 1) * per_j], r0 + i * per_d) and np.array_equal(cols[i * per_j : (i + 1) * per_j], c0)
    mu = rng.standard_normal(B.dim)
    hv = B.ctx.hess(traj.datavec, mu)
    hper = po.hess_nnz_per_interval(lay) * lay.K
    for i in range(Kk):
        close(hv[i * hper : (i + 1) * hper], po.pade4_hessian_values(Z, mu[i * per_d : (i + 1) * per_d].reshape(lay.K, -1), lay, G0, Gj, x_off=i * n), 1e-10)
    B.close()


@pytest.mark.parametrize("base", ["density", "multiket", "multidensity"])
def test_sampling_over_density_and_multi_state_bases(base):
    """BilinearIntegrator(qtraj::SamplingTrajectory, N) for the bases beside unitaries and kets [REF src/control/integrators.jl:181-226]:
    density members evaluate under their compact Lindbladian, MultiKet / MultiDensity members carry a list of sub-states and yield one
    integrator per sub-state (member-major, sub-states in order -- the reference's reduce(vcat, ...)).  Members differ in their
    drift (the robust-control use), so all of them share one batched context; every integrator against the oracle with ITS
    member's generators."""
    rng = np.random.default_rng({"density": 31, "multiket": 32, "multidensity": 33}[base])
    levels, m, N, M = 3, 2, 6, 3
    Hd = rng.standard_normal((levels, levels)) + 1j * rng.standard_normal((levels, levels))
    Hd = 0.3 * (Hd + Hd.conj().T)
    Hs = [(lambda A: A + A.conj().T)(rng.standard_normal((levels, levels)) + 1j * rng.standard_normal((levels, levels))) for _ in range(m)]
    scales = [1.0, 1.05, 0.95]
    dens = base != "multiket"
    subs = 1 if base == "density" else 2
    if dens:
        a = pa.annihilate(levels)
        Ls = [0.3 * a, 0.1 * np.diag(np.arange(levels)).astype(complex)]
        systems = [pa.OpenQuantumSystem(sc * Hd, Hs, [1.0] * m, Ls) for sc in scales]
        gens = [po.compact_lindbladian_generators(sc * Hd, Hs, Ls) for sc in scales]
        xl = levels * levels
        lay_kw = dict(d=0, cols=1, gen=xl)
        stem = "ρ⃗̃"
    else:
        systems = [pa.QuantumSystem(sc * Hd, Hs, [1.0] * m) for sc in scales]
        gens = [(o.G_drift, o.G_drives) for o in (po.quantum_system(sc * Hd, Hs, [1.0] * m) for sc in scales)]
        xl = 2 * levels
        lay_kw = dict(d=levels, cols=1)
        stem = "ψ̃"
    n_states = M * subs
    z_dim = n_states * xl + 2 + m
    Z = 0.5 * rng.standard_normal((N, z_dim))
    Z[:, n_states * xl] = 0.05 + 0.05 * rng.random(N)
    comps, names = {}, []
    for i in range(M):
        mine = []
        for j in range(subs):
            nm = "%s%d_%d" % (stem, j + 1, i + 1) if subs > 1 else "%s%d" % (stem, i + 1)
            q = i * subs + j
            comps[nm] = Z[:, q * xl : (q + 1) * xl].T
            mine.append(nm)
        names.append(mine if subs > 1 else mine[0])
    o = n_states * xl
    comps["Δt"], comps["t"], comps["u"] = Z[:, o][None], Z[:, o + 1][None], Z[:, o + 2 :].T
    traj = pa.NamedTrajectory(comps, controls=("u", "Δt"), timestep="Δt")
    assert np.array_equal(traj.datavec, Z.reshape(-1))
    lay = po.Layout(m=m, N=N, z_dim=z_dim, x_off=0, u_off=o + 2, dt_off=o, **lay_kw)
    Bs = pa.BilinearIntegrator(systems, traj, x_name=names if subs > 1 or dens else names, pade_order=4)
    assert len(Bs) == n_states and len({id(b.ensemble) for b in Bs}) == 1
    flat = [nm for p_ in names for nm in ([p_] if isinstance(p_, str) else p_)]
    for q, B in enumerate(Bs):
        i = q // subs
        G0, Gj = gens[i][0], np.array(gens[i][1])
        assert B.x_name == flat[q] and B.x_dim == xl and B.dim == xl * (N - 1)
        delta = pa.evaluate_(np.zeros(B.dim), B, traj)
        close(delta, po.pade_residual(Z, lay, G0, Gj, 4, x_off=q * xl), 1e-11)
        J = pa.eval_jacobian(B, traj).toarray()
        rows, cols = po.jac_structure(lay, x_off=q * xl)
        Jo = np.zeros_like(J)
        np.add.at(Jo, (rows, cols), po.pade_jacobian_values(Z, lay, G0, Gj, 4, x_off=q * xl).reshape(-1))
        close(J, Jo, 1e-11)
        close(B.f(Z[2, q * xl : (q + 1) * xl], Z[1, q * xl : (q + 1) * xl], Z[1, lay.u_off : lay.u_off + m], Z[1, lay.dt_off]), delta[xl : 2 * xl], 1e-11)
    assert Bs[0].ensemble.launches <= 2  # one fused launch serves every member (residual, then residual + Jacobian)
    for b in Bs:
        b.close()


@pytest.mark.parametrize("d,m,sparse", [(26, 3, False), (23, 6, True), (12, 3, False), (32, 2, True), (27, 6, True)])
def test_work_splits_on_general_shapes(d, m, sparse):
    """The work splits of kernel 3 on shapes that take the run-time-shape instances (general real generators, dense-ish
    or 2-per-row drives, per-trajectory batches, `specialize` off for config 3's shape): round-robin slices, contiguous
    ranges with both roles per workgroup, role split (where the eight-wave matrix role fits LDS) -- all against the C
    oracle and bitwise equal to each other."""
    rng = np.random.default_rng(7 * d + m)
    Bn, N = 2, 4
    lay, G0, Gj, _ = _random_case(d, m, N, rng)
    if sparse:  # two entries per row: the register-resident ELL path
        n = 2 * d
        Gj = np.zeros((m, n, n))
        for l in range(m):
            for i in range(n):
                Gj[l, i, rng.choice(n, 2, replace=False)] = rng.standard_normal(2)
    Zs = []
    for _ in range(Bn):
        Z = rng.standard_normal((N, lay.z_dim))
        Z[:, lay.dt_off] = 0.05 + 0.1 * rng.random(N)
        Zs.append(Z)
    c = make_ctx(lay, G0, Gj, batch=Bn, batch_mode=pa._lib.PCL_BATCH_TRAJ)
    refs = [ref_lib.eval_jac(Z, lay, G0, Gj) for Z in Zs]
    d_ref = np.concatenate([r[0].reshape(-1) for r in refs])
    j_ref = np.concatenate([r[1].reshape(-1) for r in refs])
    c.set_option("kernel_version", 3)
    c.set_option("specialize", 0)
    c.set_option("contiguous", 0)
    d0, v0 = c.eval_jac(np.stack(Zs))
    if d == 32:
        assert c.get_option("last_kernel") // 10 == 2  # double-buffered tiles of kernel 3 do not fit: fallback
        c.close()
        return
    assert c.get_option("last_kernel") == 30
    close(d0, d_ref, 1e-11)
    close(v0, j_ref, 1e-11)
    c.set_option("contiguous", 1)
    for grid, ns in ((0, 0), (5, 0), (0, 128), (0, 100), (9, 4), (256, 255)):
        c.set_option("grid", grid)
        c.set_option("stream_workgroups", ns)
        d1, v1 = c.eval_jac(np.stack(Zs))
        assert c.get_option("last_kernel") == 30
        fits = 2 * (16 + 3 * max(1, min(d, 16 // (2 + m)))) <= 2 * d
        assert (c.get_option("last_stream_workgroups") > 0) == (ns > 0 and fits), (ns, fits)
        assert np.array_equal(d1, d0) and np.array_equal(v1, v0), (grid, ns)
    c.close()


def _config4_share(M, N, first=0):
    """Members first..first+M-1 of BASELINE config 4 in ONE trajectory buffer (layout [U1..UM, dt, t, u, du, ddu], SURVEY 8(d):
    H_drift_i = H_drift + eps_i sum_q a_q' a_q 2 pi, eps_i ~ U(-1e-3, 1e-3), default_rng(2000 + i))."""
    base = po.config_system(3)
    d, m = base.levels, base.n_drives
    xd = 2 * d * d
    a = po.annihilate(3)
    num = sum(po.lift_operator(a.conj().T @ a, q, [3, 3, 3]) for q in (1, 2, 3))
    osys = []
    for i in range(first, first + M):
        eps = np.random.default_rng(2000 + i).uniform(-1e-3, 1e-3)
        osys.append(po.System(base.H_drift + eps * 2 * np.pi * num, base.H_drives, base.drive_bounds))
    Z1, lay1 = po.synthetic_trajectory(base, N, seed=20260929 + 4)
    lay = po.Layout(d=d, m=m, N=N, z_dim=M * xd + 2 + 3 * m, x_off=0, u_off=M * xd + 2, dt_off=M * xd)
    Z = np.zeros((N, lay.z_dim))
    rng = np.random.default_rng(44)
    for i in range(M):
        Z[:, i * xd : (i + 1) * xd] = Z1[:, :xd] + 1e-3 * rng.standard_normal((N, xd))
    Z[:, M * xd :] = Z1[:, xd:]
    comps = {}
    for i in range(M):
        comps["Ũ⃗%d" % (i + 1)] = Z[:, i * xd : (i + 1) * xd].T
    o = M * xd
    comps["Δt"], comps["t"] = Z[:, o][None], Z[:, o + 1][None]
    comps["u"], comps["du"], comps["ddu"] = Z[:, o + 2 : o + 2 + m].T, Z[:, o + 2 + m : o + 2 + 2 * m].T, Z[:, o + 2 + 2 * m :].T
    traj = pa.NamedTrajectory(comps, controls=("ddu", "Δt"), timestep="Δt")
    assert np.array_equal(traj.datavec, Z.reshape(-1))
    psys = [pa.QuantumSystem(s.H_drift, s.H_drives, [b[1] for b in s.drive_bounds]) for s in osys]
    return osys, psys, lay, Z, traj


def _fused_ensemble(psys, traj):
    """One batched context for all members (what the per-member integrators of BilinearIntegrator([...]) share)."""
    names = ["Ũ⃗%d" % (i + 1) for i in range(len(psys))]
    return pa.HipPadeIntegrator(np.array([s.G_drift for s in psys]), psys[0].G_drives_array(), traj, names, pade_order=4)


@pytest.mark.parametrize("M", [4, 8])
def test_config4_share_full_size_ensemble(M):
    """BASELINE config 4's per-GPU share, N = 100: M = 8 members is the shipped share (64 members over 8 GPUs), M = 4 a
    narrower one, through the path `auto` picks at this size (kernel 3, role split, per-member drift tiles) against the
    C oracle per member."""
    osys, psys, lay, Z, traj = _config4_share(M, 100)
    xd = lay.x_dim
    B = _fused_ensemble(psys, traj)
    B.ctx.set_option("host_path", 1)  # the full-values launch (the default host delivery launches the compact kernel)
    delta, vals = B.ctx.eval_jac(traj.datavec)
    _auto_large_launch(B.ctx)
    per_d, per_j = xd * lay.K, po.jac_nnz_per_interval(lay) * lay.K
    for i, s in enumerate(osys):
        d_ref, j_ref = ref_lib.eval_jac(Z, lay, s.G_drift, np.array(s.G_drives), x_off=i * xd)
        close(delta[i * per_d : (i + 1) * per_d], d_ref)
        close(vals[i * per_j : (i + 1) * per_j], j_ref)
    del vals
    mu = np.random.default_rng(5).standard_normal(B.dim)
    hv = B.ctx.hess(traj.datavec, mu)
    hper = po.hess_nnz_per_interval(lay) * lay.K
    for i, s in enumerate(osys):
        h_ref = ref_lib.hess(Z, mu[i * per_d : (i + 1) * per_d].reshape(lay.K, -1), lay, s.G_drift, np.array(s.G_drives), x_off=i * xd)
        close(hv[i * hper : (i + 1) * hper], h_ref, 1e-10)
    B.close()


def test_per_member_drift_fused_roles_is_race_free():
    """Per-member drift tiles on kernel 3's paths WITHOUT a workgroup barrier before the next item's G build (round-robin
    slices and contiguous ranges with both roles per workgroup; two members, short N so that a workgroup's consecutive
    items belong to different members): the four building waves rewrite the G tile concurrently with each other's G^2
    reads, so every position must only ever receive its final value.  Repeated launches must all equal the oracle."""
    osys, psys, lay, Z, traj = _config4_share(2, 7)
    xd = lay.x_dim
    B = _fused_ensemble(psys, traj)
    c = B.ctx
    per_d, per_j = xd * lay.K, po.jac_nnz_per_interval(lay) * lay.K
    refs = [ref_lib.eval_jac(Z, lay, s.G_drift, np.array(s.G_drives), x_off=i * xd) for i, s in enumerate(osys)]
    d_ref = np.concatenate([r[0].reshape(-1) for r in refs])
    j_ref = np.concatenate([r[1].reshape(-1) for r in refs])
    c.set_option("kernel_version", 3)
    first = None
    for contig, sw, grid, cps in ((0, -1, 0, 0), (0, -1, 3, 0), (0, -1, 5, 14), (1, 0, 7, 0), (1, 0, 2, 0), (1, 3, 7, 0)):
        c.set_option("contiguous", contig)
        c.set_option("stream_workgroups", sw)
        c.set_option("grid", grid)
        c.set_option("cols_per_slice", cps)
        for rep in range(25):
            delta, vals = c.eval_jac(traj.datavec)
            assert c.get_option("last_kernel") == 31
            if first is None:
                close(delta, d_ref)
                close(vals, j_ref)
                first = (delta.copy(), vals.copy())
            assert np.array_equal(delta, first[0]) and np.array_equal(vals, first[1]), (contig, sw, grid, cps, rep)
    B.close()


@pytest.mark.parametrize("M,N", [(8, 100), (2, 7), (1, 100)])
def test_fused_reduce_payload_equals_the_separate_kernels(M, N):
    """pcl_eval_jac_merit_dev: the fused kernel's matrix waves form the payload's dot products per state column while the
    column is in LDS.  delta / Jacobian values must be bit-identical to pcl_eval_jac_dev's, the payload equal to
    pcl_merit_grad_dev's (which the oracle's dense J^T lam pins in test_ensemble_merit_and_shared_gradient_on_device)
    to summation-order rounding, for lam = delta and for given multipliers with weights; every work split of a kernel
    (role split at the shipped 8-member share, round-robin slices, contiguous ranges) gives the same bits; and the payload
    itself against the C oracle's tails (J^T lam restricted to u_k, dt_k).  Both carriers: the writer wave of kernel 4 (auto) and
    the MERIT instance of kernel 3."""
    import torch

    osys, psys, lay, Z, traj = _config4_share(M, N)
    B = _fused_ensemble(psys, traj)
    c = B.ctx
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    m, K, xd, d, n = lay.m, lay.K, lay.x_dim, lay.d, 2 * lay.d
    Zd = torch.from_numpy(traj.datavec).cuda()
    dd, vd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda"), torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    dd2, vd2 = torch.empty_like(dd), torch.empty_like(vd)
    ln, sets = c.merit_grad_len()
    assert sets == 1
    rng = np.random.default_rng(11)
    lam = torch.from_numpy(rng.standard_normal(c.n_rows)).cuda()
    w = 1.0 + 0.25 * np.arange(M)
    for kv, kid in ((0, 42), (3, 31)):  # auto: the writer wave of kernel 4; kernel_version 3: the MERIT instance of kernel 3
        c.set_option("kernel_version", kv)
        for k_, v_ in (("contiguous", -1), ("stream_workgroups", -1), ("grid", 0), ("cols_per_slice", 0)):
            c.set_option(k_, v_)
        c.eval_jac_dev(Zd, dd, vd)
        for lam_d, weights in ((None, None), (lam, w)):
            c.set_weights(weights)
            ref = torch.empty(ln, dtype=torch.float64, device="cuda")
            c.merit_grad_dev(dd, lam_d, vd, ref)
            outs = []
            splits = ((-1, -1, 0, 0),) if M == 8 else ((-1, -1, 0, 0), (0, -1, 0, 0), (0, -1, 5, 14), (1, 0 if kv else -1, 7, 0), (1, 3 if kv else -1, 9, 0))
            for contig, sw, grid, cps in splits:
                c.set_option("contiguous", contig)
                c.set_option("stream_workgroups", sw)
                c.set_option("grid", grid)
                c.set_option("cols_per_slice", cps)
                dd2.zero_(), vd2.zero_()
                out = torch.full((ln,), float("nan"), dtype=torch.float64, device="cuda")
                c.eval_jac_merit_dev(Zd, lam_d, dd2, vd2, out)
                torch.cuda.synchronize()
                assert c.get_option("last_kernel") == kid and c.get_option("last_merit_fused") == 1
                assert torch.equal(dd2, dd) and torch.equal(vd2, vd)
                outs.append(out.cpu().numpy())
            for o in outs:
                assert np.array_equal(o, outs[0])  # per-column partials, added in a fixed order: independent of the work split
            r = ref.cpu().numpy()
            scale = max(1.0, float(np.abs(r).max()))
            assert np.abs(outs[0] - r).max() <= 1e-12 * scale, np.abs(outs[0] - r).max()
            # ... and against the C oracle: tails of the oracle's Jacobian dotted with the multipliers
            per_d, per_j = xd * K, po.jac_nnz_per_interval(lay) * K
            g = np.zeros((K, m + 1))
            phi = 0.0
            lam_h = None if lam_d is None else lam_d.cpu().numpy()
            for i, s in enumerate(osys):
                d_ref, j_ref = ref_lib.eval_jac(Z, lay, s.G_drift, np.array(s.G_drives), x_off=i * xd)
                d_ref = np.asarray(d_ref).reshape(K, d, n)
                li = d_ref if lam_h is None else lam_h[i * per_d : (i + 1) * per_d].reshape(K, d, n)
                tails = np.asarray(j_ref).reshape(K, -1)[:, 2 * d * n * n :].reshape(K, d, m + 1, n)
                wi = 1.0 if weights is None else weights[i]
                g += wi * np.einsum("kcln,kcn->kl", tails, li)
                phi += wi * (0.5 if lam_h is None else 1.0) * float((li * d_ref).sum())
            o = outs[0]
            assert abs(o[0] - phi) <= 1e-12 * max(1.0, abs(phi))
            close(o[1 : 1 + K * m].reshape(K, m), g[:, :m], 1e-11)
            close(o[1 + K * m :], g[:, m], 1e-11)
    c.set_weights(None)
    # a member window is not fused: the two separate calls run, same payload layout
    if M > 1:
        for k_, v_ in (("contiguous", -1), ("stream_workgroups", -1), ("grid", 0), ("cols_per_slice", 0)):
            c.set_option(k_, v_)
    B.close()


@pytest.mark.parametrize("M,N", [(8, 100), (3, 9)])
def test_ensemble_step_in_two_launches_equals_the_separate_calls(M, N):
    """pcl_eval_jac_merit_objective_dev (a rank's whole step of a sharded ensemble [REF sampling_problem.jl:381-387]: weighted
    infidelities + regularisers with their gradient, residuals, Jacobian values, reduce payload): the fused kernel + ONE launch whose
    workgroups are the regulariser rows, the terminal infidelities and the payload's finish.  Every output bit for bit what
    pcl_objective_dev + pcl_eval_jac_merit_dev write (themselves pinned on the oracle above); plain and subspace goals, weights,
    given multipliers, repeated calls; a regulariser on a state component keeps the launches apart -- same outputs."""
    import torch

    osys, psys, lay, Z, traj = _config4_share(M, N)
    Bs = pa.BilinearIntegrator(psys, traj, pade_order=4)
    c = Bs[0].ensemble.ctx
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    d = lay.d
    Zd = torch.from_numpy(traj.datavec).cuda()
    ln, _ = c.merit_grad_len()
    rng = np.random.default_rng(5)
    lam = torch.from_numpy(rng.standard_normal(c.n_rows)).cuda()
    w = (1.0 + 0.25 * np.arange(M)) / M
    U = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0]
    names = [B.x_name for B in Bs]
    goals = [U]
    if d == 27:
        sub = pa.get_subspace_indices([[0, 1]] * 3, [3, 3, 3])
        goals.append(pa.EmbeddedOperator(np.linalg.qr(rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8)))[0], sub, [3, 3, 3]))
    for goal in goals:
        for weights, lam_d, extra in ((None, None, None), (w, lam, None), (w, None, names[0])):
            J = pa.UnitaryInfidelityObjective(goal, names, traj, Q=100.0, weights=weights)
            for nm, R in (("u", 1e-2), ("du", 2e-2), ("ddu", 3e-2)):
                J = J + pa.QuadraticRegularizer(nm, traj, R)
            if extra is not None:  # a term on a member's state: the merged launch does not apply
                J = J + pa.QuadraticRegularizer(extra, traj, 1e-3, 0)
            J.bind(Bs)
            ref = [torch.full((k,), float("nan"), dtype=torch.float64, device="cuda") for k in (1, c.z_len, c.n_rows, c.jac_nnz, ln)]
            c.set_option("objective_launches", 2)  # the reference: regulariser launch, infidelity launch, fused kernel, payload finish
            J.value_and_gradient_dev(Zd, ref[0], ref[1])
            c.eval_jac_merit_dev(Zd, lam_d, ref[2], ref[3], ref[4])
            torch.cuda.synchronize()
            assert c.get_option("last_merit_fused") == 1 and c.get_option("last_objective_launches") == 2
            c.set_option("objective_launches", 0)
            one = [torch.full_like(ref[0], float("nan")), torch.full_like(ref[1], float("nan"))]  # the objective alone: ONE launch where it applies
            J.value_and_gradient_dev(Zd, one[0], one[1])
            torch.cuda.synchronize()
            assert c.get_option("last_objective_launches") == (1 if extra is None else 2)
            assert torch.equal(one[0], ref[0]) and torch.equal(one[1], ref[1])
            for rep in range(3):
                out = [torch.full_like(r, float("nan")) for r in ref]
                J.step_dev(Zd, out[0], out[1], out[2], out[3], out[4], lam_dev=lam_d)
                torch.cuda.synchronize()
                assert c.get_option("last_step_launches") == (2 if extra is None else 4)
                for a, b, nm in zip(out, ref, ("value", "gradient", "delta", "values", "payload")):
                    assert torch.equal(a, b), (nm, rep, float((a - b).abs().max()))
            if extra is None and weights is not None and lam_d is not None:
                # the relaxed ("light") arrivals of the one-launch tail under load: 200 steps back to back, the small outputs -- what a last
                # arriver assembles from other workgroups' values -- compared on the device with the two-launch result, every step
                out = [torch.full_like(r, float("nan")) for r in ref]
                bad = torch.zeros((), dtype=torch.int64, device="cuda")
                for rep in range(200):
                    J.step_dev(Zd, out[0], out[1], out[2], out[3], out[4], lam_dev=lam_d)
                    bad += (out[0] != ref[0]).sum() + (out[1] != ref[1]).sum() + (out[4] != ref[4]).sum()
                assert int(bad.item()) == 0
            assert np.isfinite(ref[0].item()) and ref[0].item() > 0
    for B in Bs:
        B.close()
    if N < 20:  # a multistart context: one objective value, one gradient and one payload set per seed
        S = 3
        lay1 = po.Layout.smooth_pulse(d, lay.m, N)
        Zs = [po.synthetic_trajectory(po.config_system(3), N, seed=300 + q)[0] for q in range(S)]
        t1 = traj_from_Z(pa, Zs[0], lay1)
        ms = pa.HipPadeMultistart(osys[0].G_drift, np.array(osys[0].G_drives), t1, S, pade_order=4)
        c = ms.ctx
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        J = (pa.UnitaryInfidelityObjective(U, "Ũ⃗", t1, Q=100.0) + pa.QuadraticRegularizer("u", t1, 1e-2) + pa.QuadraticRegularizer("ddu", t1, 1e-2)).bind(ms)
        Zd = torch.from_numpy(np.stack(Zs)).cuda()
        ln, sets = c.merit_grad_len()
        assert sets == S
        ref = [torch.full((k,), float("nan"), dtype=torch.float64, device="cuda") for k in (S, c.z_len, c.n_rows, c.jac_nnz, ln * S)]
        c.set_option("objective_launches", 2)
        J.value_and_gradient_dev(Zd, ref[0], ref[1])
        c.eval_jac_merit_dev(Zd, None, ref[2], ref[3], ref[4])
        torch.cuda.synchronize()
        c.set_option("objective_launches", 0)
        one = [torch.full_like(ref[0], float("nan")), torch.full_like(ref[1], float("nan"))]
        J.value_and_gradient_dev(Zd, one[0], one[1])
        torch.cuda.synchronize()
        assert c.get_option("last_objective_launches") == 1 and torch.equal(one[0], ref[0]) and torch.equal(one[1], ref[1])
        for rep in range(2):
            out = [torch.full_like(r, float("nan")) for r in ref]
            J.step_dev(Zd, out[0], out[1], out[2], out[3], out[4])
            torch.cuda.synchronize()
            assert c.get_option("last_step_launches") == 2
            for a, b, nm in zip(out, ref, ("value", "gradient", "delta", "values", "payload")):
                assert torch.equal(a, b), ("multistart", nm, rep)
        ms.close()


def test_fused_reduce_payload_other_shapes():
    """Other shapes: where kernel 4 applies (d = 9) its writer wave forms the payload; where `auto` takes kernel 1 (d = 4)
    pcl_eval_jac_merit_dev runs the two separate calls (same outputs); with kernel_version = 3 the MERIT instance of the shape is
    compiled on first use."""
    import torch

    two3 = po.multi_transmon_system([4.0, 4.1], [0.2, 0.21], [[0, 0.01], [0.01, 0]], levels_per_transmon=3, drive_bounds=0.1)  # d = 9
    for so, N, seed, Bn in ((po.config_system(2), 9, 3, 1), (two3, 8, 4, 3)):
        Zs = [po.synthetic_trajectory(so, N, seed=seed + 10 * i) for i in range(Bn)]
        lay = Zs[0][1]
        c = make_ctx(lay, so.G_drift, np.array(so.G_drives), batch=Bn, batch_mode=pa._lib.PCL_BATCH_TRAJ if Bn > 1 else pa._lib.PCL_BATCH_MEMBERS)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        Zd = torch.from_numpy(np.ascontiguousarray(np.stack([z for z, _ in Zs]))).cuda()
        dd, vd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda"), torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        dd2, vd2 = torch.empty_like(dd), torch.empty_like(vd)
        ln, sets = c.merit_grad_len()
        ln *= sets
        lam = torch.from_numpy(np.random.default_rng(seed).standard_normal(c.n_rows)).cuda()
        for kv in (0, 3):
            c.set_option("kernel_version", kv)
            for lam_d in (None, lam):
                ref, out = torch.empty(ln, dtype=torch.float64, device="cuda"), torch.empty(ln, dtype=torch.float64, device="cuda")
                c.eval_jac_dev(Zd, dd, vd)
                c.merit_grad_dev(dd, lam_d, vd, ref)
                c.eval_jac_merit_dev(Zd, lam_d, dd2, vd2, out)
                torch.cuda.synchronize()
                fused = c.get_option("last_merit_fused")
                assert fused == (1 if c.get_option("last_kernel") in (32, 42) else 0)  # kernel 3 compiled for the shape / kernel 4
                assert torch.equal(dd2, dd) and torch.equal(vd2, vd)
                r, o = ref.cpu().numpy(), out.cpu().numpy()
                assert np.abs(o - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max()))
        c.close()


# ---- compact-density variant ---------------------------------------------------------------------------------
@pytest.mark.parametrize("levels,order", [(2, 4), (3, 4), (3, 8), (4, 4), (5, 6)])
def test_density_variant(levels, order):
    """BilinearIntegrator(qtraj::DensityTrajectory, N) [REF src/control/integrators.jl:82-95]: compact density vector
    (levels^2 reals) under the compact Lindbladian generators (general real, odd dimension for odd `levels`:
    PCL_STATE_VECTOR).  delta / Jacobian / structure / Hessian (order 4) / rollout against the oracle; the rollout keeps
    tr(rho) = 1 and rho Hermitian positive (a physical channel)."""
    rng = np.random.default_rng(10 * levels + order)
    n, m, N = levels, 2, 7
    H = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    H = 0.5 * (H + H.conj().T)
    Hs = [(lambda A: A + A.conj().T)(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) for _ in range(m)]
    a = pa.annihilate(n)
    Ls = [0.3 * a, 0.1 * np.diag(np.arange(n)).astype(complex)]
    sys_ = pa.OpenQuantumSystem(H, Hs, [1.0] * m, Ls)
    G0o, Gjo = po.compact_lindbladian_generators(H, Hs, Ls)
    assert np.array_equal(sys_.G_drift, G0o)
    n2 = n * n
    psi = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    psi /= np.linalg.norm(psi)
    rho0 = np.outer(psi, psi.conj())
    times = np.cumsum(np.concatenate(([0.0], 0.05 + 0.05 * rng.random(N - 1))))
    traj = pa.density_trajectory(sys_, 0.5 * rng.standard_normal((m, N)), times, rho0, rho0)
    Z = traj.datavec.reshape(N, traj.dim).copy()
    Z[1:, :n2] += 0.05 * rng.standard_normal((N - 1, n2))  # an infeasible iterate: non-trivial residuals
    traj.update(Z.reshape(-1))
    lay = po.Layout(d=0, m=m, N=N, z_dim=traj.dim, x_off=0, u_off=traj.components["u"].start, dt_off=traj.components["Δt"].start,
                    cols=1, gen=n2)  # fmt: skip
    B = pa.BilinearIntegrator(sys_, traj, pade_order=order)
    assert B.x_name == "ρ⃗̃" and B.x_dim == n2 and B.dim == n2 * (N - 1)
    Gj_arr = np.array(Gjo)
    delta, vals = B.ctx.eval_jac(traj.datavec)
    close(delta, po.pade_residual(Z, lay, G0o, Gj_arr, order), 1e-11)
    close(vals, po.pade_jacobian_values(Z, lay, G0o, Gj_arr, order), 1e-11)
    close(pa.evaluate_(np.zeros(B.dim), B, traj), delta, 1e-13)
    r, c = pa.jacobian_structure(B)
    r0, c0 = po.jac_structure(lay)
    assert np.array_equal(r, r0) and np.array_equal(c, c0)
    J = pa.eval_jacobian(B, traj).toarray()
    close(J, po.pade_jacobian_dense(Z, lay, G0o, Gj_arr, order), 1e-11)
    if order == 4:
        mu = rng.standard_normal((lay.K, lay.x_dim))
        close(B.ctx.hess(traj.datavec, mu), po.pade4_hessian_values(Z, mu, lay, G0o, Gj_arr), 1e-10)
    X = pa.unitary_rollout(B, traj)  # exact propagation of the compact density vector
    close(X.T, po.exact_rollout(Z, lay, G0o, Gj_arr), 1e-11)
    for k in (1, N - 1):
        rho = pa.compact_iso_to_density(X[:, k])
        assert abs(np.trace(rho).real - 1.0) < 1e-12 and np.linalg.eigvalsh(rho).min() > -1e-12
    close(B.f(Z[2, :n2], Z[1, :n2], Z[1, lay.u_off : lay.u_off + m], Z[1, lay.dt_off]), delta[n2 : 2 * n2], 1e-11)
    B.close()


def test_device_entry_points_are_graph_capturable():
    """The *_dev entry points only enqueue kernels on the caller's stream (no allocation, no synchronisation after the
    first call of a configuration), so a caller may capture an iteration's calls in a HIP graph; replay reproduces the
    direct calls bitwise.  (Measured: replay is not faster than three direct launches on this ROCm, so the library
    itself does not build graphs.)"""
    import torch

    so = po.config_system(2)
    Z, lay = po.synthetic_trajectory(so, 30, seed=9)
    c = make_ctx(lay, so.G_drift, np.array(so.G_drives))
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        c.set_stream(stream.cuda_stream)
        Zd = torch.from_numpy(Z.reshape(-1)).cuda()
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
        hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")

        def iteration():
            c.eval_jac_dev(Zd, dd, vd)
            c.hess_dev(Zd, mu, hv)

        iteration()  # first call of each configuration sets kernel attributes / allocates scratch
        stream.synchronize()
        ref = (dd.clone(), vd.clone(), hv.clone())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            iteration()
        for t in (dd, vd, hv):
            t.zero_()
        g.replay()
        stream.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ref, (dd, vd, hv)))
        close(dd.cpu().numpy(), po.pade_residual(Z, lay, so.G_drift, np.array(so.G_drives), 4))
    c.close()


def test_auto_kernel_policy_by_shape():
    """`auto` picks the kernel measured best per shape (scripts/small_d_probe*.py): the one-wave-per-interval
    kernel for small systems (n <= 8), the persistent two-workgroup kernels for general generators, the pattern-compiled kernel 4 for
    sparse exact-iso generators with d >= 9."""
    expect = {1: 52, 2: 52, 3: 42}
    for cfg, kid in expect.items():
        so = po.config_system(cfg)
        Z, lay = po.synthetic_trajectory(so, 8, seed=1)
        c = make_ctx(lay, so.G_drift, np.array(so.G_drives))
        delta, vals = c.eval_jac(Z)
        assert c.get_option("last_kernel") == kid, (cfg, c.get_option("last_kernel"))
        d_ref, j_ref = ref_lib.eval_jac(Z, lay, so.G_drift, np.array(so.G_drives))
        close(delta, d_ref)
        close(vals, j_ref)
        c.close()
    rng = np.random.default_rng(3)
    lay, G0, Gj, Z = _random_case(20, 3, 4, rng)  # general real generators, d = 20: persistent two-workgroup kernel
    c = make_ctx(lay, G0, Gj)
    delta, vals = c.eval_jac(Z)
    assert c.get_option("last_kernel") // 10 == 2
    d_ref, j_ref = ref_lib.eval_jac(Z, lay, G0, Gj)
    close(delta, d_ref, 1e-11)
    close(vals, j_ref, 1e-11)
    c.close()


@pytest.mark.parametrize("d,m,order", [(1, 1, 4), (2, 2, 6), (3, 0, 4), (4, 4, 4), (4, 8, 10), (5, 2, 4), (5, 2, 8), (6, 3, 2), (8, 2, 4), (8, 8, 6)])
def test_small_system_kernel(d, m, order):
    """pcl_fused_small_kernel (kernel_version 5; one wave per interval): both instances (n <= 8 rows: one matrix entry per lane; n <= 16: four),
    random dense generators, every Pade order, residual + Jacobian (full and compact), residual only, TRAJ batches, per-member drifts
    (MEMBERS mode, own state offsets), any grid -- against the oracle, bitwise repeatable; and what `auto` takes at these sizes."""
    rng = np.random.default_rng(97 * d + 7 * m + order)
    lay, G0, Gj, Z = _random_case(d, m, 6, rng, x_off=2)
    d_ref = po.pade_residual(Z, lay, G0, Gj, order).reshape(-1)
    j_ref = po.pade_jacobian_values(Z, lay, G0, Gj, order).reshape(-1)
    c = make_ctx(lay, G0, Gj, pade_order=order)
    delta, vals = c.eval_jac(Z)
    assert c.get_option("last_kernel") == (50 + order // 2 if d <= 4 else (10 if order == 4 else c.get_option("last_kernel")))  # auto: n <= 8 rows
    close(c.eval(Z), d_ref, 1e-11)
    assert c.get_option("last_kernel") == 50 + order // 2  # residual only: both instances
    c.set_option("kernel_version", 5)
    first = None
    for grid, hp in ((0, 1), (1, 1), (3, 1), (1000, 1), (0, 2)):  # (host_path 2: the compact values + expansion)
        c.set_option("grid", grid)
        c.set_option("host_path", hp)
        delta, vals = c.eval_jac(Z)
        assert c.get_option("last_kernel") == 50 + order // 2
        close(delta, d_ref, 1e-11)
        close(vals, j_ref, 1e-11)
        first = (delta, vals) if first is None else first
        assert np.array_equal(delta, first[0]) and np.array_equal(vals, first[1])
        assert np.array_equal(c.eval(Z), delta)
    c.close()
    # three trajectories per launch
    Zs = [rng.standard_normal(Z.shape) for _ in range(3)]
    for Zb in Zs:
        Zb[:, lay.dt_off] = 0.05 + 0.1 * rng.random(lay.N)
    cb = make_ctx(lay, G0, Gj, pade_order=order, batch=3, batch_mode=pa._lib.PCL_BATCH_TRAJ)
    cb.set_option("kernel_version", 5)
    db, vb = cb.eval_jac(np.stack(Zs))
    close(db, np.concatenate([po.pade_residual(Zb, lay, G0, Gj, order).reshape(-1) for Zb in Zs]), 1e-11)
    close(vb, np.concatenate([po.pade_jacobian_values(Zb, lay, G0, Gj, order).reshape(-1) for Zb in Zs]), 1e-11)
    cb.close()
    # two members with their own drifts and state offsets, shared controls (MEMBERS mode)
    xd = 2 * d * d
    layE = po.Layout(d=d, m=m, N=5, z_dim=2 * xd + 2 + m, x_off=0, u_off=2 * xd + 1, dt_off=2 * xd)
    G0s = np.stack([G0, G0 + 0.1 * rng.standard_normal(G0.shape)])
    ZE = rng.standard_normal((5, layE.z_dim))
    ZE[:, layE.dt_off] = 0.05 + 0.1 * rng.random(5)
    ce = pa.integrators._PclContext(d=d, m=m, N=5, z_dim=layE.z_dim, u_off=layE.u_off, dt_off=layE.dt_off, x_offs=[0, xd], G0=G0s, Gj=Gj, batch=2,
                                    batch_mode=pa._lib.PCL_BATCH_MEMBERS, pade_order=order, per_member_G0=True)
    ce.set_option("kernel_version", 5)
    de, ve = ce.eval_jac(ZE)
    assert ce.get_option("last_kernel") == 50 + order // 2
    close(de, np.concatenate([po.pade_residual(ZE, layE, G0s[i], Gj, order, x_off=i * xd).reshape(-1) for i in range(2)]), 1e-11)
    close(ve, np.concatenate([po.pade_jacobian_values(ZE, layE, G0s[i], Gj, order, x_off=i * xd).reshape(-1) for i in range(2)]), 1e-11)
    ce.close()


@pytest.mark.parametrize("levels,batch", [(5, 1), (5, 3), (4, 1)])
def test_other_specialised_shapes(levels, batch):
    """Two 5-level (d = 25) and two 4-level (d = 16) transmons with four drives: the shape-specialised instances of
    kernel 3 (role split at d = 25 for the larger launch) and of the Hessian kernel, against the C oracle."""
    so = po.multi_transmon_system([4.0, 4.1], [0.2, 0.2], [[0, 0.01], [0.01, 0]], levels_per_transmon=levels, drive_bounds=0.1)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    N = 100 if batch > 1 else 12
    Zs, lay = [], None
    for s in range(batch):
        Z, lay = po.synthetic_trajectory(so, N, seed=60 + s)
        Zs.append(Z)
    c = make_ctx(lay, G0, Gj, batch=batch, batch_mode=pa._lib.PCL_BATCH_TRAJ)
    refs = [ref_lib.eval_jac(Z, lay, G0, Gj) for Z in Zs]
    for kv, kid in ((0, 42), (3, 31)):  # auto: the pattern-compiled kernel; kernel 3: its shape-specialised instance
        c.set_option("kernel_version", kv)
        delta, vals = c.eval_jac(np.stack(Zs))
        assert c.get_option("last_kernel") == kid
        if kv == 3 and batch > 1:
            assert c.get_option("last_stream_workgroups") > 0
        close(delta, np.concatenate([r[0].reshape(-1) for r in refs]))
        close(vals, np.concatenate([r[1].reshape(-1) for r in refs]))
    c.set_option("kernel_version", 0)
    mu = np.random.default_rng(1).standard_normal((batch, lay.K, lay.x_dim))
    h_ref = np.concatenate([ref_lib.hess(Z, mu[i], lay, G0, Gj).reshape(-1) for i, Z in enumerate(Zs)])
    hv = c.hess(np.stack(Zs), mu.reshape(-1))
    assert c.get_option("last_hess_kernel") in (6, 82)  # the pattern-compiled kernels (sparse iso generators, 9 <= d <= 32; 82: small launches)
    close(hv, h_ref, 1e-10)
    c.set_option("hess_kernel", 3)
    hv = c.hess(np.stack(Zs), mu.reshape(-1))
    assert c.get_option("last_hess_kernel") == (4 if levels == 5 else 5)  # kernel 3: static instance at d = 25
    close(hv, h_ref, 1e-10)
    c.close()


def test_runtime_compiled_shape_instances():
    """Shapes outside the static instance table are compiled on first use (hiprtc) with compile-time (d, m): fused
    kernel 3 (role split) and Hessian kernel 2.  Same values as the run-time-shape kernels (`jit` = 0) and the C oracle."""
    rng = np.random.default_rng(23)
    d, m, Bn, N = 23, 5, 4, 100
    lay, G0, _, _ = _random_case(d, m, N, rng)
    n = 2 * d
    Gj = np.zeros((m, n, n))
    for l in range(m):  # two entries per row AND per column (circulant pattern), not antisymmetric
        for i in range(n):
            Gj[l, i, (i + l + 1) % n], Gj[l, i, (i - l - 1) % n] = rng.standard_normal(2)
    Zs = []
    for _ in range(Bn):
        Z = 0.3 * rng.standard_normal((N, lay.z_dim))
        Z[:, lay.dt_off] = 0.05 + 0.05 * rng.random(N)
        Zs.append(Z)
    c = make_ctx(lay, 0.2 * G0, Gj, batch=Bn, batch_mode=pa._lib.PCL_BATCH_TRAJ)
    G0 = 0.2 * G0
    loaded = lambda cc: cc.get_option("jit_compiles") + cc.get_option("jit_cache_hits")  # (compiled here, or taken from the user's code-object cache of an earlier run on this box)
    n0 = loaded(c)
    c.set_option("kernel_version", 3)  # fused: the compiled instance replaces the run-time-shape instance of kernel 3
    delta, vals = c.eval_jac(np.stack(Zs))
    assert c.get_option("last_kernel") == 32 and c.get_option("last_stream_workgroups") > 0
    assert loaded(c) == n0 + 1
    refs = [ref_lib.eval_jac(Z, lay, G0, Gj) for Z in Zs]
    d_ref = np.concatenate([r[0].reshape(-1) for r in refs])
    j_ref = np.concatenate([r[1].reshape(-1) for r in refs])
    close(delta, d_ref, 1e-11)
    close(vals, j_ref, 1e-11)
    mu = rng.standard_normal((Bn, lay.K, lay.x_dim))
    hv = c.hess(np.stack(Zs), mu.reshape(-1))
    assert c.get_option("last_hess_kernel") == 5 and loaded(c) == n0 + 2  # one module per template instance
    h_ref = np.concatenate([ref_lib.hess(Z, mu[i], lay, G0, Gj).reshape(-1) for i, Z in enumerate(Zs)])
    close(hv, h_ref, 1e-10)
    c.set_option("jit", 0)
    c.set_option("kernel_version", 0)
    d0, v0 = c.eval_jac(np.stack(Zs))
    assert c.get_option("last_kernel") // 10 == 2
    close(d0, d_ref, 1e-11)
    close(v0, j_ref, 1e-11)
    h0 = c.hess(np.stack(Zs), mu.reshape(-1))
    assert c.get_option("last_hess_kernel") == 2
    close(h0, h_ref, 1e-10)
    c.close()
    c2 = make_ctx(lay, G0, Gj, batch=Bn, batch_mode=pa._lib.PCL_BATCH_TRAJ)  # same shape again: served from the process cache
    c2.set_option("kernel_version", 3)
    c2.eval_jac(np.stack(Zs))
    assert c2.get_option("last_kernel") == 32 and loaded(c2) == n0 + 2
    c2.close()


def test_host_delivery_paths_agree_bitwise():
    """pcl_eval_jac / pcl_jac on host buffers: the compact-over-PCIe + threaded host expansion path (default) and the
    full-values-over-PCIe path deliver bit-identical arrays, for any thread count / chunk count, single trajectory,
    multistart batch and a member window; odd interval counts make half the destination blocks 16-byte (not 32-byte)
    aligned, which exercises the streaming-store head/tail handling."""
    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Bn, N = 3, 6
    Zs = []
    for s in range(Bn):
        Z, lay = po.synthetic_trajectory(so, N, seed=900 + s)
        Zs.append(Z)
    ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Zs[0], lay), Bn, pade_order=4)
    c = ms.ctx
    c.set_option("host_path", 1)
    d_full, v_full = c.eval_jac(np.stack(Zs))
    refs = [ref_lib.eval_jac(Z, lay, G0, Gj) for Z in Zs]
    close(v_full, np.concatenate([r[1].reshape(-1) for r in refs]))
    for threads, chunks in ((0, 4), (1, 1), (2, 8), (3, 3), (7, 2), (64, 8)):
        c.set_option("host_path", 2)
        c.set_option("host_threads", threads)
        c.set_option("host_chunks", chunks)
        for rep in range(3):
            v = np.full(c.jac_nnz + 3, np.nan)[1:-2]  # deliberately 8-byte (not 32-byte) aligned destination
            d = np.empty(c.n_rows)
            c.eval_jac(np.stack(Zs), d, v)
            assert np.array_equal(v, v_full) and np.array_equal(d, d_full), (threads, chunks, rep)
        assert np.array_equal(c.jac(np.stack(Zs)), v_full)
    c.set_member_window(1, 2)
    per = c.jac_per * lay.K
    assert np.array_equal(c.jac(np.stack(Zs)), v_full[per:])
    c.set_member_window(0, Bn)
    ms.close()
    # every kernel family behind the compact mode: small d (kernel 1), general dense generators (kernel 2), higher Pade
    # order (general-order kernel), an ensemble with per-member drift, kets (one column: nothing to expand)
    rng = np.random.default_rng(77)
    cases = []
    for cfg in (1, 2):
        so_ = po.config_system(cfg)
        Zc, layc = po.synthetic_trajectory(so_, 9, seed=cfg)
        cases.append((make_ctx(layc, so_.G_drift, np.array(so_.G_drives)), Zc))
    lay2, G02, Gj2, Z2 = _random_case(20, 3, 4, rng)
    cases.append((make_ctx(lay2, G02, Gj2), Z2))
    lay3, G03, Gj3, Z3 = _random_case(12, 2, 5, rng)
    cases.append((make_ctx(lay3, G03, Gj3, pade_order=6), Z3))
    layk = po.Layout(d=5, m=2, N=6, z_dim=10 + 2 + 2, x_off=0, u_off=12, dt_off=10, cols=1)
    Zk = 0.5 * rng.standard_normal((6, layk.z_dim))
    Zk[:, layk.dt_off] = 0.05 + 0.05 * rng.random(6)
    Gk0, Gkj = rng.standard_normal((10, 10)), rng.standard_normal((2, 10, 10))
    cases.append((make_ctx(layk, Gk0, Gkj, state_cols=1), Zk))
    osys, psys, layE, ZE, trajE = _config4_share(2, 5)
    BE = _fused_ensemble(psys, trajE)
    cases.append((BE.ctx, ZE))
    for cc, Zc in cases:
        cc.set_option("host_path", 1)
        d1, v1 = cc.eval_jac(Zc)
        cc.set_option("host_path", 2)
        cc.set_option("host_threads", 5)
        d2, v2 = cc.eval_jac(Zc)
        assert np.array_equal(d1, d2) and np.array_equal(v1, v2)
        cc.close()


def test_residual_only_kernel():
    """pcl_eval[_dev] runs its own kernel for unitary states with d >= 9 (Ipopt calls eval_constraint alone in every
    line-search trial): every instance (specialised d = 27, run-time shapes, drives with 1 / 2 / many entries per union
    position, per-member drift), any grid, against the oracle and bitwise equal to the residual the fused kernel writes."""
    rng = np.random.default_rng(31)
    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Bn, N = 3, 7
    Zs = []
    for s in range(Bn):
        Z, lay = po.synthetic_trajectory(so, N, seed=400 + s)
        Z[:, lay.dt_off] = 0.08 + 0.04 * rng.random(N)
        Zs.append(Z)
    ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Zs[0], lay), Bn, pade_order=4)
    c = ms.ctx
    ref = np.concatenate([po.pade_residual(Z, lay, G0, Gj, 4).reshape(-1) for Z in Zs])
    c.set_option("eval_kernel", 1)  # the matrix-core residual kernel (any generators)
    for spec in (1, 0):
        c.set_option("specialize", spec)
        for grid in (0, 1, 2, 5, 17, 1000):
            c.set_option("grid", grid)
            d = c.eval(np.stack(Zs))
            assert c.get_option("last_kernel") == 60 + spec
            close(d, ref)
    c.set_option("grid", 0)
    c.set_option("specialize", 1)
    d_fused, _ = c.eval_jac(np.stack(Zs))
    close(c.eval(np.stack(Zs)), d_fused, 1e-13)
    # the pattern-compiled residual kernel (sparse iso generators; auto for launches with more intervals than CUs): one wave per
    # interval, any grid, the oracle's values, bitwise repeatable
    c.set_option("eval_kernel", 2)
    for grid in (0, 1, 2, 5, 1000):
        c.set_option("grid", grid)
        d = c.eval(np.stack(Zs))
        assert c.get_option("last_kernel") == 70
        close(d, ref)
        assert np.array_equal(d, c.eval(np.stack(Zs)))
    # the residual kernel on the products of kernel 4 (resident coefficients, no per-interval table; auto for sparse iso generators)
    # ... one wave per interval, or -- small launches -- four waves per interval with the products in four row ranges (eval_coop): the same bits
    first = None
    for ek in (3, 0):
        c.set_option("eval_kernel", ek)
        for coop in (-1, 0, 1):
            c.set_option("eval_coop", coop)
            for grid in (0, 1, 2, 5, 1000):
                c.set_option("grid", grid)
                d = c.eval(np.stack(Zs))
                assert c.get_option("last_kernel") == 82 and c.get_option("last_eval_coop") == (0 if coop == 0 else 1)
                close(d, ref)
                first = d if first is None else first
                assert np.array_equal(d, first)
    c.set_option("grid", 0)
    c.set_option("eval_kernel", 0)
    c.set_option("eval_coop", -1)
    ms.close()
    big = [po.synthetic_trajectory(so, 100, seed=500 + s)[0] for s in range(3)]  # full size, auto
    layb = po.synthetic_trajectory(so, 100, seed=500)[1]
    msb = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, big[0], layb), 3, pade_order=4)
    db = msb.ctx.eval(np.stack(big))
    assert msb.ctx.get_option("last_kernel") == 82
    close(db, np.concatenate([po.pade_residual(Z, layb, G0, Gj, 4).reshape(-1) for Z in big]))
    msb.close()
    # general real generators (dense drives: the union tables are read from memory), odd d, d = 32 (n = 64: the largest tile)
    for d_, m_ in ((9, 2), (20, 3), (32, 2)):
        lay2, G02, Gj2, Z2 = _random_case(d_, m_, 5, rng)
        c2 = make_ctx(lay2, G02, Gj2)
        close(c2.eval(Z2), po.pade_residual(Z2, lay2, G02, Gj2, 4).reshape(-1), 1e-11)
        assert c2.get_option("last_kernel") == 60
        c2.close()
    # per-member drift tiles (ensemble), windows
    osys, psys, layE, ZE, trajE = _config4_share(3, 6)
    B = _fused_ensemble(psys, trajE)
    B.ctx.set_option("eval_kernel", 1)
    dE = B.ctx.eval(trajE.datavec)
    assert B.ctx.get_option("last_kernel") == 61
    per = layE.x_dim * layE.K
    for i, s in enumerate(osys):
        close(dE[i * per : (i + 1) * per], po.pade_residual(ZE, layE, s.G_drift, np.array(s.G_drives), 4, x_off=i * layE.x_dim).reshape(-1))
    B.ctx.set_member_window(2, 1)
    assert np.array_equal(B.ctx.eval(trajE.datavec), dE[2 * per :])
    B.ctx.set_member_window(0, 3)
    B.ctx.set_option("eval_kernel", 2)  # per-member drifts (union pattern over the members), windows
    d2 = B.ctx.eval(trajE.datavec)
    assert B.ctx.get_option("last_kernel") == 70
    close(d2, dE, 1e-12)
    B.ctx.set_member_window(1, 2)
    assert np.array_equal(B.ctx.eval(trajE.datavec), d2[per:])
    B.ctx.set_member_window(0, 3)
    B.ctx.set_option("eval_kernel", 0)  # auto: the kernel-4 products (the members' drift value classes, some of them streamed)
    d3 = B.ctx.eval(trajE.datavec)
    assert B.ctx.get_option("last_kernel") == 82
    close(d3, dE, 1e-12)
    B.ctx.set_member_window(1, 2)
    assert np.array_equal(B.ctx.eval(trajE.datavec), d3[per:])
    B.close()


@pytest.mark.parametrize("order", [2, 4, 6, 8, 10])
def test_hessian_of_the_lagrangian_at_every_pade_order(order):
    """The general-order Hessian kernel (every diagonal Pade order; at order 4 selected with `general_pade_kernel` as a
    cross-check of the tuned kernels) against the oracle's general-order formulas (pinned by finite differences of the
    Frechet-pinned Jacobian): config 2 (sparse iso drives), a dense random case with an odd state offset, a two-member
    ensemble with per-member drift, the compact-density variant, and several column chunk widths."""
    rng = np.random.default_rng(40 + order)
    so = po.config_system(2)
    Z, lay = po.synthetic_trajectory(so, 6, seed=9, noise=1e-2)
    Z[:, lay.dt_off] = 0.2 + 0.2 * rng.random(6)  # large steps: the high-order terms matter
    G0, Gj = so.G_drift, np.array(so.G_drives)
    mu = rng.standard_normal((lay.K, lay.x_dim))
    ref = po.pade_hessian_values(Z, mu, lay, G0, Gj, order).reshape(-1)
    c = make_ctx(lay, G0, Gj, pade_order=order)
    if order == 4:
        tuned = c.hess(Z, mu.reshape(-1))
        c.set_option("general_pade_kernel", 1)
    h = c.hess(Z, mu.reshape(-1))
    assert c.get_option("last_hess_kernel") == 90 + order // 2
    close(h, ref, 1e-11)
    if order == 4:
        close(h, tuned, 1e-11)
    assert np.array_equal(h, c.hess(Z, mu.reshape(-1)))  # fixed-order sums
    hs, hc = c.hess_structure()
    r0, c0 = po.hess_structure(lay)
    assert np.array_equal(hs, r0) and np.array_equal(hc, c0)
    c.close()
    lay2, G02, Gj2, Z2 = _random_case(5, 3, 4, rng, x_off=3)
    G02, Gj2 = 0.5 * G02, 0.5 * Gj2
    mu2 = rng.standard_normal((lay2.K, lay2.x_dim))
    c2 = make_ctx(lay2, G02, Gj2, pade_order=order)
    if order == 4:
        c2.set_option("general_pade_kernel", 1)
    close(c2.hess(Z2, mu2.reshape(-1)), po.pade_hessian_values(Z2, mu2, lay2, G02, Gj2, order).reshape(-1), 1e-10)
    c2.close()
    # ensemble with per-member drift (member-major rows) and d = 27 (several column chunks)
    osys, psys, layE, ZE, trajE = _config4_share(2, 3)
    ZE = ZE.copy()
    names = ["Ũ⃗1", "Ũ⃗2"]
    BE = pa.HipPadeIntegrator(np.array([s.G_drift for s in psys]), psys[0].G_drives_array(), trajE, names, pade_order=order)
    if order == 4:
        BE.ctx.set_option("general_pade_kernel", 1)
    muE = rng.standard_normal((2, layE.K, layE.x_dim))
    hE = BE.ctx.hess(trajE.datavec, muE.reshape(-1))
    per = po.hess_nnz_per_interval(layE) * layE.K
    for i, s in enumerate(osys):
        close(hE[i * per : (i + 1) * per], po.pade_hessian_values(ZE, muE[i], layE, s.G_drift, np.array(s.G_drives), order, x_off=i * layE.x_dim).reshape(-1), 1e-10)
    BE.close()


@pytest.mark.parametrize("order", [2, 4, 6, 8, 10])
def test_pattern_compiled_general_order_hessian(order):
    """pcl_hess_sparse4_kernel (hess_kernel 7; `auto` until round 4): the Hessian of the Lagrangian at BASELINE config 3 against
    the oracle's general-order formulas (pinned by the complex-step derivative of the Frechet-pinned Jacobian), every column
    slicing and grid (the 28 scalar entries are summed over the slices in registers), repeatable bits, full size at order 8,
    and per-member drifts on a member window."""
    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    for N in ((4, 100) if order == 8 else (4,)):
        Z, lay = po.synthetic_trajectory(so, N, seed=79)
        Z[:, lay.dt_off] = 0.1 + 0.05 * np.random.default_rng(3).random(N)
        mu = np.random.default_rng(5).standard_normal((lay.K, lay.x_dim))
        ref = po.pade_hessian_values(Z, mu, lay, G0, Gj, order).reshape(-1)
        c = make_ctx(lay, G0, Gj, pade_order=order)
        c.set_option("hess_kernel", 7)
        first = {}
        for split in (0, 1, -1):  # one workgroup per interval | two, half of the drive chains each, the scalar entries assembled by the last to arrive | auto
            c.set_option("hess_split", split)
            for cps, grid in (((0, 0), (9, 0), (5, 3), (0, 1)) if N == 4 else ((0, 0),)):
                c.set_option("cols_per_slice", cps)
                c.set_option("grid", grid)
                h = c.hess(Z, mu.reshape(-1))
                assert c.get_option("last_hess_kernel") == 70 + order // 2
                if split >= 0:
                    assert c.get_option("last_hess_split") == split
                close(h, ref, 1e-11)
                for _ in range(3):  # (the arrival counters reset themselves)
                    assert np.array_equal(h, c.hess(Z, mu.reshape(-1)))
                nc_eff = cps if cps else lay.d
                if nc_eff in first:  # the same column slices: the same bits from one and from two workgroups
                    assert np.array_equal(h, first[nc_eff]) or (order == 10 and cps == 0)  # (order 10: one workgroup needs slices of 14 columns there)
                first.setdefault(nc_eff, h)
        c.set_option("hess_split", -1)
        if order != 4:  # auto
            c.set_option("hess_kernel", 0)
            c.set_option("cols_per_slice", 0)
            c.set_option("grid", 0)
            close(c.hess(Z, mu.reshape(-1)), ref, 1e-11)
            assert c.get_option("last_hess_kernel") == 80 + order // 2  # (the column-group kernel: test_column_group_hessian_kernel)
        c.close()
    osys, psys, layE, ZE, trajE = _config4_share(3, 4)
    BE = pa.HipPadeIntegrator(np.array([s.G_drift for s in psys]), psys[0].G_drives_array(), trajE, ["Ũ⃗1", "Ũ⃗2", "Ũ⃗3"], pade_order=order)
    BE.ctx.set_option("hess_kernel", 7)
    muE = np.random.default_rng(6).standard_normal((3, layE.K, layE.x_dim))
    hE = BE.ctx.hess(trajE.datavec, muE.reshape(-1))
    per = po.hess_nnz_per_interval(layE) * layE.K
    for i, s in enumerate(osys):
        close(hE[i * per : (i + 1) * per], po.pade_hessian_values(ZE, muE[i], layE, s.G_drift, np.array(s.G_drives), order, x_off=i * layE.x_dim).reshape(-1), 1e-11)
    BE.ctx.set_member_window(1, 1)
    assert np.array_equal(BE.ctx.hess(trajE.datavec, muE[1].reshape(-1)), hE[per : 2 * per])
    BE.ctx.set_member_window(0, 3)
    BE.ctx.set_option("hess_split", 1)  # two workgroups per interval, per-member drifts
    hE2 = BE.ctx.hess(trajE.datavec, muE.reshape(-1))
    assert BE.ctx.get_option("last_hess_split") == 1
    for i, s in enumerate(osys):
        close(hE2[i * per : (i + 1) * per], po.pade_hessian_values(ZE, muE[i], layE, s.G_drift, np.array(s.G_drives), order, x_off=i * layE.x_dim).reshape(-1), 1e-11)
    BE.close()


@pytest.mark.parametrize("order", [2, 4, 6, 8, 10])
def test_column_group_hessian_kernel(order):
    """pcl_hess_cols_kernel (hess_kernel 8; `auto` at every order but 4): one wave = all m + 1 chains of four state columns, a workgroup of
    its own; the scalar entries assembled by the wave of the interval that arrives last.  BASELINE config 3 against the oracle's
    general-order formulas and the chain-per-wave kernel, repeatable bits (the arrival counters reset themselves), full size at order 8,
    per-member drifts and a member window, 8 seeds in one launch against 8 launches."""
    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    for N in ((4, 100) if order == 8 else (4,)):
        Z, lay = po.synthetic_trajectory(so, N, seed=79)
        Z[:, lay.dt_off] = 0.1 + 0.05 * np.random.default_rng(3).random(N)
        mu = np.random.default_rng(5).standard_normal((lay.K, lay.x_dim))
        ref = po.pade_hessian_values(Z, mu, lay, G0, Gj, order).reshape(-1)
        c = make_ctx(lay, G0, Gj, pade_order=order)
        c.set_option("hess_kernel", 8)
        h = c.hess(Z, mu.reshape(-1))
        assert c.get_option("last_hess_kernel") == 80 + order // 2
        close(h, ref, 1e-11)
        for _ in range(3):
            assert np.array_equal(h, c.hess(Z, mu.reshape(-1)))
        c.set_option("hess_kernel", 7)
        close(h, c.hess(Z, mu.reshape(-1)), 1e-13)
        c.set_option("hess_kernel", 0)
        h0 = c.hess(Z, mu.reshape(-1))
        # (order 4: kernel 6 from two trajectories per launch on, the column-group kernel for launches of at most n_cu / 2 intervals)
        assert c.get_option("last_hess_kernel") == ((82 if 2 * lay.K <= c.get_option("n_cu") else 6) if order == 4 else 80 + order // 2)
        if c.get_option("last_hess_kernel") // 10 == 8:
            assert np.array_equal(h, h0)
        c.close()
    osys, psys, layE, ZE, trajE = _config4_share(3, 4)
    BE = pa.HipPadeIntegrator(np.array([s.G_drift for s in psys]), psys[0].G_drives_array(), trajE, ["Ũ⃗1", "Ũ⃗2", "Ũ⃗3"], pade_order=order)
    BE.ctx.set_option("hess_kernel", 8)
    muE = np.random.default_rng(6).standard_normal((3, layE.K, layE.x_dim))
    hE = BE.ctx.hess(trajE.datavec, muE.reshape(-1))
    assert BE.ctx.get_option("last_hess_kernel") == 80 + order // 2
    per = po.hess_nnz_per_interval(layE) * layE.K
    for i, s in enumerate(osys):
        close(hE[i * per : (i + 1) * per], po.pade_hessian_values(ZE, muE[i], layE, s.G_drift, np.array(s.G_drives), order, x_off=i * layE.x_dim).reshape(-1), 1e-11)
    BE.ctx.set_member_window(1, 1)
    assert np.array_equal(BE.ctx.hess(trajE.datavec, muE[1].reshape(-1)), hE[per : 2 * per])
    BE.ctx.set_member_window(0, 3)
    BE.close()
    if order == 4:  # `auto` at order 4: the column-group kernel for one trajectory (99 intervals <= n_cu / 2), kernel 6 from two on
        N = 100
        Zs = [po.synthetic_trajectory(so, N, seed=310 + i)[0] for i in range(2)]
        lay = po.synthetic_trajectory(so, N, seed=310)[1]
        mus = np.random.default_rng(9).standard_normal((2, lay.K, lay.x_dim))
        c2, c1 = make_ctx(lay, G0, Gj, batch=2, batch_mode=pa._lib.PCL_BATCH_TRAJ), make_ctx(lay, G0, Gj)
        h2 = c2.hess(np.stack(Zs), mus.reshape(-1)).reshape(2, -1)
        assert c2.get_option("last_hess_kernel") == 6
        h1 = c1.hess(Zs[0], mus[0].reshape(-1))
        assert c1.get_option("last_hess_kernel") == (82 if 2 * lay.K <= c1.get_option("n_cu") else 6)
        close(h1, h2[0], 1e-12)
        c2.close()
        c1.close()
    if order in (2, 8):  # seeds: one launch of 8 trajectories = 8 launches of one (a wave's arithmetic depends on its interval and columns only)
        # (order 2 at full length: 5544 short waves per launch -- the launch that showed a store hazard of hand-written 16-byte stores, a few
        #  wrong lines per launch at random places; the stores are the compiler's since)
        N = 100 if order == 2 else 12
        Zs = [po.synthetic_trajectory(so, N, seed=300 + i)[0] for i in range(8)]
        lay = po.synthetic_trajectory(so, N, seed=300)[1]
        mus = np.random.default_rng(8).standard_normal((8, lay.K, lay.x_dim))
        cb = make_ctx(lay, G0, Gj, pade_order=order, batch=8, batch_mode=pa._lib.PCL_BATCH_TRAJ)
        c1 = make_ctx(lay, G0, Gj, pade_order=order)
        for cc in (cb, c1):
            cc.set_option("hess_kernel", 8)
        hb = cb.hess(np.stack(Zs), mus.reshape(-1)).reshape(8, -1)
        for i in range(8):
            assert np.array_equal(hb[i], c1.hess(Zs[i], mus[i].reshape(-1)))
        for _ in range(6 if order == 2 else 1):
            assert np.array_equal(hb.reshape(-1), cb.hess(np.stack(Zs), mus.reshape(-1)))
        cb.close()
        c1.close()


def _random_sparse_iso_system(d, m, rng, n_mags=3):
    """Sparse Hermitian drift and drives (complex entries: the A and the B block of iso(-iH) are both populated), drive
    entries drawn from a few magnitudes with random signs / phases in {1, i}: what the pattern-compiled kernels specialise on."""
    def herm(mask_density, vals):
        H = np.zeros((d, d), dtype=complex)
        for i in range(d):
            for j in range(i, d):
                if rng.random() < mask_density:
                    v = vals()
                    if i == j:
                        H[i, i] = v.real if v.real != 0 else abs(v)
                    else:
                        H[i, j] = v
                        H[j, i] = np.conj(v)
        return H
    density = min(0.18, 2.5 / d)  # (a few hundred entries in the union pattern at every size)
    H0 = herm(density, lambda: complex(rng.standard_normal(), rng.standard_normal()))
    H0 += np.diag(rng.standard_normal(d))
    mags = [1.0, np.sqrt(2.0), 0.37][:n_mags]
    Hd = [herm(density * 0.5, lambda: rng.choice(mags) * rng.choice([1.0, -1.0]) * rng.choice([1.0, 1j])) for _ in range(m)]
    for H in Hd:  # every drive has entries in both blocks
        i, j = rng.choice(d, 2, replace=False)
        H[i, j] += 1j * mags[0]
        H[j, i] -= 1j * mags[0]
        H[i, i] += mags[0]
    return po.G_of_H(H0), np.array([po.G_of_H(H) for H in Hd])


@pytest.mark.parametrize("d,m,Bn,N", [(9, 1, 2, 4), (12, 2, 1, 5), (13, 6, 2, 3), (20, 3, 3, 4), (31, 5, 1, 3), (32, 4, 2, 3)])
def test_pattern_compiled_kernels_random_sparse_systems(d, m, Bn, N):
    """The pattern-compiled Hessian and residual kernels (source generated per system, hiprtc) on random sparse iso systems:
    odd and even d up to 32 (every lane of a half wave in use), 1..6 drives (3..8 waves per workgroup, odd and even counts of
    scalar entries), entries in both blocks of iso(-iH), several magnitudes and signs, more workgroups than intervals and fewer;
    against the oracle, forced and as the `auto` choice, bitwise repeatable."""
    rng = np.random.default_rng(100 * d + m)
    G0, Gj = _random_sparse_iso_system(d, m, rng)
    n, xd = 2 * d, 2 * d * d
    lay = po.Layout(d=d, m=m, N=N, z_dim=xd + 2 + m, x_off=0, u_off=xd + 1, dt_off=xd)
    Zs = []
    for _ in range(Bn):
        Z = 0.4 * rng.standard_normal((N, lay.z_dim))
        Z[:, lay.dt_off] = 0.05 + 0.1 * rng.random(N)
        Zs.append(Z)
    c = make_ctx(lay, G0, Gj, batch=Bn, batch_mode=pa._lib.PCL_BATCH_TRAJ)
    assert c.get_option("iso_structured") == 1
    mu = rng.standard_normal((Bn, lay.K, lay.x_dim))
    h_ref = np.concatenate([po.pade4_hessian_values(Z, mu[i], lay, G0, Gj).reshape(-1) for i, Z in enumerate(Zs)])
    d_ref = np.concatenate([po.pade_residual(Z, lay, G0, Gj, 4).reshape(-1) for Z in Zs])
    Zb = np.stack(Zs)
    h = c.hess(Zb, mu.reshape(-1))
    assert c.get_option("last_hess_kernel") in (6, 82)  # auto (82: the column-group kernel where the any-order generator takes the system, small launches)
    close(h, h_ref, 1e-11)
    c.set_option("hess_kernel", 4)
    c.set_option("eval_kernel", 2)
    for grid in (0, 1, 3, 1000):
        c.set_option("grid", grid)
        h2 = c.hess(Zb, mu.reshape(-1))
        assert c.get_option("last_hess_kernel") == 6
        close(h2, h_ref, 1e-11)
        assert np.array_equal(h2, c.hess(Zb, mu.reshape(-1)))
        dl = c.eval(Zb)
        assert c.get_option("last_kernel") == 70
        close(dl, d_ref, 1e-12)
        assert np.array_equal(dl, c.eval(Zb))
    c.set_option("grid", 0)
    # other controls on the same context, back and forth: the per-interval value tables live at the same addresses in every launch and
    # are read through the (non-coherent) scalar cache -- a launch must never see the previous launch's coefficients
    Zb2 = Zb.copy()
    Zb2[:, :, lay.u_off : lay.u_off + m] *= -1.7
    h_ref2 = np.concatenate([po.pade4_hessian_values(Z, mu[i], lay, G0, Gj).reshape(-1) for i, Z in enumerate(Zb2)])
    d_ref2 = np.concatenate([po.pade_residual(Z, lay, G0, Gj, 4).reshape(-1) for Z in Zb2])
    for rep in range(2):
        close(c.hess(Zb2, mu.reshape(-1)), h_ref2, 1e-11)
        close(c.eval(Zb2), d_ref2, 1e-12)
        close(c.hess(Zb, mu.reshape(-1)), h_ref, 1e-11)
        close(c.eval(Zb), d_ref, 1e-12)
    c.set_option("hess_kernel", 3)  # the matrix-core kernel on the same system
    close(c.hess(Zb, mu.reshape(-1)), h_ref, 1e-11)
    # the any-order kernel on the same system (where its resident coefficients fit), one and two workgroups per interval: odd and even
    # numbers of drives (a two-workgroup launch of m = 3, 5 has a wave without a drive), the same bits from both
    c.set_option("hess_kernel", 7)
    try:
        h7 = c.hess(Zb, mu.reshape(-1))
    except pa.PclError as e:
        assert e.code == pa._lib.PCL_ESHAPE
        h7 = None
    if h7 is not None:
        assert c.get_option("last_hess_kernel") == 72
        close(h7, h_ref, 1e-11)
        for split in (0, 1):
            c.set_option("hess_split", split)
            h8 = c.hess(Zb, mu.reshape(-1))
            assert c.get_option("last_hess_split") == (split if m >= 2 else 0)
            assert np.array_equal(h8, h7) and np.array_equal(h8, c.hess(Zb, mu.reshape(-1)))
        # the column-group kernel on the same system: 16 / 10 / 8 / 6 / 5 / 4 state columns per wave for 1 .. 6 drives, last waves with
        # fewer columns, entries of both blocks and three magnitudes in the gathers' table
        c.set_option("hess_kernel", 8)
        hc = c.hess(Zb, mu.reshape(-1))
        assert c.get_option("last_hess_kernel") == 82
        close(hc, h_ref, 1e-11)
        close(hc, h7, 1e-13)
        assert np.array_equal(hc, c.hess(Zb, mu.reshape(-1)))
        close(c.hess(Zb2, mu.reshape(-1)), h_ref2, 1e-11)
    c.close()


def test_pattern_compiled_kernels_largest_shape_falls_back():
    """d = 32 with 6 drives: the pattern-compiled Hessian kernel's 11 tiles do not fit the 160 KB of LDS -- `auto` runs the
    a matrix-core kernel (same values), forcing kernel 4 fails loudly, and the residual kernel of the same generated source (one tile
    per wave) still runs."""
    d, m, N = 32, 6, 3
    rng = np.random.default_rng(3206)
    G0, Gj = _random_sparse_iso_system(d, m, rng)
    xd = 2 * d * d
    lay = po.Layout(d=d, m=m, N=N, z_dim=xd + 2 + m, x_off=0, u_off=xd + 1, dt_off=xd)
    Z = 0.4 * rng.standard_normal((N, lay.z_dim))
    Z[:, lay.dt_off] = 0.05 + 0.1 * rng.random(N)
    c = make_ctx(lay, G0, Gj)
    mu = rng.standard_normal((lay.K, lay.x_dim))
    h_ref = po.pade4_hessian_values(Z, mu, lay, G0, Gj).reshape(-1)
    close(c.hess(Z, mu.reshape(-1)), h_ref, 1e-11)
    assert c.get_option("last_hess_kernel") != 6  # a matrix-core kernel (the tiles of kernel 3 do not fit either: kernel 2 / 1)
    c.set_option("hess_kernel", 4)
    with pytest.raises(pa.PclError):
        c.hess(Z, mu.reshape(-1))
    c.set_option("hess_kernel", 0)
    c.set_option("eval_kernel", 2)
    close(c.eval(Z), po.pade_residual(Z, lay, G0, Gj, 4).reshape(-1), 1e-12)
    assert c.get_option("last_kernel") == 70
    close(c.hess(Z, mu.reshape(-1)), h_ref, 1e-11)
    c.close()
