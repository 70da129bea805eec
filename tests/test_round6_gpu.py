"""Round-6 GPU tests (``-m gpu``): bench.py starting its own ranks, graph capture of a multi-seed context, the RCCL entry points driven
from the C++ client of the C ABI, the general-order Hessian at the default order."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import piccolo_jl_amd as pa
from helpers import traj_from_Z
from oracle import pade_oracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 1 --force-dist ...` with NO launcher (RANK unset): bench.py re-runs itself under torch.distributed.run
    (self_launch), the rank initialises RCCL, and the parent's stdout carries exactly ONE JSON line with `rccl_ranks` 1 and
    `launcher` "self" -- the branch `python3 bench.py --gpus 8` takes on an 8-GPU node."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PCL_BENCH_SELF_LAUNCHED")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--no-extras", "--no-cpu-baseline"]
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["rccl_ranks"] == 1 and out["launcher"] == "self" and out["n_gpus"] == 1 and out["value"] > 0 and out["steps"] == 3


def test_bare_bench_refuses_more_gpus_than_the_box_has():
    """One GPU on this box: `--gpus 2` ends with status 2 and the reason on stderr, nothing on stdout (no half-started ranks)."""
    import torch

    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert pr.returncode == 2 and pr.stdout.strip() == b"" and b"GPU(s) visible" in pr.stderr, (pr.returncode, pr.stderr[-500:])


def test_multi_seed_launch_is_graph_capturable_while_the_per_array_choice_is_open():
    """ADVICE (round 5): launches of several trajectories sample the static split against the slice tickets per values array with events on the
    launch stream (`v4_tune`).  A stream that is being captured must see none of that: no event record becomes a graph node, no query runs under
    capture; the captured launch takes the array's decided variant, else the tickets, and the sampling state does not move.  Capture on the
    SECOND call of a 4-seed config-3 context (the choice is still open), replay, compare bitwise with a direct launch."""
    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Bn, N = 4, 100
    Zs = [po.synthetic_trajectory(so, N, seed=1000 + i)[0] for i in range(Bn)]
    lay = po.synthetic_trajectory(so, N, seed=1000)[1]
    ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Zs[0], lay), Bn, pade_order=4)
    c = ms.ctx
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        c.set_stream(stream.cuda_stream)
        Zd = torch.from_numpy(np.stack(Zs)).cuda()
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        c.eval_jac_dev(Zd, dd, vd)  # first call: modules, attributes
        stream.synchronize()
        ref = (dd.clone(), vd.clone())
        assert c.get_option("last_v4_tune_choice") == -1  # still sampling
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            c.eval_jac_dev(Zd, dd, vd)
        assert c.get_option("last_v4_ticket") > 0  # undecided array under capture: the tickets
        for _ in range(3):
            dd.zero_()
            vd.zero_()
            g.replay()
            stream.synchronize()
            assert torch.equal(dd, ref[0]) and torch.equal(vd, ref[1])
        for _ in range(16):  # the sampling goes on afterwards, outside the capture, and still decides
            c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
        assert c.get_option("last_v4_tune_choice") in (0, 1) and torch.equal(vd, ref[1])
    ms.close()


@pytest.mark.parametrize("order", [6, 8, 10])
def test_hessian_r_chain_in_front_gives_the_same_bits(order):
    """Round 6: launches of several trajectories form R_{q-2} .. R_1 of every state column ONCE per 14 columns -- in R-chain waves at the head of the same
    launch (`hess_rpre` 1: lane = (half, column), tiles through memory, a self-resetting counter per interval) -- instead of inside each of the interval's seven column-group waves at 8 of 64 lanes (0).  Same arithmetic per column: bitwise the
    same values, `auto` takes the chain waves from more than n_cu / 2 intervals on, and the oracle agrees at 1e-11."""
    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Bn, N = 3, 60
    Zs = [po.synthetic_trajectory(so, N, seed=400 + i)[0] for i in range(Bn)]
    lay = po.synthetic_trajectory(so, N, seed=400)[1]
    ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Zs[0], lay), Bn, pade_order=order)
    c = ms.ctx
    Zd = torch.from_numpy(np.stack(Zs)).cuda()
    mu = np.random.default_rng(12).standard_normal((Bn, lay.K, lay.x_dim))
    mud = torch.from_numpy(mu.reshape(-1)).cuda()
    hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
    out = {}
    for mode in (0, 1, -1, 1, 1):  # (twice more with the chain waves: their counters reset themselves)
        c.set_option("hess_rpre", mode)
        hv.fill_(float("nan"))
        torch.cuda.synchronize()  # (the context launches on a stream of its own: the fill must have finished)
        c.hess_dev(Zd, mud, hv)
        c.sync()
        assert c.get_option("last_hess_kernel") == 80 + order // 2
        assert c.get_option("last_hess_rpre") == (0 if mode < 0 else mode)  # (auto: 177 intervals x 7 waves fit the device's wave slots -> the chain stays inside the waves)
        if mode in out:
            assert torch.equal(out[mode], hv), (mode, int((out[mode] != hv).sum()))  # (repeatable bits in every mode)
        out[mode] = hv.clone()
    # the chain waves (1) add a step's Y term behind the partner's half of the product, the column-group waves' own chain (0) before it:
    # the output vectors do not depend on R (bitwise), the (u,u) entries agree to rounding
    nsc = (lay.m + 1) * (lay.m + 2) // 2
    assert torch.equal(out[0], out[-1])
    a0, a1 = out[0].view(Bn * lay.K, -1), out[1].view(Bn * lay.K, -1)
    assert torch.equal(a0[:, nsc:], a1[:, nsc:])
    assert float((a0[:, :nsc] - a1[:, :nsc]).abs().max()) <= 1e-13 * max(1.0, float(a0[:, :nsc].abs().max()))
    per = c.hess_nnz // Bn
    ref = po.pade_hessian_values(Zs[1], mu[1], lay, G0, Gj, order).reshape(-1)
    got = out[1][per : 2 * per].cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    # one trajectory: the chain stays inside the waves
    ms1 = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Zs[0], lay), 1, pade_order=order)
    h1 = torch.empty(ms1.ctx.hess_nnz, dtype=torch.float64, device="cuda")
    ms1.ctx.hess_dev(Zd[:1].contiguous(), mud[: lay.K * lay.x_dim].contiguous(), h1)
    ms1.ctx.sync()
    assert ms1.ctx.get_option("last_hess_rpre") == 0 and torch.equal(h1, out[0][:per])
    ms1.close()
    ms.close()


@pytest.mark.parametrize("order", [8, 10])
def test_r_chain_waves_on_other_shapes(order):
    """The R-chain waves (`hess_rpre` 1) where `auto` would not take them, against the in-wave chain and the oracle: an ensemble with PER-MEMBER drifts (the chain
    wave reads its member's drift table), a member window on it, a two-transmon system (d = 9, m = 4: six columns per column-group wave, two waves per interval),
    and launches of a single interval."""
    import torch
    from test_parity_gpu import _config4_share

    nsc = lambda lay: (lay.m + 1) * (lay.m + 2) // 2

    def both(c, run):
        outs = []
        for mode in (0, 1):
            c.set_option("hess_rpre", mode)
            outs.append(run())
            assert c.get_option("last_hess_kernel") == 80 + order // 2 and c.get_option("last_hess_rpre") == mode
        return outs

    def same(h0, h1, lay, n_items):
        a0, a1 = h0.reshape(n_items, -1), h1.reshape(n_items, -1)
        assert np.array_equal(a0[:, nsc(lay):], a1[:, nsc(lay):])  # the output vectors do not depend on R
        assert np.abs(a0[:, : nsc(lay)] - a1[:, : nsc(lay)]).max() <= 1e-13 * max(1.0, np.abs(a0[:, : nsc(lay)]).max())

    # ensemble, per-member drifts
    osys, psys, layE, ZE, trajE = _config4_share(3, 5)
    BE = pa.HipPadeIntegrator(np.array([s.G_drift for s in psys]), psys[0].G_drives_array(), trajE, ["Ũ⃗1", "Ũ⃗2", "Ũ⃗3"], pade_order=order)
    BE.ctx.set_option("hess_kernel", 8)
    muE = np.random.default_rng(16).standard_normal((3, layE.K, layE.x_dim))
    h0, h1 = both(BE.ctx, lambda: BE.ctx.hess(trajE.datavec, muE.reshape(-1)))
    same(h0, h1, layE, 3 * layE.K)
    per = po.hess_nnz_per_interval(layE) * layE.K
    for i, s in enumerate(osys):
        ref = po.pade_hessian_values(ZE, muE[i], layE, s.G_drift, np.array(s.G_drives), order, x_off=i * layE.x_dim).reshape(-1)
        assert np.abs(h1[i * per : (i + 1) * per] - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    BE.ctx.set_member_window(2, 1)
    w0, w1 = both(BE.ctx, lambda: BE.ctx.hess(trajE.datavec, muE[2].reshape(-1)))
    assert np.array_equal(w1, h1[2 * per : 3 * per]) and np.array_equal(w0, h0[2 * per : 3 * per])
    BE.ctx.set_member_window(0, 3)
    BE.close()
    # two transmons, three levels each: d = 9, m = 4
    s2 = po.multi_transmon_system([4.0, 4.1], [0.2, 0.21], [[0, 0.02], [0.02, 0]], levels_per_transmon=3, drive_bounds=0.1)
    for N in (2, 9):
        Z, lay = po.synthetic_trajectory(s2, N, seed=21)
        Z[:, lay.dt_off] = 0.1 + 0.1 * np.random.default_rng(4).random(N)
        mu = np.random.default_rng(5).standard_normal((lay.K, lay.x_dim))
        G0, Gj = s2.G_drift, np.array(s2.G_drives)
        ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Z, lay), 1, pade_order=order)
        ms.ctx.set_option("hess_kernel", 8)
        g0, g1 = both(ms.ctx, lambda: ms.ctx.hess(Z[None].copy(), mu.reshape(-1)))
        same(g0, g1, lay, lay.K)
        ref = po.pade_hessian_values(Z, mu, lay, G0, Gj, order).reshape(-1)
        assert np.abs(g1 - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
        ms.close()


def test_default_order_hessian_at_full_size_is_linear_in_the_multipliers():
    """BASELINE config 5's per-GPU share at the order the default constructor picks (order 10; 8 seeds x 99 intervals in one launch, R-chain waves): size-independent
    properties of the Hessian of the Lagrangian where the oracle is too slow to follow -- exact scaling by powers of two in the multipliers (every operation on the way is
    linear in mu, and a factor 2 or 1/4 commutes with every rounding), linearity H(mu1 + mu2) = H(mu1) + H(mu2) to rounding, no dependence of a seed's values on its
    neighbours in the launch (seed 3 alone, with the in-wave chain, agrees to rounding in the (u,u) entries and bitwise elsewhere), and two intervals against the oracle."""
    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Bn, N, order = 8, 100, 10
    Zs = [po.synthetic_trajectory(so, N, seed=1000 + i)[0] for i in range(Bn)]
    lay = po.synthetic_trajectory(so, N, seed=1000)[1]
    ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Zs[0], lay), Bn, pade_order=order)
    c = ms.ctx
    stream = torch.cuda.current_stream()
    c.set_stream(stream.cuda_stream)
    Zd = torch.from_numpy(np.stack(Zs)).cuda()
    g = torch.Generator(device="cuda").manual_seed(7)
    mu1 = torch.randn(c.n_rows, dtype=torch.float64, device="cuda", generator=g)
    mu2 = torch.randn(c.n_rows, dtype=torch.float64, device="cuda", generator=g)

    def H(mu):
        out = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
        c.hess_dev(Zd, mu.contiguous(), out)
        torch.cuda.synchronize()
        return out

    h1, h2 = H(mu1), H(mu2)
    assert c.get_option("last_hess_kernel") == 85 and c.get_option("last_hess_rpre") == 1
    assert torch.isfinite(h1).all() and float(h1.abs().max()) > 0
    assert torch.equal(H(2.0 * mu1), 2.0 * h1) and torch.equal(H(0.25 * mu1), 0.25 * h1)
    h12 = H(mu1 + mu2)
    scale = float(torch.maximum(h1.abs(), h2.abs()).max())
    assert float((h12 - (h1 + h2)).abs().max()) <= 1e-12 * scale
    # seed 3 alone (one trajectory per launch: the chain inside the waves)
    per = c.hess_nnz // Bn
    rows = c.n_rows // Bn
    ms1 = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Zs[3], lay), 1, pade_order=order)
    ms1.ctx.set_stream(stream.cuda_stream)
    s3 = torch.empty(per, dtype=torch.float64, device="cuda")
    ms1.ctx.hess_dev(Zd[3:4].contiguous(), mu1[3 * rows : 4 * rows].contiguous(), s3)
    torch.cuda.synchronize()
    assert ms1.ctx.get_option("last_hess_rpre") == 0
    nsc = (lay.m + 1) * (lay.m + 2) // 2
    a, b = h1[3 * per : 4 * per].view(lay.K, -1), s3.view(lay.K, -1)
    assert torch.equal(a[:, nsc:], b[:, nsc:]) and float((a[:, :nsc] - b[:, :nsc]).abs().max()) <= 1e-13 * max(1.0, float(a[:, :nsc].abs().max()))
    ms1.close()
    # two intervals of seed 5 against the oracle (a 3-knot cut of the trajectory: intervals 40, 41)
    Zc = Zs[5][40:43].copy()
    layc = po.Layout.smooth_pulse(lay.d, lay.m, 3)
    muc = mu1[5 * rows : 6 * rows].view(lay.K, -1)[40:42].cpu().numpy()
    ref = po.pade_hessian_values(Zc, muc, layc, G0, Gj, order)
    got = h1[5 * per : 6 * per].view(lay.K, -1)[40:42].cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    ms.close()


@pytest.mark.parametrize("order", [8, 10])
def test_config3_full_size_at_the_orders_that_match_the_exp_constraint(order):
    """BASELINE config 3 at its real size (d = 27, N = 100) at the orders the policy picks (10 on config 3's bounds, 8 at the synthetic trajectories' scale) against the
    numpy oracle's general-order formulas: the residual, ALL 16,599,330 Jacobian values, the Hessian of the Lagrangian of one trajectory (the chain inside the waves) and
    of the same trajectory as seed 2 of a 4-seed launch (R-chain waves) -- the C oracle is order 4 only, so until round 6 these orders were checked at N = 4."""
    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    N = 100
    Zs = [po.synthetic_trajectory(so, N, seed=1000 + i)[0] for i in range(4)]
    lay = po.synthetic_trajectory(so, N, seed=1000)[1]
    Z = Zs[2]
    d_ref = po.pade_residual(Z, lay, G0, Gj, order).reshape(-1)
    j_ref = po.pade_jacobian_values(Z, lay, G0, Gj, order).reshape(-1)
    mu = np.random.default_rng(31).standard_normal((lay.K, lay.x_dim))
    h_ref = po.pade_hessian_values(Z, mu, lay, G0, Gj, order).reshape(-1)
    tol = lambda got, ref, t: np.abs(got - ref).max() <= t * max(1.0, np.abs(ref).max())
    ms1 = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Z, lay), 1, pade_order=order)
    c = ms1.ctx
    assert c.jac_nnz == 16599330
    delta, vals = c.eval_jac(Z[None].copy())
    assert c.get_option("last_kernel") == 40 + order // 2
    assert tol(delta, d_ref, 1e-12) and tol(vals, j_ref, 1e-12)
    h = c.hess(Z[None].copy(), mu.reshape(-1))
    assert c.get_option("last_hess_kernel") == 80 + order // 2 and c.get_option("last_hess_rpre") == 0 and tol(h, h_ref, 1e-11)
    ms1.close()
    ms4 = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Zs[0], lay), 4, pade_order=order)
    mu4 = np.random.default_rng(32).standard_normal((4, lay.K, lay.x_dim))
    mu4[2] = mu
    h4 = ms4.ctx.hess(np.stack(Zs), mu4.reshape(-1)).reshape(4, -1)
    assert ms4.ctx.get_option("last_hess_rpre") == 1 and tol(h4[2], h_ref, 1e-11)
    d4, v4 = ms4.ctx.eval_jac(np.stack(Zs))
    assert np.array_equal(v4.reshape(4, -1)[2], vals) and np.array_equal(d4.reshape(4, -1)[2], delta)  # (a seed's values do not depend on the launch it is in)
    ms4.close()


def test_config4_share_at_the_default_order():
    """BASELINE config 4's per-GPU share (8 perturbed-drift members in ONE trajectory buffer, shared controls, N = 100) at order 10 -- the order the default
    constructor picks on config 3 / 4 / 5's bounds: residual + Jacobian of two sampled members and the Hessian of the Lagrangian of one (member window; the launch of
    all eight: R-chain waves reading their member's drift table) against the numpy oracle with the member's own drift."""
    from piccolo_jl_amd import synthetic

    M, N, order = 8, 100, 10
    members = synthetic.config4_members(0, M)
    traj = synthetic.synthetic_ensemble(members, N, seed=20260929 + 4)
    d, m = members[0].levels, members[0].n_drives
    xd = 2 * d * d
    names = ["Ũ⃗%d" % (i + 1) for i in range(M)]
    Gj = members[0].G_drives_array()
    B = pa.HipPadeIntegrator(np.array([s.G_drift for s in members]), Gj, traj, names, pade_order=order)
    c = B.ctx
    delta, vals = c.eval_jac(traj.datavec)
    assert c.get_option("last_kernel") == 45
    per_v, per_d = c.jac_nnz // M, c.n_rows // M
    Z2 = traj.datavec.reshape(N, traj.dim)
    lay = po.Layout(d=d, m=m, N=N, z_dim=traj.dim, x_off=0, u_off=traj.components["u"].start, dt_off=traj.components["Δt"].start)
    tol = lambda got, ref, t: np.abs(got - ref).max() <= t * max(1.0, np.abs(ref).max())
    for i in (1, 6):
        d_ref = po.pade_residual(Z2, lay, members[i].G_drift, np.array(Gj), order, x_off=i * xd).reshape(-1)
        j_ref = po.pade_jacobian_values(Z2, lay, members[i].G_drift, np.array(Gj), order, x_off=i * xd).reshape(-1)
        assert tol(delta[i * per_d : (i + 1) * per_d], d_ref, 1e-12) and tol(vals[i * per_v : (i + 1) * per_v], j_ref, 1e-12)
    mu = np.random.default_rng(41).standard_normal((M, lay.K, lay.x_dim))
    h = c.hess(traj.datavec, mu.reshape(-1)).reshape(M, -1)
    assert c.get_option("last_hess_kernel") == 85 and c.get_option("last_hess_rpre") == 1
    h_ref = po.pade_hessian_values(Z2, mu[6], lay, members[6].G_drift, np.array(Gj), order, x_off=6 * xd).reshape(-1)
    assert tol(h[6], h_ref, 1e-11)
    c.set_member_window(6, 1)
    hw = c.hess(traj.datavec, mu[6].reshape(-1))
    assert c.get_option("last_hess_rpre") == 0 and tol(hw, h_ref, 1e-11)
    c.set_member_window(0, M)
    B.close()


@pytest.mark.parametrize("order", [2, 4, 6, 8, 10])
def test_one_trajectory_hessian_with_a_chain_wave_and_a_contribution_wave(order):
    """Round 6: launches of at most n_cu / 2 intervals (one trajectory) give every column group a workgroup of TWO waves -- the chain (product, gathers) in one, what a
    level contributes (accumulation, Y dot, gather-dot, sums; the R chain at the start; the output at the end) in the other, two buffers of chain slots and two LDS
    words between them (pcl_hess_cols_pair_kernel).  The phases' text is shared with the one-wave kernel: BITWISE the same values, repeatably; `auto` takes it for one
    trajectory and not for eight; the oracle agrees."""
    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    N = 100
    Z, lay = po.synthetic_trajectory(so, N, seed=1003)
    Z[:, lay.dt_off] = 0.08 + 0.04 * np.random.default_rng(1).random(N)
    mu = np.random.default_rng(51).standard_normal((lay.K, lay.x_dim))
    ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Z, lay), 1, pade_order=order)
    c = ms.ctx
    c.set_option("hess_kernel", 8)
    out = {}
    for mode in (0, 1, -1, 1, 1):
        c.set_option("hess_pair", mode)
        h = c.hess(Z[None].copy(), mu.reshape(-1))
        assert c.get_option("last_hess_kernel") == 80 + order // 2 and c.get_option("last_hess_pair") == (0 if mode == 0 or (mode < 0 and order == 2) else 1)
        if mode in out:
            assert np.array_equal(out[mode], h)
        out[mode] = h.copy()
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[1], out[-1])  # (auto: the pair from order 4 on)
    ref = po.pade_hessian_values(Z, mu, lay, G0, Gj, order).reshape(-1)
    assert np.abs(out[1] - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    ms.close()
    if order == 8:  # eight trajectories per launch: one wave per column group (and the R-chain waves), as before
        ms8 = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Z, lay), 8, pade_order=order)
        h8 = ms8.ctx.hess(np.stack([Z] * 8), np.tile(mu.reshape(-1), 8))
        assert ms8.ctx.get_option("last_hess_pair") == 0 and ms8.ctx.get_option("last_hess_rpre") == 1
        nsc = (lay.m + 1) * (lay.m + 2) // 2
        a, b = h8.reshape(8 * lay.K, -1), np.tile(out[1].reshape(lay.K, -1), (8, 1))
        assert np.array_equal(a[:, nsc:], b[:, nsc:]) and np.abs(a[:, :nsc] - b[:, :nsc]).max() <= 1e-13 * max(1.0, np.abs(b[:, :nsc]).max())
        ms8.close()


@pytest.mark.parametrize("order", [4, 10])
def test_two_wave_hessian_kernel_on_other_shapes(order):
    """pcl_hess_cols_pair_kernel where config 3 does not take it: an ensemble with PER-MEMBER drifts (both waves read their member's tables), a member window on it,
    a two-transmon system (d = 9, m = 4: six columns per column group, two groups per interval) at one and at eight intervals.  Bitwise the one-wave kernel's
    values on every shape (`hess_pair` 0 / 1; `auto` = 1 on all of them), and the oracle's at 1e-11."""
    from test_parity_gpu import _config4_share

    def both(c, run):
        outs = []
        for mode in (0, 1, -1):
            c.set_option("hess_pair", mode)
            outs.append(run())
            assert c.get_option("last_hess_kernel") == 80 + order // 2 and c.get_option("last_hess_pair") == (0 if mode == 0 else 1) and c.get_option("last_hess_rpre") == 0
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])
        return outs[1]

    osys, psys, layE, ZE, trajE = _config4_share(3, 5)
    BE = pa.HipPadeIntegrator(np.array([s.G_drift for s in psys]), psys[0].G_drives_array(), trajE, ["Ũ⃗1", "Ũ⃗2", "Ũ⃗3"], pade_order=order)
    BE.ctx.set_option("hess_kernel", 8)
    muE = np.random.default_rng(16).standard_normal((3, layE.K, layE.x_dim))
    h = both(BE.ctx, lambda: BE.ctx.hess(trajE.datavec, muE.reshape(-1)))
    per = po.hess_nnz_per_interval(layE) * layE.K
    for i, s in enumerate(osys):
        ref = po.pade_hessian_values(ZE, muE[i], layE, s.G_drift, np.array(s.G_drives), order, x_off=i * layE.x_dim).reshape(-1)
        assert np.abs(h[i * per : (i + 1) * per] - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    BE.ctx.set_member_window(1, 2)
    w = both(BE.ctx, lambda: BE.ctx.hess(trajE.datavec, muE[1:].reshape(-1)))
    assert np.array_equal(w, h[per:])
    BE.ctx.set_member_window(0, 3)
    BE.close()
    s2 = po.multi_transmon_system([4.0, 4.1], [0.2, 0.21], [[0, 0.02], [0.02, 0]], levels_per_transmon=3, drive_bounds=0.1)
    for N in (2, 9):
        Z, lay = po.synthetic_trajectory(s2, N, seed=21)
        Z[:, lay.dt_off] = 0.1 + 0.1 * np.random.default_rng(4).random(N)
        mu = np.random.default_rng(5).standard_normal((lay.K, lay.x_dim))
        G0, Gj = s2.G_drift, np.array(s2.G_drives)
        ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Z, lay), 1, pade_order=order)
        ms.ctx.set_option("hess_kernel", 8)
        g = both(ms.ctx, lambda: ms.ctx.hess(Z[None].copy(), mu.reshape(-1)))
        ref = po.pade_hessian_values(Z, mu, lay, G0, Gj, order).reshape(-1)
        assert np.abs(g - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
        ms.close()


def test_r_tile_copy_with_an_odd_tile_size_five_drives():
    """The R tiles travel as whole 16-byte pieces (global_load_lds_dwordx4): with five drives a column group has 5 columns x 55 doubles -- an odd count, padded to
    even in memory and in LDS (HC_RTS).  Three transmons with one drive dropped (d = 27, m = 5), orders 8 and 10, R-chain waves forced, against the in-wave chain
    and the oracle."""
    so = po.config_system(3)
    s5 = po.System(so.H_drift, list(so.H_drives)[:5], list(so.drive_bounds)[:5], so.subsystem_levels)
    G0, Gj = s5.G_drift, np.array(s5.G_drives)
    for order in (8, 10):
        N = 5
        Z5, lay5 = po.synthetic_trajectory(s5, N, seed=77)
        Z5[:, lay5.dt_off] = 0.08 + 0.04 * np.random.default_rng(2).random(N)
        mu = np.random.default_rng(9).standard_normal((lay5.K, lay5.x_dim))
        ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Z5, lay5), 1, pade_order=order)
        c = ms.ctx
        c.set_option("hess_kernel", 8)
        c.set_option("hess_pair", 0)
        outs = []
        for mode in (0, 1, 1):
            c.set_option("hess_rpre", mode)
            outs.append(c.hess(Z5[None].copy(), mu.reshape(-1)))
            assert c.get_option("last_hess_kernel") == 80 + order // 2 and c.get_option("last_hess_rpre") == mode
        assert np.array_equal(outs[1], outs[2])
        nsc = (lay5.m + 1) * (lay5.m + 2) // 2
        a0, a1 = outs[0].reshape(lay5.K, -1), outs[1].reshape(lay5.K, -1)
        assert np.array_equal(a0[:, nsc:], a1[:, nsc:]) and np.abs(a0[:, :nsc] - a1[:, :nsc]).max() <= 1e-13 * max(1.0, np.abs(a0[:, :nsc]).max())
        ref = po.pade_hessian_values(Z5, mu, lay5, G0, Gj, order).reshape(-1)
        assert np.abs(outs[1] - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
        ms.close()


def test_bench_with_two_ranks_on_one_device():
    """bench.py with WORLD_SIZE = 2 for real -- the branches only a world > 1 takes: 64 / N seeds per rank (strong scaling), the ensemble step with its
    all-reduce beside the multistart line, the max over ranks, one JSON line from rank 0.  RCCL refuses two ranks on one device, so the test hook
    (PCL_BENCH_TEST_ONE_DEVICE: every rank on device 0, gloo) carries the collective; the kernels and the script's logic are the production ones."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.update(PCL_BENCH_TEST_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]  # fmt: skip
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1500)
    assert pr.returncode == 0, pr.stderr.decode()[-3000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["collective_backend"].startswith("gloo")
    cfg = out["config"]
    assert cfg["workload_id"] == "multistart" and cfg["units_per_gpu"] == 32 and cfg["total_units"] == 64
    assert out["value"] > 0 and out["steps"] == 3 and 0 < out["roofline"]["frac"] < 1
    es = out["ensemble_share"]
    assert es["members_per_gpu"] == 32 and es["members_total"] == 64 and es["all_reduce"] is True and es["rccl_ranks"] == 2 and es["all_reduce_us"] > 0
    assert out["rccl_ranks"] == 2
