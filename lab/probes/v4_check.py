#!/usr/bin/env python3
"""Pattern-compiled fused kernel (kernel_version 4): parity against the oracle at BASELINE config 3 (orders 2..10, round-robin and
contiguous splits, compact form), then rates for 1 and 8 trajectories per launch next to the default kernels."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
from oracle import pade_oracle as po

quick = "--quick" in sys.argv
so = po.config_system(3)
G0, Gj = so.G_drift, np.array(so.G_drives)


def ctx(lay, order, batch=1, mode=None):
    return pa.integrators._PclContext(d=lay.d, m=lay.m, N=lay.N, z_dim=lay.z_dim, u_off=lay.u_off, dt_off=lay.dt_off, x_offs=[lay.x_off], G0=G0, Gj=Gj,
                                      batch=batch, batch_mode=pa._lib.PCL_BATCH_MEMBERS if mode is None else mode, pade_order=order)


def err(a, b):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


ok = True
for N in ((5,) if quick else (5, 100)):
    Z, lay = po.synthetic_trajectory(so, N, seed=77)
    Z[:, lay.dt_off] = 0.1 + 0.05 * np.random.default_rng(3).random(N)
    for order in (4, 2, 8) if N == 5 else (4, 8):
        d_ref = po.pade_residual(Z, lay, G0, Gj, order)
        j_ref = po.pade_jacobian_values(Z, lay, G0, Gj, order)
        c = ctx(lay, order)
        c.set_option("kernel_version", 4)
        first = None
        for contig, cps, hp, tm in ((-1, 0, 1, 0), (0, 0, 1, 0), (1, 0, 1, 0), (0, 5, 1, 0), (0, 27, 1, 0), (0, 1, 1, 0), (-1, 0, 2, 0), (1, 0, 1, 3), (0, 5, 1, 3), (0, 0, 1, 2), (1, 0, 1, 1)):
            c.set_option("contiguous", contig)
            c.set_option("cols_per_slice", cps)
            c.set_option("v4_tail_mode", tm)
            c.set_option("host_path", hp)  # 1: full values (the streamed blocks), 2: compact values + host expansion
            t0 = time.time()
            delta, vals = c.eval_jac(Z)
            dt = time.time() - t0
            ed, ej = err(delta, d_ref), err(vals, j_ref)
            K = lay.K
            V = vals.reshape(K, -1)
            nb = 2 * lay.d * lay.n * lay.n
            eb = err(V[:, :nb], j_ref.reshape(K, -1)[:, :nb])
            et = err(V[:, nb:], j_ref.reshape(K, -1)[:, nb:])
            same = first is None or (np.array_equal(delta, first[0]) and np.array_equal(vals, first[1]))
            if first is None:
                first = (delta, vals)
            good = ed < 1e-11 and ej < 1e-11 and same and c.get_option("last_kernel") == 40 + order // 2
            ok &= good
            print("N=%3d order %2d contig %2d cps %2d host_path %d tail_mode %d: delta %.1e blocks %.1e tails %.1e bitwise-equal-splits %s kernel %d (%.2f s) %s" % (
                N, order, contig, cps, hp, tm, ed, eb, et, same, c.get_option("last_kernel"), dt, "ok" if good else "FAIL"), flush=True)
        c.close()
# ensemble: per-member drifts (config 4's members: 27 drift classes, some of them streamed), members of one trajectory buffer
from oracle import ref_lib
for M, N in ((3, 6), (8, 100)):
    psys = synthetic.config4_members(0, M)
    osys = [po.System(s.H_drift, s.H_drives, s.drive_bounds) for s in psys]
    traj = synthetic.synthetic_ensemble(psys, N, seed=20260929 + 4)
    xd = 2 * 27 * 27
    lay = po.Layout(d=27, m=6, N=N, z_dim=traj.dim, x_off=0, u_off=traj.components["u"].start, dt_off=traj.components["Δt"].start)
    Z = traj.datavec.reshape(N, traj.dim)
    names = ["Ũ⃗%d" % (i + 1) for i in range(M)]
    Bi = pa.HipPadeIntegrator(np.array([s.G_drift for s in psys]), psys[0].G_drives_array(), traj, names, pade_order=4)
    c = Bi.ctx
    c.set_option("host_path", 1)
    per_d, per_j = xd * lay.K, po.jac_nnz_per_interval(lay) * lay.K
    refs = [ref_lib.eval_jac(Z, lay, s.G_drift, np.array(s.G_drives), x_off=i * xd) for i, s in enumerate(osys)]
    d_ref = np.concatenate([r[0].reshape(-1) for r in refs])
    j_ref = np.concatenate([r[1].reshape(-1) for r in refs])
    c.set_option("kernel_version", 4)
    for contig, tm in ((-1, 0), (1, 3), (0, 3)):
        c.set_option("contiguous", contig)
        c.set_option("v4_tail_mode", tm)
        delta, vals = c.eval_jac(traj.datavec)
        good = err(delta, d_ref) < 1e-12 and err(vals, j_ref) < 1e-12 and c.get_option("last_kernel") == 42
        ok &= good
        print("ensemble M=%d N=%d contig %d tail_mode %d: delta %.1e jac %.1e kernel %d %s" % (M, N, contig, tm, err(delta, d_ref), err(vals, j_ref), c.get_option("last_kernel"), "ok" if good else "FAIL"), flush=True)
    c.set_member_window(1, 1)  # one member of the ensemble (what a per-member Hessian / Jacobian call runs on)
    delta, vals = c.eval_jac(traj.datavec)
    good = err(delta, d_ref[per_d:2 * per_d]) < 1e-12 and err(vals, j_ref[per_j:2 * per_j]) < 1e-12
    ok &= good
    print("ensemble M=%d member window 1: delta %.1e jac %.1e %s" % (M, err(delta, d_ref[per_d:2 * per_d]), err(vals, j_ref[per_j:2 * per_j]), "ok" if good else "FAIL"), flush=True)
    Bi.close()
print("PARITY", "OK" if ok else "FAILED", flush=True)

# ---- rates ----------------------------------------------------------------------------------------------------------------
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for B in (1, 8):
        trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
        t0 = trajs[0]
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        for order in (4, 8, 10):
            c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                           dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                           Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
            c.set_stream(stream.cuda_stream)
            dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
            vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
            for kv, contig, tm in ((0, -1, 0), (4, -1, 0), (4, -1, 1), (4, -1, 2), (4, -1, 3), (4, 1, 0), (4, 1, 3)):
                if (B == 8 and contig == 1) or (order == 10 and tm in (1, 2)):
                    continue
                c.set_option("kernel_version", kv)
                c.set_option("contiguous", contig)
                c.set_option("v4_tail_mode", tm)
                for _ in range(5):
                    c.eval_jac_dev(Zd, dd, vd)
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                reps = 50 if B == 1 else 20
                for _ in range(reps):
                    c.eval_jac_dev(Zd, dd, vd)
                e1.record(stream)
                stream.synchronize()
                us = e0.elapsed_time(e1) / reps * 1e3
                print("B=%d order %2d kernel_version %d contig %2d tail_mode %d: %.1f us/launch (%.2f TB/s algorithmic) kernel id %d" % (
                    B, order, kv, contig, tm, us, B * 135119952 / us / 1e6, c.get_option("last_kernel")), flush=True)
            c.close()
