#!/usr/bin/env python3
"""General-order pattern-compiled Hessian kernel (hess_kernel 7; auto for every order but 4): parity against the oracle at config 3,
ensemble members, column slices; rates next to the round-2 kernels."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
from oracle import pade_oracle as po

so = po.config_system(3)
G0, Gj = so.G_drift, np.array(so.G_drives)
ok = True
for N in (4, 100):
    Z, lay = po.synthetic_trajectory(so, N, seed=79)
    Z[:, lay.dt_off] = 0.1 + 0.05 * np.random.default_rng(3).random(N)
    mu = np.random.default_rng(5).standard_normal((lay.K, lay.x_dim))
    for order in ((2, 4, 6, 8, 10) if N == 4 else (4, 8)):
        c = pa.integrators._PclContext(d=lay.d, m=lay.m, N=lay.N, z_dim=lay.z_dim, u_off=lay.u_off, dt_off=lay.dt_off, x_offs=[lay.x_off], G0=G0, Gj=Gj,
                                      batch=1, batch_mode=pa._lib.PCL_BATCH_MEMBERS, pade_order=order)
        h_ref = po.pade_hessian_values(Z, mu, lay, G0, Gj, order).reshape(-1)
        nsc = (lay.m + 1) * (lay.m + 2) // 2
        per = po.hess_nnz_per_interval(lay)
        for cps, grid in ((0, 0), (9, 0), (5, 3), (0, 1)):
            c.set_option("hess_kernel", 7)
            c.set_option("cols_per_slice", cps)
            c.set_option("grid", grid)
            hv = c.hess(Z, mu.reshape(-1))
            Hm, Rm = hv.reshape(lay.K, per), h_ref.reshape(lay.K, per)
            scale = max(1.0, np.abs(h_ref).max())
            es, ev = np.abs(Hm[:, :nsc] - Rm[:, :nsc]).max() / scale, np.abs(Hm[:, nsc:] - Rm[:, nsc:]).max() / scale
            good = es < 1e-11 and ev < 1e-11 and c.get_option("last_hess_kernel") == 70 + order // 2
            ok &= good
            print("N=%3d order %2d cps %d grid %d: scalars %.1e vectors %.1e kernel %d %s" % (N, order, cps, grid, es, ev, c.get_option("last_hess_kernel"), "ok" if good else "FAIL"), flush=True)
        c.close()
print("PARITY", "OK" if ok else "FAILED", flush=True)
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
            mud = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
            hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
            for hk in ((0, 7) if order == 4 else (1, 7)):
                c.set_option("hess_kernel", hk)
                reps = 20 if (order == 4 or hk == 7) else 3
                for _ in range(2):
                    c.hess_dev(Zd, mud, hv)
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps):
                    c.hess_dev(Zd, mud, hv)
                e1.record(stream)
                stream.synchronize()
                us = e0.elapsed_time(e1) / reps * 1e3
                print("B=%d order %2d hess_kernel %d: %.1f us/launch, %.2f us/eval (kernel id %d)" % (B, order, hk, us, us / B, c.get_option("last_hess_kernel")), flush=True)
            c.close()
