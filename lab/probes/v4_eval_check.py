#!/usr/bin/env python3
"""Residual-only kernel on the pattern-compiled products (eval_kernel 3 / auto): parity against the oracle at config 3 for every order,
then rates next to the matrix-core (1) and the round-2 pattern-compiled (2) residual kernels."""
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
    Z, lay = po.synthetic_trajectory(so, N, seed=78)
    Z[:, lay.dt_off] = 0.1 + 0.05 * np.random.default_rng(3).random(N)
    for order in (2, 4, 6, 8, 10):
        c = pa.integrators._PclContext(d=lay.d, m=lay.m, N=lay.N, z_dim=lay.z_dim, u_off=lay.u_off, dt_off=lay.dt_off, x_offs=[lay.x_off], G0=G0, Gj=Gj,
                                      batch=1, batch_mode=pa._lib.PCL_BATCH_MEMBERS, pade_order=order)
        d_ref = po.pade_residual(Z, lay, G0, Gj, order)
        for ek, grid in ((3, 0), (0, 0), (3, 3)):
            c.set_option("eval_kernel", ek)
            c.set_option("grid", grid)
            delta = c.eval(Z)
            e = np.abs(delta.ravel() - d_ref.ravel()).max() / max(1.0, np.abs(d_ref).max())
            good = e < 1e-12 and c.get_option("last_kernel") == 80 + order // 2
            ok &= good
            print("N=%3d order %2d eval_kernel %d grid %d: %.1e kernel %d %s" % (N, order, ek, grid, e, c.get_option("last_kernel"), "ok" if good else "FAIL"), flush=True)
        c.close()
print("PARITY", "OK" if ok else "FAILED", flush=True)
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for B in (1, 8, 32):
        trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
        t0 = trajs[0]
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        for order in (4, 8):
            c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                           dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                           Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
            c.set_stream(stream.cuda_stream)
            dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
            for ek in ((1, 2, 3) if order == 4 else (1, 3)):
                c.set_option("eval_kernel", ek)
                if order != 4 and ek == 1:
                    c.set_option("general_kernel_version", 1)  # the general-order residual kernel of round 2
                else:
                    c.set_option("general_kernel_version", 0)
                for _ in range(5):
                    c.eval_dev(Zd, dd)
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(50):
                    c.eval_dev(Zd, dd)
                e1.record(stream)
                stream.synchronize()
                us = e0.elapsed_time(e1) / 50 * 1e3
                print("B=%2d order %d eval_kernel %d: %.1f us/launch, %.2f us/eval (kernel id %d)" % (B, order, ek, us, us / B, c.get_option("last_kernel")), flush=True)
            c.close()
