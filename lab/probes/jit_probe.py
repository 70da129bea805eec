#!/usr/bin/env python3
"""Run-time compiled shape instances vs the run-time-shape kernels on shapes outside the static table (two drive entries
per row and column), fused residual+Jacobian and Hessian, 8 trajectories per launch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
rng = np.random.default_rng(0)
stream = torch.cuda.Stream()
N, batch = 100, 8
with torch.cuda.stream(stream):
    for d, m in ((22, 4), (24, 4), (26, 3), (30, 2), (24, 6)):
        n = 2 * d
        Hd = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        G0 = pa.quantum.G(0.3 * (Hd + Hd.conj().T))
        Gj = np.zeros((m, n, n))
        for l in range(m):
            for i in range(n):
                a, b = rng.standard_normal(2)
                Gj[l, i, (i + l + 1) % n] += a
                Gj[l, (i + l + 1) % n, i] -= a  # antisymmetric, two entries per row and column
        xd = 2 * d * d
        comps = {"Ũ⃗": 0.1 * rng.standard_normal((xd, N)), "Δt": np.full((1, N), 0.1), "t": 0.1 * np.arange(N)[None], "u": 0.1 * rng.standard_normal((m, N))}
        traj = pa.NamedTrajectory(comps, controls=("u", "Δt"), timestep="Δt")
        ms = pa.HipPadeMultistart(G0, Gj, traj, batch, pade_order=4)
        c = ms.ctx
        c.set_stream(stream.cuda_stream)
        Z = torch.from_numpy(np.tile(traj.datavec, batch)).cuda()
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
        hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
        out = []
        for jit in (1, 0):
            c.set_option("jit", jit)
            t0 = time.perf_counter()
            c.eval_jac_dev(Z, dd, vd); c.hess_dev(Z, mu, hv); stream.synchronize()
            first = time.perf_counter() - t0
            res = []
            for fn in (lambda: c.eval_jac_dev(Z, dd, vd), lambda: c.hess_dev(Z, mu, hv)):
                for _ in range(5):
                    fn()
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(50):
                    fn()
                e1.record(stream)
                stream.synchronize()
                res.append(e0.elapsed_time(e1) / 50 * 1e3)
            out.append((first, res[0], res[1], c.get_option("last_kernel"), c.get_option("last_hess_kernel")))
        mb = c.jac_nnz * 8 / 1e6
        print("d %d m %d (%.0f MB): compiled: first call %.2f s, fused %.1f us (%.2f TB/s, kernel %d), Hessian %.1f us (kernel %d) | run-time shapes: fused %.1f us (kernel %d), Hessian %.1f us"
              % (d, m, mb, out[0][0], out[0][1], mb / out[0][1] , out[0][3], out[0][2], out[0][4], out[1][1], out[1][3], out[1][2]), flush=True)
        ms.close()
