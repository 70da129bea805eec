#!/usr/bin/env python3
"""Which fused kernel wins for small Hilbert dimensions?  us/launch of eval+Jacobian by kernel_version over d, batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
rng = np.random.default_rng(0)
stream = torch.cuda.Stream()
N, m = 100, 4
with torch.cuda.stream(stream):
    for d in (2, 4, 6, 9, 12, 16, 20, 24):
        Hd = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        Hs = []
        for _ in range(m):
            A = np.zeros((d, d), complex)
            for i in range(d - 1):
                A[i, i + 1] = rng.standard_normal() + 1j * rng.standard_normal()
            Hs.append(A + A.conj().T)
        sys_ = pa.QuantumSystem(0.3 * (Hd + Hd.conj().T), Hs, [1.0] * m)
        traj = pa.unitary_trajectory(sys_, 0.1 * rng.standard_normal((m, N)), 0.1 * np.arange(N), np.eye(d))
        for batch in (1, 16):
            ms = pa.HipPadeMultistart(sys_.G_drift, sys_.G_drives_array(), traj, batch, pade_order=4)
            c = ms.ctx
            c.set_stream(stream.cuda_stream)
            Z = torch.from_numpy(np.tile(traj.datavec, batch)).cuda()
            dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
            vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
            res = {}
            for kv in (0, 1, 2, 3):
                c.set_option("kernel_version", kv)
                for _ in range(10):
                    c.eval_jac_dev(Z, dd, vd)
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(200):
                    c.eval_jac_dev(Z, dd, vd)
                e1.record(stream)
                stream.synchronize()
                res[kv] = e0.elapsed_time(e1) / 200 * 1e3
            print("d %2d batch %2d (%.1f MB/launch): auto %.1f | v1 %.1f | v2 %.1f | v3 %.1f us/launch" % (d, batch, c.jac_nnz * 8 / 1e6, res[0], res[1], res[2], res[3]), flush=True)
            ms.close()
