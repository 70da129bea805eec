#!/usr/bin/env python3
"""Hessian kernel choice (hess_kernel 1 / 2) on multi-transmon systems of several sizes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
rng = np.random.default_rng(0)
stream = torch.cuda.Stream()
N = 100
cases = [([4.0, 4.1], 5)] if "--few" in sys.argv else [([4.0, 4.1], 2), ([4.0, 4.1], 3), ([4.0, 4.1], 4), ([4.0, 4.1], 5), ([4.0, 4.1, 4.2], 2), ([4.0, 4.1, 4.2], 3)]
with torch.cuda.stream(stream):
    for oms, lev in cases:
        q = len(oms)
        gs = 0.01 * (np.ones((q, q)) - np.eye(q))
        sys_ = pa.MultiTransmonSystem(oms, [0.2] * q, gs, levels_per_transmon=lev, drive_bounds=0.1)
        d, m = sys_.levels, sys_.n_drives
        traj = pa.unitary_trajectory(sys_, 0.02 * rng.standard_normal((m, N)), 0.1 * np.arange(N), np.eye(d))
        for batch in (1, 8):
            ms = pa.HipPadeMultistart(sys_.G_drift, sys_.G_drives_array(), traj, batch, pade_order=4)
            c = ms.ctx
            c.set_stream(stream.cuda_stream)
            Z = torch.from_numpy(np.tile(traj.datavec, batch)).cuda()
            mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
            hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
            res = {}
            for hk in (0, 1, 2):
                c.set_option("hess_kernel", hk)
                for _ in range(5):
                    c.hess_dev(Z, mu, hv)
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(50):
                    c.hess_dev(Z, mu, hv)
                e1.record(stream)
                stream.synchronize()
                res[hk] = e0.elapsed_time(e1) / 50 * 1e3
            print("%d transmons x %d levels: d %2d m %d batch %d: auto %.1f | v1 %.1f | v2 %.1f us/launch" % (q, lev, d, m, batch, res[0], res[1], res[2]), flush=True)
            ms.close()
