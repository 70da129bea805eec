#!/usr/bin/env python3
"""Kernel choice on multi-transmon systems of other sizes (two drive entries per row, like BASELINE config 3)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
rng = np.random.default_rng(0)
stream = torch.cuda.Stream()
N = 100
cases = [([4.0, 4.1], 4), ([4.0, 4.1], 5), ([4.0, 4.1, 4.2], 3)] if "--few" in sys.argv else [([4.0, 4.1], 2), ([4.0, 4.1], 3), ([4.0, 4.1], 4), ([4.0, 4.1], 5), ([4.0, 4.1, 4.2], 2), ([4.0, 4.1, 4.2], 3), ([4.0, 4.1, 4.2, 4.3], 2)]
with torch.cuda.stream(stream):
    for oms, lev in cases:
        q = len(oms)
        gs = 0.01 * (np.ones((q, q)) - np.eye(q))
        sys_ = pa.MultiTransmonSystem(oms, [0.2] * q, gs, levels_per_transmon=lev, drive_bounds=0.1)
        d, m = sys_.levels, sys_.n_drives
        traj = pa.unitary_trajectory(sys_, 0.02 * rng.standard_normal((m, N)), 0.1 * np.arange(N), np.eye(d))
        for batch in (1, 8):
            if d * d * 4 * d * 99 * batch * 8 > 3e9:
                continue
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
                for _ in range(100):
                    c.eval_jac_dev(Z, dd, vd)
                e1.record(stream)
                stream.synchronize()
                res[kv] = e0.elapsed_time(e1) / 100 * 1e3
            mb = c.jac_nnz * 8 / 1e6
            print("%d transmons x %d levels: d %2d m %d ell %d batch %d (%.0f MB): auto %.1f | v1 %.1f | v2 %.1f | v3 %.1f us/launch  (best %.2f TB/s)"
                  % (q, lev, d, m, c.get_option("ell_width"), batch, mb, res[0], res[1], res[2], res[3], mb / min(res.values()) / 1e6 * 1e6 / 1e6), flush=True)
            ms.close()
