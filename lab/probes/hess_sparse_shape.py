#!/usr/bin/env python3
"""Pattern-compiled Hessian kernel on the d = 16 / d = 25 two-transmon systems against kernel 3, per output section."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import piccolo_jl_amd as pa
from piccolo_jl_amd import quantum, synthetic
for levels in (4, 5, 3):
    so = quantum.MultiTransmonSystem([4.0, 4.1], [0.2, 0.2], [[0, 0.01], [0.01, 0]], levels_per_transmon=levels, drive_bounds=0.1)
    G0, Gj = so.G_drift, so.G_drives_array()
    d, m = G0.shape[0] // 2, len(Gj)
    for batch, N in ((1, 12), (3, 30)):
        trajs = [synthetic.synthetic_trajectory(so, N, seed=60 + s) for s in range(batch)]
        ms = pa.HipPadeMultistart(G0, Gj, trajs[0], batch, pade_order=4)
        c = ms.ctx
        Zb = np.stack([t.datavec for t in trajs])
        mu = np.random.default_rng(1).standard_normal(c.n_rows)
        c.set_option("hess_kernel", 3); h3 = c.hess(Zb, mu)
        c.set_option("hess_kernel", 4); h4 = c.hess(Zb, mu); k = c.get_option("last_hess_kernel")
        h4b = c.hess(Zb, mu)
        per = c.hess_nnz // (batch * (N - 1))
        xd = 2 * d * d
        nsc = (m + 1) * (m + 2) // 2
        blk = np.abs(h3 - h4).reshape(-1, per)
        secs = [nsc, nsc + m * xd, nsc + (m + 1) * xd, nsc + (2 * m + 1) * xd, per]
        prev = 0
        msg = []
        for nm, e in zip(["scal", "H3", "H4", "H5", "H6"], secs):
            msg.append("%s %.1e" % (nm, blk[:, prev:e].max())); prev = e
        print("levels %d d %d m %d batch %d N %d kernel %d repeat-equal %s | %s | bad intervals %d of %d" % (levels, d, m, batch, N, k, np.array_equal(h4, h4b), " ".join(msg), int((blk.max(1) > 1e-9).sum()), blk.shape[0]), flush=True)
        ms.close()
