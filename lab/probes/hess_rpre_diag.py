#!/usr/bin/env python3
"""Diagnostics of the R-chain modes of the column-group Hessian kernel (hess_rpre 0 in-wave | 1 chain waves in the launch; 2, a launch in front, was removed after the measurement): repeatability and
where the modes differ.  usage: hess_rpre_diag.py [order=6] [B=3] [N=60]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
order, B, N = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 6), (2, 3), (3, 60)))
system = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(system, N, seed=400 + i) for i in range(B)]
t0 = trajs[0]
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), t0, B, pade_order=order)
c = ms.ctx
Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
mud = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
K = N - 1
nsc = 28
outs = {}
for mode in (0, 1, -1, 1, 1, 0, 1):
    c.set_option("hess_rpre", mode)
    hv.fill_(float("nan"))
    torch.cuda.synchronize()  # (the context launches on a stream of its own: the fill must have finished)
    c.hess_dev(Zd, mud, hv)
    c.sync()
    cur = hv.clone().view(B * K, -1)
    msg = "mode %2d -> rpre %d nan %d" % (mode, c.get_option("last_hess_rpre"), int(torch.isnan(cur).sum()))
    for k_, prev in outs.items():
        df = (prev != cur)
        if df.any():
            rows = df.any(dim=1).nonzero().flatten()
            cols = df.any(dim=0).nonzero().flatten()
            msg += " | vs mode %d: %d entries differ in %d intervals (first %s), columns %s, max |diff| %.2e" % (
                k_, int(df.sum()), len(rows), rows[:5].tolist(), cols[:8].tolist(), float((prev - cur).abs().nan_to_num(1e300).max()))
        else:
            msg += " | == mode %d" % k_
    print(msg, flush=True)
    outs.setdefault(mode, cur)
