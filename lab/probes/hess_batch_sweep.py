#!/usr/bin/env python3
"""Hessian kernel 3: launch time against the number of trajectories per launch and against the grid (items per workgroup)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic

system = synthetic.config_system(3)
def tm(c, Zd, mu, hv, reps=40):
    for _ in range(5): c.hess_dev(Zd, mu, hv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): c.hess_dev(Zd, mu, hv)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for batch in (1, 2, 3, 4, 5, 6, 8, 10, 16):
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], batch, pade_order=4)
    c = ms.ctx
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
    hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_option("hess_kernel", 3)
    items = batch * 99
    line = "batch %2d items %4d: default %.1f us" % (batch, items, tm(c, Zd, mu, hv))
    for g in (items, (items + 1) // 2, (items + 2) // 3, 256, 128):
        if g <= items:
            c.set_option("grid", g)
            line += " | grid %d: %.1f" % (g, tm(c, Zd, mu, hv))
    c.set_option("grid", 0)
    print(line, flush=True)
    ms.close()
