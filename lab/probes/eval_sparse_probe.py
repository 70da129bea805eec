#!/usr/bin/env python3
"""Pattern-compiled residual kernel (eval_kernel=2) against the matrix-core kernel (eval_kernel=1): agreement and times."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
def tm(c, Zd, dd, reps=60):
    for _ in range(5): c.eval_dev(Zd, dd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): c.eval_dev(Zd, dd)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for batch in (1, 2, 8, 16, 32):
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], batch, pade_order=4)
    c = ms.ctx
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    d1 = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    d2 = torch.full((c.n_rows,), float("nan"), dtype=torch.float64, device="cuda")
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_option("eval_kernel", 1); c.eval_dev(Zd, d1); c.sync(); k1 = c.get_option("last_kernel"); t1 = tm(c, Zd, d1)
    c.set_option("eval_kernel", 2); c.eval_dev(Zd, d2); c.sync(); k2 = c.get_option("last_kernel"); t2 = tm(c, Zd, d2)
    err = (d1 - d2).abs().max().item()
    print("batch %2d: kernel %d %.1f us (%.2f us/eval) | kernel %d %.1f us (%.2f us/eval) | max|diff| %.2e (max %.2e) nan %d"
          % (batch, k1, t1, t1 / batch, k2, t2, t2 / batch, err, d1.abs().max().item(), int(torch.isnan(d2).sum().item())), flush=True)
    ms.close()
