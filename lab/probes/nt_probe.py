#!/usr/bin/env python3
"""nt_stores on/off for the fused kernel, batch 8 and 1, interleaved A/B in one process."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
for batch in (8, 1):
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], batch, pade_order=4)
    c = ms.ctx; c.set_stream(stream.cuda_stream)
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda"); vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    def tm(n=60):
        for _ in range(5): c.eval_jac_dev(Zd, dd, vd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): c.eval_jac_dev(Zd, dd, vd)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    for rep in range(3):
        for nt in (0, -1, 2) if batch == 1 else (0, -1):
            c.set_option("nt_stores", nt)
            print("batch %d nt %d: %.2f us/launch" % (batch, nt, tm()), flush=True)
    c.set_option("nt_stores", 0)
    if batch == 8:
        for sw in (96, 112, 128, 144, 160):
            c.set_option("stream_workgroups", sw)
            print("batch 8 stream_workgroups %d: %.2f us/launch" % (sw, tm()), flush=True)
        c.set_option("stream_workgroups", -1)
    ms.close()
