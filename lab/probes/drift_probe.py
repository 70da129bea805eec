#!/usr/bin/env python3
"""Kernel time of the default batch-8 launch over ~40 s of continuous running (is there a slow and a fast state?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = 8
system = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, pade_order=4)
c = ms.ctx
stream = torch.cuda.Stream()
t00 = time.time()
with torch.cuda.stream(stream):
    c.set_stream(stream.cuda_stream)
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(2000):
            c.eval_jac_dev(Zd, dd, vd)
        e1.record(stream)
        stream.synchronize()
        print("t=%5.1fs  %.2f us/eval" % (time.time() - t00, e0.elapsed_time(e1) / 2000 / B * 1e3), flush=True)
ms.close()
