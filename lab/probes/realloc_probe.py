#!/usr/bin/env python3
"""Within one process: free and re-allocate the Jacobian buffer (with perturbing allocations in between) and time the
default batch-8 launch each time.  Separates 'placement of the buffer' from 'state of the process/GPU'."""
import os, sys, random
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
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
with torch.cuda.stream(stream):
    c.set_stream(stream.cuda_stream)
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    keep = []
    for trial in range(10):
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        for _ in range(5):
            c.eval_jac_dev(Zd, dd, vd)
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(30):
            c.eval_jac_dev(Zd, dd, vd)
        e1.record(stream)
        stream.synchronize()
        print("trial %d: vd @ 0x%x  %.2f us/eval" % (trial, vd.data_ptr(), e0.elapsed_time(e1) / 30 / B * 1e3), flush=True)
        del vd
        stream.synchronize()
        torch.cuda.empty_cache()
        if trial % 2 == 1:
            keep.append(torch.empty(random.randrange(1 << 20, 300 << 20), dtype=torch.uint8, device="cuda"))
ms.close()
