#!/usr/bin/env python3
"""Kernel time of the default batch-8 launch vs the byte offset of the Jacobian buffer inside one allocation."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
system = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, pade_order=4)
c = ms.ctx
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    c.set_stream(stream.cuda_stream)
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    big = torch.empty(c.jac_nnz + (64 << 20) // 8, dtype=torch.float64, device="cuda")
    print("base 0x%x" % big.data_ptr())
    for off in (0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 2 << 20, 3 << 20, (4 << 20) + 4096, 17 << 20, 0):
        vd = big[off // 8 : off // 8 + c.jac_nnz]
        for _ in range(5):
            c.eval_jac_dev(Zd, dd, vd)
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(40):
            c.eval_jac_dev(Zd, dd, vd)
        e1.record(stream)
        stream.synchronize()
        print("offset %9d B: %.2f us/eval" % (off, e0.elapsed_time(e1) / 40 / B * 1e3), flush=True)
ms.close()
