#!/usr/bin/env python3
"""Does the kernel time depend on where the 1 GB output buffer lies?  Times the default batch-8 launch into several
distinct output buffers (all alive at once), printing each buffer's address."""
import os, sys
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
with torch.cuda.stream(stream):
    c.set_stream(stream.cuda_stream)
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    bufs = []
    pad = []
    for i in range(6):
        bufs.append(torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda"))
        pad.append(torch.empty((i + 1) * 12345677, dtype=torch.uint8, device="cuda"))  # shift the next allocation
    for rep in range(2):
        for i, vd in enumerate(bufs):
            for _ in range(5):
                c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(40):
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            print("buffer %d @ 0x%x (mod 2MiB = %d KiB): %.2f us/eval" % (i, vd.data_ptr(), (vd.data_ptr() % (2 << 20)) // 1024, e0.elapsed_time(e1) / 40 / B * 1e3), flush=True)
ms.close()
