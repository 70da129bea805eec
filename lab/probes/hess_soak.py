#!/usr/bin/env python3
"""Soak test of the Hessian kernel's cross-workgroup scalar reduction (relaxed agent-scope atomics, self-resetting arrival
counters): thousands of launches, every result compared bitwise with the first."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for batch, reps in ((8, 3000), (1, 6000), (3, 3000)):
        trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
        ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], batch, pade_order=4)
        c = ms.ctx
        c.set_stream(stream.cuda_stream)
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
        hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        c.hess_dev(Zd, mu, hv)
        stream.synchronize()
        ref = hv.clone()
        bad = 0
        for i in range(reps):
            hv.zero_()
            if i % 3 == 0:
                c.eval_jac_dev(Zd, dd, vd)  # interleave the big store stream (L2 pressure between launches)
            c.hess_dev(Zd, mu, hv)
            if i % 50 == 49:
                stream.synchronize()
                if not torch.equal(hv, ref):
                    bad += 1
        stream.synchronize()
        print("batch %d: %d launches, %d mismatching checks" % (batch, reps, bad), flush=True)
        ms.close()
