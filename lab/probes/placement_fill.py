#!/usr/bin/env python3
"""Does a bare fill of the buffer (torch zero_(), hipMemsetAsync) see the placement the way the fused kernel does?"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = 8
nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 10
system = synthetic.config_system(3)
m = system.n_drives
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
    t0 = trajs[0]
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start, dt_off=t0.components["Δt"].start,
                                   x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift, Gj=system.G_drives_array(), batch=B,
                                   batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=4)
    c.set_stream(stream.cuda_stream)
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    bufs = [torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
    def timeit(f, reps):
        for _ in range(2):
            f()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            f()
        e1.record(stream)
        stream.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for i, vd in enumerate(bufs):
        tk = np.median([timeit(lambda: c.eval_jac_dev(Zd, dd, vd), 10) for _ in range(3)])
        tz = np.median([timeit(lambda: vd.zero_(), 10) for _ in range(3)])
        print("buffer %2d: fused kernel %.1f us, zero_() %.1f us" % (i, tk, tz))
