#!/usr/bin/env python3
"""8 trajectories per launch: does the launch time depend on WHERE the 1.08 GB of Jacobian values live?  Six output buffers allocated one after the
other (all alive: six different places), each timed in alternating rounds in one process; then the same with buffers from hipMalloc directly."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
    bufs = [torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda") for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6)]
    pad = [torch.empty(12345 * (i + 1), dtype=torch.float64, device="cuda") for i in range(3)]
    bufs += [torch.empty(c.jac_nnz + 7 * 1024 * (i + 1), dtype=torch.float64, device="cuda")[7 * 1024 * (i + 1):] for i in range(2)]  # odd offsets inside an allocation
    res = [[] for _ in bufs]
    for rnd in range(3):
        for i, vd in enumerate(bufs):
            for _ in range(3):
                c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(20):
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            res[i].append(e0.elapsed_time(e1) / 20 * 1e3)
    for i, vd in enumerate(bufs):
        print("buffer %d at 0x%x (mod 2 MiB: 0x%06x): %s  median %.1f us" % (i, vd.data_ptr(), vd.data_ptr() % (2 << 20), " ".join("%.0f" % x for x in res[i]), np.median(res[i])))
