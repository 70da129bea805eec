#!/usr/bin/env python3
"""Is the one-trajectory launch bound by the memory system or by the CUs that store?  Kernel 4, one item per workgroup (round-robin slices),
knot counts from 65 (128 items) to 129 (256 items): a launch that takes the same time for 198 and for 256 items is bound per CU."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
order = int(sys.argv[1]) if len(sys.argv) > 1 else 4
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ctxs = {}
    for N in (65, 81, 100, 113, 129):
        t0 = synthetic.synthetic_trajectory(system, N, seed=1000)
        Zd = torch.from_numpy(t0.datavec.copy()).cuda()
        c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                       dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                       Gj=system.G_drives_array(), batch=1, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
        c.set_stream(stream.cuda_stream)
        c.set_option("kernel_version", 4)
        c.set_option("cols_per_slice", 14)
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        ctxs[N] = (c, Zd, dd, vd)
        for _ in range(5):
            c.eval_jac_dev(Zd, dd, vd)
        stream.synchronize()
    res = {k: [] for k in ctxs}
    for rnd in range(6):
        for N in (list(ctxs) if rnd % 2 == 0 else list(ctxs)[::-1]):
            c, Zd, dd, vd = ctxs[N]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(100):
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            res[N].append(e0.elapsed_time(e1) / 100 * 1e3)
    for N, v in res.items():
        c = ctxs[N][0]
        mb = c.jac_nnz * 8 / 1e6
        print("order %d N %3d (%3d items, %6.1f MB): median %.2f us/launch = %.2f TB/s (kernel id %d, cols/slice %d)" % (order, N, 2 * (N - 1), mb, np.median(v), mb / np.median(v), c.get_option("last_kernel"), c.get_option("effective_cols_per_slice")), flush=True)
