#!/usr/bin/env python3
"""Residual+Jacobian rate of the general-order kernel (pade_order 2..10) at BASELINE config 3, 8 trajectories per launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = 8
system = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
t0 = trajs[0]
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    for order in (4, 2, 6, 8, 10):
        c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                       dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                       Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
        c.set_stream(stream.cuda_stream)
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        for gen in ((0, 1, 2) if order == 4 else (1, 2)):
            c.set_option("general_pade_kernel", 1 if gen else 0)
            c.set_option("general_threads", {1: 512, 2: 256, 3: 1024}.get(gen, 512))
            for _ in range(3):
                c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(20):
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            print("order %2d%s: %.1f us/eval (kernel id %d)" % (order, (" (general kernel, %d threads)" % {1: 512, 2: 256, 3: 1024}[gen]) if gen else "", e0.elapsed_time(e1) / 20 / B * 1e3, c.get_option("last_kernel")), flush=True)
        c.close()
