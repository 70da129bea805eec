#!/usr/bin/env python3
"""A few launches of the general-order kernel for rocprofv3: order_run.py <order> <batch> <compact 0|1> [version]."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
from piccolo_jl_amd.trajectory import STATE
order, B, compact = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
version = int(sys.argv[4]) if len(sys.argv) > 4 else 0
system = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
t0 = trajs[0]
Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                               dt_off=t0.components["Δt"].start, x_offs=[t0.components[STATE].start], G0=system.G_drift,
                               Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
c.set_option("general_kernel_version", version)
dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
vd = torch.empty(c.compact_nnz if compact else c.jac_nnz, dtype=torch.float64, device="cuda")
for _ in range(10):
    (c.eval_jac_compact_dev if compact else c.eval_jac_dev)(Zd, dd, vd)
c.sync()
