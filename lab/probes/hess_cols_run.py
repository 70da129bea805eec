#!/usr/bin/env python3
"""A few launches of the general-order Hessian kernels at a given order (target of rocprofv3 passes): hess_cols_run.py [batch=8] [hess_kernel=8] [order=8] [key=value ...] [launches=12]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
order = int(sys.argv[3]) if len(sys.argv) > 3 else 8
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
t0 = trajs[0]
c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                               dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                               Gj=system.G_drives_array(), batch=batch, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_option("hess_kernel", hk)
launches = 12
for kv in sys.argv[4:]:
    if kv.startswith("launches="):
        launches = int(kv.split("=")[1])
    else:
        c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for _ in range(launches): c.hess_dev(Zd, mu, hv)
torch.cuda.synchronize()
c.close()
