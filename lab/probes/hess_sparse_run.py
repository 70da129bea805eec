#!/usr/bin/env python3
"""A few launches of the pattern-compiled Hessian kernel at 8 trajectories per launch (target of the counter passes)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], batch, pade_order=4)
c = ms.ctx
Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_option("hess_kernel", int(sys.argv[2]) if len(sys.argv) > 2 else 4)
for _ in range(12): c.hess_dev(Zd, mu, hv)
torch.cuda.synchronize()
ms.close()
