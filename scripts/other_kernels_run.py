#!/usr/bin/env python3
"""One pass over the kernels `bench.py` does not launch, for the rocprofv3 kernel trace (scripts/profile_other.sh):
general-order Pade (orders 6, 8, 10: residual+Jacobian, residual only, Hessian), rollout, derivative rows; BASELINE config 3,
one trajectory per launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
from piccolo_jl_amd.trajectory import STATE

REP = 10
system = synthetic.config_system(3)
t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
G0, Gj = system.G_drift, system.G_drives_array()
Zd = torch.from_numpy(t0.datavec).cuda()
for order in (6, 8, 10):
    it = pa.HipPadeIntegrator(G0, Gj, t0, pade_order=order)
    c = it.ctx
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
    mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
    for _ in range(REP):
        c.eval_jac_dev(Zd, dd, vd)
    jk = c.get_option("last_kernel")
    for _ in range(REP):
        c.eval_dev(Zd, dd)
    for _ in range(2):
        c.hess_dev(Zd, mu, hv)
    c.sync()
    print("order", order, "jac kernel", jk, "residual kernel", c.get_option("last_kernel"), "hess kernel", c.get_option("last_hess_kernel"), flush=True)
    it.close()
    del dd, vd, hv, mu
it = pa.HipPadeIntegrator(G0, Gj, t0, pade_order=4)
c = it.ctx
roll = torch.empty(t0.N * 2 * system.levels ** 2, dtype=torch.float64, device="cuda")
for _ in range(REP):
    c.rollout_dev(Zd, roll)
c.sync()
# derivative rows du = (u_{k+1} - u_k) / dt and the time-consistency row, if the synthetic trajectory carries them
comps = t0.components
if "du" in comps:
    dim = comps["u"].stop - comps["u"].start
    rows, nnz = c.deriv_dims(comps["du"].start, dim)
    dd = torch.empty(rows, dtype=torch.float64, device="cuda")
    vd = torch.empty(nnz, dtype=torch.float64, device="cuda")
    for _ in range(REP):
        c.deriv_eval_jac_dev(comps["u"].start, comps["du"].start, dim, Zd, dd, vd)
    c.sync()
it.close()
print("done", flush=True)
