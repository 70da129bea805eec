import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch, piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
t0 = synthetic.synthetic_trajectory(system, 100, seed=1)
Zd = torch.from_numpy(t0.datavec.copy()).cuda()
for order in (4, 8, 10):
    c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start, dt_off=t0.components["Δt"].start,
                                   x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift, Gj=system.G_drives_array(), batch=1, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda"); vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda"); hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
    t = time.perf_counter(); c.eval_jac_dev(Zd, dd, vd); torch.cuda.synchronize(); t1 = time.perf_counter()
    c.eval_dev(Zd, dd); torch.cuda.synchronize(); t2 = time.perf_counter()
    c.hess_dev(Zd, mu, hv); torch.cuda.synchronize(); t3 = time.perf_counter()
    print("order %d: first eval_jac %.2f s, first eval %.2f s, first hess %.2f s, jit_compiles %d" % (order, t1 - t, t2 - t1, t3 - t2, c.get_option("jit_compiles")), flush=True)
    c.close()
