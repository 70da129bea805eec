import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
m = system.n_drives
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
    Zd = torch.from_numpy(t0.datavec.copy()[None]).cuda()
    for order in (4, 8):
        ctxs = {}
        for name, opts in (("auto", {}), ("k7 split0", dict(hess_kernel=7, hess_split=0)), ("k7 split1", dict(hess_kernel=7, hess_split=1))):
            c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start, dt_off=t0.components["Δt"].start,
                                           x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift, Gj=system.G_drives_array(), batch=1,
                                           batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
            c.set_stream(stream.cuda_stream)
            for k, v in opts.items():
                c.set_option(k, v)
            ctxs[name] = c
        mud = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
        hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
        res = {k: [] for k in ctxs}
        for rnd in range(5):
            for name, c in ctxs.items():
                for _ in range(5):
                    c.hess_dev(Zd, mud, hv)
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(50):
                    c.hess_dev(Zd, mud, hv)
                e1.record(stream)
                stream.synchronize()
                res[name].append(e0.elapsed_time(e1) / 50 * 1e3)
        for name, v in res.items():
            print("order %d %-10s: %.2f us per call (kernel id %d)" % (order, name, np.median(v), ctxs[name].get_option("last_hess_kernel")))
