#!/usr/bin/env python3
"""General-order kernel (pade_order 6, 8, 10) at BASELINE config 3: full output in one launch vs compact output + expansion."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
from piccolo_jl_amd.trajectory import STATE

system = synthetic.config_system(3)
for B in (1, 8):
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
    t0 = trajs[0]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        for order in (6, 8, 10):
            c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                           dt_off=t0.components["Δt"].start, x_offs=[t0.components[STATE].start], G0=system.G_drift,
                                           Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
            c.set_stream(stream.cuda_stream)
            dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
            vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
            cv = torch.empty(c.compact_nnz, dtype=torch.float64, device="cuda")
            def timeit(fn, n=20):
                for _ in range(3):
                    fn()
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(n):
                    fn()
                e1.record(stream)
                stream.synchronize()
                return e0.elapsed_time(e1) / n * 1e3
            c.set_option("general_kernel_version", 1)
            tf = timeit(lambda: c.eval_jac_dev(Zd, dd, vd))
            tc = timeit(lambda: c.eval_jac_compact_dev(Zd, dd, cv))
            te = timeit(lambda: c.jac_expand_dev(cv, vd))
            tr = timeit(lambda: c.eval_dev(Zd, dd))
            print("B %d order %2d: reference formulation: full %.1f us | compact %.1f + expand %.1f = %.1f us | residual only %.1f us  (per launch)" % (B, order, tf, tc, te, tc + te, tr), flush=True)
            c.set_option("general_kernel_version", 2)
            for sl in (0, 1, 2, 3):
                c.set_option("general_slices", sl)
                tf = timeit(lambda: c.eval_jac_dev(Zd, dd, vd))
                tc = timeit(lambda: c.eval_jac_compact_dev(Zd, dd, cv))
                print("   lock-step kernel, slices %d: full %.1f us | compact %.1f us   (kernel id %d)" % (sl, tf, tc, c.get_option("last_kernel")), flush=True)
            c.close()
