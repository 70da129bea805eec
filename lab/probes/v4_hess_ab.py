#!/usr/bin/env python3
"""General-order Hessian kernel (hess_kernel 7), profile build: source variants (option v4_variant: 8 the all-drive gather-dot reads nine
columns of z at a time, 16 the output vectors leave one column per LDS round trip) alternating in one process."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    m = system.n_drives
    order = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    split = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 1: two workgroups per interval
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for B in ((1,) if split else (1, 8)):
            trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
            t0 = trajs[0]
            Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
            ctxs = {}
            for var in (0, 8, 16, 24):
                c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                               dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                               Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
                c.set_stream(stream.cuda_stream)
                c.set_option("hess_kernel", 7)
                c.set_option("hess_split", split)
                c.set_option("v4_variant", var)
                ctxs[var] = c
            mud = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
            hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
            ref = None
            for var, c in ctxs.items():
                for _ in range(3):
                    c.hess_dev(Zd, mud, hv)
                stream.synchronize()
                if ref is None:
                    ref = hv.clone()
                print("variant %d: max |diff| to variant 0 %.1e" % (var, (hv - ref).abs().max().item()))
            res = {k: [] for k in ctxs}
            for rnd in range(6):
                for var in (list(ctxs) if rnd % 2 == 0 else list(ctxs)[::-1]):
                    c = ctxs[var]
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(10):
                        c.hess_dev(Zd, mud, hv)
                    e1.record(stream)
                    stream.synchronize()
                    res[var].append(e0.elapsed_time(e1) / 10 * 1e3)
            for var, v in res.items():
                print("B=%d order %d variant %2d: %s  median %.1f us/launch = %.2f us/eval" % (B, order, var, " ".join("%.1f" % x for x in v), np.median(v), np.median(v) / B), flush=True)
            for c in ctxs.values():
                c.close()
finally:
    pa.build_library(force=True)
