#!/usr/bin/env python3
"""Compact Jacobian (unique tiles only: what the host-delivery path launches): kernel 3 (every workgroup in the matrix role) against
kernel 4 (the chains alone), 8 and 1 trajectories per launch, alternating in one process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for B in (8, 1):
        trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
        t0 = trajs[0]
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        ctxs = {}
        for name, opts in (("v3", dict(kernel_version=3)), ("v4", dict(kernel_version=4)), ("v4-rr", dict(kernel_version=4, contiguous=0)), ("v4-rr-cps27", dict(kernel_version=4, cols_per_slice=27))):
            c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                           dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                           Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=4)
            c.set_stream(stream.cuda_stream)
            for k, v in opts.items():
                c.set_option(k, v)
            ctxs[name] = c
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        cv = torch.empty(c.compact_nnz, dtype=torch.float64, device="cuda")
        ref = None
        for name, c in ctxs.items():
            for _ in range(3):
                c.eval_jac_compact_dev(Zd, dd, cv)
            stream.synchronize()
            if ref is None:
                ref = cv.clone()
            print(name, "kernel", c.get_option("last_kernel"), "max diff to v3 %.1e" % (cv - ref).abs().max().item())
        res = {k: [] for k in ctxs}
        for rnd in range(6):
            for name in (list(ctxs) if rnd % 2 == 0 else list(ctxs)[::-1]):
                c = ctxs[name]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(20):
                    c.eval_jac_compact_dev(Zd, dd, cv)
                e1.record(stream)
                stream.synchronize()
                res[name].append(e0.elapsed_time(e1) / 20 * 1e3)
        for name, v in res.items():
            print("B=%d %-12s: %s  median %.1f us/launch = %.2f us/eval" % (B, name, " ".join("%.1f" % x for x in v), np.median(v), np.median(v) / B), flush=True)
