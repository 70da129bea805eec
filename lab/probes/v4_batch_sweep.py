#!/usr/bin/env python3
"""Kernel 4 (and kernel 3) against the number of trajectories per launch, contiguous column ranges against intervals dealt round-robin
to the workgroups: us per launch, us per trajectory, fraction of the 8 TB/s peak.  Alternating in one process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for B in (8, 16, 32, 64):
        trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
        t0 = trajs[0]
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        ctxs = {}
        for name, opts in (("v4-auto", dict(kernel_version=4)), ("v4-rr", dict(kernel_version=4, contiguous=0)), ("v3", dict(kernel_version=3))):
            c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                           dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                           Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=4)
            c.set_stream(stream.cuda_stream)
            for k, v in opts.items():
                c.set_option(k, v)
            ctxs[name] = c
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        for c in ctxs.values():
            for _ in range(3):
                c.eval_jac_dev(Zd, dd, vd)
        stream.synchronize()
        res = {k: [] for k in ctxs}
        for rnd in range(4):
            for name in (list(ctxs) if rnd % 2 == 0 else list(ctxs)[::-1]):
                c = ctxs[name]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                reps = max(2, 80 // B)
                for _ in range(reps):
                    c.eval_jac_dev(Zd, dd, vd)
                e1.record(stream)
                stream.synchronize()
                res[name].append(e0.elapsed_time(e1) / reps * 1e3)
        for name, v in res.items():
            med = np.median(v)
            print("B=%2d %-8s: %s  median %.1f us/launch, %.2f us/trajectory (%.3f of 8 TB/s)" % (B, name, " ".join("%.1f" % x for x in v), med, med / B, B * 135119952 / med / 8e6), flush=True)
        for c in ctxs.values():
            c.close()
        del Zd, dd, vd
