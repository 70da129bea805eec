#!/usr/bin/env python3
"""Kernel 4 at orders 4 / 8 / 10 and kernel 3 (order 4), 8 trajectories per launch, alternating in one process (clock ramps and
thermal drift hit every variant alike)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
system = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
t0 = trajs[0]
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    ctxs = {}
    for name, order, kv, np_, fl in (("v3-o4", 4, 0, 0, 0), ("v4-o2", 2, 4, 0, 0), ("v4-o4", 4, 4, 0, 0), ("v4-o4-np2", 4, 4, 2, 0), ("v4-o4-np1", 4, 4, 1, 0),
                                    ("v4-o6", 6, 4, 0, 0), ("v4-o6-np3", 6, 4, 3, 0), ("v4-o8", 8, 4, 0, 0), ("v4-o8-np2", 8, 4, 2, 0), ("v4-o10", 10, 4, 0, 0)):
        c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                       dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                       Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
        c.set_stream(stream.cuda_stream)
        c.set_option("kernel_version", kv)
        if kv == 4:
            c.set_option("v4_power_tiles", np_)
            c.set_option("v4_flags", fl)
        ctxs[name] = c
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    for c in ctxs.values():
        for _ in range(5):
            c.eval_jac_dev(Zd, dd, vd)
    stream.synchronize()
    res = {k: [] for k in ctxs}
    for rnd in range(6):
        for name in (list(ctxs) if rnd % 2 == 0 else list(ctxs)[::-1]):
            c = ctxs[name]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            reps = 20 if B > 1 else 100
            for _ in range(reps):
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            res[name].append(e0.elapsed_time(e1) / reps * 1e3)
    for name, v in res.items():
        print("B=%d %-14s: %s  median %.1f us/launch (%.3f of 8 TB/s)" % (B, name, " ".join("%.1f" % x for x in v), np.median(v), B * 135119952 / np.median(v) / 8e6), flush=True)
