#!/usr/bin/env python3
"""Kernel 4, ONE trajectory per launch: round-robin slices (198 workgroups of 13-14 columns) against contiguous ranges over all
256 CUs (10-11 columns, some spanning two intervals), tail modes, ring sizes; kernel 3 beside them.  Alternating in one process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
order = int(sys.argv[1]) if len(sys.argv) > 1 else 4
system = synthetic.config_system(3)
t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    Zd = torch.from_numpy(t0.datavec.copy()).cuda()
    ctxs = {}
    variants = [("v3", dict(kernel_version=0)) if order == 4 else ("v4-auto", dict()),
                ("v4", dict(kernel_version=4)), ("v4-nocoop", dict(kernel_version=4, v4_flags=4)), ("v4-coop-nowait", dict(kernel_version=4, v4_flags=16)),
                ("v4-unbalanced", dict(kernel_version=4, v4_flags=32)), ("v4-nt0", dict(kernel_version=4, nt_stores=0))]
    for name, opts in variants:
        c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                       dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                       Gj=system.G_drives_array(), batch=1, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
        c.set_stream(stream.cuda_stream)
        for k, v in opts.items():
            c.set_option(k, v)
        ctxs[name] = c
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    ref = None
    for name, c in ctxs.items():
        for _ in range(5):
            c.eval_jac_dev(Zd, dd, vd)
        stream.synchronize()
        if name.startswith("v4"):
            if ref is None:
                ref = (dd.clone(), vd.clone())
            assert torch.equal(ref[0], dd) and torch.equal(ref[1], vd), name
    res = {k: [] for k in ctxs}
    for rnd in range(6):
        for name in (list(ctxs) if rnd % 2 == 0 else list(ctxs)[::-1]):
            c = ctxs[name]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(100):
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            res[name].append(e0.elapsed_time(e1) / 100 * 1e3)
    for name, v in res.items():
        print("order %d %-18s: %s  median %.2f us/launch (kernel id %d)" % (order, name, " ".join("%.1f" % x for x in v), np.median(v), ctxs[name].get_option("last_kernel")), flush=True)
