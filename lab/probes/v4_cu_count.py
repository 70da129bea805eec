#!/usr/bin/env python3
"""Is the block stream of a one-trajectory launch bound by the memory system or by the number of CUs that store?  Profile build, chains off
(profile_flags 4: P and the stream only -- wrong tails, same block bytes): 198 workgroups of 13-14 columns against 256 of 10-11 columns
(contiguous ranges), and grids in between."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        Zd = torch.from_numpy(t0.datavec.copy()).cuda()
        c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                       dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                       Gj=system.G_drives_array(), batch=1, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=4)
        c.set_stream(stream.cuda_stream)
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        c.set_option("kernel_version", 4)
        variants = [("round-robin, 198 workgroups, chains on", dict(profile_flags=0, contiguous=-1, grid=0)),
                    ("round-robin, 198 workgroups, chains off", dict(profile_flags=4, contiguous=-1, grid=0)),
                    ("contiguous, 256 workgroups, chains off", dict(profile_flags=4, contiguous=1, grid=0)),
                    ("contiguous, 224 workgroups, chains off", dict(profile_flags=4, contiguous=1, grid=224)),
                    ("contiguous, 198 workgroups, chains off", dict(profile_flags=4, contiguous=1, grid=198)),
                    ("contiguous, 160 workgroups, chains off", dict(profile_flags=4, contiguous=1, grid=160)),
                    ("contiguous, 128 workgroups, chains off", dict(profile_flags=4, contiguous=1, grid=128))]
        res = {k: [] for k, _ in variants}
        for rnd in range(5):
            for name, opts in (variants if rnd % 2 == 0 else variants[::-1]):
                for k, v in opts.items():
                    c.set_option(k, v)
                for _ in range(5):
                    c.eval_jac_dev(Zd, dd, vd)
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(100):
                    c.eval_jac_dev(Zd, dd, vd)
                e1.record(stream)
                stream.synchronize()
                res[name].append(e0.elapsed_time(e1) / 100 * 1e3)
        for name, v in res.items():
            print("%-42s: median %.2f us/launch" % (name, np.median(v)), flush=True)
finally:
    pa.build_library(force=True)
