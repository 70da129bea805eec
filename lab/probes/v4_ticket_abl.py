#!/usr/bin/env python3
"""Kernel 4 in ticket mode, profile build: launch times under the ablation flags (WRONG results, timing only).
usage: v4_ticket_abl.py [trajectories=8] [order=4] [cols,cols,...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
order = int(sys.argv[2]) if len(sys.argv) > 2 else 4
colss = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [3, 6, 9]
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    m = system.n_drives
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
        t0 = trajs[0]
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                       dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                       Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
        c.set_stream(stream.cuda_stream)
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        def t(reps=20):
            for _ in range(3):
                c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        c.set_option("v4_ticket", 0)
        c.set_option("profile_flags", 0)
        print("static split: %.1f us" % t())
        c.set_option("v4_ticket", 1)
        FL = ((0, "everything"), (4, "no chains"), (8, "no tail stores"), (6, "no chains, no block stores"), (2, "no block stores"))
        for cols in colss:
            for np_ in (0, 2):
                c.set_option("v4_ticket_cols", cols)
                c.set_option("v4_power_tiles", np_)
                print("cols %d tiles %d: " % (cols, np_) + " | ".join("%s %.1f" % (nm, (c.set_option("profile_flags", f), t())[1]) for f, nm in FL), flush=True)
        c.close()
finally:
    pa.build_library(force=True)
