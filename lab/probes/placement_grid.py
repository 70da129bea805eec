#!/usr/bin/env python3
"""8 trajectories per launch on several differently placed output buffers x grid sizes (the stride between the workgroups' store streams):
is the slow placement (224 against 188 us) channel camping of 256 regular streams?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nbuf = int(sys.argv[2]) if len(sys.argv) > 2 else 6
grids = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 255, 251, 248, 240, 224]
key = sys.argv[4] if len(sys.argv) > 4 else "grid"
system = synthetic.config_system(3)
m = system.n_drives
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
    t0 = trajs[0]
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start, dt_off=t0.components["Δt"].start,
                                   x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift, Gj=system.G_drives_array(), batch=B,
                                   batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=4)
    c.set_stream(stream.cuda_stream)
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    bufs = [torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
    res = {}
    for rnd in range(3):
        for g in grids:
            c.set_option(key, g)
            for i, vd in enumerate(bufs):
                for _ in range(3):
                    c.eval_jac_dev(Zd, dd, vd)
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(15):
                    c.eval_jac_dev(Zd, dd, vd)
                e1.record(stream)
                stream.synchronize()
                res.setdefault((g, i), []).append(e0.elapsed_time(e1) / 15 * 1e3)
    print("grid   " + " ".join("buf%-3d" % i for i in range(nbuf)))
    for g in grids:
        print("%4d   " % g + " ".join("%6.1f" % np.median(res[(g, i)]) for i in range(nbuf)))
