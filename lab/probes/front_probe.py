#!/usr/bin/env python3
"""8 trajectories per launch, 198 workgroups both ways: contiguous ranges (every workgroup 4 consecutive intervals: the whole 1 GB is the front) against
whole intervals dealt round-robin (198 consecutive intervals = 265 MB are the front, four full rounds), on several placements."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B, nbuf = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
    variants = [("contiguous, 256 workgroups (default)", dict(grid=0, contiguous=-1, cols_per_slice=0)),
                ("contiguous, 198 workgroups", dict(grid=198, contiguous=1, cols_per_slice=0)),
                ("round-robin intervals, 198 workgroups", dict(grid=198, contiguous=0, cols_per_slice=27)),
                ("round-robin intervals, 132 workgroups", dict(grid=132, contiguous=0, cols_per_slice=27)),
                ("round-robin half intervals, 198 workgroups", dict(grid=198, contiguous=0, cols_per_slice=14))]
    res = {}
    for rnd in range(3):
        for name, opts in variants:
            for k, v in opts.items():
                c.set_option(k, v)
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
                res.setdefault((name, i), []).append(e0.elapsed_time(e1) / 15 * 1e3)
    for name, _ in variants:
        print("%-46s " % name + " ".join("%6.1f" % np.median(res[(name, i)]) for i in range(nbuf)))
