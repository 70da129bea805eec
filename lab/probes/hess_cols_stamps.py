#!/usr/bin/env python3
"""Column-group Hessian kernel (hess_kernel 8), profile build: cycle stamps of ONE wave (block `prof`): entry | inputs stored | power chains done |
per level: product done, contributions done | sums stored + counter | output vectors stored | end.  usage: hess_cols_stamps.py [order=8] [B=1] [block=0]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
order = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
blocks = [int(a) for a in (sys.argv[3] if len(sys.argv) > 3 else "0,3,300").split(",")]
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
        mud = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
        hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
        c.set_option("hess_kernel", 8)
        c.set_option("debug_timing", 1)
        for blk in blocks:
            if blk < 0:  # a timing variant of the generated product (WRONG results): -1 no ds_add | -2 no LDS operation in the epilogues | -3 half-depth chains
                c.set_option("v4_variant", -blk)
                blk = 0
            c.set_option("profile_flags", blk)
            for _ in range(3):
                c.hess_dev(Zd, mud, hv)
            stream.synchronize()
            out = (ctypes.c_int64 * 64)()
            c._chk(c._L.pcl_debug_timing(c._h, out, 64))
            t = np.array(out[:32], dtype=np.int64)
            t = t[t > 0]
            print("order %d B=%d block %d: %s  (total %d)" % (order, B, blk, " ".join("%d" % x for x in np.diff(t)), t[-1] - t[0]), flush=True)
        c.close()
finally:
    pa.build_library(force=True)
