#!/usr/bin/env python3
"""Cycle stamps of the pattern-compiled Hessian kernel (builds the -DPCL_PROFILE library, then rebuilds the shipped one)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    batch = 8
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], batch, pade_order=4)
    c = ms.ctx
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
    hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_option("hess_kernel", 4)
    if len(sys.argv) > 1: c.set_option("profile_flags", int(sys.argv[1]))
    for _ in range(3): c.hess_dev(Zd, mu, hv)
    c.sync()
    c.set_option("debug_timing", 1)
    c.hess_dev(Zd, mu, hv); c.sync()
    out = (ctypes.c_int64 * (64 + 2048))()
    c._chk(c._L.pcl_debug_timing(c._h, out, 64 + 2048))
    wg = np.array(out[64:]).reshape(1024, 2)
    wg = wg[wg[:, 0] > 0]
    w0 = wg[:, 0].min()
    print("workgroups", len(wg), "start (us): min %.1f max %.1f | end (us): min %.1f max %.1f" % (0.0, (wg[:, 0].max() - w0) / 100.0, (wg[:, 1].min() - w0) / 100.0, (wg[:, 1].max() - w0) / 100.0))
    t = np.array(out[:64]).reshape(4, 16)
    t0 = t[t > 0].min()
    for w in range(4):
        row = t[w]; row = row[row > 0]
        print("wave", w, "stamps (cycles since first):", (row - t0).tolist())
    ms.close()
finally:
    pa.build_library(force=True)
