#!/usr/bin/env python3
"""Cycle stamps of a matrix-role workgroup of fused kernel 3 (compact launch = every workgroup in the matrix role), second item
of workgroup 0, wave 0: top | per chunk: S,D in LDS | G_l D | G^2 D | MFMA + accumulators -> LDS | outputs issued | next G built | barrier.
Builds the -DPCL_PROFILE library, then the shipped one again."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], batch, pade_order=4)
    c = ms.ctx
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    cv = torch.empty(c.compact_nnz, dtype=torch.float64, device="cuda")
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): c.eval_jac_compact_dev(Zd, dd, cv)
    c.sync()
    c.set_option("debug_timing", 1)
    c.eval_jac_compact_dev(Zd, dd, cv); c.sync()
    out = (ctypes.c_int64 * (64 + 2048))()
    c._chk(c._L.pcl_debug_timing(c._h, out, 64 + 2048))
    wg = np.array(out[64:]).reshape(2, 1024).T
    wg = wg[wg[:, 0] > 0]
    w0 = wg[:, 0].min()
    print("last_kernel", c.get_option("last_kernel"), "workgroups", len(wg), "start max %.1f us | end min %.1f max %.1f us (100 MHz clock)" % ((wg[:, 0].max() - w0) / 100.0, (wg[:, 1].min() - w0) / 100.0, (wg[:, 1].max() - w0) / 100.0))
    w = np.array(out[52:60])
    print("chunks done per wave (cycles after wave 0's item top):", (w - out[0]).tolist())
    t = np.array(out[:40]); t = t[t > 0]
    print("stamps (s_memtime, 100 MHz ticks) deltas:", np.diff(t).tolist(), "total", int(t[-1] - t[0]))
    ms.close()
finally:
    pa.build_library(force=True)
