#!/usr/bin/env python3
"""Cycle stamps of the pattern-compiled column kernel (workgroup 0, second interval; waves 0, 1, 2 and the last): top | D,S read | table
lines touched | G D | own product(s) | outputs issued | staged | barrier passed.  Builds the -DPCL_PROFILE library, then the shipped one."""
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
    c.set_option("column_kernel", 2)
    for _ in range(3): c.eval_jac_compact_dev(Zd, dd, cv)
    c.sync()
    c.set_option("debug_timing", 1)
    c.eval_jac_compact_dev(Zd, dd, cv); c.sync()
    out = (ctypes.c_int64 * 64)()
    c._chk(c._L.pcl_debug_timing(c._h, out, 64))
    t = np.array(out[:]).reshape(4, 16)
    t0 = t[t > 0].min()
    print("last_kernel", c.get_option("last_kernel"))
    for w, name in enumerate(("wave 0 (d/ddt)", "wave 1 (delta)", "wave 2 (d/du_0)", "last wave")):
        row = t[w]; row = row[row > 0]
        print(name, "stamps (cycles since the first):", (row - t0).tolist())
    ms.close()
finally:
    pa.build_library(force=True)
