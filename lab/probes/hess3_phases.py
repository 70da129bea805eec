#!/usr/bin/env python3
"""Phase stamps of Hessian kernel 3 (needs the -DPCL_PROFILE build: this script builds it, runs, and rebuilds the shipped library)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    for batch in (8, 1):
        trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
        ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], batch, pade_order=4)
        c = ms.ctx
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
        hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_option("hess_kernel", int(sys.argv[1]) if len(sys.argv) > 1 else 3)
        if batch == 1:
            c.set_option("grid", 49)  # two items per workgroup so that the stamped (second) item exists
        c.set_option("debug_timing", 1)
        for _ in range(3):
            c.hess_dev(Zd, mu, hv)
        c.sync()
        out = (ctypes.c_int64 * 64)()
        c._chk(c._L.pcl_debug_timing(c._h, out, 64))
        t = np.array(out[:]); t = t[t > 0]
        print("batch", batch, "stamps:", len(t), "deltas:", np.diff(t).tolist(), "total", int(t[-1] - t[0]) if len(t) else 0)
        c.set_option("debug_timing", 0)
        c.set_option("grid", 0)
        def tm(label):
            for _ in range(5): c.hess_dev(Zd, mu, hv)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40): c.hess_dev(Zd, mu, hv)
            e1.record(); torch.cuda.synchronize()
            print("   batch %d %-22s %.2f us/eval" % (batch, label, e0.elapsed_time(e1) * 1e3 / 40 / batch))
        tm("default")
        c.set_option("profile_flags", 1); tm("no output stores"); c.set_option("profile_flags", 0)
        c.set_option("nt_stores", 1); tm("nontemporal stores"); c.set_option("nt_stores", 0)
        ms.close()
finally:
    pa.build_library(force=True)
