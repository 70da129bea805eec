#!/usr/bin/env python3
"""General-order Hessian kernel (hess_kernel 7), profile build: cycle stamps of workgroup 0 per wave over its first interval(s).
Stamps per wave: inputs issued | barrier | after the Z phase | per level j: [drive waves: gather done | barrier A] | product done |
barrier B | contributions done | ... | output vectors stored."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    m = system.n_drives
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for B in (1,):
            trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
            t0 = trajs[0]
            Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
            for order, split in ((8, 0), (8, 1), (4, 1)):
                c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                               dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                               Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
                c.set_stream(stream.cuda_stream)
                mud = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
                hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
                c.set_option("hess_kernel", 7)
                c.set_option("hess_split", split)
                c.set_option("debug_timing", 1)
                for _ in range(3):
                    c.hess_dev(Zd, mud, hv)
                stream.synchronize()
                W = 64 + 2 * 1024
                out = (ctypes.c_int64 * W)()
                c._chk(c._L.pcl_debug_timing(c._h, out, W))
                t = np.array(out[:], dtype=np.int64)
                base = min(int(t[32 * w]) for w in range(m + 1 if not split else (m + 1) // 2 + 1) if t[32 * w] > 0)
                print("---- B=%d order %d, %d workgroup(s) per interval: stamps of workgroup 0 (cycles after its first stamp)" % (B, order, 1 + split))
                for w in range(m + 1 if not split else (m + 1) // 2 + 1):
                    st = t[32 * w:32 * w + 32]
                    st = st[st > 0]
                    print("%5s: %s" % ("W" if w == 0 else "V%d" % (w - 1), " ".join("%d" % (x - base) for x in st)))
                c.close()
finally:
    pa.build_library(force=True)
