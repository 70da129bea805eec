#!/usr/bin/env python3
"""Kernel 4, profile build, one trajectory per launch: fine stamps of the four cooperative waves per level of the first item's powers
(profile flag 512: wait for the previous power | operand in registers | product part | arrived), order given as argument (default 10)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
order = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    m = system.n_drives
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
        Zd = torch.from_numpy(t0.datavec.copy()).cuda()
        c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start, dt_off=t0.components["Δt"].start,
                                       x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift, Gj=system.G_drives_array(), batch=1,
                                       batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
        c.set_stream(stream.cuda_stream)
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        c.set_option("debug_timing", 1)
        c.set_option("profile_flags", 512)
        for kv in sys.argv[2:]:
            c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        for _ in range(20):
            c.eval_jac_dev(Zd, dd, vd)
        stream.synchronize()
        W = 64 + 2 * 1024
        out = (ctypes.c_int64 * W)()
        c._chk(c._L.pcl_debug_timing(c._h, out, W))
        t = np.array(out[:], dtype=np.int64)
        base = min(int(t[32 * w]) for w in range(4) if t[32 * w] > 0)
        print("order %d: stamps of the cooperative waves of workgroup 0 (cycles after the first one's first stamp); per level j >= 2: wait, operand, product, arrived" % order)
        for w in range(4):
            st = t[32 * w:32 * w + 32]
            st = st[st > 0] - base
            print("wave %d:" % w, " ".join("%d" % x for x in st))
            # the first stamp after the scalars is G's (one stamp); then groups of 4
        c.close()
finally:
    pa.build_library(force=True)
