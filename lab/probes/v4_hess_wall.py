#!/usr/bin/env python3
"""General-order Hessian kernel, profile build, one trajectory per launch: 100 MHz wall stamps of every workgroup of two consecutive launches
(gap between kernels, spread of the starts, workgroup life, spread of the ends), one and two workgroups per interval."""
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
        t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
        Zd = torch.from_numpy(t0.datavec.copy()[None]).cuda()
        for order, split in ((4, 1), (4, 0), (8, 1), (8, 0)):
            c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                           dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                           Gj=system.G_drives_array(), batch=1, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
            c.set_stream(stream.cuda_stream)
            mud = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
            hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
            c.set_option("hess_kernel", 7)
            c.set_option("hess_split", split)
            c.set_option("debug_timing", 1)
            for i in range(6):
                c.set_option("profile_flags", 64 if i & 1 else 0)
                c.hess_dev(Zd, mud, hv)
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            reps = 40
            for i in range(reps):
                c.set_option("profile_flags", 64 if i & 1 else 0)
                c.hess_dev(Zd, mud, hv)
            e1.record(stream)
            stream.synchronize()
            per = e0.elapsed_time(e1) / reps * 1e3
            W = 64 + 2 * 1024
            out = (ctypes.c_int64 * W)()
            c._chk(c._L.pcl_debug_timing(c._h, out, W))
            t = np.array(out[:], dtype=np.int64)
            a = t[512:512 + 768].reshape(256, 3)
            b = t[512 + 768:512 + 1536].reshape(256, 3)
            a, b = a[a[:, 0] > 0], b[b[:, 0] > 0]
            us = lambda x: x / 100.0
            print("order %d, %d workgroup(s) per interval: %.2f us launch to launch; %d workgroups" % (order, 1 + split, per, len(b)))
            print("   gap (last end of launch n -> first entry of launch n+1): %.2f us; entries spread over %.2f us" % (us(b[:, 0].min() - a[:, 2].max()), us(b[:, 0].max() - b[:, 0].min())))
            print("   workgroup life: median %.2f, min %.2f, max %.2f us; ends spread over %.2f us; kernel first entry -> last out %.2f us" %
                  (us(np.median(b[:, 2] - b[:, 0])), us((b[:, 2] - b[:, 0]).min()), us((b[:, 2] - b[:, 0]).max()), us(b[:, 2].max() - b[:, 2].min()), us(b[:, 2].max() - b[:, 0].min())), flush=True)
            c.close()
finally:
    pa.build_library(force=True)
