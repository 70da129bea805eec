#!/usr/bin/env python3
"""Kernel 4, profile build: 100 MHz wall stamps of every workgroup (entry, tiles zeroed, last wave out) of two consecutive launches in a
back-to-back train -- how a launch-to-launch period splits into the gap between kernels, the spread of workgroup starts, the body and the
spread of the ends."""
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
        for B, order, extra in ((1, 4, {}), (1, 4, dict(profile_flags=2)), (1, 4, dict(profile_flags=10)), (1, 4, dict(profile_flags=14)), (1, 8, {}), (8, 4, {})):
            trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
            t0 = trajs[0]
            Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
            c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                           dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                           Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
            c.set_stream(stream.cuda_stream)
            dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
            vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
            c.set_option("kernel_version", 4)
            c.set_option("debug_timing", 1)
            base = extra.get("profile_flags", 0)
            for i in range(6):
                c.set_option("profile_flags", base | (64 if i & 1 else 0))
                c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            reps = 40
            for i in range(reps):
                c.set_option("profile_flags", base | (64 if i & 1 else 0))
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            per = e0.elapsed_time(e1) / reps * 1e3
            W = 64 + 2 * 1024
            out = (ctypes.c_int64 * W)()
            c._chk(c._L.pcl_debug_timing(c._h, out, W))
            t = np.array(out[:], dtype=np.int64)
            g = min(256, c.get_option("n_cu"))
            a = t[512:512 + 768].reshape(256, 3)[:g]
            b = t[512 + 768:512 + 1536].reshape(256, 3)[:g]
            a, b = a[a[:, 0] > 0], b[b[:, 0] > 0]
            us = lambda x: x / 100.0
            print("B=%d order %d flags %d: %.2f us launch to launch; %d workgroups" % (B, order, base, per, len(b)))
            print("   gap (last end of launch n -> first entry of launch n+1): %.2f us" % us(b[:, 0].min() - a[:, 2].max()))
            print("   entries spread over %.2f us (median %.2f after the first); tiles zeroed %.2f us after entry (median)" %
                  (us(b[:, 0].max() - b[:, 0].min()), us(np.median(b[:, 0]) - b[:, 0].min()), us(np.median(b[:, 1] - b[:, 0]))))
            print("   workgroup life entry -> last wave out: median %.2f, min %.2f, max %.2f us" % (us(np.median(b[:, 2] - b[:, 0])), us((b[:, 2] - b[:, 0]).min()), us((b[:, 2] - b[:, 0]).max())))
            print("   ends spread over %.2f us (median %.2f before the last); kernel first entry -> last out %.2f us" %
                  (us(b[:, 2].max() - b[:, 2].min()), us(b[:, 2].max() - np.median(b[:, 2])), us(b[:, 2].max() - b[:, 0].min())), flush=True)
            c.close()
finally:
    pa.build_library(force=True)
