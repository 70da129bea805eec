#!/usr/bin/env python3
"""Kernel 4, profile build, 8 (or argv[1]) trajectories per launch: the life of every workgroup against the number of items (pieces of
intervals) its contiguous range touches, against its XCD (blockIdx mod 8) and against its position in the grid."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
order = int(sys.argv[2]) if len(sys.argv) > 2 else 4
extra = dict(kv.split("=") for kv in sys.argv[3:])
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    m, d = system.n_drives, system.levels
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
        t0 = trajs[0]
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        c = pa.integrators._PclContext(d=d, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                       dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                       Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
        c.set_stream(stream.cuda_stream)
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        c.set_option("kernel_version", 4)
        c.set_option("debug_timing", 1)
        for k, v in extra.items():
            if k != "profile_flags":
                c.set_option(k, int(v))
        base = int(extra.get("profile_flags", 0))
        for i in range(6):
            c.set_option("profile_flags", base | (64 if i & 1 else 0))
            c.eval_jac_dev(Zd, dd, vd)
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        reps = 20
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
        g = 256
        b = t[512 + 768:512 + 1536].reshape(256, 3)[:g]
        life = (b[:, 2] - b[:, 0]) / 100.0
        end = (b[:, 2] - b[:, 0].min()) / 100.0
        K = t0.N - 1
        tot = B * K * d
        bx = np.arange(g)
        lo, hi = tot * bx // g, tot * (bx + 1) // g
        nmy = (hi - 1) // d - lo // d + 1
        print("B=%d order %d %s: %.2f us launch to launch; contiguous %d, last kernel %d" % (B, order, extra, per, c.get_option("contiguous"), c.get_option("last_kernel")))
        print("life: median %.1f min %.1f max %.1f" % (np.median(life), life.min(), life.max()))
        for v in sorted(set(nmy)):
            s = life[nmy == v]
            print("  items %d: %3d workgroups, life median %.1f min %.1f max %.1f" % (v, len(s), np.median(s), s.min(), s.max()))
        for x in range(8):
            s = life[bx % 8 == x]
            print("  xcd %d: life median %.1f min %.1f max %.1f" % (x, np.median(s), s.min(), s.max()))
        for q in range(8):
            s = life[q * 32:(q + 1) * 32]
            print("  bx %3d..%3d: life median %.1f min %.1f max %.1f" % (q * 32, q * 32 + 31, np.median(s), s.min(), s.max()))
        first_cols = (lo // d + 1) * d - lo  # columns of the first (partial) item
        print("  corr(life, columns in the first item) = %.2f" % np.corrcoef(life, first_cols)[0, 1])
        print("  lives:", " ".join("%.0f" % x for x in life))
        c.close()
finally:
    pa.build_library(force=True)
