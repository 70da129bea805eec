#!/usr/bin/env python3
"""Kernel 4 in ticket mode, profile build: cycle stamps of workgroup 0 per wave (first 32 each), and launch times with the
ablation flags.  usage: v4_ticket_stamps.py [trajectories=8] [order=4] [key=value ...]"""
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
    m = system.n_drives
    roles = ["P", "W", "V"] + ["dW%d" % l for l in range(m)] + ["load", "write"] + ["str%d" % i for i in range(4)] + ["disp"]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
        t0 = trajs[0]
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                       dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                       Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
        c.set_stream(stream.cuda_stream)
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        c.set_option("v4_ticket", 1)
        for k, v in extra.items():
            c.set_option(k, int(v))
        W = 64 + 2 * 1024
        out = (ctypes.c_int64 * W)()
        for flags in (0,):
            c.set_option("debug_timing", 1)
            c.set_option("profile_flags", flags)
            for _ in range(3):
                c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
            c._chk(c._L.pcl_debug_timing(c._h, out, W))
            t = np.array(out[:], dtype=np.int64)
            base = min(int(t[32 * w]) for w in range(len(roles)) if t[32 * w] > 0)
            print("---- B=%d order %d %s flags %d: stamps of workgroup 0 (first: cycles after the workgroup's first stamp; then differences)" % (B, order, extra, flags))
            for w, nm in enumerate(roles):
                st = t[32 * w:32 * w + 32]
                st = st[st > 0]
                if len(st):
                    print("%5s: %d | %s" % (nm, st[0] - base, " ".join("%d" % x for x in np.diff(st))))
            c.set_option("debug_timing", 0)
        for flags, what in ((0, "everything"), (2, "no block stores"), (4, "no column chains"), (8, "no tail stores"), (6, "block pipeline without stores")):
            c.set_option("profile_flags", flags)
            for _ in range(3):
                c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            reps = 20
            for _ in range(reps):
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            print("B=%d order %d %-30s %.1f us/launch" % (B, order, what + ":", e0.elapsed_time(e1) / reps * 1e3), flush=True)
        c.close()
finally:
    pa.build_library(force=True)
