#!/usr/bin/env python3
"""Kernel 4, profile build, one trajectory per launch: fine cycle stamps of workgroup 0 (profile flag 128: kernel entry, barrier, scalars
arrived, cooperative powers, fold set-up, folds, first two block columns, end)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    m = system.n_drives
    roles = ["P", "W", "V"] + ["dW%d" % l for l in range(m)] + ["load", "write"] + ["str%d" % i for i in range(4)]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
        Zd = torch.from_numpy(t0.datavec.copy()).cuda()
        for order in (4, 8, 10):
            for extra in sys.argv[1:] or ["-"]:
                opts = {} if extra == "-" else {kv.split("=")[0]: int(kv.split("=")[1]) for kv in extra.split(",")}
                c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                               dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                               Gj=system.G_drives_array(), batch=1, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
                c.set_stream(stream.cuda_stream)
                dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
                vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
                c.set_option("kernel_version", 4)
                c.set_option("debug_timing", 1)
                for k, v in opts.items():
                    c.set_option(k, v)
                c.set_option("profile_flags", 128 | opts.get("profile_flags", 0))
                for _ in range(20):
                    c.eval_jac_dev(Zd, dd, vd)
                stream.synchronize()
                W = 64 + 2 * 1024
                out = (ctypes.c_int64 * W)()
                c._chk(c._L.pcl_debug_timing(c._h, out, W))
                t = np.array(out[:], dtype=np.int64)
                base = min(int(t[32 * w]) for w in range(len(roles)) if t[32 * w] > 0)
                print("---- order %d %s: stamps of workgroup 0 (cycles after the first wave's entry)" % (order, opts))
                for w, nm in enumerate(roles):
                    st = t[32 * w:32 * w + 32]
                    st = st[st > 0]
                    print("%5s: %s" % (nm, " ".join("%d" % (x - base) for x in st)))
                c.close()
finally:
    pa.build_library(force=True)
