#!/usr/bin/env python3
"""Kernel 4, profile build, one trajectory per launch: launch-to-launch period, gap between kernels, workgroup life and end spread for option
sets given as arguments ("key=value,key=value" each; "-" = defaults)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
profile = os.environ.get("PCL_PROBE_PROFILE", "1") == "1"
if profile:
    pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    m = system.n_drives
    stream = torch.cuda.Stream()
    order = int(os.environ.get("PCL_PROBE_ORDER", "4"))
    B = int(os.environ.get("PCL_PROBE_B", "1"))
    with torch.cuda.stream(stream):
        trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
        t0 = trajs[0]
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        sets = [dict() if a == "-" else {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",")} for a in sys.argv[1:]] or [dict()]
        ctxs = []
        for extra in sets:
            c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                           dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                           Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
            c.set_stream(stream.cuda_stream)
            c.set_option("kernel_version", 4)
            if profile:
                c.set_option("debug_timing", 1)
            for k, v in extra.items():
                if k != "profile_flags":
                    c.set_option(k, v)
            ctxs.append(c)
        dd = torch.empty(ctxs[0].n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(ctxs[0].jac_nnz, dtype=torch.float64, device="cuda")
        res = [[] for _ in sets]
        info = [None] * len(sets)
        for rnd in range(5):
            order_ = list(range(len(sets)))
            if rnd & 1:
                order_.reverse()
            for i in order_:
                c, extra = ctxs[i], sets[i]
                base = extra.get("profile_flags", 0)
                for j in range(6):
                    if profile:
                        c.set_option("profile_flags", base | (64 if j & 1 else 0))
                    c.eval_jac_dev(Zd, dd, vd)
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                reps = 60
                for j in range(reps):
                    if profile:
                        c.set_option("profile_flags", base | (64 if j & 1 else 0))
                    c.eval_jac_dev(Zd, dd, vd)
                e1.record(stream)
                stream.synchronize()
                res[i].append(e0.elapsed_time(e1) / reps * 1e3)
                if profile:
                    W = 64 + 2 * 1024
                    out = (ctypes.c_int64 * W)()
                    c._chk(c._L.pcl_debug_timing(c._h, out, W))
                    t = np.array(out[:], dtype=np.int64)
                    a = t[512:512 + 768].reshape(256, 3)
                    b = t[512 + 768:512 + 1536].reshape(256, 3)
                    a, b = a[a[:, 0] > 0], b[b[:, 0] > 0]
                    life = (b[:, 2] - b[:, 0]) / 100.0
                    info[i] = (len(b), (b[:, 0].min() - a[:, 2].max()) / 100.0, np.median(life), life.max(), (b[:, 2].max() - np.median(b[:, 2])) / 100.0, (b[:, 2].max() - b[:, 0].min()) / 100.0)
        for i, extra in enumerate(sets):
            s = "%-44s: %6.2f us launch to launch (%s)" % (extra or "defaults", np.median(res[i]), " ".join("%.1f" % x for x in res[i]))
            if info[i]:
                s += "; %d wgs, gap %.2f, life median %.2f max %.2f, median end %.2f before the last, kernel %.2f" % info[i]
            print(s, flush=True)
finally:
    if profile:
        pa.build_library(force=True)
