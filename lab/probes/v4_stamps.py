#!/usr/bin/env python3
"""Pattern-compiled fused kernel (kernel_version 4), profile build: cycle stamps of workgroup 0 per wave, and ablations (no block
stores / no column chains / no tail stores: WRONG results, timing only)."""
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
        for B in (1, 8):
            trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
            t0 = trajs[0]
            Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
            for order in (4, 8):
                c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                               dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                               Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
                c.set_stream(stream.cuda_stream)
                dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
                vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
                c.set_option("kernel_version", 4)
                c.set_option("debug_timing", 1)
                c.set_option("profile_flags", 16)
                for _ in range(3):
                    c.eval_jac_dev(Zd, dd, vd)
                stream.synchronize()
                W = 64 + 2 * 1024
                out = (ctypes.c_int64 * W)()
                c._chk(c._L.pcl_debug_timing(c._h, out, W))
                t = np.array(out[:], dtype=np.int64)
                base = min(int(t[32 * w]) for w in range(len(roles)) if t[32 * w] > 0)
                print("---- B=%d order %d: stamps of workgroup 0 (cycles after the first stamp of the workgroup)" % (B, order))
                for w, nm in enumerate(roles):
                    st = t[32 * w:32 * w + 32]
                    st = st[st > 0]
                    print("%5s: %s" % (nm, " ".join("%d" % (x - base) for x in st)))
                if B == 1:  # the product's warm rate: the P wave repeats its second product 16 more times (flag 32), alone (flag 4: no chains)
                    for fl, var in ((32 + 4, 0), (32, 0), (32 + 4, 1), (32 + 4, 2), (32 + 4, 3)):
                        c.set_option("v4_variant", var)
                        c.set_option("profile_flags", fl)
                        for _ in range(2):
                            c.eval_jac_dev(Zd, dd, vd)
                        stream.synchronize()
                        c._chk(c._L.pcl_debug_timing(c._h, out, W))
                        st = np.array(out[:32], dtype=np.int64)
                        st = st[st > 0]
                        print("P wave, flags %d variant %d: %s" % (fl, var, " ".join("%d" % (x - st[0]) for x in st)))
                    c.set_option("v4_variant", 0)
                c.set_option("debug_timing", 0)
                for flags, what in ((0, "everything"), (2, "no block stores"), (4, "no column chains"), (8, "no tail stores"), (10, "no stores at all"), (6, "P chain only")):
                    c.set_option("profile_flags", flags)
                    for _ in range(3):
                        c.eval_jac_dev(Zd, dd, vd)
                    stream.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    reps = 50 if B == 1 else 20
                    for _ in range(reps):
                        c.eval_jac_dev(Zd, dd, vd)
                    e1.record(stream)
                    stream.synchronize()
                    print("B=%d order %d %-18s %.1f us/launch" % (B, order, what + ":", e0.elapsed_time(e1) / reps * 1e3), flush=True)
                c.close()
finally:
    pa.build_library(force=True)
