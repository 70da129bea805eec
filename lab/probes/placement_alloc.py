#!/usr/bin/env python3
"""The launch time against the ALLOCATION of the Jacobian values: torch's allocator, plain hipMalloc, hipExtMallocWithFlags(hipDeviceMallocContiguous),
several buffers of each kind alive at once; argv: trajectories per launch (1 or 8)."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
class Raw:
    dtype = "torch.float64"
    def __init__(self, p):
        self.p = p
    def data_ptr(self):
        return self.p
    def is_contiguous(self):
        return True
system = synthetic.config_system(3)
m = system.n_drives
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
    t0 = trajs[0]
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    c = pa.integrators._PclContext(d=system.levels, m=m, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start, dt_off=t0.components["Δt"].start,
                                   x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift, Gj=system.G_drives_array(), batch=B,
                                   batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=4)
    c.set_stream(stream.cuda_stream)
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    nbytes = c.jac_nnz * 8
    bufs = []
    for i in range(3):
        bufs.append(("torch %d" % i, torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")))
    for i in range(3):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), nbytes) == 0
        bufs.append(("hipMalloc %d" % i, Raw(p.value)))
    for i in range(3):
        p = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, 0x4)
        if rc != 0:
            print("hipExtMallocWithFlags(contiguous) failed:", rc)
            break
        bufs.append(("contiguous %d" % i, Raw(p.value)))
    res = [[] for _ in bufs]
    reps = 20 if B > 1 else 100
    for rnd in range(5):
        for i, (nm, vd) in enumerate(bufs):
            for _ in range(3):
                c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            res[i].append(e0.elapsed_time(e1) / reps * 1e3)
    for i, (nm, vd) in enumerate(bufs):
        print("%-14s at 0x%x: %s  median %.2f us" % (nm, vd.data_ptr(), " ".join("%.1f" % x for x in res[i]), np.median(res[i])))
